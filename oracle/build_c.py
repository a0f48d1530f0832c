"""Compile oracle/oracle_c.c (the plain-C part of the CPU oracle) with gcc into oracle/liboracle_c.so.

Test infrastructure: `__graft_entry__.build()` builds it, `tests/test_oracle_c.py` uses it.
"""

from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "oracle_c.c"
LIB = HERE / "liboracle_c.so"


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        # -ffp-contract=off / no fast-math: every operation rounds exactly once, as written
        subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                        str(SRC), "-o", str(LIB), "-lm"], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
