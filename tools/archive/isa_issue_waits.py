"""List the kernels whose ISSUE phase contains a wait: an `s_waitcnt vmcnt(N)` between the first and the last vector-memory load that
precede the kernel's first MFMA (development aid).  In a "put the whole slice in flight, then consume it" kernel such a wait is a
memory round trip -- or a drain of everything in flight -- before the remaining loads are even requested; hipcc inserts one at
every branch of the issue code, for every value that is loaded and converted / copied before the loop, and for every register two
branches share.

    python tools/isa_issue_waits.py gemm_skinny.hip gemm_mixed_skinny.hip bnb.hip
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from conch_amd._build import CSRC, FLAGS, HIPCC  # noqa: E402


def scan(src: Path) -> list[tuple[str, int, int, list[str]]]:
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / (src.stem + ".s")
        subprocess.run([HIPCC, *[f for f in FLAGS if f != "-fPIC"], "--cuda-device-only", "-S", str(src), "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
    rows = []
    for fn in re.split(r"\n(?=_Z[^\n:]*:)", text):
        name = fn.split(":")[0]
        if not name.startswith("_Z") or name.endswith(".kd"):
            continue
        lines = [l.strip() for l in fn.split("\n") if l.strip() and not l.strip().startswith(";")]
        mfma = next((i for i, l in enumerate(lines) if l.startswith("v_mfma")), None)
        if mfma is None:
            continue
        loads = [i for i, l in enumerate(lines[:mfma]) if re.match(r"(buffer_load|global_load)", l)]
        if len(loads) < 8:
            continue
        waits = [l for i, l in enumerate(lines[:mfma]) if loads[0] < i < loads[-1] and l.startswith("s_waitcnt") and "vmcnt" in l]
        if waits:
            rows.append((name, len(loads), len(waits), waits[:4]))
    return rows


if __name__ == "__main__":
    total = 0
    for arg in sys.argv[1:] or ["gemm_skinny.hip", "gemm_mixed_skinny.hip", "bnb.hip"]:
        rows = scan(CSRC / arg)
        total += len(rows)
        print(f"{arg}: {len(rows)} kernels with a wait inside their issue phase")
        for name, n_loads, n_waits, first in rows[:40]:
            print(f"  {name[:110]}  loads {n_loads}  waits {n_waits}  {first}")
    sys.exit(1 if total else 0)
