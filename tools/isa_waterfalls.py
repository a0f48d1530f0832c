"""Count hipcc's waterfall loops around buffer instructions in the library's gfx950 code (cdna_hip_programming.md T20).

    python tools/isa_waterfalls.py [source.hip ...]

Compiles each HIP source device-only to assembly (no GPU needed) and lists every kernel in which an `s_and_saveexec_b64` is
preceded by `v_readfirstlane_b32`s and followed within two lines by a `buffer_*` instruction: a buffer descriptor hipcc keeps in
VGPRs, re-read lane by lane around every use.  Exit status 1 if any kernel has one.
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from conch_amd._build import CSRC, FLAGS, HIPCC, SOURCES  # noqa: E402


def waterfalls(src: Path) -> list[tuple[int, str]]:
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / (src.stem + ".s")
        subprocess.run([HIPCC, *[f for f in FLAGS if f != "-fPIC"], "--cuda-device-only", "-S", str(src), "-o", str(out)], check=True,
                       capture_output=True)
        text = out.read_text()
    rows = []
    for fn in re.split(r"\n(?=_Z[^\n:]*:)", text):
        name = fn.split(":")[0]
        if not name.startswith("_Z"):
            continue
        lines = fn.split("\n")
        # a waterfall: v_readfirstlane + v_cmp_eq just above the saveexec (the descriptor words, or a per-lane soffset), the buffer
        # instruction just below, `s_xor_b64 exec, exec` just behind it
        n = sum(1 for i, l in enumerate(lines) if "s_and_saveexec_b64" in l and any("buffer_" in x for x in lines[i + 1:i + 4])
                and any("v_readfirstlane_b32" in x for x in lines[max(0, i - 10):i])
                and any("v_cmp_eq" in x for x in lines[max(0, i - 6):i])
                and any("s_xor_b64 exec, exec" in x for x in lines[i + 1:i + 8]))
        if n:
            rows.append((n, name))
    return rows


def main() -> int:
    srcs = [Path(a) for a in sys.argv[1:]] or [CSRC / s for s in SOURCES]
    bad = 0
    for src in srcs:
        rows = waterfalls(src)
        names = subprocess.run(["c++filt"], input="\n".join(n for _, n in rows), capture_output=True, text=True).stdout.splitlines()
        print(f"{src.name}: {len(rows)} kernel(s) with waterfalled buffer instructions")
        for (n, _), d in zip(rows, names):
            print(f"    {n:3d}  {d[:150]}")
        bad += len(rows)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
