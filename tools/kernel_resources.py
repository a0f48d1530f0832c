"""Per-kernel register / scratch / LDS table of the library's gfx950 code objects.

    python tools/kernel_resources.py [--all] [source.hip ...]

Compiles each HIP source device-only with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per
kernel.  Kernels with scratch (register spills) are marked `SPILL`; exit status 1 if any kernel of the MFMA hot loops
spills.  `--all` lists every kernel, the default only those with > 128 VGPRs or scratch.
"""

from __future__ import annotations

import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from conch_amd._build import CSRC, FLAGS, HIPCC, SOURCES  # noqa: E402


def demangle(names: list[str]) -> list[str]:
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return out.stdout.splitlines()


def resources(src: Path) -> list[dict]:
    cmd = [HIPCC, *[f for f in FLAGS if f != "-fPIC"], "--cuda-device-only", "-c", str(src), "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: \s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" ")[0]] = int(val)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"conch::\(anonymous namespace\)::", "", n).split("(")[0]
    return rows


def main() -> int:
    show_all = "--all" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")] or SOURCES
    bad = 0
    print(f"{'kernel':100s} {'VGPR':>5s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
    for f in files:
        for r in resources(CSRC / f if not Path(f).exists() else Path(f)):
            spill = r.get("ScratchSize", 0) > 0
            if show_all or spill or r.get("VGPRs", 0) > 128:
                print(f"{r['name'][:100]:100s} {r.get('VGPRs', 0):5d} {r.get('ScratchSize', 0):7d} {r.get('LDS', 0):7d} "
                      f"{r.get('Occupancy', 0):3d}{'  SPILL' if spill else ''}")
            bad += spill
    print(f"{bad} kernel(s) with scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
