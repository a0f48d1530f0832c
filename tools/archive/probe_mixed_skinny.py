"""Where the time of the one-launch int4 decode kernel goes (development aid; gemm_mixed_skinny.hip MS_STAMP).

Diagnostic twin of the library (python -m conch_amd._build --probe); eight 100 MHz stamps of thread 0 of every workgroup:
  0 entry, 1 loads issued, 2 loads landed (barrier), 3 slice consumed, 4 own slab stores drained, 5 ticket drawn,
  last arriver: 6 slabs loaded and summed, 7 C stores drained.

usage: python tools/probe_mixed_skinny.py [MxKxN] [launches]
"""
import ctypes
import os
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from conch_amd import _C, _build  # noqa: E402

_C.LIB_PATH = Path(os.environ["CONCH_PROBE_LIB"]) if os.environ.get("CONCH_PROBE_LIB") else _build.PROBE_LIB
from conch_amd.ops.quantization.gemm import mixed_precision_gemm  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "16x4096x4096"
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    m, k, n = (int(v) for v in wl.split("x"))
    torch.manual_seed(0)
    x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.05 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    for _ in range(launches):
        mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
    torch.cuda.synchronize()
    fn = _C.load().conch_debug_probe_mixed_skinny
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    blocks = 4096
    buf = (ctypes.c_ulonglong * (8 * blocks))()
    assert fn(buf, blocks) == 0
    rows = [tuple(buf[8 * b:8 * b + 8]) for b in range(blocks) if buf[8 * b]]
    if not rows:
        print("no stamps: the workload did not go through mixed_skinny_kernel")
        return
    t0 = min(r[0] for r in rows)
    print(f"int4 x fp16 {wl}: {len(rows)} workgroups stamped; entries span {(max(r[0] for r in rows) - t0) * 0.01:.2f} us")
    names = ["entry -> loads issued", "issued -> landed (barrier)", "landed -> slice consumed", "consumed -> slab stores drained", "drained -> ticket drawn"]
    for i, name in enumerate(names):
        d = sorted((r[i + 1] - r[i]) * 0.01 for r in rows if r[i + 1] >= r[i])
        if d:
            print(f"  {name:34s} median {statistics.median(d):5.2f} us  p10 {d[len(d) // 10]:5.2f}  p90 {d[len(d) * 9 // 10]:5.2f}  max {d[-1]:5.2f}")
    last = [r for r in rows if r[7] >= r[6] > r[5] and (r[6] - r[5]) * 0.01 < 100]
    for i, name in ((5, "ticket -> slabs summed"), (6, "summed -> C stores drained")):
        d = sorted((r[i + 1] - r[i]) * 0.01 for r in last)
        if d:
            print(f"  {name:34s} median {statistics.median(d):5.2f} us  p10 {d[len(d) // 10]:5.2f}  p90 {d[len(d) * 9 // 10]:5.2f}  max {d[-1]:5.2f}   ({len(d)} last arrivers)")
    end = max([r[5] for r in rows] + [r[7] for r in last])
    print(f"  first entry -> last stamp {(end - t0) * 0.01:.2f} us")


if __name__ == "__main__":
    main()
