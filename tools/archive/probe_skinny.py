"""Where the time of the one-launch split-K kernel goes (development aid; BASELINE config C2 by default).

Loads the diagnostic twin of the library (python -m conch_amd._build --probe), launches the workload back to back and
reads the eight 100 MHz stamps thread 0 of every workgroup took (gemm_skinny.hip SK_STAMP):
  0 entry, 1 loads issued, 2 slice consumed (last MFMA issued), 3 own slab stores drained, 4 ticket drawn (second barrier),
  last arriver only: 5 other slabs loaded and summed, 6 C stores issued, 7 C stores drained.
Prints, for the LAST launch, the medians of the intervals and the launch's span (first entry -> last stamp).

usage: python tools/probe_skinny.py [MxKxN[:int8|fp8]] [launches]
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
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "128x4096x4096:int8"
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dims, _, dt = wl.partition(":")
    m, k, n = (int(v) for v in dims.split("x"))
    torch.manual_seed(0)
    if (dt or "int8") == "int8":
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
        bt = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, out.dtype)
    for _ in range(launches):
        kg.scaled_gemm_launcher(out, a, bt.T, sa, sb, md)
    torch.cuda.synchronize()
    lib = _C.load()
    fn = lib.conch_debug_probe_skinny
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    blocks = 4096
    buf = (ctypes.c_ulonglong * (8 * blocks))()
    assert fn(buf, blocks) == 0
    rows = [tuple(buf[8 * b:8 * b + 8]) for b in range(2048) if buf[8 * b]]
    # shader clock between entry and "slice consumed": s_memtime ticks per 100 MHz tick (rows b + 2048 hold the s_memtime pair)
    ghz = [(buf[8 * (b + 2048) + 1] - buf[8 * (b + 2048)]) / (buf[8 * b + 2] - buf[8 * b]) * 0.1 for b in range(2048)
           if buf[8 * b] and buf[8 * b + 2] > buf[8 * b] and buf[8 * (b + 2048) + 1] > buf[8 * (b + 2048)]]
    if ghz:
        print(f"  shader clock between entry and 'slice consumed': median {statistics.median(ghz):.2f} GHz")
    if rows and not any(r[4] > r[0] for r in rows):  # two-launch form: stamps 0..2 only
        for i, name in enumerate(["entry -> loads issued", "loads issued -> slice consumed"]):
            d = sorted((r[i + 1] - r[i]) * 0.01 for r in rows)
            print(f"  {name:34s} median {statistics.median(d):5.2f} us  p10 {d[len(d) // 10]:5.2f}  p90 {d[len(d) * 9 // 10]:5.2f}  max {d[-1]:5.2f}   (two-launch form)")
        return
    if not rows:
        print("no stamps: the workload did not go through skinny_splitk_kernel")
        return
    # stamps of earlier launches stay in slots 5..7 of workgroups that were not the last arriver this time: keep what follows slot 4
    t0 = min(r[0] for r in rows)
    us = lambda v: v * 0.01
    names = ["entry -> loads issued", "loads issued -> slice consumed", "consumed -> slab stores drained", "drained -> ticket drawn"]
    print(f"{wl}: {len(rows)} workgroups stamped; entries span {us(max(r[0] for r in rows) - t0):.2f} us")
    for i, name in enumerate(names):
        d = sorted(us(r[i + 1] - r[i]) for r in rows)
        print(f"  {name:34s} median {statistics.median(d):5.2f} us  p10 {d[len(d) // 10]:5.2f}  p90 {d[len(d) * 9 // 10]:5.2f}  max {d[-1]:5.2f}")
    last = [r for r in rows if r[5] > r[4] and r[7] >= r[6] >= r[5] and us(r[5] - r[4]) < 100]
    for i, name in ((4, "ticket -> slabs summed"), (5, "summed -> C stores issued"), (6, "C stores issued -> drained")):
        d = sorted(us(r[i + 1] - r[i]) for r in last)
        if d:
            print(f"  {name:34s} median {statistics.median(d):5.2f} us  p10 {d[len(d) // 10]:5.2f}  p90 {d[len(d) * 9 // 10]:5.2f}  max {d[-1]:5.2f}   ({len(d)} last arrivers)")
    end = max([r[4] for r in rows] + [r[7] for r in last])
    print(f"  first entry -> last stamp {us(end - t0):.2f} us;  ticket drawn at (since first entry) median {statistics.median(us(r[4] - t0) for r in rows):.2f} "
          f"max {us(max(r[4] for r in rows) - t0):.2f} us")


if __name__ == "__main__":
    main()
