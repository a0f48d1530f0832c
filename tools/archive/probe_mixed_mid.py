"""Stamps of the strip kernel's K-split forms at batched-decode sizes (diagnostic twin of the library): kernel entry -> K loop,
K loop, K loop -> slab stores issued, first entry -> last exit.  usage: python tools/archive/probe_mixed_mid.py"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from conch_amd import _C, _build  # noqa: E402

_C.LIB_PATH = _build.PROBE_LIB
from conch_amd.ops.quantization.gemm import mixed_precision_gemm  # noqa: E402


def case(m, k, n):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    for _ in range(3000):
        mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
    torch.cuda.synchronize()
    lib = _C.load()
    fn = lib.conch_debug_probe_mixed_strip
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    blocks = 1024
    buf = (ctypes.c_ulonglong * (8 * blocks))()
    assert fn(buf, blocks) == 0
    rows = [tuple(buf[8 * b + i] for i in range(8)) for b in range(blocks)]
    rows = [r for r in rows if r[3] > r[1] > 0 and r[7] > 0]
    # (t0, r0) loop start, (t1, r1) loop end, (t2, r2) entry, (t3, r3) exit; r in 100 MHz ticks
    last = max(r[7] for r in rows)
    rows = [r for r in rows if last - r[7] < 5000]  # the last launch only (50 us window)
    pro = statistics.median((r[1] - r[5]) * 0.01 for r in rows)
    loop = statistics.median((r[3] - r[1]) * 0.01 for r in rows)
    epi = statistics.median((r[7] - r[3]) * 0.01 for r in rows)
    ghz = statistics.median((r[2] - r[0]) / (r[3] - r[1]) * 0.1 for r in rows)
    first, end = min(r[5] for r in rows), max(r[7] for r in rows)
    ins = sorted((r[5] - first) * 0.01 for r in rows)
    print(f"{m}x{k}x{n}: {len(rows)} workgroups stamped; entry -> K loop {pro:.2f} us, K loop {loop:.2f} us at {ghz:.2f} GHz, K loop -> slab stores issued {epi:.2f} us; "
          f"entries span {ins[-1]:.2f} us (median {ins[len(ins) // 2]:.2f}); first entry -> last exit {(end - first) * 0.01:.2f} us", flush=True)


if __name__ == "__main__":
    for shape in ((64, 4096, 11008), (128, 4096, 11008), (64, 8192, 28672), (256, 4096, 11008)):
        case(*shape)
