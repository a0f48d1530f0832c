"""int4 decode kernel: CONCH_TUNE_SKINNY_GATHER 1 (words into registers) / 2 (LDS-DMA staging) / 0 (automatic), interleaved in one
process.  usage: python tools/ab_mixed_decode_gather.py [MxKxN ...]"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

shapes = [a for a in sys.argv[1:] if "x" in a] or ["48x4096x4096", "64x4096x4096", "64x4096x11008", "64x8192x8192", "128x4096x4096", "128x4096x11008", "256x4096x4096",
                                                   "64x11008x4096", "40x4096x28672"]
ms = ctypes.c_float()
for wl in shapes:
    m, k, n = (int(v) for v in wl.split("x"))
    torch.manual_seed(0)
    x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.05 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    out = torch.zeros((m, n), dtype=torch.float16, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)
    _C.set_gemm_variant(_C.VARIANT_MFMA_SKINNY)

    def run(g, iters):
        _C.set_tuning(_C.TUNE_SKINNY_GATHER, g)
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms))), "time")
        return ms.value

    iters = int(min(2000, max(100, 20.0 / max(run(1, 20), 1e-4))))
    res = {1: [], 2: [], 0: []}
    for _ in range(3):
        for g in res:
            run(g, iters)
    for _ in range(7):
        for g in res:
            res[g].append(run(g, iters))
    med = {g: statistics.median(v) * 1e3 for g, v in res.items()}
    print(f"int4 x fp16 {wl:16s} registers {med[1]:6.2f} us   LDS-DMA {med[2]:6.2f} us (x{med[2] / med[1]:.3f})   automatic {med[0]:6.2f} us", flush=True)
_C.set_tuning(_C.TUNE_SKINNY_GATHER, 0)
_C.set_gemm_variant(_C.VARIANT_AUTO)
