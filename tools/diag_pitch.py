"""Diagnostic: does the power-of-two row pitch of A / B^T (K = 4096 bytes) cost the scaled GEMM anything (L2 channel / set
conflicts)?  Same data in tensors whose rows are K and K + 128 / + 256 bytes apart, interleaved timing in one process."""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

for (m, k, n) in [(4096, 4096, 11008), (8192, 8192, 8192)]:
    torch.manual_seed(0)
    a0 = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
    b0 = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    cases = {}
    for pad in (0, 128, 256, 1024 + 128):
        a = torch.zeros((m, k + pad), dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)[:, :k]
        bt = torch.zeros((n, k + pad), dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)[:, :k]
        a.copy_(a0)
        bt.copy_(b0)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
        cases[pad] = (a, bt, out, md)
    ms = ctypes.c_float()

    def run(pad, iters):
        a, bt, out, md = cases[pad]
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
        return ms.value

    iters = int(30.0 / max(run(0, 20), 1e-4))
    for _ in range(8):
        for pad in cases:
            run(pad, iters)
    torch.cuda.synchronize()
    same = {pad: bool(torch.equal(cases[pad][2], cases[0][2])) for pad in cases}
    res = {pad: [] for pad in cases}
    for _ in range(9):
        for pad in cases:
            res[pad].append(run(pad, iters))
    for pad, r in res.items():
        med = statistics.median(r)
        print(f"{m}x{k}x{n} row pitch K + {pad:4d} B: median {med * 1e3:8.2f} us  min {min(r) * 1e3:8.2f} us  {2.0 * m * n * k / med / 1e9:7.1f} TFLOP/s  bits {'equal' if same[pad] else 'DIFFER'}",
              flush=True)
