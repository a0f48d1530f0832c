"""The int8 instantiation of the 256 x 256-tile kernel at the C3 shape on operands of different switching activity (same binary, same
launch): zeros, a small range, the full range.  Device time, back-to-back launches after a clock ramp.  -> stdout"""
import ctypes
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

M, K, N = 4096, 4096, 11008


def case(name, a, bt):
    sa, sb = 0.25 * torch.rand((M, 1), device="cuda"), 0.25 * torch.rand((N, 1), device="cuda")
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()
    t_end = time.perf_counter() + 2.0
    vals = []
    while time.perf_counter() < t_end:
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (200, ctypes.byref(ms))), "time")
        vals.append(ms.value)
    t = statistics.median(vals[len(vals) // 2:])
    ops = 2.0 * M * N * K
    print(f"int8 {name:44s}: {t * 1e3:7.1f} us  {ops / t / 1e9:6.0f} TOP/s ({ops / t / 1e9 / 5000:.3f} of 5 POP/s)", flush=True)


torch.manual_seed(0)
ri = lambda lo, hi, r: torch.randint(lo, hi, (r, K), dtype=torch.int8, device="cuda")  # noqa: E731
case("all zero", torch.zeros((M, K), dtype=torch.int8, device="cuda"), torch.zeros((N, K), dtype=torch.int8, device="cuda"))
case("one constant (3)", torch.full((M, K), 3, dtype=torch.int8, device="cuda"), torch.full((N, K), 3, dtype=torch.int8, device="cuda"))
case("uniform in [0, 8)", ri(0, 8, M), ri(0, 8, N))
case("uniform in [-32, 32) (the tests' recipe)", ri(-32, 32, M), ri(-32, 32, N))
case("uniform in [-128, 128) (full range)", ri(-128, 128, M), ri(-128, 128, N))
