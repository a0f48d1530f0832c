"""Interleaved A/B: mixed_precision_gemm on the plain [K/pf][N] weights against the pre-packed image (same process, same
data, rounds interleaved: cdna_hip_programming.md rule 24).  usage: python tools/ab_prepack.py [rounds]"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402
from conch_amd.ops.quantization.prepack import _prepacked_args, prepack_mixed_weights  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 9


def case(m, k, n, dtype, bits, zp):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    wzp = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda") if zp else None
    bias = 0 if zp else (8 if bits == 4 else 128)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, wzp, bits, bias, 128)
    pre = prepack_mixed_weights(wq, bits, m_hint=m, per_group_zero_points=zp)
    ms = ctypes.c_float()

    def plain(iters):
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, wzp, md, (iters, ctypes.byref(ms))), "time")
        return ms.value

    args, _keep = _prepacked_args(out, x, pre, ws, wzp, bias, 128)

    def packed(iters):
        _C.check(_C.load().conch_time_mixed_precision_gemm_prepacked(*args, iters, ctypes.byref(ms)), "time")
        return ms.value

    iters = int(min(2000, max(50, 30.0 / max(plain(20), 1e-4))))
    for _ in range(10):
        plain(iters)
    res = {"plain": [], "prepacked": []}
    for _ in range(ROUNDS):
        res["plain"].append(plain(iters))
        res["prepacked"].append(packed(iters))
    flops = 2.0 * m * n * k
    line = "  ".join(f"{name}: {statistics.median(v) * 1e3:7.1f} us ({flops / statistics.median(v) / 1e9 / 2500:.3f})" for name, v in res.items())
    print(f"mixed {m}x{k}x{n} int{bits}{'+zp' if zp else '   '} {str(dtype)[6:]:9s} tile_nt={pre.tile_nt}: {line}   "
          f"prepacked/plain = {statistics.median(res['prepacked']) / statistics.median(res['plain']):.3f}", flush=True)


if __name__ == "__main__":
    case(1024, 4096, 11008, torch.float16, 4, False)
    case(1024, 4096, 11008, torch.bfloat16, 4, False)
    case(1024, 4096, 11008, torch.float16, 4, True)
    case(1024, 4096, 11008, torch.float16, 8, False)
    case(4096, 8192, 4096, torch.float16, 4, False)
    case(8192, 8192, 8192, torch.float16, 4, False)
    case(512, 4096, 11008, torch.float16, 4, False)
