"""Interleaved A/B of the two mixed-GEMM tile kernels in ONE process: the LDS-tiled kernel (gemm_mixed.hip, CONCH_TUNE_MIXED_KERNEL = 1)
against the column-strip kernel (gemm_mixed_strip.hip, = 2), per tile width.  usage: python tools/ab_mixed_strip.py [rounds]"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def case(m, k, n, dtype, bits, zp, widths=(0,)):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    wzp = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda") if zp else None
    bias = 0 if zp else (8 if bits == 4 else 128)
    outs = {}
    md = create_mixed_precision_metadata(x, wq, ws, wzp, bits, bias, 128)
    ms = ctypes.c_float()
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1)

    def run(kernel, nt, iters):
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
        out = outs.setdefault((kernel, nt), torch.empty((m, n), dtype=dtype, device="cuda"))
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, wzp, md, (iters, ctypes.byref(ms))), "time")
        return ms.value

    legs = [(1, 0)] + [(2, w) for w in widths]
    iters = int(min(2000, max(30, 30.0 / max(run(1, 0, 20), 1e-4))))
    for leg in legs:
        for _ in range(3):
            run(*leg, iters)
    res = {leg: [] for leg in legs}
    for _ in range(ROUNDS):
        for leg in legs:
            res[leg].append(run(*leg, iters))
    flops = 2.0 * m * n * k
    base = statistics.median(res[(1, 0)])
    same = all(torch.equal(outs[(1, 0)], outs[leg]) for leg in legs[1:])
    line = f"mixed {m}x{k}x{n} int{bits}{'+zp' if zp else '   '} {str(dtype)[6:]:9s}: tiled {base * 1e3:7.1f} us ({flops / base / 1e9 / 2500:.3f})"
    for leg in legs[1:]:
        t = statistics.median(res[leg])
        line += f" | strip nt={leg[1]}: {t * 1e3:7.1f} us ({flops / t / 1e9 / 2500:.3f}) x{t / base:.3f}"
    print(line + f"  bits {'equal' if same else 'DIFFER'}", flush=True)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)


if __name__ == "__main__":
    case(1024, 4096, 11008, torch.float16, 4, False, (0, 4, 3, 2))
    case(4096, 8192, 4096, torch.float16, 4, False, (0, 4, 3))
    case(1024, 4096, 11008, torch.bfloat16, 4, False, (0,))
    case(4096, 8192, 4096, torch.bfloat16, 4, False, (0,))
    case(1024, 4096, 11008, torch.float16, 8, False, (0,))
    case(4096, 8192, 4096, torch.float16, 8, False, (0,))
    case(1024, 4096, 11008, torch.float16, 4, True, (0,))
    case(4096, 8192, 4096, torch.float16, 4, True, (0, 4))
    case(8192, 8192, 8192, torch.float16, 4, False, (0,))
    case(512, 4096, 11008, torch.float16, 4, False, (0, 2))
    case(2048, 4096, 4096, torch.float16, 4, False, (0, 4, 3, 2))
