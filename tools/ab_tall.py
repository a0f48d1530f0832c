"""Interleaved A/B of the mixed GEMM's tile shapes: the dispatcher's 256-row tile against the 512 x 128 tile (tuning key 1 = 5),
plain and pre-packed weights (same process, same data, rounds interleaved: cdna_hip_programming.md rule 24).
usage: python tools/ab_tall.py [rounds]"""
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

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def case(m, k, n, dtype, bits, zp):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    wzp = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda") if zp else None
    bias = 0 if zp else (8 if bits == 4 else 128)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, wzp, bits, bias, 128)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    pre_auto = prepack_mixed_weights(wq, bits, m_hint=m, per_group_zero_points=zp)
    pre_128 = prepack_mixed_weights(wq, bits, tile_nt=2)
    ms = ctypes.c_float()
    args_auto, _k1 = _prepacked_args(out, x, pre_auto, ws, wzp, bias, 128)
    args_128, _k2 = _prepacked_args(out, x, pre_128, ws, wzp, bias, 128)

    def plain(iters, nt):
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, wzp, md, (iters, ctypes.byref(ms))), "time")
        return ms.value

    def packed(iters, nt, args):
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
        _C.check(_C.load().conch_time_mixed_precision_gemm_prepacked(*args, iters, ctypes.byref(ms)), "time")
        return ms.value

    legs = {
        f"auto(nt={pre_auto.tile_nt})": lambda it: plain(it, pre_auto.tile_nt),
        "tall": lambda it: plain(it, 5),
        "auto+pre": lambda it: packed(it, pre_auto.tile_nt, args_auto),
        "tall+pre": lambda it: packed(it, 5, args_128),
    }
    iters = int(min(2000, max(30, 30.0 / max(plain(20, 0), 1e-4))))
    for f in legs.values():
        for _ in range(3):
            f(iters)
    res = {name: [] for name in legs}
    for _ in range(ROUNDS):
        for name, f in legs.items():
            res[name].append(f(iters))
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    flops = 2.0 * m * n * k
    med = {name: statistics.median(v) for name, v in res.items()}
    line = "  ".join(f"{name}: {t * 1e3:7.1f} us ({flops / t / 1e9 / 2500:.3f})" for name, t in med.items())
    base = med[next(iter(med))]
    print(f"mixed {m}x{k}x{n} int{bits}{'+zp' if zp else '   '} {str(dtype)[6:]:9s}: {line}   tall/auto = {med['tall'] / base:.3f}, "
          f"tall+pre/auto+pre = {med['tall+pre'] / med['auto+pre']:.3f}", flush=True)


if __name__ == "__main__":
    case(4096, 8192, 4096, torch.float16, 4, False)
    case(4096, 8192, 4096, torch.bfloat16, 4, False)
    case(4096, 8192, 4096, torch.float16, 4, True)
    case(4096, 8192, 4096, torch.float16, 8, False)
    case(8192, 8192, 8192, torch.float16, 4, False)
    case(4096, 4096, 11008, torch.float16, 4, False)
    case(2048, 4096, 11008, torch.float16, 4, False)
    case(1024, 4096, 11008, torch.float16, 4, False)
    case(2048, 4096, 4096, torch.float16, 4, False)
    case(8192, 4096, 28672, torch.float16, 4, False)
    case(4096, 11008, 4096, torch.float16, 4, False)
    case(1536, 4096, 4096, torch.float16, 4, False)
