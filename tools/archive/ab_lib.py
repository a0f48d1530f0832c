"""Interleaved A/B of two BUILDS of the library on the mixed GEMM (same process, same data, rounds interleaved).
usage: python tools/ab_lib.py <variant> [rounds]     (variant = name given to `python -m conch_amd._build --variant NAME -D...`)"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402
from conch_amd.ops.quantization.prepack import _prepacked_args, prepack_mixed_weights  # noqa: E402

VARIANT = sys.argv[1]
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 7
BASE = _C.load()
EXP = _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{VARIANT}.so"))


def case(m, k, n, dtype, bits, zp, nt=0):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    wzp = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda") if zp else None
    bias = 0 if zp else (8 if bits == 4 else 128)
    outs = [torch.empty((m, n), dtype=dtype, device="cuda") for _ in range(2)]
    md = create_mixed_precision_metadata(x, wq, ws, wzp, bits, bias, 128)
    pre = prepack_mixed_weights(wq, bits, m_hint=m, per_group_zero_points=zp)
    ms = ctypes.c_float()
    for lib in (BASE, EXP):
        lib.conch_set_tuning(1, nt)

    def plain(lib, out, iters):
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, wzp, md, (iters, ctypes.byref(ms)), lib=lib), "time")
        return ms.value

    def packed(lib, out, iters):
        args, _keep = _prepacked_args(out, x, pre, ws, wzp, bias, 128)
        _C.check(lib.conch_time_mixed_precision_gemm_prepacked(*args, iters, ctypes.byref(ms)), "time")
        return ms.value

    legs = {"base": lambda it: plain(BASE, outs[0], it), VARIANT: lambda it: plain(EXP, outs[1], it),
            "base+pre": lambda it: packed(BASE, outs[0], it), VARIANT + "+pre": lambda it: packed(EXP, outs[1], it)}
    iters = int(min(2000, max(30, 30.0 / max(plain(BASE, outs[0], 20), 1e-4))))
    for f in legs.values():
        for _ in range(3):
            f(iters)
    same = torch.equal(outs[0], outs[1])
    res = {name: [] for name in legs}
    for _ in range(ROUNDS):
        for name, f in legs.items():
            res[name].append(f(iters))
    flops = 2.0 * m * n * k
    med = {name: statistics.median(v) for name, v in res.items()}
    line = "  ".join(f"{name}: {t * 1e3:7.1f} us ({flops / t / 1e9 / 2500:.3f})" for name, t in med.items())
    print(f"mixed {m}x{k}x{n} int{bits}{'+zp' if zp else '   '} {str(dtype)[6:]:9s} nt={pre.tile_nt}: {line}   {VARIANT}/base = "
          f"{med[VARIANT] / med['base']:.3f}, +pre {med[VARIANT + '+pre'] / med['base+pre']:.3f}  bits {'equal' if same else 'DIFFER'}", flush=True)


def scaled(m, k, n, dtype, silu=False):
    from conch_amd.ops.quantization.gemm import create_scaled_metadata

    torch.manual_seed(0)
    if dtype == torch.int8:
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(dtype)
        bt = (0.25 * torch.rand((n, k), device="cuda")).to(dtype)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    bias = torch.rand((n,), device="cuda").to(torch.bfloat16)
    outs = [torch.empty((m, n), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()

    def run(lib, out, iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, bias, (iters, ctypes.byref(ms)), lib=lib), "time")
        return ms.value

    iters = int(min(3000, max(30, 30.0 / max(run(BASE, outs[0], 20), 1e-4))))
    for _ in range(3):
        run(BASE, outs[0], iters), run(EXP, outs[1], iters)
    same = torch.equal(outs[0], outs[1])
    res = {"base": [], VARIANT: []}
    for _ in range(ROUNDS):
        res["base"].append(run(BASE, outs[0], iters))
        res[VARIANT].append(run(EXP, outs[1], iters))
    med = {k_: statistics.median(v) for k_, v in res.items()}
    print(f"scaled {str(dtype)[6:]:13s} {m}x{k}x{n} (+bias): base {med['base'] * 1e3:7.1f} us  {VARIANT} {med[VARIANT] * 1e3:7.1f} us   {VARIANT}/base = "
          f"{med[VARIANT] / med['base']:.3f}  bits {'equal' if same else 'DIFFER'}", flush=True)


if __name__ == "__main__":
    if "--scaled" in sys.argv:
        scaled(4096, 4096, 11008, torch.float8_e4m3fn)
        scaled(4096, 4096, 11008, torch.int8)
        scaled(8192, 8192, 8192, torch.float8_e4m3fn)
        scaled(128, 4096, 4096, torch.int8)
        scaled(512, 4096, 4096, torch.float8_e4m3fn)
        scaled(16, 4096, 11008, torch.int8)
    case(1024, 4096, 11008, torch.float16, 4, False)
    case(4096, 8192, 4096, torch.float16, 4, False)
    case(1024, 4096, 11008, torch.bfloat16, 4, False)
    case(4096, 8192, 4096, torch.bfloat16, 4, False)
    case(4096, 8192, 4096, torch.float16, 8, False)
    case(8192, 8192, 8192, torch.float16, 4, False)
    case(4096, 8192, 4096, torch.float16, 4, True)
    case(512, 4096, 11008, torch.float16, 4, False)
