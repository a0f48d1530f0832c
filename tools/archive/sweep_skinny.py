"""Skinny-M scaled GEMM: time every (CONCH_TUNE_SKINNY_MODE, CONCH_TUNE_SKINNY_GATHER) pair against the automatic choice on decode
shapes (development aid; the table behind the rules in gemm_skinny.hip launch_splitk).

    python tools/sweep_skinny.py [MxKxN[:int8|fp8] ...]
"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

DEFAULT = ["1x4096x4096:int8", "16x4096x4096:int8", "32x4096x4096:int8", "64x4096x4096:int8", "96x4096x4096:int8", "128x4096x4096:int8",
           "128x4096x4096:fp8", "16x4096x11008:int8", "32x4096x11008:fp8", "64x4096x11008:fp8", "128x4096x11008:int8", "1x8192x8192:fp8",
           "16x8192x8192:int8", "32x8192x8192:fp8", "64x8192x8192:int8", "128x8192x8192:fp8", "16x11008x4096:fp8", "128x11008x4096:int8",
           "16x4096x28672:fp8", "128x8192x4096:fp8", "8x2048x2048:int8", "128x2048x2048:int8"]


def main():
    shapes = [a for a in sys.argv[1:] if "x" in a] or DEFAULT
    ms = ctypes.c_float()
    for wl in shapes:
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
        out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)

        def run(iters):
            _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
            return ms.value

        configs = [(0, 0)] + [(mode, g) for mode in (1, 2, 3) for g in (1, 2)]
        ref = None
        res = {}
        for _ in range(3):
            for mode, g in configs:
                _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
                _C.set_tuning(_C.TUNE_SKINNY_GATHER, g)
                run(200)
                res.setdefault((mode, g), []).append(run(1000) * 1e3)
                if ref is None:
                    ref = out.clone()
                elif (dt or "int8") == "int8" and not torch.equal(ref, out):  # fp8: the slice size changes the fp32 summation order
                    print(f"{wl}: mode {mode} gather {g} DIFFERS from the automatic choice")
        _C.set_tuning(_C.TUNE_SKINNY_MODE, 0)
        _C.set_tuning(_C.TUNE_SKINNY_GATHER, 0)
        med = {c: statistics.median(v) for c, v in res.items()}
        best = min((c for c in med if c != (0, 0)), key=lambda c: med[c])
        cells = "  ".join(f"m{mode}g{g} {med[(mode, g)]:6.2f}" for mode, g in configs[1:])
        print(f"{wl:22s} auto {med[(0, 0)]:6.2f} us | {cells} | best m{best[0]}g{best[1]}  auto/best {med[(0, 0)] / med[best]:.3f}", flush=True)


if __name__ == "__main__":
    main()
