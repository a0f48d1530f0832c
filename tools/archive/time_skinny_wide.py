"""fp8 decode shapes on a very wide N (8192 x 28672): every (launch form, chunks, fetch order) of the split-K skinny kernel against the automatic pick.
usage: python tools/archive/time_skinny_wide.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

ms = ctypes.c_float()
LIB = _C.load()
for m in (16, 64):
    k, n = 8192, 28672
    torch.manual_seed(0)
    count = 3
    a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((count, n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    b = bt[0].T
    md = create_scaled_metadata(a, b, sa, sb, torch.bfloat16)

    def run(iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, b, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
        return ms.value * 1e3

    for regime, rot in (("warm", 1), ("cold", count)):
        LIB.conch_set_timing_rotation(rot, n * k)
        res = {}
        for mode in (0, 1, 2, 3):
            for chunks in (0, 1, 2, 4, 8):
                for gather in (0, 1, 2):
                    _C.set_gemm_variant(4 if (mode, chunks, gather) != (0, 0, 0) else 0)
                    _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
                    _C.set_tuning(_C.TUNE_SKINNY_CHUNKS, chunks)
                    _C.set_tuning(_C.TUNE_SKINNY_GATHER, gather)
                    try:
                        run(20)
                        res[(mode, chunks, gather)] = sorted(run(100) for _ in range(3))[1]
                    except Exception:  # noqa: BLE001
                        pass
        _C.set_gemm_variant(0)
        for key in (_C.TUNE_SKINNY_MODE, _C.TUNE_SKINNY_CHUNKS, _C.TUNE_SKINNY_GATHER):
            _C.set_tuning(key, 0)
        best = sorted(res.items(), key=lambda kv: kv[1])[:6]
        print(f"{regime} fp8 {m}x{k}x{n}: auto {res[(0, 0, 0)]:.1f} us; best (mode, chunks, gather): " + "  ".join(f"{kk} {v:.1f}" for kk, v in best), flush=True)
    LIB.conch_set_timing_rotation(1, 0)
