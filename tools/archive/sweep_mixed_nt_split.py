"""mixed_precision_gemm (int4 x fp16) on the LDS-tiled kernel: tile shape (CONCH_TUNE_MIXED_TILE_NT 2..5) x K slices (CONCH_TUNE_MIXED_SPLITK)
against the dispatcher's pick, at 768-2048 rows.  Development aid.  -> stdout"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

SHAPES = [(1024, 4096, 4096), (1024, 8192, 8192), (1024, 28672, 8192), (768, 4096, 4096), (1536, 4096, 4096), (2048, 4096, 4096), (1024, 11008, 4096),
          (1024, 4096, 11008), (4096, 4096, 4096), (1024, 13824, 5120)]


def main():
    for m, k, n in SHAPES:
        torch.manual_seed(0)
        x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
        wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
        ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
        md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        ms = ctypes.c_float()

        def run(nt, split, kernel, iters=30):
            _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
            _C.set_tuning(_C.TUNE_MIXED_SPLITK, split)
            _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
            try:
                rc = kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms)))
                return float("nan") if rc else ms.value * 1e3
            finally:
                _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
                _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
                _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)

        legs = [("auto", 0, 0, 0)] + [(f"nt{nt}x{s}", nt, s, 1) for nt in (2, 3, 4, 5) for s in (1, 2, 4)] + [("strip", 0, 1, 2)]
        res = {}
        for name, nt, s, kern in legs:
            for _ in range(2):
                run(nt, s, kern, 8)
            res[name] = statistics.median(run(nt, s, kern) for _ in range(5))
        best = min((t, nm) for nm, t in res.items() if nm != "auto" and t == t)
        print(f"{m:5d}x{k:5d}x{n:5d}: " + " ".join(f"{nm} {t:6.1f}" for nm, t in res.items()) + f"  best {best[1]} ({res['auto'] / best[0]:.2f}x auto)", flush=True)


if __name__ == "__main__":
    main()
