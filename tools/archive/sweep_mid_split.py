"""The 128x128-tile scaled GEMM with K split over 1..8 workgroups per tile (CONCH_TUNE_MID_SPLITK) against the dispatcher's pick and the
split-K skinny kernel, on shapes with few tiles and a long K.  Device time of the whole op (both launches of a split form), weights
re-read back to back.  -> stdout (profiles/r04/mid_splitk_sweep.txt)"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

SHAPES = [(256, 11008, 4096), (512, 11008, 4096), (384, 11008, 4096), (256, 13824, 5120), (512, 13824, 5120), (256, 8192, 8192), (512, 8192, 8192),
          (256, 28672, 8192), (512, 28672, 8192), (1024, 28672, 8192), (256, 4096, 4096), (512, 4096, 4096), (1024, 11008, 4096), (768, 8192, 4096),
          (1024, 8192, 1024), (2048, 8192, 1024), (128, 11008, 4096), (192, 11008, 4096)]


def main():
    dt = torch.float8_e4m3fn if (len(sys.argv) < 2 or sys.argv[1] != "int8") else torch.int8
    for m, k, n in SHAPES:
        torch.manual_seed(0)
        if dt == torch.int8:
            a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device="cuda")
            bt = torch.randint(-128, 128, (n, k), dtype=torch.int8, device="cuda")
        else:
            a, bt = (0.25 * torch.rand((m, k), device="cuda")).to(dt), (0.25 * torch.rand((n, k), device="cuda")).to(dt)
        sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
        ms = ctypes.c_float()

        def run(variant, split, stages=0, iters=100):
            _C.set_gemm_variant(variant)
            _C.set_tuning(_C.TUNE_MID_SPLITK, split)
            _C.set_tuning(_C.TUNE_MID_STAGES, stages)
            try:
                rc = kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms)))
                if rc:
                    return float("nan")
                return ms.value * 1e3
            finally:
                _C.set_gemm_variant(0)
                _C.set_tuning(_C.TUNE_MID_SPLITK, 0)
                _C.set_tuning(_C.TUNE_MID_STAGES, 0)

        legs = [("auto", 0, 0, 0), ("skinny", 4, 0, 0), ("tiled", _C.VARIANT_MFMA_PINGPONG2, 0, 0), ("mid x1", 6, 1, 0)]
        legs += [(f"mid x{s}", 6, s, 0) for s in (2, 3, 4, 6, 8)] + [("mid x4 2st", 6, 4, 2), ("mid x8 4st", 6, 8, 4)]
        res = {}
        for name, v, s, st in legs:
            if name == "skinny" and m > 256:
                continue
            for _ in range(2):
                run(v, s, st, 30)
            res[name] = statistics.median(run(v, s, st) for _ in range(5))
        best = min((t, nm) for nm, t in res.items() if nm != "auto" and t == t)
        print(f"{str(dt)[6:]:13s} {m:5d}x{k:5d}x{n:5d}: " + "  ".join(f"{nm} {t:6.1f}" for nm, t in res.items()) + f"   best {best[1]} ({res['auto'] / best[0]:.2f}x auto)", flush=True)


if __name__ == "__main__":
    main()
