"""The split-K skinny-M scaled GEMM with 1..n passes per slice (CONCH_TUNE_SKINNY_CHUNKS) on decode batches with a long K, against the
dispatcher's pick.  Device time of the whole op, back-to-back launches.  -> stdout (profiles/r04/skinny_chunks_sweep.txt)"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

SHAPES = [(16, 28672, 8192), (32, 28672, 8192), (64, 28672, 8192), (128, 28672, 8192), (16, 8192, 28672), (64, 8192, 28672), (128, 8192, 28672),
          (16, 11008, 4096), (64, 11008, 4096), (128, 11008, 4096), (16, 13824, 5120), (64, 13824, 5120), (64, 5120, 13824), (32, 8192, 8192),
          (64, 8192, 8192), (128, 8192, 8192), (8, 4096, 4096), (128, 4096, 4096), (64, 4096, 11008), (16, 4096, 28672)]


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

        def run(variant, chunks, mode, iters=100):
            _C.set_gemm_variant(variant)
            _C.set_tuning(_C.TUNE_SKINNY_CHUNKS, chunks)
            _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
            try:
                rc = kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms)))
                return float("nan") if rc else ms.value * 1e3
            finally:
                _C.set_gemm_variant(0)
                _C.set_tuning(_C.TUNE_SKINNY_CHUNKS, 0)
                _C.set_tuning(_C.TUNE_SKINNY_MODE, 0)

        legs = [("auto", 0, 0, 0), ("sk auto", 4, 0, 0)]
        for mode in (1, 2, 3):
            legs += [(f"m{mode}c{c}", 4, c, mode) for c in (1, 2, 4, 7, 14)]
        res = {}
        for name, v, c, mode in legs:
            for _ in range(2):
                run(v, c, mode, 20)
            res[name] = statistics.median(run(v, c, mode) for _ in range(5))
        best = min((t, nm) for nm, t in res.items() if nm != "auto" and t == t)
        print(f"{str(dt)[6:]:13s} {m:4d}x{k:5d}x{n:5d}: " + " ".join(f"{nm} {t:5.1f}" for nm, t in res.items()) + f"  best {best[1]} ({res['auto'] / best[0]:.2f}x auto)", flush=True)


if __name__ == "__main__":
    main()
