"""mixed_precision_gemm (int4 x fp16, group 128) at 96..512 rows: the decode-batch kernel against the LDS-tiled kernel with K split over
1..8 workgroups per tile (CONCH_TUNE_MIXED_SPLITK) and the dispatcher's pick.  Device time of the whole op.
-> stdout (profiles/r04/mixed_splitk_sweep.txt)"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

DIMS = [(4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (28672, 8192), (8192, 28672), (5120, 13824), (13824, 5120)]
ROWS = (96, 128, 192, 256, 384, 512)


def main():
    bits = 8 if (len(sys.argv) > 1 and sys.argv[1] == "int8") else 4
    for k, n in DIMS:
        for m in ROWS:
            torch.manual_seed(0)
            x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
            wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
            ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
            md = create_mixed_precision_metadata(x, wq, ws, None, bits, 8 if bits == 4 else 128, 128)
            out = torch.empty((m, n), dtype=torch.float16, device="cuda")
            ms = ctypes.c_float()

            def run(variant, split, iters=50):
                _C.set_gemm_variant(variant)
                _C.set_tuning(_C.TUNE_MIXED_SPLITK, split)
                try:
                    rc = kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms)))
                    return float("nan") if rc else ms.value * 1e3
                finally:
                    _C.set_gemm_variant(0)
                    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)

            legs = [("auto", 0, 0)] + ([("decode", 4, 0)] if m <= 256 else []) + [(f"tiles x{s}", 5, s) for s in (1, 2, 4, 8)]
            res = {}
            for name, v, s in legs:
                for _ in range(2):
                    run(v, s, 10)
                res[name] = statistics.median(run(v, s) for _ in range(5))
            best = min((t, nm) for nm, t in res.items() if nm != "auto" and t == t)
            print(f"int{bits} x fp16 {m:4d}x{k:5d}x{n:5d}: " + "  ".join(f"{nm} {t:6.1f}" for nm, t in res.items()) + f"   best {best[1]} ({res['auto'] / best[0]:.2f}x auto)", flush=True)


if __name__ == "__main__":
    main()
