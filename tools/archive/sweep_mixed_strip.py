"""Where does the column-strip mixed GEMM (gemm_mixed_strip.hip) beat the LDS-tiled one?  Interleaved timings over a grid of shapes,
each kernel at its own automatic tile width; `--zero` repeats two shapes on all-zero activations (the chip then holds its full
clock: what the loop's SCHEDULE costs, without the power limit).  -> stdout (profiles/r04/mixed_strip_sweep.txt)"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402


def case(m, k, n, dtype=torch.float16, bits=4, zero=False, rounds=5):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    if zero:
        x.zero_()
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, bits, 8 if bits == 4 else 128, 128)
    ms = ctypes.c_float()

    def run(kernel, iters):
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms))), "time")
        return ms.value

    iters = int(min(2000, max(20, 20.0 / max(run(1, 10), 1e-4))))
    res = {1: [], 2: []}
    for kern in (1, 2):
        run(kern, iters)
    for _ in range(rounds):
        for kern in (1, 2):
            res[kern].append(run(kern, iters))
    t1, t2 = statistics.median(res[1]), statistics.median(res[2])
    fl = 2.0 * m * n * k
    print(f"{m:5d} x {k:5d} x {n:5d} int{bits} {str(dtype)[6:]:8s}{' ZERO X' if zero else ''}: tiled {t1 * 1e3:7.1f} us ({fl / t1 / 1e9 / 2500:.3f})  strip {t2 * 1e3:7.1f} us "
          f"({fl / t2 / 1e9 / 2500:.3f})  strip/tiled {t2 / t1:.3f}", flush=True)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)


if __name__ == "__main__":
    if "--zero" in sys.argv:
        for z in (False, True):
            case(4096, 8192, 4096, zero=z)
            case(1024, 4096, 11008, zero=z)
        sys.exit(0)
    for (k, n) in ((4096, 4096), (4096, 11008), (8192, 4096), (11008, 4096), (8192, 8192), (4096, 28672)):
        for m in (384, 512, 768, 1024, 1536, 2048, 4096, 8192):
            case(m, k, n)
    for dt, bits in ((torch.bfloat16, 4), (torch.float16, 8)):
        for (m, k, n) in ((1024, 4096, 11008), (4096, 8192, 4096), (512, 4096, 11008), (2048, 4096, 4096), (4096, 4096, 11008)):
            case(m, k, n, dt, bits)
