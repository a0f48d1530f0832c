"""Interleaved A/B of tuning-key settings on the mixed-precision GEMM in ONE process (development aid).

    python tools/ab_mixed_tuning.py <workload> <key>=<v1,v2,...> [rounds] [--nt N] [--pre]

workload: c4 | readme | sq8k | MxKxN;  key: a tuning-key number (6 = epilogue, 1 = tile shape, 9 = split-K).
"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

SHAPES = {"c4": (1024, 4096, 11008), "readme": (4096, 8192, 4096), "sq8k": (8192, 8192, 8192)}


def main():
    wl = sys.argv[1]
    key, vals = sys.argv[2].split("=")
    key, vals = int(key), [int(v) for v in vals.split(",")]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 9
    m, k, n = SHAPES[wl] if wl in SHAPES else (int(v) for v in wl.split("x"))
    if "--nt" in sys.argv:
        _C.set_tuning(1, int(sys.argv[sys.argv.index("--nt") + 1]))
    torch.manual_seed(0)
    dtype, bits = torch.float16, 4
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, bits, 8, 128)
    ms = ctypes.c_float()

    def run(iters):
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms))), "time")
        return ms.value

    iters = int(min(2000, max(30, 30.0 / max(run(20), 1e-4))))
    for _ in range(10):
        run(iters)
    res = {v: [] for v in vals}
    for _ in range(rounds):
        for v in vals:
            _C.set_tuning(key, v)
            res[v].append(run(iters))
    _C.set_tuning(key, 0)
    flops = 2.0 * m * n * k
    for v, r in res.items():
        med = statistics.median(r)
        print(f"mixed {wl:8s} key{key}={v}: median {med * 1e3:8.2f} us  min {min(r) * 1e3:8.2f} us  {flops / med / 1e9:7.1f} TFLOP/s "
              f"({flops / med / 1e9 / 2500:.3f} of 2.5 PF)", flush=True)


if __name__ == "__main__":
    main()
