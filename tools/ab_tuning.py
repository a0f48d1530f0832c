"""Interleaved A/B of tuning-key settings in ONE process (cdna_hip_programming.md rule 24): development aid.

    python tools/ab_tuning.py <workload> <key>=<v1,v2,...> [rounds] [--variant V]

workload: c3 | c3i8 | c2 | c2fp8 | c5shard | sq8k | MxKxN[:int8|fp8]     key: variant | nt | nosplitk | skinny | sched
Keeps the chip loaded: every round times `iters` back-to-back launches of each setting in turn; prints median / min.
"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

KEYS = {"variant": 0, "nt": 1, "nosplitk": 2, "skinny": 3, "sched": 4, "persist": 5, "epi": 6}
SHAPES = {"c3": (4096, 4096, 11008, "fp8"), "c3i8": (4096, 4096, 11008, "int8"), "c2": (128, 4096, 4096, "int8"),
          "c2fp8": (128, 4096, 4096, "fp8"), "c5shard": (8192, 8192, 3584, "fp8"), "sq8k": (8192, 8192, 8192, "fp8")}


def main():
    wl = sys.argv[1]
    key, vals = sys.argv[2].split("=")
    vals = [int(v) for v in vals.split(",")]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 9
    if "--tune-epi2" in sys.argv:  # the row-major epilogue on every leg
        _C.set_tuning(6, 2)
    if "--variant" in sys.argv:
        _C.set_gemm_variant(int(sys.argv[sys.argv.index("--variant") + 1]))
    if wl in SHAPES:
        m, k, n, dt = SHAPES[wl]
    else:
        dims, _, dt = wl.partition(":")
        m, k, n = (int(v) for v in dims.split("x"))
        dt = dt or "fp8"
    torch.manual_seed(0)
    if dt == "int8":
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
        bt = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()

    def run(iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
        return ms.value

    probe = max(run(20), 1e-4)
    iters = int(min(2000, max(50, 30.0 / probe)))  # ~30 ms per sample
    for _ in range(20):  # ramp
        run(iters)
    res = {v: [] for v in vals}
    for _ in range(rounds):
        for v in vals:
            _C.set_tuning(KEYS[key], v)
            res[v].append(run(iters))
    _C.set_tuning(KEYS[key], 0)
    flops = 2.0 * m * n * k
    for v, r in res.items():
        med, mn = statistics.median(r), min(r)
        print(f"{wl:10s} {key}={v}: median {med * 1e3:8.2f} us  min {mn * 1e3:8.2f} us   {flops / (med * 1e-3) / 1e12:7.1f} TFLOP/s "
              f"({flops / (med * 1e-3) / 1e12 / 5000:.3f} of 5 PF)", flush=True)


if __name__ == "__main__":
    main()
