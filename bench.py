"""Headline benchmark: scaled GEMM fp8(e4m3fn) x fp8 -> bf16, M=4096 K=4096 N=11008 (BASELINE.json C3).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|c1|c4readme|c4decode|midm]

One "step" = one `scaled_gemm` call (public op: output allocation + fused GEMM/scale/cast kernel) on
synthetic inputs that are already resident in HBM, built with the reference benchmark's recipe
(benchmarks/scaled_gemm_benchmark.py:198-210).  Prints ONE JSON line on rank 0.

  value      effective TFLOP/s = 2*M*N*K * steps * n_gpus / wall time of the timed region
             (barrier + synchronize on both sides, max over ranks).
  roofline   dominant kernel (the MFMA GEMM): algorithmic FLOPs per launch / average launch
             duration measured with HIP events recorded on the launch stream inside the library
             (conch_time_scaled_gemm), against the dense fp8 MFMA peak of
             /opt/skills/guides/MI355X_MICROARCH.md (~5 PFLOP/s).  `traffic` is the PMC-measured HBM
             bytes per launch taken from profiles/ (separate rocprofv3 --pmc passes), or null.
  cpu_baseline  the CPU oracle (oracle.scaled_gemm_ref: the reference's PyTorch-only path restated)
             timed on this host on the same workload, rank 0, N=1 only.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): the default is data-parallel over the
token dimension M -- every rank multiplies its own 4096-token batch by replicated weights, no
data-path collective ("scaling": "weak").  The N-sharded C5 mode with its RCCL all-gather
(`--workload c5`, "scaling": "strong") is additionally timed and reported under "nshard_c5".
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kgemm  # noqa: E402
from conch_amd.ops.quantization.gemm import (  # noqa: E402
    create_mixed_precision_metadata,
    create_scaled_metadata,
    mixed_precision_gemm,
    scaled_gemm,
)

FP8_PEAK_TFLOPS = 5000.0  # dense fp8 MFMA, MI355X_MICROARCH.md "Peak FP8 MFMA ~5 PF dense"
F16_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA
I8_PEAK_TFLOPS = 5000.0  # i8 = 2x bf16 per clock
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (kind, M, K, N)
    "c1": ("quant_int8", 4096, 0, 4096),
    "c3": ("scaled_fp8", 4096, 4096, 11008),
    "c2": ("scaled_int8", 128, 4096, 4096),
    "c4": ("mixed_int4", 1024, 4096, 11008),
    "c4readme": ("mixed_int4", 4096, 8192, 4096),
    "c5": ("scaled_fp8", 8192, 8192, 28672),
    # not BASELINE configs: profiling handles for the small-M kernels (decode-batch mixed kernel, 128x128-tile scaled kernel)
    "c4decode": ("mixed_int4", 16, 4096, 11008),
    "midm": ("scaled_int8", 256, 4096, 11008),
}
CLOCK_RAMP_S = 0.15  # the GPU needs ~20 ms of load to leave its idle clocks; ramp before the W warm-up steps


def dist_env() -> tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def make_scaled(kind: str, m: int, k: int, n: int, device: torch.device, seed: int):
    """Inputs as in benchmarks/scaled_gemm_benchmark.py:198-210 (vector scales, no bias)."""
    torch.manual_seed(seed)
    if kind == "scaled_int8":
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device=device)
        b = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=device).T
    else:
        a = (0.25 * torch.rand((m, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fn)
        b = (0.25 * torch.rand((n, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fn).T
    sa = 0.25 * torch.rand((m, 1), dtype=torch.float32, device=device)
    sb = 0.25 * torch.rand((n, 1), dtype=torch.float32, device=device)
    return a, b, sa, sb


def make_mixed(m: int, k: int, n: int, device: torch.device, seed: int):
    """Inputs as in benchmarks/mixed_precision_gemm_benchmark.py:205-214 (uint4b8, group 128)."""
    from conch_amd.third_party.vllm.quant_utils import pack_rows, quantize_weights
    from conch_amd.third_party.vllm.scalar_type import scalar_types

    torch.manual_seed(seed)
    a = (10 * (torch.rand((m, k), dtype=torch.float32) - 0.3)).to(torch.float16)
    b = (10 * (torch.rand((k, n), dtype=torch.float32) - 0.3)).to(torch.float16)
    wt = scalar_types.uint4b8
    w_ref, w_q, w_s, _ = quantize_weights(b, wt, 128, zero_points=False)
    packed = pack_rows(w_q, wt.size_bits, *w_q.shape)
    return a.to(device), packed.to(device), w_s.to(device), w_ref, wt


def kernel_avg_ms_scaled(a, b, sa, sb, out, iters: int) -> float:
    """Average GEMM launch duration: HIP events on the launch stream, inside the C library."""
    md = create_scaled_metadata(a, b, sa, sb, out.dtype)
    ms = ctypes.c_float()
    extra = (iters, ctypes.byref(ms))
    _C.check(kgemm._scaled_gemm_call("conch_time_scaled_gemm", out, a, b, sa, sb, md, None, extra), "time")
    return ms.value


def kernel_avg_ms_mixed(x, packed, w_s, out, wt, iters: int) -> float:
    md = create_mixed_precision_metadata(x, packed, w_s, None, wt.size_bits, wt.bias, 128)
    ms = ctypes.c_float()
    extra = (iters, ctypes.byref(ms))
    _C.check(kgemm._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, packed, w_s, None, md, extra), "time")
    return ms.value


def cpu_baseline_scaled(kind: str, m: int, k: int, n: int) -> dict:
    """Time the CPU oracle on the same workload (bounded: a few repetitions of the full problem)."""
    import oracle

    a, b, sa, sb = make_scaled(kind, m, k, n, torch.device("cpu"), 0)
    oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16)  # warm-up (thread pool, allocator)
    times = []
    budget_end = time.perf_counter() + 20.0
    while len(times) < 3 or (len(times) < 10 and time.perf_counter() < budget_end):
        t0 = time.perf_counter()
        oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > budget_end and len(times) >= 3:
            break
    med = sorted(times)[len(times) // 2]
    return {
        "value": round(2.0 * m * n * k / med / 1e12, 4),
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"full {m}x{k}x{n} problem through oracle.scaled_gemm_ref (fp32 torch matmul + scales + cast), "
        f"median of {len(times)} runs, {med * 1e3:.1f} ms each, host has {os.cpu_count()} logical CPUs",
    }


def cpu_baseline_mixed(a, w_ref, m: int, k: int, n: int) -> dict:
    """The reference's mixed-precision CPU path is `torch.matmul(a_fp16, w_ref_fp16)`; half-precision
    matmul is very slow on CPUs (tens of seconds for C4), so a BOUNDED sample of activation rows is
    timed (calibrated to ~10 s of CPU work) and the rate is reported for that sample."""
    import oracle

    a = a.cpu()
    t0 = time.perf_counter()
    oracle.mixed_precision_gemm_ref(a[:8], w_ref)
    probe = max(time.perf_counter() - t0, 1e-4)
    rows = int(min(m, max(8, 8 * (3.0 / probe))))  # ~3 s per repetition
    rows -= rows % 8
    sample = a[:rows].contiguous()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.mixed_precision_gemm_ref(sample, w_ref)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[1]
    return {
        "value": round(2.0 * rows * n * k / med / 1e12, 5),
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"first {rows} of {m} activation rows: torch.matmul(a[:{rows}], w_ref) in fp16, {rows}x{k}x{n} "
        f"(weights dequantised once, not timed), median of 3 runs, {med * 1e3:.1f} ms each, "
        f"host has {os.cpu_count()} logical CPUs",
    }


def load_traffic(workload: str) -> float | None:
    """HBM bytes per launch measured by rocprofv3 --pmc (see profiles/README.md), if recorded."""
    f = ROOT / "profiles" / "traffic.json"
    if f.exists():
        try:
            return json.loads(f.read_text()).get(workload, {}).get("hbm_bytes_per_launch")
        except (ValueError, AttributeError):
            return None
    return None


def barrier_sync(world: int) -> None:
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def timed_region(fn, steps: int, warmup: int, world: int, device: torch.device) -> float:
    """Clock ramp, W warm-up calls, then exactly `steps` calls between barrier+synchronize; max over ranks (s)."""
    t_end = time.perf_counter() + CLOCK_RAMP_S
    while time.perf_counter() < t_end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier_sync(world)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


def nshard_c5(world: int, rank: int, device: torch.device, steps: int, warmup: int) -> dict:
    """C5: M=8192 K=8192 N=28672 with N sharded over the ranks, C all-gathered over xGMI (RCCL)."""
    from conch_amd.distributed import NShardedScaledGemm

    _, m, k, n = WORKLOADS["c5"]
    torch.manual_seed(0)
    a = (0.25 * torch.rand((m, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), dtype=torch.float32, device=device)
    n_loc = n // world
    torch.manual_seed(1 + rank)
    bt_loc = (0.25 * torch.rand((n_loc, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fn)
    sb_loc = 0.25 * torch.rand((n_loc, 1), dtype=torch.float32, device=device)
    op = NShardedScaledGemm(m, n, torch.bfloat16, device)
    t_gemm = timed_region(lambda: op.local_gemm(a, bt_loc.T, sa, sb_loc), steps, warmup, world, device)
    t_full = timed_region(lambda: op(a, bt_loc.T, sa, sb_loc), steps, warmup, world, device)
    flops = 2.0 * m * n * k
    return {
        "workload": f"scaled_gemm fp8 {m}x{k}x{n}, N/{world} columns per rank",
        "scaling": "strong",
        "tflops_gemm_only": round(flops * steps / t_gemm / 1e12, 2),
        "tflops_gemm_plus_allgather": round(flops * steps / t_full / 1e12, 2),
        "ms_gemm_only": round(t_gemm / steps * 1e3, 4),
        "ms_gemm_plus_allgather": round(t_full / steps * 1e3, 4),
    }


def run_quant_bench(args, tokens: int, hidden: int, device: torch.device, world: int, rank: int) -> None:
    """C1: static int8 quantisation of a 4096x4096 fp16 tensor (HBM-bound feeder)."""
    from conch_amd.ops.quantization.int8 import scaled_int8_quant

    torch.manual_seed(rank)
    x = torch.rand(tokens, hidden, dtype=torch.float16, device=device) * 1000
    scale = torch.tensor([2.1], dtype=torch.float32, device=device)
    elapsed = timed_region(lambda: scaled_int8_quant(x, scale), args.steps, args.warmup, world, device)
    bytes_alg = tokens * hidden * 3
    # kernel-only time: events on the current stream around back-to-back launches into a reused output
    from conch_amd.ops.quantization.int8 import static_scaled_int8_quant

    out = torch.empty_like(x, dtype=torch.int8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        static_scaled_int8_quant(out, x, scale)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 100
    result = {
        "metric": "GB/s, static_scaled_int8_quant 4096x4096 fp16 (bytes = 2 in + 1 out per element)",
        "value": round(bytes_alg * args.steps * world / elapsed / 1e9, 1), "unit": "GB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"static_scaled_int8_quant fp16 [{tokens}x{hidden}], scale 2.1"},
        "roofline": {"bound": "hbm", "achieved": round(bytes_alg / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(bytes_alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": load_traffic("c1"), "kernel_avg_ms": round(k_ms, 5), "algorithmic_bytes": bytes_alg},
    }
    if world == 1 and not args.no_cpu_baseline:
        import oracle

        xc, sc = x.cpu(), scale.cpu()
        oracle.scaled_int8_quant_ref(xc, sc)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            oracle.scaled_int8_quant_ref(xc, sc)
            times.append(time.perf_counter() - t0)
        med = sorted(times)[2]
        result["cpu_baseline"] = {"value": round(bytes_alg / med / 1e9, 2), "unit": "GB/s", "cores": torch.get_num_threads(),
                                  "kind": "port", "sample": f"full tensor, median of 5 runs, {med * 1e3:.1f} ms each"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="dry-run aid: put every rank on cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    rank, local_rank, world = dist_env()
    if args.gpus > 1 and world == 1:
        print(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`",
              file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path is the only path)"
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(args.backend)

    kind, m, k, n = WORKLOADS[args.workload]
    result: dict = {}
    if args.workload == "c5":
        res = nshard_c5(world, rank, device, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({
                "metric": "effective TFLOP/s, scaled-GEMM fp8xbf16 8192x8192x28672 N-sharded + all-gather",
                "value": res["tflops_gemm_plus_allgather"], "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": res["ms_gemm_plus_allgather"], "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "fp8_e4m3fn", "data": "synthetic",
                "config": {"workload": res["workload"]}, "detail": res}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    if kind == "quant_int8":
        run_quant_bench(args, m, n, device, world, rank)
        return

    flops = 2.0 * m * n * k
    if kind.startswith("scaled"):
        a, b, sa, sb = make_scaled(kind, m, k, n, device, seed=rank)
        step = lambda: scaled_gemm(a, b, sa, sb, torch.bfloat16)  # noqa: E731
        elapsed = timed_region(step, args.steps, args.warmup, world, device)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=device)
        k_ms = kernel_avg_ms_scaled(a, b, sa, sb, out, max(20, min(args.steps, 200)))
        peak, dtype_name = (FP8_PEAK_TFLOPS, "fp8_e4m3fn") if kind == "scaled_fp8" else (I8_PEAK_TFLOPS, "int8")
        desc = f"scaled_gemm {dtype_name} x {dtype_name} -> bf16, per-row/per-column fp32 scales, M={m} K={k} N={n}"
        bytes_alg = m * k + k * n + 2 * m * n + 4 * (m + n)
    else:
        x, packed, w_s, w_ref, wt = make_mixed(m, k, n, device, seed=rank)
        step = lambda: mixed_precision_gemm(x, packed, w_s, None, wt.size_bits, wt.bias, 128)  # noqa: E731
        elapsed = timed_region(step, args.steps, args.warmup, world, device)
        out = torch.empty((m, n), dtype=torch.float16, device=device)
        k_ms = kernel_avg_ms_mixed(x, packed, w_s, out, wt, max(20, min(args.steps, 200)))
        peak, dtype_name = F16_PEAK_TFLOPS, "fp16"
        desc = f"mixed_precision_gemm uint4b8(g128) x fp16 -> fp16, M={m} K={k} N={n}"
        bytes_alg = 2 * m * k + k * n // 2 + 2 * (k // 128) * n + 2 * m * n

    achieved = flops / (k_ms * 1e-3) / 1e12
    # C2-like shapes are HBM/latency bound: report against the HBM roof instead of the MFMA roof
    ai = flops / bytes_alg
    if ai < peak * 1e12 / (HBM_PEAK_GBS * 1e9):
        roofline = {"bound": "hbm", "achieved": round(bytes_alg / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(bytes_alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    else:
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4)}
    roofline["traffic"] = load_traffic(args.workload)
    roofline["kernel_avg_ms"] = round(k_ms, 5)
    roofline["algorithmic_bytes"] = bytes_alg
    roofline["algorithmic_flops"] = flops

    result = {
        "metric": "effective TFLOP/s + % MFMA roofline, scaled-GEMM fp8xbf16 4096x4096x11008"
        if args.workload == "c3" else f"effective TFLOP/s, {desc}",
        "value": round(flops * args.steps * world / elapsed / 1e12, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic",
        "config": {
            "workload": desc,
            "parallelism": "single GPU" if world == 1 else f"dp{world} over tokens (M={m} per rank, weights replicated)",
            "gemm_variant": "auto",
            "clock_ramp_s": CLOCK_RAMP_S,
        },
        "roofline": roofline,
    }
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        if kind.startswith("scaled"):
            result["cpu_baseline"] = cpu_baseline_scaled(kind, m, k, n)
        else:
            result["cpu_baseline"] = cpu_baseline_mixed(x, w_ref, m, k, n)
    if world > 1 and args.workload == "c3":
        try:
            result["nshard_c5"] = nshard_c5(world, rank, device, max(10, args.steps // 10), 3)
        except Exception as exc:  # noqa: BLE001 - the headline line must still be printed
            result["nshard_c5"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
