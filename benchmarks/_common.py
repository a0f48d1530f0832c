"""Shared pieces of the microbenchmark CLIs (the reference's benchmarks/*.py, Triton-free).

Each benchmark does what the reference's does (benchmarks/scaled_gemm_benchmark.py:216-252): a
correctness check of the conch_amd op against a plain-PyTorch baseline running on the same GPU
("Results matched :)" or a WARNING), then `benchmark_it` of both.  The baselines below are the
reference's PyTorch-only formulas executed on the device; they exist for timing comparison only.
"""

from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from conch_amd import envs  # noqa: E402
from conch_amd.platforms import current_platform  # noqa: E402
from conch_amd.utils.benchmark import BenchmarkMetadata, benchmark_it  # noqa: E402

DTYPES = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}


def torch_scaled_gemm(a, b, scale_a, scale_b, out_dtype, bias=None):
    out = torch.matmul(a.to(torch.float32), b.to(torch.float32))
    out = (scale_b.T * (scale_a * out)).to(out_dtype)
    return out if bias is None else out + bias


def torch_int8_quant(x, scale):
    return (x * scale.reciprocal()).clamp(-128, 127).to(torch.int8)


def torch_fp8_quant(x, scale, dtype):
    lim = torch.finfo(dtype).max
    return (x.to(torch.float32) * scale.reciprocal()).clamp(-lim, lim).to(dtype)


def report_match(ok: bool, detail: str = "") -> None:
    if ok:
        print("Results matched :)", file=sys.stderr)
    else:
        print(f"WARNING: results did not match the PyTorch baseline {detail}", file=sys.stderr)


def run_pair(name: str, ours, baseline, params: dict, iteration_time_ms: int, warmup_time_ms: int, csv: bool,
             flops: float | None = None, nbytes: float | None = None, extra: dict | None = None) -> None:
    """`extra`: further comparators {tag: callable}, timed only under CONCH_BENCH_ENABLE_ALL_REF=1 (the reference gates its
    vLLM / bitsandbytes legs the same way, benchmarks/scaled_gemm_benchmark.py:196-214)."""
    md = BenchmarkMetadata(platform=current_platform.name(), params=params)
    extras = []
    if extra and envs.CONCH_BENCH_ENABLE_ALL_REF:
        for tag, fn in extra.items():
            try:
                fn()
            except Exception as exc:  # noqa: BLE001 -- a vendor path this build lacks is reported, not fatal
                print(f"{tag}: unavailable ({str(exc)[:100]})", file=sys.stderr)
                continue
            extras.append(benchmark_it(fn, tag=tag, metadata=md, iteration_time_ms=iteration_time_ms, warmup_time_ms=warmup_time_ms))
    base = benchmark_it(baseline, tag="Baseline (PyTorch on GPU)", metadata=md, iteration_time_ms=iteration_time_ms,
                        warmup_time_ms=warmup_time_ms)
    mine = benchmark_it(ours, tag="conch_amd (HIP)", metadata=md, iteration_time_ms=iteration_time_ms,
                        warmup_time_ms=warmup_time_ms)
    mine.print_parameters(csv=csv)
    mine.print_results(csv=csv)
    base.print_results(csv=csv)
    for r in extras:
        r.print_results(csv=csv)
    if not csv:
        if flops:
            print(f"{name}: {flops / (mine.median_ * 1e-3) / 1e12:.1f} TFLOP/s (median)")
        if nbytes:
            print(f"{name}: {nbytes / (mine.median_ * 1e-3) / 1e9:.1f} GB/s (median)")
        print(f"speedup over the PyTorch baseline: {base.median_ / mine.median_:.2f}x")
