"""Benchmark utilities (Triton-free equivalent of conch/utils/benchmark.py).

`benchmark_it` keeps the reference's protocol (:82-112): a warm-up budget, then repeated timed
runs for a fixed time budget, per-run device timings, min/max/mean/median reported.  The reference
delegates to `triton.testing.do_bench`; here the timing is done with HIP events on torch's current
stream and the L2 / Infinity Cache are flushed between runs by overwriting a buffer larger than
the 256 MiB Infinity Cache, which is what do_bench's cache clearing does on a smaller scale.
"""

from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass
from typing import Any

import torch

_FLUSH_BYTES = 512 * 1024 * 1024


@dataclass
class BenchmarkMetadata:
    platform: str
    params: dict[str, Any]


@dataclass
class BenchmarkResult:
    tag: str
    metadata: BenchmarkMetadata
    num_iterations: int
    min_: float
    max_: float
    mean_: float
    median_: float

    def print_parameters(self, csv: bool = False) -> None:
        if csv:
            print(f"tag,platform,num_iterations,{','.join(self.metadata.params.keys())},runtime_ms")
            return
        print(f"Parameters: {self.metadata.params}")

    def csv_print(self) -> None:
        values = ",".join(str(v) for v in self.metadata.params.values())
        print(f"{self.tag},{self.metadata.platform},{self.num_iterations},{values},{self.median_:.3f}")

    def pretty_print(self) -> None:
        fmt = "{:.3f} ms".format
        print(
            f"{self.tag}: num_iterations={self.num_iterations}, min={fmt(self.min_)}, max={fmt(self.max_)}, "
            f"mean={fmt(self.mean_)}, median={fmt(self.median_)}"
        )

    def print_results(self, csv: bool = False) -> None:
        if csv:
            self.csv_print()
        else:
            self.pretty_print()


def device_times_ms(
    fn: Callable[[], Any],
    iteration_time_ms: float,
    warmup_time_ms: float,
    flush_cache: bool = True,
    max_iterations: int = 10000,
) -> list[float]:
    """Per-call device times (ms) of `fn` measured with events on the current stream."""
    fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(5):
        fn()
    end.record()
    torch.cuda.synchronize()
    estimate = max(start.elapsed_time(end) / 5, 1e-3)
    n_warm = max(1, int(warmup_time_ms / estimate))
    n_rep = max(1, min(max_iterations, int(iteration_time_ms / estimate)))
    flush = torch.empty(_FLUSH_BYTES, dtype=torch.int8, device="cuda") if flush_cache else None
    for _ in range(min(n_warm, max_iterations)):
        fn()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep)]
    for i in range(n_rep):
        if flush is not None:
            flush.zero_()
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in zip(starts, ends)]


def benchmark_it(
    fn: Callable[[], Any],
    tag: str,
    metadata: BenchmarkMetadata,
    iteration_time_ms: int = 10000,
    warmup_time_ms: int = 1000,
) -> BenchmarkResult:
    """Benchmark `fn` (same arguments and result type as the reference's benchmark_it)."""
    results = device_times_ms(fn, iteration_time_ms, warmup_time_ms)
    n = len(results)
    return BenchmarkResult(tag, metadata, n, min(results), max(results), sum(results) / n, sorted(results)[n // 2])
