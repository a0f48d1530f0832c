"""Headline benchmark: scaled GEMM fp8(e4m3fn) x fp8 -> bf16, M=4096 K=4096 N=11008 (BASELINE.json C3).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|c1|c4readme|c4decode|midm]

One "step" = one `scaled_gemm` call (public op: output allocation + fused GEMM/scale/cast kernel) on
synthetic inputs that are already resident in HBM, built with the reference benchmark's recipe
(benchmarks/scaled_gemm_benchmark.py:198-210).  Prints ONE JSON line on rank 0.

  value      effective TFLOP/s = 2*M*N*K * steps / wall time of the timed region (barrier + synchronize on both sides,
             max over ranks): exactly `--steps` calls after `--warmup` untimed ones.  On one GPU the chip is kept under
             the kernel's own load for SUSTAINED_S (2 s) BEFORE the warm-up steps, so the timed steps run at the clock
             the part settles to (the MFMA-dense kernels run against the board's power limit; a cold chip is up to 4 %
             faster for its first milliseconds).  The figure of a cold start rides along as the side field `burst`.
  roofline   dominant kernel (the MFMA GEMM): algorithmic FLOPs per launch / average launch duration measured with HIP
             events recorded on the launch stream inside the library (conch_time_*), against the dense MFMA peak of
             /opt/skills/guides/MI355X_MICROARCH.md.
               frac            launches averaged over the second half of >= 2 s of continuous load (steady-state clock:
                               the number a rocprofv3 --kernel-trace --stats pass of this command reproduces)
               frac_burst      back-to-back launches after 0.15 s of load (what rounds 1-4 reported as `frac`)
               frac_cold       the reference's benchmark protocol (conch/utils/benchmark.py:82-112 = do_bench): op-level
                               (allocation + kernel), L2 / Infinity Cache flushed before every run, median
             plus the clock the chip holds inside the K loop (`held_clock_mhz`: s_memtime / s_memrealtime stamps of the
             diagnostic twin of the library under the same sustained load; null if the twin is not built), and
               measured_peak   what THIS box sustains on THIS data in a bare MFMA loop fed from LDS (libconch_micro.so,
                               csrc_diag/micro_peak.hip: the product kernel's per-wave tile, fragment reads and MFMAs, no
                               global traffic, no barrier, no epilogue; LDS filled with the benchmark's operand bytes):
                               TFLOP/s, held clock, and `frac_of_measured` = achieved / that.
             `traffic` = PMC-measured HBM bytes per launch, read from profiles/traffic.json (separate rocprofv3 --pmc
             passes; `traffic_source` names the profile and the commit it was taken at), or null.
  cpu_baseline  the CPU oracle (oracle.scaled_gemm_ref: the reference's PyTorch-only path restated)
             timed on this host on the same workload, rank 0, N=1 only.
  fnuz       (N = 1, workload c3) e4m3fnuz operands at the C3 size and at 16 rows, clean and saturated (+-240 in every row / column):
             op-level ms per call, each form checked against the oracle first.  --no-side-legs leaves it out.
  c1 c2 c4 c4readme   (N = 1, workload c3) the other BASELINE configs and the reference README's mixed-GEMM shape, each
             measured the same way in the same run: op-level `ms_per_step` / `value` after sustained load, `roofline`
             (kernel events, sustained), `cpu_baseline` (bounded sample).  --no-side-legs leaves them out.

Multi-GPU (`--gpus N` > 1; launched by torch.distributed.run, or called bare -- `python bench.py --gpus N` then starts its own
N ranks as child processes and relays their one JSON line): the headline is BASELINE config C5 -- scaled GEMM fp8
8192x8192x28672 with N sharded over the ranks and C all-gathered over xGMI (RCCL), `"scaling": "strong"`: value = total
FLOPs / time of (GEMM + gather + unpack to row-major).  GEMM-only throughput, the gather-free block-major form and a
weak-scaling data-parallel C3 run (every rank its own 4096-token batch, replicated weights, no collective) are side fields.
The N = 1 line (C3, the BASELINE metric) always carries `c5_one_gpu`, the one-GPU figure of that same C5 problem, so that
strong-scaling efficiency follows from the lines alone: line(N).value / line(1).c5_one_gpu.value / N.
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
CLOCK_RAMP_S = 0.15  # the GPU needs ~20 ms of load to leave its idle clocks: the ramp of the `burst` side figure
SUSTAINED_S = 2.0  # continuous load in front of the timed region and behind `roofline.frac`
SIDE_SUSTAINED_S = 0.8  # the same for the side legs (c1, c2, c4, c4readme)


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


def kernel_avg_ms_scaled(a, b, sa, sb, out, iters: int, lib=None) -> float:
    """Average GEMM launch duration: HIP events on the launch stream, inside the C library."""
    md = create_scaled_metadata(a, b, sa, sb, out.dtype)
    ms = ctypes.c_float()
    extra = (iters, ctypes.byref(ms))
    _C.check(kgemm._scaled_gemm_call("conch_time_scaled_gemm", out, a, b, sa, sb, md, None, extra, lib=lib), "time")
    return ms.value


def kernel_avg_ms_mixed(x, packed, w_s, out, wt, iters: int, lib=None) -> float:
    md = create_mixed_precision_metadata(x, packed, w_s, None, wt.size_bits, wt.bias, 128)
    ms = ctypes.c_float()
    extra = (iters, ctypes.byref(ms))
    _C.check(kgemm._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, packed, w_s, None, md, extra, lib=lib), "time")
    return ms.value


def sustained_kernel_ms(timer, seconds: float = SUSTAINED_S) -> tuple[float, float]:
    """`timer(iters)` -> average ms per launch of `iters` back-to-back launches.  Keeps the chip loaded for `seconds` in
    chunks of ~50 ms and returns (average over the chunks of the SECOND half, total seconds loaded)."""
    probe = max(timer(20), 1e-4)
    chunk = int(min(4000, max(20, 50.0 / probe)))
    t0 = time.perf_counter()
    samples: list[tuple[float, float]] = []
    while time.perf_counter() - t0 < seconds or len(samples) < 4:
        samples.append((time.perf_counter() - t0, timer(chunk)))
    total = time.perf_counter() - t0
    late = [ms for t, ms in samples if t >= total / 2] or [samples[-1][1]]
    return sum(late) / len(late), total


def cold_op_ms(step, budget_ms: float = 400.0) -> dict:
    """The reference's benchmark protocol (conch/utils/benchmark.py:82-112): per-run device times of the PUBLIC op with
    the L2 / Infinity Cache flushed before every run; median (and min / mean / count)."""
    from conch_amd.utils.benchmark import device_times_ms

    times = device_times_ms(step, iteration_time_ms=budget_ms, warmup_time_ms=50.0, flush_cache=True, max_iterations=200)
    times.sort()
    n = len(times)
    return {"median_ms": times[n // 2], "min_ms": times[0], "mean_ms": sum(times) / n, "runs": n}


def held_clock_asm1w(timer, m: int, n: int) -> dict | None:
    """The same measurement for the one-wave-per-SIMD assembly kernel (csrc/asm/gen_gemm1w.py), which carries its own stamped
    twin INSIDE the product library (conch_debug_gemm1w_probe switches the next launches to it; the product kernel itself
    contains no stamp): wave 0 of every workgroup stamps (s_memtime, s_memrealtime) at entry, K loop start, K loop end, stores
    issued, stores retired.  None when the launches did not go to that kernel (nothing was stamped)."""
    try:
        fn = _C.load().conch_debug_gemm1w_probe
    except AttributeError:
        return None
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    nwg = -(-m // 256) * -(-n // 352)
    buf = torch.zeros((nwg, 5, 2), dtype=torch.int64, device="cuda")
    fn(buf.data_ptr())
    try:
        sustained_kernel_ms(timer, 1.0)
        torch.cuda.synchronize()
    finally:
        fn(None)
    st = buf.cpu().double()
    clk, rt = st[:, :, 0], st[:, :, 1]
    ok = rt[:, 2] > rt[:, 1]
    if int(ok.sum()) < nwg // 2:
        return None
    loop_clk = ((clk[ok, 2] - clk[ok, 1]) / (rt[ok, 2] - rt[ok, 1]) * 100.0).median().item()
    seg = lambda i: ((rt[ok, i + 1] - rt[ok, i]) / 100.0).median().item()  # noqa: E731  (us; the constant clock runs at 100 MHz)
    return {"held_clock_mhz": round(loop_clk, 1),
            "held_clock_source": "median over workgroups of d(s_memtime)/d(s_memrealtime) x 100 MHz around the K loop, stamped twin of the "
                                 "assembly kernel (conch_debug_gemm1w_probe), after >= 1 s of back-to-back launches",
            "workgroup_us": {"entry_to_k_loop": round(seg(0), 2), "k_loop": round(seg(1), 2), "epilogue_until_stores_issued": round(seg(2), 2),
                             "stores_retired": round(seg(3), 2)}}


def held_clock_mhz(kind: str, timer_for_lib, blocks: int) -> dict | None:
    """Clock inside the K loop under sustained load: the DIAGNOSTIC twin of the library (-DCONCH_CLOCK_PROBE: thread 0
    of every workgroup stamps s_memtime and the 100 MHz s_memrealtime around its K loop into a buffer nothing else
    reads; MI355X_MICROARCH.md, DVFS item 6).  The twin is only ever used here, after the product library was timed."""
    from conch_amd import _build

    if not _build.PROBE_LIB.exists():
        return None
    try:
        lib = _C.load_library(_build.PROBE_LIB)
        reader = getattr(lib, "conch_debug_probe_scaled" if kind.startswith("scaled") else "conch_debug_probe_mixed")
        reader.restype = ctypes.c_int
        reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
        timer = timer_for_lib(lib)
        sustained_kernel_ms(timer, 1.0)
        blocks = min(blocks, 4096)
        buf = (ctypes.c_ulonglong * (8 * blocks))()
        if reader(buf, blocks) != 0:
            return None
        clocks = sorted((buf[8 * b + 2] - buf[8 * b]) / (buf[8 * b + 3] - buf[8 * b + 1]) * 100.0
                        for b in range(blocks) if buf[8 * b + 3] > buf[8 * b + 1])
        if not clocks:
            return None
        return {"held_clock_mhz": round(clocks[len(clocks) // 2], 1),
                "held_clock_source": "median over workgroups of d(s_memtime)/d(s_memrealtime) x 100 MHz around the K loop, "
                                     "diagnostic twin libconch_amd_probe.so, after >= 1 s of back-to-back launches"}
    except Exception as exc:  # noqa: BLE001 - a diagnostic must never take the headline down
        return {"held_clock_mhz": None, "held_clock_source": f"probe failed: {exc!r}"}


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


def load_traffic(workload: str) -> tuple[float | None, str | None]:
    """(HBM-side bytes per launch measured by rocprofv3 --pmc, where that number comes from), from profiles/traffic.json -- which
    tools/write_traffic.py WRITES from the counter passes of tools/gpu_profile.sh.  An entry is used only if it was measured on
    the kernel sources this run executes (`sources_sha256` over the files that define the workload's dominant kernel): anything
    else is reported as null with the reason, not carried along."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None, None
    try:
        rec = json.loads(f.read_text()).get(workload)
        if not isinstance(rec, dict):
            return None, None
        want = rec.get("sources_sha256")
        if not want:
            return None, f"refused: the entry of {workload!r} carries no sources_sha256 (hand-maintained until round 5); re-run tools/gpu_profile.sh"
        sys.path.insert(0, str(ROOT / "tools"))
        from write_traffic import sources_sha256

        have = sources_sha256(workload, ROOT)
        if have != want:
            return None, (f"refused: measured at commit {rec.get('commit')} on sources {want}, this run executes {have} "
                          f"({', '.join(rec.get('sources', []))} changed since); re-run tools/gpu_profile.sh")
        return rec.get("hbm_bytes_per_launch"), rec.get("source")
    except Exception as exc:  # noqa: BLE001 -- a side figure
        return None, f"unreadable: {exc!r}"


def traffic_commit(workload: str) -> str | None:
    try:
        return json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(workload, {}).get("commit")
    except Exception:  # noqa: BLE001
        return None


def device_sync(device: torch.device | None = None) -> None:
    if device is None or device.type == "cuda":  # (a CPU device only in tests/test_distributed_cpu.py's rank script)
        torch.cuda.synchronize()


def barrier_sync(world: int, device: torch.device | None = None) -> None:
    if world > 1:
        torch.distributed.barrier()
    device_sync(device)


def timed_region(fn, steps: int, warmup: int, world: int, device: torch.device, ramp_s: float = CLOCK_RAMP_S, load=None) -> float:
    """Load phase (`load()` if given -- seconds of the kernel's own launches -- else `ramp_s` of calls), W warm-up calls, then
    exactly `steps` calls between barrier+synchronize; max over ranks (s)."""
    if load is not None:
        load()
    else:
        # `fn` may hold collectives (the N-sharded legs: all-gather per panel, dist.barrier in direct mode), so EVERY rank must
        # make the same number of calls: a loop bounded by each rank's own wall clock lets one rank leave a batch earlier than its
        # peer, whose extra all-gathers then pair with the first rank's warm-up calls and, at the end, with its barrier -- a
        # mismatched collective, i.e. a hang (the intermittent stall of round 5's two-rank test; it would have hung the N = 8 run
        # the same way).  With more than one rank the ranks vote after every batch: the loop ends on all of them in the same batch.
        t_end = time.perf_counter() + ramp_s
        go = torch.ones(1, dtype=torch.int32, device=device) if world > 1 else None
        while True:
            for _ in range(20):
                fn()
            device_sync(device)
            more = time.perf_counter() < t_end
            if world > 1:
                go.fill_(1 if more else 0)
                torch.distributed.all_reduce(go, op=torch.distributed.ReduceOp.MIN)
                more = bool(go.item())
            if not more:
                break
    for _ in range(warmup):
        fn()
    barrier_sync(world, device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier_sync(world, device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


def nshard_c5(world: int, rank: int, device: torch.device, steps: int, warmup: int, shape: tuple[int, int, int] | None = None,
              direct: bool = False) -> dict:
    """C5: M=8192 K=8192 N=28672 with N sharded over the ranks, C all-gathered over xGMI (RCCL).

    Timed: (1) the local GEMM only, (2) GEMM + all-gather in the block-major layout the collective produces, (3) GEMM +
    all-gather + unpack to the row-major [M, N] tensor (the headline: what a caller of the single-GPU op gets back).  Before
    timing, the exchange is CHECKED: every rank regenerates a band of every other rank's weight shard from its seed and
    compares the gathered columns with a local product of that band, bit for bit."""
    from conch_amd.distributed import NShardedScaledGemm

    m, k, n = shape or WORKLOADS["c5"][1:]
    torch.manual_seed(0)
    a = (0.25 * torch.rand((m, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), dtype=torch.float32, device=device)
    n_loc = n // world

    def shard(r: int, cols: int):
        g = torch.Generator(device=device).manual_seed(1 + r)
        bt = (0.25 * torch.rand((n_loc, k), dtype=torch.float32, device=device, generator=g)).to(torch.float8_e4m3fn)
        sb = 0.25 * torch.rand((n_loc, 1), dtype=torch.float32, device=device, generator=g)
        return bt[:cols], sb[:cols]

    bt_loc, sb_loc = shard(rank, n_loc)
    op = NShardedScaledGemm(m, n, torch.bfloat16, device, k=k, in_dtype=torch.float8_e4m3fn)
    c = op(a, bt_loc.T, sa, sb_loc)
    torch.cuda.synchronize()
    band = 256  # rows checked per foreign block; 64 of its columns
    mismatches = 0
    for r in range(world):
        bt_r, sb_r = shard(r, 64)
        want = scaled_gemm(a[:band], bt_r.T, sa[:band], sb_r, torch.bfloat16)
        got = c[:band, r * n_loc : r * n_loc + 64]
        mismatches += int((want.view(torch.int16) != got.view(torch.int16)).sum().item())
    t_gemm = timed_region(lambda: op.local_gemm(a, bt_loc.T, sa, sb_loc), steps, warmup, world, device)
    t_blocks = timed_region(lambda: op.gathered_panels(a, bt_loc.T, sa, sb_loc), steps, warmup, world, device)
    t_full = timed_region(lambda: op(a, bt_loc.T, sa, sb_loc), steps, warmup, world, device)
    flops = 2.0 * m * n * k
    direct_fields = {}
    if direct and world > 1:
        # opt-in (--c5-direct): the epilogue-writes-to-peers form (conch_scaled_gemm_multi; no collective on the data path).  Never
        # on by default: it has not met real xGMI links in any round, and a hang here would take the headline with it.
        try:
            op_d = NShardedScaledGemm(m, n, torch.bfloat16, device, direct=True, k=k, in_dtype=torch.float8_e4m3fn)
            c_d = op_d(a, bt_loc.T, sa, sb_loc)
            torch.cuda.synchronize()
            same = bool(torch.equal(c_d, c))
            t_direct = timed_region(lambda: op_d(a, bt_loc.T, sa, sb_loc), steps, warmup, world, device)
            direct_fields = {"tflops_direct_rowmajor": round(2.0 * m * n * k * steps / t_direct / 1e12, 2), "ms_direct_rowmajor": round(t_direct / steps * 1e3, 4),
                             "direct_equals_allgather_result": same}
        except Exception as exc:  # noqa: BLE001
            direct_fields = {"direct_error": repr(exc)}
    return {
        **direct_fields,
        "workload": f"scaled_gemm fp8 {m}x{k}x{n}, N/{world} = {n_loc} columns per rank, {op.panels} row panel(s)",
        "scaling": "strong",
        "exchange": "RCCL all_gather_into_tensor per row panel on a side stream, in place, + unpack to row-major" if world > 1 else "none (1 rank)",
        "exchange_check_mismatches": mismatches,
        "tflops_gemm_only": round(flops * steps / t_gemm / 1e12, 2),
        "tflops_gemm_plus_allgather_blockmajor": round(flops * steps / t_blocks / 1e12, 2),
        "tflops_gemm_plus_allgather_rowmajor": round(flops * steps / t_full / 1e12, 2),
        "ms_gemm_only": round(t_gemm / steps * 1e3, 4),
        "ms_gemm_plus_allgather_blockmajor": round(t_blocks / steps * 1e3, 4),
        "ms_gemm_plus_allgather_rowmajor": round(t_full / steps * 1e3, 4),
        "c_bytes_gathered_per_rank": m * n * 2,
    }


def c5_headline(res: dict, world: int, steps: int, warmup: int, extra: dict | None = None) -> dict:
    line = {
        "metric": "effective TFLOP/s, scaled-GEMM fp8xbf16 8192x8192x28672 N-sharded + RCCL all-gather (BASELINE C5)",
        "value": res["tflops_gemm_plus_allgather_rowmajor"], "unit": "TFLOP/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": res["ms_gemm_plus_allgather_rowmajor"], "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "fp8_e4m3fn", "data": "synthetic",
        "config": {"workload": res["workload"], "parallelism": f"N-sharded over {world} GPU(s), A replicated"},
        "roofline": {"bound": "mfma", "achieved": round(res["tflops_gemm_only"] / world, 2), "peak": FP8_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(res["tflops_gemm_only"] / world / FP8_PEAK_TFLOPS, 4), "traffic": None,
                     "note": "per-GPU GEMM-only rate from the op-level timed region (not kernel events)"},
        "detail": res,
    }
    if extra:
        line.update(extra)
    return line


class Leg:
    """One workload: the public op as `step`, a kernel-event timer, its algorithmic work and the roof that bounds it."""

    def __init__(self, workload: str, device: torch.device, seed: int):
        kind, m, k, n = WORKLOADS[workload]
        self.workload, self.kind, self.m, self.k, self.n, self.device = workload, kind, m, k, n, device
        self.micro_mode = None  # libconch_micro.so mode of the measured ceiling (MFMA-bound legs only)
        if kind == "quant_int8":
            from conch_amd.ops.quantization.int8 import scaled_int8_quant, static_scaled_int8_quant

            torch.manual_seed(seed)
            self.x = torch.rand(m, n, dtype=torch.float16, device=device) * 1000
            self.scale = torch.tensor([2.1], dtype=torch.float32, device=device)
            self.step = lambda: scaled_int8_quant(self.x, self.scale)
            out = self.out = torch.empty_like(self.x, dtype=torch.int8)

            def timer(iters: int) -> float:  # the quantiser launches on torch's current stream: torch events see it
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    static_scaled_int8_quant(out, self.x, self.scale)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / iters

            self.timer_for_lib = lambda lib: timer
            self.flops, self.bytes_alg = 0.0, m * n * 3
            self.peak, self.dtype_name = HBM_PEAK_GBS, "f32"
            self.desc = f"static_scaled_int8_quant fp16 [{m}x{n}], scale 2.1"
            self.hbm_bound, self.probe_blocks = True, 0
        elif kind.startswith("scaled"):
            self.a, self.b, self.sa, self.sb = make_scaled(kind, m, k, n, device, seed)
            self.step = lambda: scaled_gemm(self.a, self.b, self.sa, self.sb, torch.bfloat16)
            out = self.out = torch.empty((m, n), dtype=torch.bfloat16, device=device)
            self.timer_for_lib = lambda lib: (lambda iters: kernel_avg_ms_scaled(self.a, self.b, self.sa, self.sb, out, iters, lib))
            self.flops = 2.0 * m * n * k
            self.bytes_alg = m * k + k * n + 2 * m * n + 4 * (m + n)
            self.peak, self.dtype_name = (FP8_PEAK_TFLOPS, "fp8_e4m3fn") if kind == "scaled_fp8" else (I8_PEAK_TFLOPS, "int8")
            self.desc = f"scaled_gemm {self.dtype_name} x {self.dtype_name} -> bf16, per-row/per-column fp32 scales, M={m} K={k} N={n}"
            self.probe_blocks = -(-m // 256) * -(-n // 192)
            self.micro_mode, self.micro_src = (0 if kind == "scaled_fp8" else None), self.a
        else:
            self.x, self.packed, self.w_s, self.w_ref, self.wt = make_mixed(m, k, n, device, seed)
            self.step = lambda: mixed_precision_gemm(self.x, self.packed, self.w_s, None, self.wt.size_bits, self.wt.bias, 128)
            out = self.out = torch.empty((m, n), dtype=torch.float16, device=device)
            self.timer_for_lib = lambda lib: (lambda iters: kernel_avg_ms_mixed(self.x, self.packed, self.w_s, out, self.wt, iters, lib))
            self.flops = 2.0 * m * n * k
            self.bytes_alg = 2 * m * k + k * n // 2 + 2 * (k // 128) * n + 2 * m * n
            self.peak, self.dtype_name = F16_PEAK_TFLOPS, "fp16"
            self.desc = f"mixed_precision_gemm uint4b8(g128) x fp16 -> fp16, M={m} K={k} N={n}"
            self.probe_blocks = -(-m // 256) * -(-n // 128)
            self.micro_mode, self.micro_src = 1, self.x
        if kind != "quant_int8":
            # C2-like shapes are HBM/latency bound: report against the HBM roof instead of the MFMA roof
            self.hbm_bound = self.flops / self.bytes_alg < self.peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        self.timer = self.timer_for_lib(None)

    def rate(self, ms: float) -> float:
        return self.bytes_alg / (ms * 1e-3) / 1e9 if self.hbm_bound else self.flops / (ms * 1e-3) / 1e12

    @property
    def roof_peak(self) -> float:
        return HBM_PEAK_GBS if self.hbm_bound else self.peak

    @property
    def unit(self) -> str:
        return "GB/s" if self.hbm_bound else "TFLOP/s"

    def value(self, steps: int, world: int, elapsed: float) -> float:
        """Whole-job throughput in the metric's unit: TFLOP/s for the GEMMs, GB/s for the quantiser."""
        work = self.bytes_alg / 1e9 if self.kind == "quant_int8" else self.flops / 1e12
        return work * steps * world / elapsed

    def picked_kernel(self) -> str:
        """The kernel the automatic dispatcher launches for this leg's call (conch_debug_*_plan: the same pure function of the
        problem that run_scaled_fast / run_mixed act on), by its symbol name."""
        lib = _C.load()
        if self.kind == "quant_int8":
            return "quant_flat_kernel (quant.hip)"
        if self.kind.startswith("scaled"):
            fn = lib.conch_debug_scaled_plan
            fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
            out = (ctypes.c_int * 2)()
            if fn(self.m, self.n, self.k, 3 if self.kind == "scaled_fp8" else 4, out) != 0:
                return "unknown"
            tag = "fp8" if self.kind == "scaled_fp8" else "i8"
            width = {352: "", 288: "_n9", 224: "_n7"}.get(out[1], "?")
            return {0: "scaled_gemm_pp2_persistent_kernel 256x256 (gemm_mfma.hip)", 1: "scaled_gemm_mid_kernel 128x128 (gemm_mid.hip)",
                    2: "skinny_splitk_kernel + skinny_reduce_kernel (gemm_skinny.hip)",
                    3: f"conch_gemm1w_{tag}_bf16{width} 256x{out[1]} (asm/gen_gemm1w.py)"}.get(out[0], f"pick {out[0]}")
        fn = lib.conch_debug_mixed_plan
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)]
        out = (ctypes.c_int * 4)()
        if fn(self.m, self.n, self.k, 4, 1, 0, out) != 0:
            return "unknown"
        return {0: "mixed generic kernel (gemm_generic.hip)", 1: "mixed_skinny_kernel (gemm_mixed_skinny.hip)",
                2: "mixed_gemm_kernel 256x256 (gemm_mixed.hip)", 3: f"mixed_strip_kernel {out[1]}x{out[2]} tiles, {out[3]} K slice(s) (gemm_mixed_strip.hip)",
                4: "conch_mixed1w (asm/gen_mixed1w.py)"}.get(out[0], f"pick {out[0]}")

    def parity(self, got: torch.Tensor, rows: int = 256) -> dict:
        """`got` = an output of THIS leg's `step` (the public op under the automatic dispatcher: the call that is timed) against the
        CPU oracle, as the reference's benchmarks check before they time (benchmarks/scaled_gemm_benchmark.py:216-227).  The
        oracle is the checker here, never the thing measured.
          quantiser   every element, bit for bit (oracle.scaled_int8_quant_ref)
          scaled      `rows` rows spread over M (every row tile of the kernels is hit), every column: int8 bit for bit; fp8 within
                      the per-element bound of tests/test_gpu_gemm.py::check_scaled -- two roundings to bf16 plus the worst-case
                      fp32 accumulation-order error of that element's own products
          mixed       the same rows against oracle.mixed_precision_gemm_ref (reference bar: rtol 1e-1, atol min(5e-2 sqrt K, 1))
                      and against the fp64 product of the reference's own operands within check_mixed's per-element bound
        max_excess = max(|got - ref| - bound) over the checked elements (<= 0 passes; mismatching elements for bit-exact legs)."""
        import oracle

        t0 = time.perf_counter()
        if self.kind == "quant_int8":
            q = got[0] if isinstance(got, tuple) else got
            want = oracle.scaled_int8_quant_ref(self.x.cpu(), self.scale.cpu())
            bad = int((q.cpu() != want).sum().item())
            res = {"checked_rows": self.m, "bit_exact": True, "max_excess": bad, "ok": bad == 0, "oracle": "oracle.scaled_int8_quant_ref"}
        else:
            g = torch.Generator().manual_seed(1234)
            take = min(rows, self.m)
            if self.m <= rows:
                idx = torch.arange(self.m)
            else:
                idx = torch.unique(torch.cat([torch.linspace(0, self.m - 1, take // 2).long(), torch.randint(0, self.m, (take - take // 2,), generator=g)]))
            got_rows = got[idx.to(got.device)].cpu()
            if self.kind.startswith("scaled"):
                a, b, sa, sb = self.a[idx.to(self.device)].cpu(), self.b.cpu(), self.sa[idx.to(self.device)].cpu(), self.sb.cpu()
                ref = oracle.scaled_gemm_ref(a, b, sa, sb, torch.bfloat16)
                if self.kind == "scaled_int8":
                    bad = int((got_rows.view(torch.int16) != ref.view(torch.int16)).sum().item())
                    res = {"bit_exact": True, "max_excess": bad, "ok": bad == 0}
                else:
                    s_abs = a.float().abs() @ b.float().abs()
                    bound = 2.0 * 2.0**-7 * ref.float().abs() + (sa.reshape(-1, 1) * sb.reshape(1, -1)).abs() * (self.k * 2.0**-24) * s_abs + 2.0 * 2.0**-133
                    excess = ((got_rows.float() - ref.float()).abs() - bound).max().item()
                    res = {"bit_exact": False, "max_excess": excess, "ok": excess <= 0,
                           "max_abs_diff": (got_rows.float() - ref.float()).abs().max().item(), "max_abs_ref": ref.float().abs().max().item()}
                res["oracle"] = "oracle.scaled_gemm_ref"
            else:
                a = self.x[idx.to(self.device)].cpu()
                exact = a.double() @ self.w_ref.double()
                bound = 2.0**-10 * exact.abs() + (self.k * 2.0**-24) * (a.double().abs() @ self.w_ref.double().abs()) + 1e-30
                excess = ((got_rows.double() - exact).abs() - bound).max().item()
                atol = min(5e-2 * self.k**0.5, 1.0)
                # the oracle's own fp16 matmul on a spread of the sampled rows (this host class runs fp16 CPU matmuls at
                # ~0.2 GMAC/s: 256 rows of c4readme took 47 s in round 6's first run), the fp64 product above on all of them
                sub = torch.linspace(0, idx.numel() - 1, min(idx.numel(), max(4, (1 << 28) // (self.k * self.n)))).long().unique()
                ref = oracle.mixed_precision_gemm_ref(a[sub], self.w_ref).float()
                ref_bar = bool(((got_rows[sub].float() - ref).abs() <= atol + 1e-1 * ref.abs()).all().item())
                res = {"bit_exact": False, "max_excess": excess, "ok": excess <= 0 and ref_bar, "reference_tolerance_met": ref_bar,
                       "max_abs_diff": (got_rows[sub].float() - ref).abs().max().item(), "max_abs_ref": ref.abs().max().item(),
                       "oracle_rows": int(sub.numel()), "oracle": "oracle.mixed_precision_gemm_ref (oracle_rows rows) + fp64 product of w_ref (checked_rows rows)"}
            res["checked_rows"] = int(idx.numel())
        res["kernel"] = self.picked_kernel()
        res["seconds"] = round(time.perf_counter() - t0, 2)
        return res

    def cpu_baseline(self) -> dict:
        if self.kind == "quant_int8":
            import oracle

            xc, sc = self.x.cpu(), self.scale.cpu()
            oracle.scaled_int8_quant_ref(xc, sc)
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                oracle.scaled_int8_quant_ref(xc, sc)
                times.append(time.perf_counter() - t0)
            med = sorted(times)[2]
            return {"value": round(self.bytes_alg / med / 1e9, 2), "unit": "GB/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": f"full tensor through oracle.scaled_int8_quant_ref, median of 5 runs, {med * 1e3:.1f} ms each, host has {os.cpu_count()} logical CPUs"}
        if self.kind.startswith("scaled"):
            return cpu_baseline_scaled(self.kind, self.m, self.k, self.n)
        return cpu_baseline_mixed(self.x, self.w_ref, self.m, self.k, self.n)


_MICRO = None


def micro_lib():
    """libconch_micro.so (csrc_diag/micro_peak.hip): the achievable-peak loop.  None if it was not built."""
    global _MICRO
    if _MICRO is None:
        from conch_amd import _build

        if not _build.MICRO_LIB.exists():
            _MICRO = False
        else:
            lib = ctypes.CDLL(str(_build.MICRO_LIB))
            lib.conch_micro_mfma_peak.restype = ctypes.c_int
            lib.conch_micro_mfma_peak.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)]
            _MICRO = lib
    return _MICRO or None


def measured_peak(leg: Leg, seconds: float) -> dict | None:
    """What this box sustains on this leg's operand bytes in the bare LDS-fed MFMA loop: launches of ~4 ms for `seconds`, the
    second half averaged (TFLOP/s and in-loop clock)."""
    lib = micro_lib()
    if lib is None or leg.micro_mode is None:
        return None
    try:
        src = leg.micro_src.contiguous().view(torch.uint8).reshape(-1)
        words = 1 << ((src.numel() // 4).bit_length() - 1)
        iters = 3000 if leg.micro_mode == 0 else 3000
        ms, fl, clk = ctypes.c_float(), ctypes.c_double(), ctypes.c_float()
        stream = torch.cuda.current_stream().cuda_stream
        samples = []
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds or len(samples) < 4:
            rc = lib.conch_micro_mfma_peak(src.data_ptr(), words, leg.micro_mode, iters, stream, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(clk))
            if rc != 0:
                return {"error": f"conch_micro_mfma_peak returned {rc}"}
            samples.append((time.perf_counter() - t0, fl.value / (ms.value * 1e-3) / 1e12, clk.value))
        total = time.perf_counter() - t0
        late = [x for x in samples if x[0] >= total / 2] or samples[-1:]
        tf = sum(x[1] for x in late) / len(late)
        return {"tflops": round(tf, 1), "frac_of_datasheet": round(tf / leg.peak, 4), "held_clock_mhz": round(sum(x[2] for x in late) / len(late), 1),
                "loop": ("32 x v_mfma_f32_16x16x128_f8f6f4" if leg.micro_mode == 0 else "64 x v_mfma_f32_16x16x32_f16") +
                        " per 12 ds_read_b128 fragment pairs, 128 x 64 per wave, 2 waves per SIMD, every CU, operands = this workload's bytes",
                "seconds": round(total, 2), "launches": len(samples)}
    except Exception as exc:  # noqa: BLE001 - a diagnostic must never take the headline down
        return {"error": repr(exc)}


def measure_leg(leg: Leg, steps: int, warmup: int, world: int, sustained_s: float, with_burst: bool, with_cold: bool, with_probe: bool,
                with_peak: bool, with_cpu: bool, with_parity: bool = True) -> dict:
    """One workload, measured: op-level timed region after `sustained_s` of the kernel's own load, kernel-level roofline
    (sustained / burst / cold), the measured ceiling and the CPU oracle's rate.  `with_parity`: BEFORE anything is timed an output
    of the op that will be timed is checked against the CPU oracle (Leg.parity), and AFTER the timed region the buffer the
    kernel-event timer wrote under sustained load (`leg.out`: the launches behind `roofline`) is checked the same way."""
    device = leg.device
    burst = None
    parity = None
    if with_parity:
        leg.out.zero_()
        parity = leg.parity(leg.step())
    if with_burst:  # a cold chip's figure (rounds 1-4's headline protocol), taken FIRST
        t = timed_region(leg.step, steps, warmup, world, device)
        k_burst_ms = leg.timer(max(20, min(steps, 200)))
        burst = {"ms_per_step": round(t / steps * 1e3, 5), "value": round(leg.value(steps, world, t), 2), "kernel_avg_ms": round(k_burst_ms, 5),
                 "clock_ramp_s": CLOCK_RAMP_S}
    sus: dict = {}

    def load() -> None:
        sus["ms"], sus["s"] = sustained_kernel_ms(leg.timer, sustained_s)

    if sustained_s > 0:
        elapsed = timed_region(leg.step, steps, warmup, world, device, load=load)
    else:
        elapsed = timed_region(leg.step, steps, warmup, world, device)
        sus["ms"], sus["s"] = leg.timer(max(20, min(steps, 200))), 0.0
    k_ms = sus["ms"]
    roofline = {"bound": "hbm" if leg.hbm_bound else "mfma", "achieved": round(leg.rate(k_ms), 2), "peak": leg.roof_peak, "unit": leg.unit,
                "frac": round(leg.rate(k_ms) / leg.roof_peak, 4)}
    roofline["traffic"], roofline["traffic_source"] = load_traffic(leg.workload)
    roofline["traffic_commit"] = traffic_commit(leg.workload)
    roofline["kernel_avg_ms"] = round(k_ms, 5)
    roofline["sustained_seconds"] = round(sus["s"], 2)
    if burst is not None:
        roofline["kernel_burst_ms"] = burst["kernel_avg_ms"]
        roofline["frac_burst"] = round(leg.rate(burst["kernel_avg_ms"]) / leg.roof_peak, 4)
    if with_cold:
        cold = cold_op_ms(leg.step)
        roofline["op_cold_median_ms"] = round(cold["median_ms"], 5)
        roofline["op_cold_min_ms"] = round(cold["min_ms"], 5)
        roofline["op_cold_runs"] = cold["runs"]
        roofline["frac_cold"] = round(leg.rate(cold["median_ms"]) / leg.roof_peak, 4)
        roofline["cold_protocol"] = "public op (allocation + kernel), 512 MiB cache flush before every run, median (reference: conch/utils/benchmark.py:82-112)"
    if with_probe and leg.probe_blocks:
        clock = held_clock_asm1w(leg.timer, leg.m, leg.n) if leg.kind == "scaled_fp8" else None
        if clock is None:
            clock = held_clock_mhz(leg.kind, leg.timer_for_lib, leg.probe_blocks)
        if clock:
            roofline.update(clock)
    if with_peak and not leg.hbm_bound:
        mp = measured_peak(leg, 1.5 if sustained_s >= SUSTAINED_S else 1.0)
        if mp:
            roofline["measured_peak"] = mp
            if "tflops" in mp:
                roofline["frac_of_measured"] = round(leg.rate(k_ms) / mp["tflops"], 4)
    roofline["algorithmic_bytes"] = leg.bytes_alg
    roofline["algorithmic_flops"] = leg.flops
    res = {"value": round(leg.value(steps, world, elapsed), 2), "unit": "GB/s" if leg.kind == "quant_int8" else "TFLOP/s",
           "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 5), "dtype": leg.dtype_name,
           "config": {"workload": leg.desc}, "roofline": roofline}
    if burst is not None:
        res["burst"] = burst
    if parity is not None:
        torch.cuda.synchronize()
        after = leg.parity(leg.out)
        parity["timed_buffer_max_excess"], parity["timed_buffer_ok"] = after["max_excess"], after["ok"]
        parity["seconds"] = round(parity["seconds"] + after["seconds"], 2)
        parity["ok"] = bool(parity["ok"] and after["ok"])
        res["parity"] = parity
    if with_cpu:
        res["cpu_baseline"] = leg.cpu_baseline()
    return res


def fnuz_side_legs(device: torch.device) -> dict:
    """e4m3fnuz operands -- the dtype the reference hands AMD users (conch/ops/quantization/fp8.py:54) -- at the C3 size and at a
    decode size, op-level (torch events around 30 calls after 10): `clean` = no code that OCP e4m3 cannot hold (scan + the fp8
    kernel on the raw bytes), `saturated` = every row of A and every column of B holds one +-240 (0x7F / 0xFF: what a clamping
    absmax quantiser emits; the exact bf16 expansion serves it).  Each form is checked on 64 rows against the oracle
    with the tests' per-element bound before it is timed."""
    import oracle

    out: dict = {}
    for name, (m, k, n) in (("c3_size", (4096, 4096, 11008)), ("decode_16_rows", (16, 4096, 11008))):
        torch.manual_seed(0)
        a = (0.25 * torch.rand((m, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fnuz)
        bt = (0.25 * torch.rand((n, k), dtype=torch.float32, device=device)).to(torch.float8_e4m3fnuz)
        sa = 0.25 * torch.rand((m, 1), dtype=torch.float32, device=device)
        sb = 0.25 * torch.rand((n, 1), dtype=torch.float32, device=device)
        a_sat, bt_sat = a.clone(), bt.clone()
        ra, rb = torch.arange(m, device=device), torch.arange(n, device=device)
        a_sat.view(torch.uint8)[ra, (ra * 37) % k] = torch.where(ra % 2 == 0, 0x7F, 0xFF).to(torch.uint8)
        bt_sat.view(torch.uint8)[rb, (rb * 53) % k] = torch.where(rb % 3 == 0, 0xFF, 0x7F).to(torch.uint8)
        rows = torch.linspace(0, m - 1, min(m, 64)).long().unique()
        leg: dict = {"workload": f"scaled_gemm e4m3fnuz x e4m3fnuz -> bf16, M={m} K={k} N={n}"}
        for form, (aa, bb) in (("clean", (a, bt)), ("saturated", (a_sat, bt_sat))):
            got = scaled_gemm(aa, bb.T, sa, sb, torch.bfloat16)
            ac, bc, sac, sbc = aa[rows.to(device)].cpu(), bb.cpu().T, sa[rows.to(device)].cpu(), sb.cpu()
            ref = oracle.scaled_gemm_ref(ac, bc, sac, sbc, torch.bfloat16).float()
            bound = 2.0 * 2.0**-7 * ref.abs() + (sac * sbc.T).abs() * (k * 2.0**-24) * (ac.float().abs() @ bc.float().abs()) + 2.0**-132
            excess = ((got[rows.to(device)].float().cpu() - ref).abs() - bound).max().item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(10):
                scaled_gemm(aa, bb.T, sa, sb, torch.bfloat16)
            e0.record()
            for _ in range(30):
                scaled_gemm(aa, bb.T, sa, sb, torch.bfloat16)
            e1.record()
            torch.cuda.synchronize()
            leg[form] = {"ms_per_call": round(e0.elapsed_time(e1) / 30, 5), "parity_ok": bool(excess <= 0), "max_excess": excess, "checked_rows": int(rows.numel())}
        out[name] = leg
    return out


def _flush_c_stdio() -> None:
    """RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would otherwise be flushed at process
    exit -- BEHIND the JSON line.  Flush it now."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001, S110 -- cosmetic
        pass


_REAL_STDOUT: int | None = None


def claim_stdout() -> None:
    """From here on whatever any library writes to stdout (RCCL's banner, gloo's "Expected number of connected peer ranks",
    torch warnings that pick stdout) lands on stderr: file descriptor 1 is pointed at stderr and the JSON line alone is written to
    the descriptor that WAS stdout -- the driver parses one line, whatever the libraries feel like printing."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _flush_c_stdio()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(result: dict, world: int, rank: int) -> None:
    """The ONE JSON line, as the only thing any rank writes to stdout: every rank flushes what the libraries buffered, the ranks
    meet, rank 0 prints, then the group is torn down."""
    sys.stdout.flush()
    _flush_c_stdio()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        line = (json.dumps(result) + "\n").encode()
        if _REAL_STDOUT is not None:
            os.write(_REAL_STDOUT, line)
        else:
            sys.stdout.buffer.write(line)
            sys.stdout.flush()
    if world > 1:
        torch.distributed.destroy_process_group()
        _flush_c_stdio()


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv: list[str], script: Path | None = None, launch_timeout_s: float = 0.0) -> int:
    """`python bench.py --gpus N` called bare (no torchrun environment): start the N ranks as plain CHILD processes of this one
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment -- the env:// rendezvous
    torch.distributed.run would set up, without its elastic agent) and relay the one JSON line rank 0 prints.  This parent never
    touches the GPU -- no HIP call, no torch.cuda.is_available(), no exec: on this pool replacing a process that has initialised
    the GPU takes the machine down.

    The parent OWNS its ranks: it knows their PIDs (printed to stderr as `bench.py ranks: <pid> ...`), they share its process
    group, and it is the one that ends them --
      * a rank that exits non-zero takes the others down (SIGTERM, then SIGKILL) and its code is returned;
      * with `launch_timeout_s` > 0 (--launch-timeout / $CONCH_BENCH_LAUNCH_TIMEOUT_S) ranks still alive after that long are sent
        SIGUSR1 (each writes every thread's stack into its own file: faulthandler, async-signal-safe, so a rank blocked inside a
        collective answers too), then SIGKILL; the stacks are printed to stderr and 124 is returned.
    (Round 5 started `python -m torch.distributed.run` here; its workers live in sessions of their own, so killing the launcher's
    process group from a test left the ranks alive, holding the pipes and the GPU.)"""
    import signal
    import subprocess
    import tempfile

    target = str(script or Path(__file__).resolve())
    with tempfile.TemporaryDirectory(prefix="conch_bench_") as tmp:
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   CONCH_BENCH_STACKS_DIR=tmp)
        env.setdefault("OMP_NUM_THREADS", "1")
        out0 = open(os.path.join(tmp, "rank0.stdout"), "w+b")  # a file, not a pipe: nothing a rank writes can ever block it
        procs: list[subprocess.Popen] = []
        for r in range(n):
            renv = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, target, *argv], env=renv, stdout=out0 if r == 0 else subprocess.DEVNULL))
        print("bench.py ranks: " + " ".join(str(p.pid) for p in procs), file=sys.stderr, flush=True)

        def end_all(sig: int) -> None:
            for p in procs:
                if p.poll() is None:
                    try:
                        p.send_signal(sig)
                    except ProcessLookupError:
                        pass

        def reap(grace_s: float) -> None:
            t_stop = time.monotonic() + grace_s
            for p in procs:
                try:
                    p.wait(timeout=max(0.05, t_stop - time.monotonic()))
                except subprocess.TimeoutExpired:
                    pass
            end_all(signal.SIGKILL)
            for p in procs:
                p.wait()

        rc, t0 = 0, time.monotonic()
        try:
            while True:
                codes = [p.poll() for p in procs]
                bad = [c for c in codes if c not in (None, 0)]
                if bad:
                    rc = bad[0] if bad[0] > 0 else 128 - bad[0]
                    end_all(signal.SIGTERM)
                    reap(5.0)
                    break
                if all(c == 0 for c in codes):
                    break
                if launch_timeout_s > 0 and time.monotonic() - t0 > launch_timeout_s:
                    print(f"bench.py: ranks still running after {launch_timeout_s:.0f} s -- collecting stacks, then ending them",
                          file=sys.stderr, flush=True)
                    end_all(signal.SIGUSR1)
                    time.sleep(1.5)
                    end_all(signal.SIGKILL)
                    reap(5.0)
                    for r in range(n):
                        f = Path(tmp) / f"rank{r}.stacks"
                        txt = f.read_text(errors="replace") if f.exists() else "(no stack file: the rank had not reached main())"
                        print(f"---- rank {r} (pid {procs[r].pid}) stacks ----\n{txt[-6000:]}", file=sys.stderr, flush=True)
                    rc = 124
                    break
                time.sleep(0.05)
        finally:
            end_all(signal.SIGKILL)  # whatever ended the loop (KeyboardInterrupt included): no rank outlives its launcher
            for p in procs:
                p.wait()
        out0.seek(0)
        text = out0.read().decode("utf-8", "replace")
        out0.close()
    line = None
    for raw in text.splitlines():
        raw = raw.strip()
        if raw.startswith("{") and raw.endswith("}"):
            try:
                json.loads(raw)
            except ValueError:
                continue
            line = raw  # the last JSON object wins (there is one)
    if line is not None and rc == 0:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        print("bench.py: the ranks exited 0 without printing a JSON line", file=sys.stderr)
        return 1
    return rc


def stall_if_asked(rank: int) -> None:
    """Test switch: $CONCH_BENCH_STALL_RANK=<r> makes that rank sit here for ever after the rendezvous, so its peers block in
    their next collective -- the failure tests/test_gpu_distributed.py must see end in ONE failed test, with stacks, no orphan."""
    want = os.environ.get("CONCH_BENCH_STALL_RANK", "")
    if want != "" and int(want) == rank:
        while True:
            time.sleep(1.0)


def arm_stack_dumps(rank: int) -> None:
    """A rank of a multi-process run writes every thread's stack into $CONCH_BENCH_STACKS_DIR/rank<r>.stacks when its launcher
    sends SIGUSR1 (spawn_ranks' launch timeout) and, if $CONCH_BENCH_WATCHDOG_S is set, by itself after that many seconds."""
    import faulthandler
    import signal

    d = os.environ.get("CONCH_BENCH_STACKS_DIR")
    f = open(os.path.join(d, f"rank{rank}.stacks"), "w") if d and os.path.isdir(d) else sys.stderr  # noqa: SIM115 -- lives as long as the process
    faulthandler.register(signal.SIGUSR1, file=f, all_threads=True)
    watchdog = float(os.environ.get("CONCH_BENCH_WATCHDOG_S", "0") or 0)
    if watchdog > 0:
        faulthandler.dump_traceback_later(watchdog, exit=False, file=f)
    arm_stack_dumps.file = f  # keep the descriptor open


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="skip the sustained / cold / clock legs (PMC profiling passes)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed op's output (rocprofv3 passes: fewer kernels in the trace)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cache-flushed op-level leg")
    ap.add_argument("--no-probe", action="store_true", help="skip the diagnostic-twin clock probe (kernel-trace profiling: its kernels share names)")
    ap.add_argument("--no-peak", action="store_true", help="skip the measured-ceiling microbenchmark (libconch_micro.so)")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="N = 1, workload c3: leave out the c1 / c2 / c4 / c4readme side fields (rocprofv3 passes of the headline kernel)")
    ap.add_argument("--c5-base", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-c5-base", action="store_true",
                    help="N = 1, workload c3: leave out the one-GPU figure of BASELINE config C5 (the strong-scaling base of --gpus N > 1); "
                         "for rocprofv3 kernel-trace passes, whose per-kernel average it would skew")
    ap.add_argument("--c5-direct", action="store_true",
                    help="with --gpus N > 1: also time the direct form of C5 (the GEMM's epilogue stores into every peer's result; side field)")
    ap.add_argument("--dp", action="store_true",
                    help="with --gpus N > 1: make the weak-scaling data-parallel C3 run the headline instead of N-sharded C5")
    ap.add_argument("--c5-shape", default=None, metavar="M,K,N",
                    help="dry-run aid: another (smaller) problem for the N-sharded C5 path; the line says so in config.workload")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("CONCH_BENCH_LAUNCH_TIMEOUT_S", "0") or 0), metavar="S",
                    help="bare --gpus N > 1 only: ranks still alive after S seconds dump their stacks and are killed; exit code 124 (0 = wait for ever)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="dry-run aid: put every rank on cuda:0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1") or 1) == 1:
        # called bare, as the driver calls N = 1 (no launcher's environment, or a stale one-rank one inherited from the caller):
        # be the launcher (before anything touches the GPU)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], launch_timeout_s=args.launch_timeout))
    claim_stdout()

    rank, local_rank, world = dist_env()
    if args.gpus != world:
        print(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `--nproc-per-node {args.gpus}` (or call bench.py bare: it "
              "starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path is the only path)"
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        arm_stack_dumps(rank)
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            # every rank of this benchmark is on one node: gloo's pairs go over loopback, whatever other interfaces the box has
            # and whatever its hostname resolves to
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            torch.distributed.init_process_group(args.backend)

        stall_if_asked(rank)

    kind, m, k, n = WORKLOADS[args.workload]
    result: dict = {}
    if args.workload == "c5" or (world > 1 and args.workload == "c3" and not args.dp):
        # multi-GPU headline (and `--workload c5` on one GPU): BASELINE config C5, strong scaling over N
        shape = tuple(int(v) for v in args.c5_shape.split(",")) if args.c5_shape else None
        res = nshard_c5(world, rank, device, args.steps, args.warmup, shape=shape, direct=args.c5_direct)
        extra = {}
        if world > 1 and args.workload == "c3":
            try:  # side field: weak-scaling data parallel C3 (no collective on the data path)
                a, b, sa, sb = make_scaled("scaled_fp8", m, k, n, device, seed=rank)
                t = timed_region(lambda: scaled_gemm(a, b, sa, sb, torch.bfloat16), args.steps, args.warmup, world, device)
                extra["dp_weak_c3"] = {"scaling": "weak", "tflops_total": round(2.0 * m * n * k * args.steps * world / t / 1e12, 2),
                                       "ms_per_step": round(t / args.steps * 1e3, 5),
                                       "workload": f"scaled_gemm fp8 {m}x{k}x{n} per rank, weights replicated, no collective"}
            except Exception as exc:  # noqa: BLE001
                extra["dp_weak_c3"] = {"error": repr(exc)}
        emit(c5_headline(res, world, args.steps, args.warmup, extra), world, rank)
        return

    full = not args.quick
    leg = Leg(args.workload, device, seed=rank)
    sustained_s = (SUSTAINED_S if world == 1 else 0.0) if full else 0.0
    res = measure_leg(leg, args.steps, args.warmup, world, sustained_s, with_burst=full and world == 1,
                      with_cold=full and not args.no_cold and leg.kind != "quant_int8", with_probe=world == 1 and full and not args.no_probe,
                      with_peak=world == 1 and full and not args.no_peak, with_cpu=world == 1 and rank == 0 and not args.no_cpu_baseline,
                      with_parity=not args.no_parity)
    if args.workload == "c3":
        metric = "effective TFLOP/s + % MFMA roofline, scaled-GEMM fp8xbf16 4096x4096x11008"
    elif leg.kind == "quant_int8":
        metric = "GB/s, static_scaled_int8_quant 4096x4096 fp16 (bytes = 2 in + 1 out per element)"
    else:
        metric = f"effective TFLOP/s, {leg.desc}"
    result = {
        "metric": metric, "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": leg.dtype_name,
        "data": "synthetic",
        "config": {
            "workload": leg.desc,
            "parallelism": "single GPU" if world == 1 else f"dp{world} over tokens (M={m} per rank, weights replicated)",
            "gemm_variant": "auto",
            "load_before_timed_region_s": sustained_s if sustained_s else CLOCK_RAMP_S,
        },
        "roofline": res["roofline"],
    }
    if "parity" in res:
        result["parity"] = res["parity"]
    if "burst" in res:
        result["burst"] = res["burst"]
    if world == 1 and args.workload == "c3" and full and not args.no_side_legs:
        # the other BASELINE configs and the README comparator, driver-timed in the same run
        del leg
        for name in ("c1", "c2", "c4", "c4readme"):
            try:
                side = Leg(name, device, seed=0)
                result[name] = measure_leg(side, 50, 10, 1, SIDE_SUSTAINED_S, with_burst=False, with_cold=False, with_probe=False,
                                           with_peak=not args.no_peak and name == "c4", with_cpu=not args.no_cpu_baseline,
                                           with_parity=not args.no_parity)
                del side
            except Exception as exc:  # noqa: BLE001 -- a side field must not take the headline down
                result[name] = {"error": repr(exc)}
            torch.cuda.empty_cache()
    if world == 1 and args.workload == "c3" and not (args.no_c5_base or args.quick):
        # The strong-scaling base of the multi-GPU lines: `--gpus N` (N > 1) reports BASELINE config C5 (a different metric than
        # this line's C3), so its one-GPU figure rides along on EVERY N = 1 line and strong-scaling efficiency is computable from
        # driver lines alone: speed-up(N) = line(N).value / line(1).c5_one_gpu.value (total work fixed), efficiency = that / N.
        # It runs LAST, behind every leg whose numbers this line reports; --no-c5-base leaves it out (it launches the headline's
        # kernel symbol on another shape, which would skew a rocprofv3 per-kernel average of this command).
        try:
            r5 = nshard_c5(1, 0, device, 10, 3)
            result["c5_one_gpu"] = {"metric": "effective TFLOP/s, scaled-GEMM fp8xbf16 8192x8192x28672 on ONE GPU (strong-scaling base of --gpus N > 1)",
                                    "value": r5["tflops_gemm_plus_allgather_rowmajor"], "unit": "TFLOP/s",
                                    "ms_per_step": r5["ms_gemm_plus_allgather_rowmajor"], "steps": 10, "warmup": 3}
        except Exception as exc:  # noqa: BLE001 -- a side field must not take the headline down
            result["c5_one_gpu"] = {"error": repr(exc)}
    if world == 1 and args.workload == "c3" and full and not args.no_side_legs:
        try:
            result["fnuz"] = fnuz_side_legs(device)
        except Exception as exc:  # noqa: BLE001 -- a side field must not take the headline down
            result["fnuz"] = {"error": repr(exc)}
        torch.cuda.empty_cache()
    if "cpu_baseline" in res:
        result["cpu_baseline"] = res["cpu_baseline"]
    # a fast kernel whose results differ from the oracle's is not a result: the line still prints (with parity.ok false), the
    # exit code says so (what the reference's benchmarks do before they time: benchmarks/scaled_gemm_benchmark.py:216-227)
    checks = [result.get("parity")] + [result[nm].get("parity") for nm in ("c1", "c2", "c4", "c4readme") if isinstance(result.get(nm), dict)]
    for leg_ in (result.get("fnuz") or {}).values():
        if isinstance(leg_, dict):
            checks += [{"ok": f["parity_ok"], "what": "fnuz"} for f in leg_.values() if isinstance(f, dict) and "parity_ok" in f]
    failed = [c for c in checks if c is not None and not c["ok"]]
    emit(result, world, rank)
    if failed:
        print(f"bench.py: PARITY FAILED against the oracle: {failed}", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
