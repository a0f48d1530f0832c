"""int4 decode GEMM (GEMV sizes): two launches (decode kernel + reduce kernel) against one launch (the last-arriving workgroup of a
tile reduces it).  GPU time per call from HIP-graph replay (20 calls per graph), and eager op-level time."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.ops.quantization.gemm import mixed_precision_gemm  # noqa: E402

STREAM = torch.cuda.Stream()
with torch.cuda.stream(STREAM):
    _C.reserve_scratch(256 << 20)
STREAM.synchronize()
CALLS = 20


def graph_time(fn, replays=30, rounds=5):
    with torch.cuda.stream(STREAM):
        for _ in range(5):
            fn()
    STREAM.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=STREAM):
        for _ in range(CALLS):
            fn()
    for _ in range(10):
        graph.replay()
    torch.cuda.synchronize()
    vals = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(replays):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        vals.append(s.elapsed_time(e) / (replays * CALLS) * 1e3)
    return sorted(vals)[len(vals) // 2]


def eager_time(fn, iters=400, rounds=5):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    vals = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        vals.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(vals)[len(vals) // 2]


for m, k, n in ((1, 4096, 4096), (8, 4096, 4096), (16, 4096, 4096), (32, 4096, 4096), (1, 4096, 11008), (16, 4096, 11008), (32, 4096, 11008),
                (1, 11008, 4096), (16, 11008, 4096), (1, 8192, 8192), (16, 8192, 8192), (16, 4096, 28672)):
    torch.manual_seed(0)
    x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.05 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    fn = lambda: mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)  # noqa: E731
    res = {}
    for name, mode in (("two launches", 1), ("one launch", 2)):
        _C.set_tuning(_C.TUNE_SKINNY_MODE, mode)
        res[name] = (graph_time(fn), eager_time(fn))
    _C.set_tuning(_C.TUNE_SKINNY_MODE, 0)
    (g2, e2), (g1, e1) = res["two launches"], res["one launch"]
    print(f"int4 x fp16 {m:3d}x{k}x{n}: GPU time two launches {g2:5.1f} us, one launch {g1:5.1f} us ({g2 / g1:.2f}x);   eager op level {e2:5.1f} -> {e1:5.1f} us", flush=True)
