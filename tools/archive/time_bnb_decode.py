"""bitsandbytes-style 4-bit matmul at decode sizes: GPU time per call (HIP-graph replay, 20 calls per graph) and eager op-level time."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.ops.quantization.bitsandbytes.functional import matmul_4bit, quantize_4bit  # noqa: E402

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


def eager_time(fn, iters=300, rounds=5):
    for _ in range(50):
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


for qt in ("nf4", "fp4"):
    for m, k, n in ((1, 4096, 4096), (16, 4096, 4096), (16, 4096, 11008), (64, 4096, 11008), (16, 11008, 4096), (128, 4096, 4096)):
        torch.manual_seed(0)
        w = torch.randn((n, k), dtype=torch.float16, device="cuda")
        q, state = quantize_4bit(w, blocksize=64, quant_type=qt)
        x = (torch.rand((m, k), device="cuda") - 0.5).to(torch.float16)
        res = {}
        for name, variant in (("decode kernel", _C.VARIANT_MFMA_SKINNY), ("dequantise first", _C.VARIANT_MFMA_SIMPLE)):
            _C.set_gemm_variant(variant)
            fn = lambda: matmul_4bit(x, q, state)  # noqa: E731
            res[name] = (graph_time(fn), eager_time(fn))
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        (gd, ed), (gf, ef) = res["decode kernel"], res["dequantise first"]
        print(f"{qt} {m:4d}x{k}x{n}: GPU time decode kernel {gd:6.1f} us, dequantise first {gf:6.1f} us ({gf / gd:.1f}x);   eager op level {ed:6.1f} / {ef:6.1f} us", flush=True)
