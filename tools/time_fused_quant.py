"""Decode-size activation quantiser + scaled_gemm: the pair against static_quant_scaled_gemm.  Each form is captured into a HIP graph
(20 calls per graph) and replayed, so the figure is GPU time per call, not the Python cost of one or two op calls."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant  # noqa: E402
from conch_amd.ops.quantization.gemm import scaled_gemm, static_quant_scaled_gemm  # noqa: E402
from conch_amd.ops.quantization.int8 import scaled_int8_quant  # noqa: E402


from conch_amd import _C  # noqa: E402

STREAM = torch.cuda.Stream()
with torch.cuda.stream(STREAM):
    _C.reserve_scratch(256 << 20)
STREAM.synchronize()
CALLS = 20


def timeit(fn, replays=30, rounds=5):
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


for dt, quant in ((torch.int8, scaled_int8_quant), (torch.float8_e4m3fn, scaled_fp8_quant)):
    for m, k, n in ((1, 4096, 4096), (16, 4096, 4096), (16, 4096, 11008), (64, 4096, 11008), (16, 11008, 4096), (128, 4096, 4096), (128, 8192, 8192)):
        torch.manual_seed(0)
        x = (torch.rand((m, k), device="cuda") - 0.5).to(torch.float16)
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda") if dt == torch.int8 else (0.25 * torch.rand((n, k), device="cuda")).to(dt)
        sx = torch.tensor([0.01], device="cuda")
        sx2 = sx.reshape(1, 1)
        sb = 0.25 * torch.rand((n, 1), device="cuda")

        def pair():
            q, _ = quant(x, sx)
            return scaled_gemm(q, bt.T, sx2, sb, torch.bfloat16)

        t_pair = timeit(pair)
        t_fused = timeit(lambda: static_quant_scaled_gemm(x, bt.T, sx, sb, torch.bfloat16))
        print(f"{str(dt)[6:]:13s} {m:4d}x{k}x{n}: quantiser + scaled_gemm {t_pair:6.1f} us   static_quant_scaled_gemm {t_fused:6.1f} us   "
              f"({t_pair / t_fused:.2f}x)", flush=True)
