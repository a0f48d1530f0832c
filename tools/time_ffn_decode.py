"""GPU time (HIP-graph replay) of scaled_gemm_silu_and_mul at decode sizes (the split-K kernel + the fused silu reduce kernel)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C
from conch_amd.ops.quantization.gemm import scaled_gemm_silu_and_mul
STREAM = torch.cuda.Stream()
with torch.cuda.stream(STREAM):
    _C.reserve_scratch(512 << 20)
STREAM.synchronize()
def graph_time(fn, calls=20, replays=30, rounds=5):
    with torch.cuda.stream(STREAM):
        for _ in range(5): fn()
    STREAM.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=STREAM):
        for _ in range(calls): fn()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    vals = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(replays): g.replay()
        e.record(); torch.cuda.synchronize()
        vals.append(s.elapsed_time(e) / (replays * calls) * 1e3)
    return sorted(vals)[len(vals) // 2]
for m, k, d in ((1, 4096, 11008), (16, 4096, 11008), (64, 4096, 11008), (16, 8192, 14336), (128, 4096, 5504)):
    for dt in (torch.int8, torch.float8_e4m3fn):
        torch.manual_seed(0)
        if dt == torch.int8:
            a = torch.randint(-32, 32, (m, k), dtype=dt, device="cuda"); bt = torch.randint(-32, 32, (2 * d, k), dtype=dt, device="cuda")
        else:
            a = (0.25 * torch.rand((m, k), device="cuda")).to(dt); bt = (0.25 * torch.rand((2 * d, k), device="cuda")).to(dt)
        sa = 0.05 * torch.rand((m, 1), device="cuda"); sb = 0.05 * torch.rand((2 * d, 1), device="cuda")
        t = graph_time(lambda: scaled_gemm_silu_and_mul(a, bt.T, sa, sb, torch.bfloat16))
        print(f"scaled_gemm_silu_and_mul {str(dt)[6:]:14s} {m:3d} x {k} x 2*{d}: {t:6.1f} us", flush=True)
