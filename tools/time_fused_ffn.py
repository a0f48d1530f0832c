"""GPU time of the fused FFN ops (GEMM + silu_and_mul epilogue) under CONCH_TUNE_EPILOGUE = 1 (direct stores) and 2 (whole-line
image), HIP-graph replayed, interleaved in one process.  usage: python tools/time_fused_ffn.py [rounds]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.ops.quantization.gemm import mixed_precision_gemm_silu_and_mul, scaled_gemm_silu_and_mul  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
torch.manual_seed(0)


def scaled_case(m, k, d):
    a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((2 * d, k), device="cuda")).to(torch.float8_e4m3fn)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((2 * d, 1), device="cuda")
    return lambda: scaled_gemm_silu_and_mul(a, bt.T, sa, sb, torch.bfloat16), 2.0 * m * 2 * d * k, f"scaled fp8 {m}x{k}x(2x{d})"


def mixed_case(m, k, d):
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, 2 * d), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, 2 * d), device="cuda") + 0.01).to(torch.float16)
    return lambda: mixed_precision_gemm_silu_and_mul(x, wq, ws, None, 4, 8, 128), 2.0 * m * 2 * d * k, f"mixed int4 {m}x{k}x(2x{d})"


for fn, flops, name in (scaled_case(4096, 4096, 11008), scaled_case(8192, 8192, 14336), mixed_case(4096, 4096, 11008)):
    graphs, outs = {}, {}
    stream = torch.cuda.Stream()
    for mode in (1, 2):
        _C.set_tuning(6, mode)
        with torch.cuda.stream(stream):
            outs[mode] = fn()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(10):
                    y = fn()
            graphs[mode] = g
    _C.set_tuning(6, 0)
    torch.cuda.synchronize()
    same = bool(torch.equal(outs[1], outs[2]))

    def run(mode):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graphs[mode].replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 50

    for _ in range(3):
        run(1), run(2)
    res = {1: [], 2: []}
    for _ in range(rounds):
        for mode in (1, 2):
            res[mode].append(run(mode))
    for mode in (1, 2):
        med = statistics.median(res[mode])
        print(f"{name:34s} epilogue {mode}: median {med * 1e3:8.1f} us  min {min(res[mode]) * 1e3:8.1f} us  {flops / med / 1e9:7.1f} TFLOP/s  bits {'equal' if same else 'DIFFER'}",
              flush=True)
