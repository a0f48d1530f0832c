"""The fused ops across batch sizes against the compositions they replace (all op level, back-to-back):
scaled_gemm_silu_and_mul vs scaled_gemm + (silu * up) in torch; mixed_precision_gemm_silu_and_mul vs mixed_precision_gemm + torch;
static_quant_scaled_gemm vs quant + scaled_gemm; matmul_4bit vs dequantize_4bit + torch.matmul.  Development aid: finds batch sizes
where a fused form is SLOWER than its own parts.  -> stdout (profiles/r04/fused_ops_sweep.txt)"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd.ops.quantization import gemm as G  # noqa: E402
from conch_amd.ops.quantization.bitsandbytes import functional as B  # noqa: E402
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant  # noqa: E402

MS = (1, 16, 64, 128, 256, 512, 1024, 2048, 4096)


def timeit(fn, iters=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[1]


def line(tag, m, k, n, fused, parts):
    flag = "  <-- fused slower than its parts" if fused > parts * 1.03 else ""
    print(f"{tag:34s} M={m:5d} K={k:5d} N={n:5d}: fused {fused:8.1f} us   parts {parts:8.1f} us   x{fused / parts:.2f}{flag}", flush=True)


def main():
    f8 = torch.float8_e4m3fn
    k, d = 4096, 11008  # Llama-7B gate/up: N = 2d
    w = (0.25 * torch.rand((2 * d, k), device="cuda")).to(f8)
    sb = 0.25 * torch.rand((2 * d, 1), device="cuda")
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, 2 * d), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, 2 * d), device="cuda") + 0.01).to(torch.float16)
    for m in MS:
        it = 100 if m <= 1024 else 30
        a = (0.25 * torch.rand((m, k), device="cuda")).to(f8)
        sa = 0.25 * torch.rand((m, 1), device="cuda")
        fused = timeit(lambda: G.scaled_gemm_silu_and_mul(a, w.T, sa, sb, torch.bfloat16), it)

        def parts():
            g = G.scaled_gemm(a, w.T, sa, sb, torch.bfloat16)
            return F.silu(g[:, :d]) * g[:, d:]
        line("scaled_gemm_silu_and_mul fp8", m, k, 2 * d, fused, timeit(parts, it))
    for m in MS:
        it = 100 if m <= 1024 else 30
        x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
        fused = timeit(lambda: G.mixed_precision_gemm_silu_and_mul(x, wq, ws, None, 4, 8, 128), it)

        def parts():
            g = G.mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
            return F.silu(g[:, :d]) * g[:, d:]
        line("mixed_precision_gemm_silu int4", m, k, 2 * d, fused, timeit(parts, it))
    n = 4096
    for kk in (4096, 11008):
        w2 = (0.25 * torch.rand((n, kk), device="cuda")).to(f8)
        sb2 = 0.25 * torch.rand((n, 1), device="cuda")
        sx = torch.tensor([0.01], device="cuda")
        for m in MS:
            it = 100 if m <= 1024 else 30
            x = (torch.rand((m, kk), device="cuda") - 0.3).to(torch.float16)
            fused = timeit(lambda: G.static_quant_scaled_gemm(x, w2.T, sx, sb2, torch.bfloat16), it)

            def parts():
                q, _ = scaled_fp8_quant(x, sx)
                return G.scaled_gemm(q, w2.T, sx, sb2, torch.bfloat16)
            line("static_quant_scaled_gemm fp8", m, kk, n, fused, timeit(parts, it))
    for kk, nn in ((4096, 4096), (4096, 11008), (11008, 4096)):
        wf = (torch.randn((nn, kk), device="cuda") * 0.02).to(torch.float16)
        packed, state = B.quantize_4bit(wf, blocksize=64, quant_type="nf4")
        for m in MS:
            it = 100 if m <= 1024 else 30
            x = (torch.rand((m, kk), device="cuda") - 0.3).to(torch.float16)
            fused = timeit(lambda: B.matmul_4bit(x, packed, state), it)

            def parts():
                wd = B.dequantize_4bit(packed, state)
                return x @ wd.T
            line("bnb matmul_4bit nf4", m, kk, nn, fused, timeit(parts, it))


if __name__ == "__main__":
    main()
