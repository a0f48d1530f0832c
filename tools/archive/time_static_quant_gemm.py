import sys, torch
sys.path.insert(0, '/root/repo')
from conch_amd.ops.quantization import gemm as G
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant
def timeit(fn, iters=200):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return ts
f8 = torch.float8_e4m3fn
for kk, n in ((4096, 4096), (11008, 4096)):
    w2 = (0.25 * torch.rand((n, kk), device="cuda")).to(f8)
    sb2 = 0.25 * torch.rand((n, 1), device="cuda")
    sx = torch.tensor([0.01], device="cuda")
    for m in (1, 16, 64, 128, 256):
        x = (torch.rand((m, kk), device="cuda") - 0.3).to(torch.float16)
        q, _ = scaled_fp8_quant(x, sx)
        a = timeit(lambda: G.static_quant_scaled_gemm(x, w2.T, sx, sb2, torch.bfloat16))
        b = timeit(lambda: scaled_fp8_quant(x, sx))
        c = timeit(lambda: G.scaled_gemm(q, w2.T, sx, sb2, torch.bfloat16))
        print(f"M={m:4d} K={kk} N={n}: fused {['%.1f'%v for v in a]}  quant {['%.1f'%v for v in b]}  gemm {['%.1f'%v for v in c]}", flush=True)
