import sys, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from conch_amd.ops.quantization import gemm as G
sys.path.insert(0, '/root/repo/tools/archive')
def timeit(fn, iters=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[1]
for k, d in ((4096, 11008), (8192, 14336), (4096, 4096)):
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, 2 * d), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, 2 * d), device="cuda") + 0.01).to(torch.float16)
    for m in (16, 32, 48, 64, 96, 128, 192, 256, 512):
        x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
        fused = timeit(lambda: G.mixed_precision_gemm_silu_and_mul(x, wq, ws, None, 4, 8, 128))
        def parts():
            g = G.mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
            return F.silu(g[:, :d]) * g[:, d:]
        pt = timeit(parts)
        a = G.mixed_precision_gemm_silu_and_mul(x, wq, ws, None, 4, 8, 128).float(); b = parts().float()
        print(f"mixed silu M={m:4d} K={k} N=2x{d}: fused {fused:7.1f} us  parts {pt:7.1f} us  x{fused/pt:.2f}  max|diff| {(a-b).abs().max().item():.3g} of {b.abs().max().item():.3g}", flush=True)
