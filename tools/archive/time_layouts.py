"""scaled_gemm with the weights in the layouts a caller may hand over: K-contiguous (the MFMA contract: b = w.T of an [N, K] tensor)
against row-major [K, N] (repacked by the library on every call).  Op level, us.  Development aid."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd.ops.quantization.gemm import scaled_gemm  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(10):
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


for dt in (torch.float8_e4m3fn, torch.int8):
    for m, k, n in [(16, 4096, 4096), (128, 4096, 4096), (1024, 4096, 4096), (4096, 4096, 11008)]:
        if dt == torch.int8:
            a, w = torch.randint(-32, 32, (m, k), dtype=dt, device="cuda"), torch.randint(-32, 32, (n, k), dtype=dt, device="cuda")
        else:
            a, w = (0.25 * torch.rand((m, k), device="cuda")).to(dt), (0.25 * torch.rand((n, k), device="cuda")).to(dt)
        sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
        b_kmajor = w.T                      # (K, N) view, K-contiguous
        b_rowmajor = w.T.contiguous()       # (K, N) contiguous: N-contiguous
        t1 = timeit(lambda: scaled_gemm(a, b_kmajor, sa, sb, torch.bfloat16))
        t2 = timeit(lambda: scaled_gemm(a, b_rowmajor, sa, sb, torch.bfloat16))
        same = torch.equal(scaled_gemm(a, b_kmajor, sa, sb, torch.bfloat16), scaled_gemm(a, b_rowmajor, sa, sb, torch.bfloat16))
        print(f"{str(dt)[6:]:13s} {m:5d}x{k}x{n:5d}: K-contiguous weights {t1:7.1f} us   row-major [K, N] weights {t2:7.1f} us (x{t2 / t1:.2f})   same bits: {same}", flush=True)
