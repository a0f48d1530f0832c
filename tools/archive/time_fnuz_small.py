"""scaled_gemm on e4m3fnuz operands against e4m3fn at decode / mid sizes (op level, back-to-back).  Development aid."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd.ops.quantization.gemm import scaled_gemm  # noqa: E402


def timeit(fn, iters=200):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[2]


for m, k, n in [(1, 4096, 4096), (16, 4096, 4096), (64, 4096, 11008), (128, 4096, 4096), (256, 11008, 4096), (512, 4096, 4096), (1024, 4096, 4096), (4096, 4096, 11008)]:
    torch.manual_seed(0)
    x, w = 0.25 * torch.rand((m, k), device="cuda"), 0.25 * torch.rand((n, k), device="cuda")
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    res = {}
    for name, dt in (("fn", torch.float8_e4m3fn), ("fnuz", torch.float8_e4m3fnuz), ("sat", torch.float8_e4m3fnuz)):
        a, bt = x.to(dt), w.to(dt)
        if name == "sat":  # one saturated code in the weights: the exact path
            bt.view(torch.uint8)[5, 17] = 0x7F
        res[name] = timeit(lambda: scaled_gemm(a, bt.T, sa, sb, torch.bfloat16), 200 if m < 2048 else 50)
    print(f"{m:5d}x{k:5d}x{n:5d}: e4m3fn {res['fn']:7.1f} us   e4m3fnuz {res['fnuz']:7.1f} us (x{res['fnuz'] / res['fn']:.2f})   "
          f"e4m3fnuz with a saturated weight {res['sat']:7.1f} us (x{res['sat'] / res['fn']:.2f})", flush=True)

# GPU-side time of the e4m3fnuz flow alone: the same call replayed from a HIP graph (no host launch cost)
from conch_amd import _C  # noqa: E402

for m, k, n in [(16, 4096, 4096), (128, 4096, 4096), (64, 4096, 11008)]:
    x, w = 0.25 * torch.rand((m, k), device="cuda"), 0.25 * torch.rand((n, k), device="cuda")
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    for name, dt in (("fn", torch.float8_e4m3fn), ("fnuz", torch.float8_e4m3fnuz)):
        a, bt = x.to(dt), w.to(dt)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            _C.reserve_scratch(_C.load().conch_scaled_gemm_workspace_bytes(m, n, k))
            scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(20):
                out = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
        t = timeit(g.replay, 20) / 20
        print(f"graph replay {m:4d}x{k}x{n} {name:5s}: {t:6.1f} us per call (GPU side)", flush=True)
