"""Run mixed_precision_gemm (int4 x fp16, group 128) N times at one shape -- the program to put behind `rocprofv3 --kernel-trace --stats --`.
usage: python3 tools/archive/run_mixed_once.py M K N [iters]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd.ops.quantization.gemm import mixed_precision_gemm  # noqa: E402

m, k, n = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
torch.manual_seed(0)
x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
for _ in range(iters):
    mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
torch.cuda.synchronize()
