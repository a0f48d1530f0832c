"""One process for `rocprofv3 --kernel-trace --stats`: 200 C3-size scaled_gemm calls on CLEAN e4m3fnuz operands (the scan + gated
kernels flow of the tile sizes), so that the per-kernel averages show what each launch of the flow costs."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd.ops.quantization.gemm import scaled_gemm  # noqa: E402

dev = torch.device("cuda")
m, k, n = 4096, 4096, 11008
torch.manual_seed(0)
a = (0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fnuz)
bt = (0.25 * torch.rand((n, k), device=dev)).to(torch.float8_e4m3fnuz)
sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
sat = "--saturated" in sys.argv
if sat:
    ra, rb = torch.arange(m, device=dev), torch.arange(n, device=dev)
    a.view(torch.uint8)[ra, (ra * 37) % k] = 0x7F
    bt.view(torch.uint8)[rb, (rb * 53) % k] = 0xFF
for _ in range(200):
    scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
torch.cuda.synchronize()
print("done", "saturated" if sat else "clean")
