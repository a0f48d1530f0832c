"""Soak test of the one-launch split-K hand-offs (scaled M <= 32 / C2, mixed decode <= 32 rows): N launches of eleven shapes in random
order, a second stream keeping the chip unevenly busy, every result compared bit for bit with the two-launch form.
usage: python tools/soak_one_launch.py [launches]"""
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.ops.quantization.gemm import scaled_gemm, mixed_precision_gemm
torch.manual_seed(0); random.seed(0)
cases = []
for (m, k, n, dt) in [(1, 4096, 4096, torch.int8), (16, 4096, 11008, torch.float8_e4m3fn), (32, 8192, 8192, torch.int8), (8, 11008, 4096, torch.float8_e4m3fn),
                      (128, 4096, 4096, torch.int8), (16, 2944, 520, torch.float8_e4m3fn)]:
    if dt == torch.int8:
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda"); bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(dt); bt = (0.25 * torch.rand((n, k), device="cuda")).to(dt)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    fn = lambda a=a, bt=bt, sa=sa, sb=sb: scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
    cases.append((f"scaled {m}x{k}x{n}", fn))
for (m, k, n) in [(1, 4096, 4096), (16, 4096, 11008), (32, 8192, 8192), (8, 11008, 4096), (24, 1152, 256)]:
    x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.05 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    fn = lambda x=x, wq=wq, ws=ws: mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
    cases.append((f"mixed {m}x{k}x{n}", fn))
_C.set_tuning(_C.TUNE_SKINNY_MODE, 1)
want = [fn().clone() for _, fn in cases]
_C.set_tuning(_C.TUNE_SKINNY_MODE, 0)
side = torch.cuda.Stream(); noise = torch.randn((3000, 3000), device="cuda")
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for i in range(N):
    j = random.randrange(len(cases))
    if i % 7 == 0:
        with torch.cuda.stream(side):
            noise @ noise
    got = cases[j][1]()
    if i % 4 == 0 or True:
        if not torch.equal(got, want[j]):
            bad += 1
            print("MISMATCH", i, cases[j][0], flush=True)
            if bad > 5: break
torch.cuda.synchronize()
print(f"{N} launches of the automatic (one-launch) forms in random order under uneven load: {bad} mismatches against the two-launch results")
