import sys, numpy as np, torch
sys.path.insert(0, ".")
from conch_amd import _C
from conch_amd.ops.quantization.gemm import mixed_precision_gemm
torch.manual_seed(0)
for nt, (m, k, n) in [(4, (1024, 256, 1536)), (4, (256, 4096, 256)), (2, (256, 256, 128))]:
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k * 4 // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    _C.set_tuning(1, nt)
    _C.set_tuning(6, 1); d = mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
    tot = 0
    for rep in range(20):
        _C.set_tuning(6, 2); r = mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
        torch.cuda.synchronize()
        bad = (d.view(torch.int16) != r.view(torch.int16)).nonzero().cpu().numpy()
        tot += len(bad)
        if len(bad) and rep < 3:
            print("nt", nt, (m, k, n), "rep", rep, "mismatches", len(bad))
            for (i, j) in bad[:6].tolist():
                print("   ", i, j, "got", float(r[i, j]), "want", float(d[i, j]), " neighbours got", [float(v) for v in r[i, max(0, j - 2):j + 3]],
                      "want", [float(v) for v in d[i, max(0, j - 2):j + 3]])
    print("nt", nt, (m, k, n), "total mismatches over 20 reps", tot)
