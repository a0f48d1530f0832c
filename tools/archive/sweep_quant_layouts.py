"""Static / dynamic quantisers on contiguous against row-strided inputs (a column slice of a wider activation, e.g. one third of a
fused qkv projection) and a strided output.  Op level, us.  Development aid -> stdout (profiles/r04/quant_layouts.txt)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd.ops.quantization.fp8 import scaled_fp8_quant  # noqa: E402
from conch_amd.ops.quantization.int8 import scaled_int8_quant  # noqa: E402


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


scale = torch.tensor([0.01], device="cuda")
for tokens, hidden in [(16, 4096), (128, 4096), (1024, 4096), (4096, 4096), (4096, 11008), (16384, 4096)]:
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        wide = (torch.rand((tokens, 3 * hidden), device="cuda") - 0.5).to(dt)
        cont = wide[:, :hidden].contiguous()
        sl = wide[:, hidden:2 * hidden]  # row stride 3 x hidden, 16-byte aligned rows
        odd = wide[:, 1:hidden + 1]      # misaligned rows
        for name, op in (("int8 static", lambda t: scaled_int8_quant(t, scale)), ("fp8 static", lambda t: scaled_fp8_quant(t, scale)),
                         ("int8 dynamic", lambda t: scaled_int8_quant(t, None)), ("fp8 dynamic", lambda t: scaled_fp8_quant(t, None))):
            a, b, c = timeit(lambda: op(cont)), timeit(lambda: op(sl)), timeit(lambda: op(odd))
            flag = "  <-- strided >1.5x" if b > 1.5 * a else ""
            print(f"{name:13s} {tokens:6d}x{hidden:5d} {str(dt)[6:]:9s}: contiguous {a:7.1f}  column slice {b:7.1f} (x{b / a:.2f})  misaligned slice {c:7.1f} (x{c / a:.2f}){flag}", flush=True)
