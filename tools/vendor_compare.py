"""Vendor-library comparators on the GPU box (development aid, not part of the product or of bench.py):
hipBLASLt through torch._scaled_mm (fp8, per-row / per-column scales) and through torch.matmul (fp16 GEMM on weights
that are ALREADY dequantised -- the upper bound for a fused int4 kernel), timed like tools/quick_bench.py
(back-to-back launches after a clock ramp, HIP events)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd.ops.quantization.gemm import mixed_precision_gemm, scaled_gemm  # noqa: E402


def timeit(fn, iters=100, rounds=5):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / iters)
    return sorted(best)[len(best) // 2]


def fp8_case(m, k, n):
    torch.manual_seed(0)
    a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
    bt = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    ours = timeit(lambda: scaled_gemm(a, bt.T, sa, sb, torch.bfloat16))
    line = f"fp8 {m}x{k}x{n}: conch_amd {ours * 1e3:.1f} us ({2.0 * m * n * k / ours / 1e9:.0f} TFLOP/s)"
    try:
        ref = timeit(lambda: torch._scaled_mm(a, bt.T, scale_a=sa, scale_b=sb.T.contiguous(), out_dtype=torch.bfloat16))
        got = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16).float()
        lib = torch._scaled_mm(a, bt.T, scale_a=sa, scale_b=sb.T.contiguous(), out_dtype=torch.bfloat16).float()
        line += (f"   torch._scaled_mm (hipBLASLt, row-wise scales) {ref * 1e3:.1f} us ({2.0 * m * n * k / ref / 1e9:.0f} TFLOP/s)"
                 f"   max|diff| {(got - lib).abs().max().item():.3g} of max|C| {lib.abs().max().item():.3g}")
    except Exception as exc:  # noqa: BLE001
        line += f"   torch._scaled_mm unavailable: {str(exc)[:120]}"
    print(line, flush=True)


def fp16_case(m, k, n):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
    w = (0.1 * torch.rand((k, n), device="cuda")).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    ours = timeit(lambda: mixed_precision_gemm(x, wq, ws, None, 4, 8, 128))
    ref = timeit(lambda: torch.matmul(x, w))
    print(f"int4 x fp16 {m}x{k}x{n}: conch_amd (fused dequantisation) {ours * 1e3:.1f} us ({2.0 * m * n * k / ours / 1e9:.0f} TFLOP/s)"
          f"   torch.matmul fp16 on dequantised weights (hipBLASLt / rocBLAS) {ref * 1e3:.1f} us ({2.0 * m * n * k / ref / 1e9:.0f} TFLOP/s)", flush=True)


def int8_case(m, k, n):
    """C2-style decode shape: ours (scales + cast fused) against torch._int_mm (hipBLASLt int8 -> int32, no epilogue)."""
    torch.manual_seed(0)
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
    bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    ours = timeit(lambda: scaled_gemm(a, bt.T, sa, sb, torch.bfloat16))
    line = f"int8 {m}x{k}x{n}: conch_amd (scales + bf16 cast fused) {ours * 1e3:.1f} us"
    try:
        ref = timeit(lambda: torch._int_mm(a, bt.T))
        line += f"   torch._int_mm (int32 result, no epilogue) {ref * 1e3:.1f} us"
    except Exception as exc:  # noqa: BLE001
        line += f"   torch._int_mm unavailable: {str(exc)[:100]}"
    print(line, flush=True)


def grid():
    """LLM projection shapes x batch sizes, ours against the vendor's, one line per shape: where does the dispatcher leave time?"""
    dims = [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (8192, 8192), (8192, 28672), (28672, 8192), (5120, 13824), (13824, 5120)]
    for k, n in dims:
        for m in (16, 64, 256, 512, 1024, 2048, 4096, 8192):
            if m * n * k > 8192 * 8192 * 28672 // 2:
                continue
            torch.manual_seed(0)
            a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
            bt = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
            sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
            sbt = sb.T.contiguous()
            it = 200 if m * n * k < 2**36 else 40
            ours = timeit(lambda: scaled_gemm(a, bt.T, sa, sb, torch.bfloat16), iters=it, rounds=3)
            try:
                ref = timeit(lambda: torch._scaled_mm(a, bt.T, scale_a=sa, scale_b=sbt, out_dtype=torch.bfloat16), iters=it, rounds=3)
            except Exception:  # noqa: BLE001
                ref = float("nan")
            fl = 2.0 * m * n * k
            flag = "  <-- vendor ahead" if ref < ours * 0.97 else ""
            print(f"fp8 M={m:5d} K={k:5d} N={n:5d}: ours {ours * 1e3:8.1f} us ({fl / ours / 1e9 / 5000:.3f})  vendor {ref * 1e3:8.1f} us  x{ours / ref:.2f}{flag}", flush=True)
            del a, bt


def grid_mixed():
    """int4 x fp16 (group 128, symmetric): the fused kernel against torch.matmul on weights dequantised beforehand (4x the weight bytes)."""
    dims = [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (8192, 8192), (8192, 28672), (28672, 8192), (5120, 13824), (13824, 5120)]
    for k, n in dims:
        for m in (1, 16, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
            if m * n * k > 8192 * 8192 * 8192:
                continue
            torch.manual_seed(0)
            x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
            w = (0.1 * torch.rand((k, n), device="cuda")).to(torch.float16)
            wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
            ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
            it = 200 if m * n * k < 2**36 else 40
            ours = timeit(lambda: mixed_precision_gemm(x, wq, ws, None, 4, 8, 128), iters=it, rounds=3)
            ref = timeit(lambda: torch.matmul(x, w), iters=it, rounds=3)
            fl = 2.0 * m * n * k
            flag = "  <-- vendor ahead" if ref < ours * 0.97 else ""
            print(f"int4xfp16 M={m:5d} K={k:5d} N={n:5d}: ours {ours * 1e3:8.1f} us ({fl / ours / 1e9 / 2500:.3f})  vendor fp16 {ref * 1e3:8.1f} us  x{ours / ref:.2f}{flag}", flush=True)
            del x, w, wq, ws


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--grid":
        grid()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--grid-mixed":
        grid_mixed()
        sys.exit(0)
    for shape in [(128, 4096, 4096), (32, 8192, 8192), (256, 4096, 11008)]:
        int8_case(*shape)
    for shape in [(4096, 4096, 11008), (8192, 8192, 8192), (8192, 8192, 3584)]:
        fp8_case(*shape)
    for shape in [(1024, 4096, 11008), (4096, 8192, 4096), (8192, 8192, 8192)]:
        fp16_case(*shape)
