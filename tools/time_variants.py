"""Kernel times of the instantiations BASELINE's own configurations do not touch (development / evidence aid):
C4-size mixed GEMM with zero points / 8-bit weights / bf16, the e4m3fnuz compat path at C3 size, the dynamic quantisers, the
bitsandbytes kernels, the kernel-mode path.  Sustained (>= 0.5 s each), HIP-event or torch-event averages.  -> stdout."""
import ctypes
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata, create_scaled_metadata, scaled_gemm  # noqa: E402


def sustained(timer, seconds=0.6):
    probe = max(timer(10), 1e-4)
    chunk = int(min(2000, max(10, 30.0 / probe)))
    t0, vals = time.perf_counter(), []
    while time.perf_counter() - t0 < seconds:
        vals.append(timer(chunk))
    late = vals[len(vals) // 2:]
    return sum(late) / len(late)


def torch_timer(fn):
    def run(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    return run


def mixed(m, k, n, dtype, bits, zp):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    pf = 32 // bits
    wq = torch.randint(-2**31, 2**31 - 1, (k // pf, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    wzp = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda") if zp else None
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, wzp, bits, 0 if zp else (8 if bits == 4 else 128), 128)
    ms = ctypes.c_float()

    def run(iters):
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, wzp, md, (iters, ctypes.byref(ms))), "time")
        return ms.value
    t = sustained(run)
    print(f"mixed {m}x{k}x{n} int{bits}{'+zp' if zp else '   '} {str(dtype)[6:]:9s}: {t * 1e3:8.1f} us  {2.0 * m * n * k / t / 1e9:7.0f} TFLOP/s "
          f"({2.0 * m * n * k / t / 1e9 / 2500:.3f} of 2.5 PF)", flush=True)


def scaled(m, k, n, dtype):
    torch.manual_seed(0)
    if dtype == torch.int8:  # the reference tests' distribution (tests/scaled_gemm_test.py); NOT (0.25 * rand).to(int8), which is all zeros
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(dtype)
        bt = (0.25 * torch.rand((n, k), device="cuda")).to(dtype)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()

    def run(iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
        return ms.value
    t = sustained(run)
    peak = 5000  # fp8 / int8 dense MFMA peak (e4m3fnuz too since round 4: raw bytes on the OCP fp8 MFMA behind a device-side flag)
    print(f"scaled {m}x{k}x{n} {str(dtype)[6:]:16s}: {t * 1e3:8.1f} us  {2.0 * m * n * k / t / 1e9:7.0f} TFLOP/s ({2.0 * m * n * k / t / 1e9 / peak:.3f} of "
          f"{peak / 1000} PF){'  [scan + fp8 kernel on the raw bytes; no saturated code in this data]' if dtype == torch.float8_e4m3fnuz else ''}", flush=True)


def quantizers():
    from conch_amd.ops.quantization.fp8 import scaled_fp8_quant
    from conch_amd.ops.quantization.int8 import scaled_int8_quant

    for tokens, hidden in ((4096, 4096), (4096, 8192), (128, 4096)):
        x = torch.randn(tokens, hidden, device="cuda", dtype=torch.float16)
        s = torch.tensor([2.1], device="cuda")
        for name, fn, nbytes in (("static int8", lambda: scaled_int8_quant(x, s), 3), ("dynamic int8", lambda: scaled_int8_quant(x), 3),
                                 ("static fp8", lambda: scaled_fp8_quant(x, s), 3), ("dynamic fp8", lambda: scaled_fp8_quant(x), 3)):
            t = sustained(torch_timer(fn), 0.3)
            print(f"{name:13s} quant {tokens}x{hidden} fp16 (op level): {t * 1e3:7.2f} us  {tokens * hidden * nbytes / t / 1e6:7.0f} GB/s", flush=True)


def bnb():
    from conch_amd.ops.quantization.bitsandbytes.functional import dequantize_4bit, matmul_4bit, quantize_4bit

    w = torch.randn(11008, 4096, device="cuda", dtype=torch.float16)
    for qt in ("nf4", "fp4"):
        q, st = quantize_4bit(w, blocksize=64, quant_type=qt)
        t = sustained(torch_timer(lambda: quantize_4bit(w, blocksize=64, quant_type=qt)), 0.3)
        print(f"bnb quantize_4bit {qt} 11008x4096 fp16 b64 (op level): {t * 1e3:7.1f} us  {w.numel() * 2.5 / t / 1e6:7.0f} GB/s", flush=True)
        t = sustained(torch_timer(lambda: dequantize_4bit(q, st, blocksize=64, quant_type=qt)), 0.3)
        print(f"bnb dequantize_4bit {qt} 11008x4096 -> fp16 (op level): {t * 1e3:7.1f} us  {w.numel() * 2.5 / t / 1e6:7.0f} GB/s", flush=True)
        for m in (16, 1024):
            x = torch.randn(m, 4096, device="cuda", dtype=torch.float16)
            t = sustained(torch_timer(lambda: matmul_4bit(x, q, st)), 0.3)
            print(f"bnb matmul_4bit {qt} M={m} K=4096 N=11008 ({'decode kernel: codes -> MFMA operands' if m <= 128 else 'dequantise + 16-bit MFMA GEMM'}): {t * 1e3:7.1f} us  {2.0 * m * 4096 * 11008 / t / 1e9:7.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    mixed(1024, 4096, 11008, torch.float16, 4, False)
    mixed(1024, 4096, 11008, torch.float16, 4, True)
    mixed(1024, 4096, 11008, torch.float16, 8, False)
    mixed(1024, 4096, 11008, torch.bfloat16, 4, False)
    mixed(1024, 4096, 11008, torch.bfloat16, 4, True)
    mixed(1024, 4096, 11008, torch.bfloat16, 8, True)
    mixed(4096, 8192, 4096, torch.float16, 4, False)
    scaled(4096, 4096, 11008, torch.float8_e4m3fn)
    scaled(4096, 4096, 11008, torch.float8_e4m3fnuz)
    scaled(4096, 4096, 11008, torch.int8)
    quantizers()
    bnb()
