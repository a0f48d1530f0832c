"""Randomised cross-check of the automatic kernel choice against the generic device kernel (development aid).

scaled_gemm: int8 bit-for-bit, fp8 within 2 eps of max|C|; mixed_precision_gemm: 2 eps of max|C|.  Shapes are drawn so
that every dispatcher branch (split-K 32/64/128 rows, 128x128 tiles, 256x256 tiles, repack, generic; decode-batch and tiled
mixed kernels) is hit.  usage: python tools/fuzz_dispatch.py [cases] [seed] [--mixed-mid | --asm]
--mixed-mid: mixed_precision_gemm only, at batched-decode sizes on wide problems (the K-split strip forms of round 5).
--asm: scaled_gemm only, FORCED onto the one-wave-per-SIMD assembly kernel (round 6: with and without a bias -- the *_bias twins --,
       all three tile widths, both input and output dtypes, scalar and vector scales, ragged M and N inside its contract).
"""
import ctypes
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.ops.quantization.gemm import mixed_precision_gemm, scaled_gemm  # noqa: E402

MIXED_MID = "--mixed-mid" in sys.argv
ASM = "--asm" in sys.argv
_width_hook = _C.load().conch_debug_gemm1w_width
_width_hook.restype, _width_hook.argtypes = ctypes.c_int, [ctypes.c_int]
sys.argv = [a for a in sys.argv if not a.startswith("--")]
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = {torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}
bad = 0
for case in range(CASES):
    m = random.choice([1, 3, 16, 17, 32, 33, 48, 64, 65, 100, 128, 129, 192, 256, 257, 300, 512, 700, 1024, 1500])
    k = random.choice([128, 256, 384, 1024, 1152, 2048, 2944, 3072, 4096, 8192, 11008, 13824, 28672])  # long K: the split-K forms of round 4
    n = random.choice([8, 24, 64, 100, 128, 260, 520, 1376, 2048, 4096, 4100, 11008])
    out_dt = random.choice([torch.float16, torch.bfloat16])
    torch.manual_seed(case)
    if MIXED_MID:
        m = random.choice([33, 40, 48, 64, 65, 96, 100, 128, 129, 192, 200, 256])
        n = random.choice([1376, 2048, 4096, 4100, 5120, 8192, 11008, 13824])
    if ASM:
        m = random.choice([16, 128, 256, 257, 300, 512, 1000, 1024, 2304])
        k = random.choice([512, 768, 1024, 2048])
        n = random.choice([16, 176, 352, 368, 400, 1008, 1056, 2816, 3584])
    if ASM or (not MIXED_MID and random.random() < 0.6):
        in_dt = random.choice([torch.int8, torch.float8_e4m3fn] if ASM else [torch.int8, torch.float8_e4m3fn, torch.float8_e4m3fnuz])
        special = ""
        if in_dt == torch.int8:
            a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
            bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
        else:
            a = (0.25 * torch.rand((m, k), device="cuda") - 0.1).to(in_dt)
            bt = (0.25 * torch.rand((n, k), device="cuda") - 0.1).to(in_dt)
            if in_dt == torch.float8_e4m3fnuz and random.random() < 0.6:
                # e4m3fnuz: +-240 (0x7F / 0xFF) scattered over both operands, now and then a NaN (0x80): the in-register split at decode
                # sizes, the scan + expansion flow on the tile kernels (round 6)
                for t, name in ((a, "a"), (bt, "b")):
                    if random.random() < 0.8:
                        cnt = random.choice([1, 7, 64])
                        idx = torch.randint(0, t.numel(), (cnt,), device="cuda")
                        t.view(torch.uint8).view(-1)[idx] = torch.where(torch.rand(cnt, device="cuda") < 0.5, 0x7F, 0xFF).to(torch.uint8)
                        special += f" {name}:{cnt}sat"
                if random.random() < 0.2:
                    a.view(torch.uint8)[random.randrange(m), random.randrange(k)] = 0x80
                    special += " a:nan"
        sa = 0.01 * torch.rand((m, 1), device="cuda") if random.random() < 0.7 else torch.tensor([[0.01]], device="cuda")
        sb = 0.01 * torch.rand((n, 1), device="cuda") if random.random() < 0.7 else torch.tensor([[0.02]], device="cuda")
        bias = torch.rand((n,), device="cuda").to(out_dt) if random.random() < 0.5 else None
        if ASM:
            width = random.choice([0, 0, 9, 7])
            _C.set_gemm_variant(_C.VARIANT_MFMA_ASM1W)
            _width_hook(width)
        try:
            got = scaled_gemm(a, bt.T, sa, sb, out_dt, bias)
        finally:
            if ASM:
                _width_hook(0)
        _C.set_gemm_variant(_C.VARIANT_GENERIC)
        ref = scaled_gemm(a, bt.T, sa, sb, out_dt, bias)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        if in_dt == torch.int8:
            ok = torch.equal(got, ref)
        else:
            gf, rf = got.float(), ref.float()
            nan = torch.isnan(rf)
            ok = torch.equal(torch.isnan(gf), nan)  # a fnuz NaN poisons exactly its row (none otherwise)
            if ok and not bool(nan.all()):
                ok = (gf[~nan] - rf[~nan]).abs().max().item() <= 2 * EPS[out_dt] * max(rf[~nan].abs().max().item(), 1e-6)
            if bool(nan.any()):
                got = torch.where(nan, torch.zeros_like(got), got)  # (the finiteness check below is about the rest)
        what = f"scaled {in_dt} {m}x{k}x{n} -> {out_dt} bias={bias is not None}{special}" + (f" assembly kernel, width hook {width}" if ASM else "")
    else:
        bits = random.choice([4, 8])
        k = max(k // 128 * 128, 128)
        x = (torch.rand((m, k), device="cuda") - 0.3).to(out_dt)
        wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
        ws = (0.05 * torch.rand((k // 128, n), device="cuda") + 0.01).to(out_dt)
        zp = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda") if random.random() < 0.5 else None
        got = mixed_precision_gemm(x, wq, ws, zp, bits, 0 if zp is not None else 2 ** (bits - 1), 128)
        _C.set_gemm_variant(_C.VARIANT_GENERIC)
        ref = mixed_precision_gemm(x, wq, ws, zp, bits, 0 if zp is not None else 2 ** (bits - 1), 128)
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        ok = (got.float() - ref.float()).abs().max().item() <= 2 * EPS[out_dt] * max(ref.float().abs().max().item(), 1e-6)
        what = f"mixed int{bits} {m}x{k}x{n} {out_dt} zp={zp is not None}"
    if not ok or not torch.isfinite(got.float()).all():
        bad += 1
        print("MISMATCH", what, flush=True)
print(f"{CASES} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
