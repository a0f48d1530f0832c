"""mixed_precision_gemm (int4 x fp16, group 128) at batched-decode sizes (M = 32 .. 512): the column-strip kernel's K-split forms
(tile rows 64 / 128 / 256 x tile width x K slices: CONCH_TUNE_MIXED_STRIP_ROWS / _TILE_NT / _SPLITK) against the decode-batch
kernel, the LDS-tiled kernel's own split and the dispatcher's pick -- the table gemm_mixed_strip.hip's strip_plan and
dispatch_fit.hpp's MixedStripSplit are fitted on.  Weights warm (one buffer) and cold (rotating through > 600 MB).

    python tools/sweep_mixed_mid.py [--check] [--quick]        -> stdout (profiles/r05/mixed_mid_sweep.txt)

--check first compares every forced form with the always-correct generic kernel (same dequantised weights, fp32 sums) on small and
ragged shapes and exits non-zero on a mismatch."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata, mixed_precision_gemm  # noqa: E402

LIB = _C.load()
COLD_BYTES = 640 << 20
ms = ctypes.c_float()
QUICK = "--quick" in sys.argv
BITS = 8 if "--int8" in sys.argv else 4   # weight width of the timing sweep (uint4b8 / uint8b128)
XDT = torch.bfloat16 if "--bf16" in sys.argv else torch.float16  # activation / result dtype of the timing sweep


def force(rows=0, nt=0, split=0, kernel=0, variant=0):
    _C.set_gemm_variant(variant)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
    _C.set_tuning(_C.TUNE_MIXED_STRIP_ROWS, rows)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, split)


def make(m, k, n, dtype=torch.float16, bits=4, zp=False, count=1, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    words = k * bits // 32
    x = (10 * (torch.rand((m, k), device="cuda", generator=g) - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (count, words, n), dtype=torch.int32, device="cuda", generator=g)
    ws = (0.1 * torch.rand((k // 128, n), device="cuda", generator=g) + 0.01).to(dtype)
    wz = torch.randint(0, 2**bits, (k // 128, n), dtype=torch.int32, device="cuda", generator=g) if zp else None
    return x, wq, ws, wz


def check() -> int:
    bad = 0
    cases = [(64, 1024, 512), (33, 512, 200), (128, 2048, 704), (100, 1024, 1000), (256, 1024, 384), (200, 768, 260), (16, 512, 128), (1, 256, 64)]
    for dtype, bits, zp in ((torch.float16, 4, False), (torch.float16, 4, True), (torch.bfloat16, 4, False), (torch.float16, 8, False), (torch.float16, 8, True)):
        for m, k, n in cases:
            x, wq, ws, wz = make(m, k, n, dtype, bits, zp, seed=m + n)
            bias = (8 if bits == 4 else 128) if not zp else 0
            force(variant=1)
            ref = mixed_precision_gemm(x, wq[0], ws, wz, bits, bias, 128).float()
            for rows in (64, 128, 256):
                if rows < 256 and m > rows * 4:
                    continue
                for nt in (2, 3, 4):
                    for split in (1, 2, 3, 4, 8):
                        if split > 1 and (n % 4 or k // 128 < split):
                            continue
                        force(rows, nt, split, kernel=2, variant=5)
                        try:
                            got = mixed_precision_gemm(x, wq[0], ws, wz, bits, bias, 128).float()
                        except Exception as exc:  # noqa: BLE001
                            print(f"ERROR {dtype} int{bits} zp={zp} {m}x{k}x{n} rows {rows} nt {nt} split {split}: {str(exc)[:100]}")
                            bad += 1
                            continue
                        # one rounding of the fp32 sum to the 16-bit output on either side, sums in another order
                        tol = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * ref.abs().max().item() + 1e-3
                        err = (got - ref).abs().max().item()
                        if not err <= tol:
                            print(f"MISMATCH {dtype} int{bits} zp={zp} {m}x{k}x{n} rows {rows} nt {nt} split {split}: max|diff| {err:.4g} (tol {tol:.4g})")
                            bad += 1
    force()
    print(f"check: {bad} bad case(s)", flush=True)
    return bad


def timed(fn, est_us=30.0):
    iters = int(min(2000, max(40, 15000.0 / max(est_us, 1.0))))
    fn(20)
    vals = []
    for _ in range(3):
        fn(iters)
        vals.append(ms.value * 1e3)
    return sorted(vals)[1]


def sweep():
    dims = [(4096, 4096), (4096, 11008)] if QUICK else [(4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (8192, 28672), (5120, 13824)]
    rows_list = (64, 128, 256) if QUICK else (32, 48, 64, 96, 128, 192, 256, 384, 512)
    for k, n in dims:
        for m in rows_list:
            words = k * BITS // 32
            count = max(2, -(-COLD_BYTES // (words * n * 4)))
            x, wq, ws, _ = make(m, k, n, XDT, bits=BITS, count=count)
            out = torch.empty((m, n), dtype=XDT, device="cuda")
            md = create_mixed_precision_metadata(x, wq[0], ws, None, BITS, 8 if BITS == 4 else 128, 128)

            def run(iters):
                _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq[0], ws, None, md, (iters, ctypes.byref(ms))), "time")

            for regime, rot in (("warm", 1), ("cold", count)):
                LIB.conch_set_timing_rotation(rot, words * n * 4)
                res = {}
                force()
                res["auto"] = timed(run)
                if m <= 256:
                    force(variant=4)
                    res["decode"] = timed(run)
                force(kernel=1, variant=5)
                res["tiles"] = timed(run)
                best = (1e9, "")
                strips = []
                for nt in (2, 3, 4):
                    for split in (1, 2, 3, 4, 6, 8):
                        force(0, nt, split, kernel=2, variant=5)
                        try:
                            t = timed(run)
                        except Exception:  # noqa: BLE001
                            continue
                        strips.append(f"{64 * nt}/{split} {t:5.1f}")
                        best = min(best, (t, f"{64 * nt}/{split}"))
                force(kernel=2, variant=5)
                res["strip(plan)"] = timed(run)
                force()
                LIB.conch_set_timing_rotation(1, 0)
                others = min(v for kk, v in res.items() if kk not in ("auto", "strip(plan)"))
                print(f"{regime} {m:4d}x{k:5d}x{n:5d}: " + "  ".join(f"{kk} {v:6.1f}" for kk, v in res.items()) +
                      f"   best strip {best[1]} {best[0]:6.1f} ({others / best[0]:.2f}x the best other)   [" + "  ".join(strips) + "]", flush=True)


if __name__ == "__main__":
    rc = 0
    if "--check" in sys.argv:
        rc = check()
        if rc:
            sys.exit(1)
    sweep()
