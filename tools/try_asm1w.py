"""Bring-up / A-B driver of the one-wave-per-SIMD assembly kernel (variant 7) against the 256 x 256-tile kernel (variant 5).

  python tools/try_asm1w.py [--check-only] [--time-only] [--rounds R]

Check: bit equality with variant 5 on single tiles, ragged M / N, several K, both output dtypes, scalar / vector scales;
where the two differ, WHERE (rows, columns, 16 x 16 tile coordinates inside the 256 x 352 tile) is printed.
Time: interleaved rounds in one process (guide rule 24) on C3 and two more shapes, kernel-only (conch_time_scaled_gemm).
"""
from __future__ import annotations

import argparse
import ctypes
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kgemm  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata, scaled_gemm  # noqa: E402


def inputs(m, k, n, vec_a=True, vec_b=True, seed=0, int8=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if int8:
        a = torch.randint(-32, 32, (m, k), device="cuda", generator=g, dtype=torch.int8)
        bt = torch.randint(-32, 32, (n, k), device="cuda", generator=g, dtype=torch.int8)
    else:
        a = (0.25 * torch.rand((m, k), device="cuda", generator=g)).to(torch.float8_e4m3fn)
        bt = (0.25 * torch.rand((n, k), device="cuda", generator=g)).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1) if vec_a else (1, 1), device="cuda", generator=g) + 0.01
    sb = 0.25 * torch.rand((n, 1) if vec_b else (1, 1), device="cuda", generator=g) + 0.01
    return a, bt.T, sa, sb


def run(variant, a, b, sa, sb, dt):
    _C.set_gemm_variant(variant)
    try:
        out = scaled_gemm(a, b, sa, sb, dt)
        torch.cuda.synchronize()
        return out
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)


def where(diff):
    rows = diff.any(dim=1).nonzero().flatten()
    cols = diff.any(dim=0).nonzero().flatten()
    tiles = {}
    idx = diff.nonzero()
    for r, c in idx[:200000].tolist():
        key = ((r % 256) // 16, (c % 352) // 16)
        tiles[key] = tiles.get(key, 0) + 1
    top = sorted(tiles.items(), key=lambda kv: -kv[1])[:12]
    return (f"rows {rows.min().item()}..{rows.max().item()} ({rows.numel()} rows), cols {cols.min().item()}..{cols.max().item()} "
            f"({cols.numel()} cols); (m tile, n tile) inside the 256x352 tile -> count: {top}")


def check() -> int:
    bad = 0
    cases = [(256, 512, 352), (256, 512, 176), (256, 1024, 704), (512, 512, 352), (128, 512, 352), (300, 768, 400), (1000, 2048, 1008),
             (2048, 4096, 2816), (4096, 4096, 11008)]
    for m, k, n in cases:
        for dt in (torch.bfloat16, torch.float16):
            for vec_a, vec_b in ((True, True), (False, False)):
                if (dt is torch.float16 or not vec_a) and m * n > 3e6:
                    continue
                a, b, sa, sb = inputs(m, k, n, vec_a, vec_b, seed=m + n)
                ref = run(_C.VARIANT_MFMA_PINGPONG2, a, b, sa, sb, dt)
                got = run(_C.VARIANT_MFMA_ASM1W, a, b, sa, sb, dt)
                diff = ref.view(torch.int16) != got.view(torch.int16)
                nd = int(diff.sum().item())
                tag = f"{m}x{k}x{n} {str(dt)[6:]} sa{'v' if vec_a else 's'} sb{'v' if vec_b else 's'}"
                if nd:
                    bad += 1
                    err = (ref.float() - got.float()).abs().max().item()
                    print(f"MISMATCH {tag}: {nd} of {m * n} elements differ, max |diff| {err:.4g} (max |ref| {ref.float().abs().max().item():.4g}); {where(diff)}", flush=True)
                else:
                    print(f"ok       {tag}", flush=True)
    for m, k, n in ((256, 512, 352), (300, 768, 400), (1000, 2048, 1008), (4096, 4096, 11008)):
        for dt in (torch.bfloat16, torch.float16):
            a, b, sa, sb = inputs(m, k, n, seed=7, int8=True)
            ref = run(_C.VARIANT_MFMA_PINGPONG2, a, b, sa, sb, dt)
            got = run(_C.VARIANT_MFMA_ASM1W, a, b, sa, sb, dt)
            diff = ref.view(torch.int16) != got.view(torch.int16)
            nd = int(diff.sum().item())
            bad += 1 if nd else 0
            print((f"MISMATCH {m}x{k}x{n} int8 {str(dt)[6:]}: {nd} differ; {where(diff)}" if nd else f"ok       {m}x{k}x{n} int8 {str(dt)[6:]}"), flush=True)
    for nt in (9, 7):
        for m, k, n in ((256, 512, 32 * nt), (300, 768, 400), (1000, 2048, 1008), (4096, 4096, 11008), (8192, 1024, 3584)):
            for int8 in (False, True):
                a, b, sa, sb = inputs(m, k, n, seed=11, int8=int8)
                ref = run(_C.VARIANT_MFMA_PINGPONG2, a, b, sa, sb, torch.bfloat16)
                set_width(nt)
                try:
                    got = run(_C.VARIANT_MFMA_ASM1W, a, b, sa, sb, torch.bfloat16)
                finally:
                    set_width(0)
                diff = ref.view(torch.int16) != got.view(torch.int16)
                nd = int(diff.sum().item())
                bad += 1 if nd else 0
                print((f"MISMATCH {m}x{k}x{n} width {32 * nt} {'int8' if int8 else 'fp8'}: {nd} differ; {where(diff)}" if nd
                       else f"ok       {m}x{k}x{n} width {32 * nt} {'int8' if int8 else 'fp8'}"), flush=True)
    for m, k, n in ((512, 1024, 704), (4096, 4096, 11008)):
        a, b, sa, sb = inputs(m, k, n, seed=5)
        ref = run(_C.VARIANT_MFMA_PINGPONG2, a, b, sa, sb, torch.bfloat16)
        set_alt(1)
        try:
            got = run(_C.VARIANT_MFMA_ASM1W, a, b, sa, sb, torch.bfloat16)
        finally:
            set_alt(0)
        ok = torch.equal(ref.view(torch.int16), got.view(torch.int16))
        bad += 0 if ok else 1
        print(f"{'ok      ' if ok else 'MISMATCH'} {m}x{k}x{n} alt twin", flush=True)
    return bad


def set_alt(on: int) -> None:
    fn = _C.load().conch_debug_gemm1w_alt
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int]
    fn(on)


def set_width(nt: int) -> None:
    fn = _C.load().conch_debug_gemm1w_width
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int]
    fn(nt)


def timer(variant, a, b, sa, sb, out, iters):
    """variant 7 = the assembly kernel (automatic width), 8 = its A/B twin (generator's `alt` choice), 711 / 709 / 707 = forced width"""
    md = create_scaled_metadata(a, b, sa, sb, out.dtype)
    ms = ctypes.c_float()
    set_alt(1 if variant == 8 else 0)
    set_width(variant - 700 if variant > 700 else 0)
    _C.set_gemm_variant(7 if variant >= 8 else variant)
    try:
        _C.check(kgemm._scaled_gemm_call("conch_time_scaled_gemm", out, a, b, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
    finally:
        _C.set_gemm_variant(_C.VARIANT_AUTO)
        set_alt(0)
        set_width(0)
    return ms.value


def bench(rounds: int, alt_only: bool = False) -> None:
    for m, k, n in ((4096, 4096, 11008), (8192, 8192, 3584), (4096, 8192, 4096), (8192, 8192, 28672), (8192, 8192, 7168), (2048, 4096, 11008),
                    (8192, 4096, 4096)):
        a, b, sa, sb = inputs(m, k, n)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        flops = 2.0 * m * n * k
        iters = max(10, int(0.15 / (flops / 3.0e15)))
        arms = (5, 711, 8) if alt_only else (5, 711, 709, 707)
        for v in arms:  # load the chip first
            t_end = time.perf_counter() + 0.7
            while time.perf_counter() < t_end:
                timer(v, a, b, sa, sb, out, iters)
        res = {v: [] for v in arms}
        for _ in range(rounds):
            for v in arms:
                res[v].append(timer(v, a, b, sa, sb, out, iters))
        for v, name in ((5, "256x256 two waves/SIMD (variant 5)"), (711, "256x352 one wave/SIMD  (variant 7)"), (709, "256x288 one wave/SIMD             "),
                        (707, "256x224 one wave/SIMD             "), (8, "256x352, the A/B twin (alt)       ")):
            if v not in res:
                continue
            xs = sorted(res[v])
            med = xs[len(xs) // 2]
            print(f"{m}x{k}x{n}  {name}: median {med * 1e3:8.1f} us  min {xs[0] * 1e3:8.1f} us   {flops / med / 1e9:7.1f} TFLOP/s = {flops / med / 1e9 / 5000:.3f} of 5 PF", flush=True)


def bench_int8(rounds: int) -> None:
    m, k, n = 4096, 4096, 11008
    a, b, sa, sb = inputs(m, k, n, int8=True)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    flops = 2.0 * m * n * k
    for v in (5, 7):
        t_end = time.perf_counter() + 0.7
        while time.perf_counter() < t_end:
            timer(v, a, b, sa, sb, out, 200)
    res = {5: [], 7: []}
    for _ in range(rounds):
        for v in (5, 7):
            res[v].append(timer(v, a, b, sa, sb, out, 200))
    for v in (5, 7):
        xs = sorted(res[v])
        med = xs[len(xs) // 2]
        print(f"int8 {m}x{k}x{n} variant {v}: median {med * 1e3:8.1f} us   {flops / med / 1e9:7.1f} TOP/s = {flops / med / 1e9 / 5000:.3f} of 5 POP/s", flush=True)


def probe(m=4096, k=4096, n=11008, seconds=1.5, alt=False, int8=False, grid=0) -> None:
    """In-kernel stamps of the diagnostic twin under sustained load: where a workgroup's time goes and the clock it holds."""
    lib = _C.load()
    fn = lib.conch_debug_gemm1w_probe
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    a, b, sa, sb = inputs(m, k, n, int8=int8)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    nwg = -(-m // 256) * -(-n // 352)
    buf = torch.zeros((nwg, 5, 2), dtype=torch.int64, device="cuda")
    fn(buf.data_ptr())
    lib.conch_debug_gemm1w_grid(grid)
    try:
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            timer(8 if alt else _C.VARIANT_MFMA_ASM1W, a, b, sa, sb, out, 50)
        torch.cuda.synchronize()
    finally:
        fn(None)
        lib.conch_debug_gemm1w_grid(0)
    st = buf.cpu().double()
    clk, rt = st[:, :, 0], st[:, :, 1]
    names = ["entry -> K loop", "K loop", "K loop end -> stores issued", "stores issued -> retired"]
    steps = k // 128
    print(f"probe{' (alt twin)' if alt else ''}{' int8' if int8 else ''}{f' grid {grid}' if grid else ''} {m}x{k}x{n}: {nwg} workgroups, {steps} K steps (medians over workgroups)")
    for i, name in enumerate(names):
        us = ((rt[:, i + 1] - rt[:, i]) / 100.0).median().item()
        cyc = (clk[:, i + 1] - clk[:, i]).median().item()
        extra = f"  = {cyc / steps:7.0f} cycles per step ({SLOT_CYC * 88} are MFMA)" if i == 1 else ""
        print(f"  {name:32s} {us:8.2f} us  {cyc:10.0f} cycles  clock {cyc / max(us, 1e-9) / 1e3:5.2f} GHz{extra}")
    wg = torch.arange(nwg)
    per_xcd = []
    for x in range(8):
        sel = (wg % 8) == x
        loop_us = ((rt[sel, 2] - rt[sel, 1]) / 100.0)
        ghz = ((clk[sel, 2] - clk[sel, 1]) / (rt[sel, 2] - rt[sel, 1]) / 10.0)
        total = ((rt[sel, 4] - rt[sel, 0]) / 100.0)
        per_xcd.append(f"{x}: loop {loop_us.median().item():.1f} (max {loop_us.max().item():.1f}) us @ {ghz.median().item():.2f} GHz, workgroup {total.median().item():.1f} (max {total.max().item():.1f})")
    print("  per XCD (workgroup id % 8): " + "; ".join(per_xcd))
    t0 = rt[:, 0].min()
    first = ((rt[:, 0] - t0) / 100.0)
    last = ((rt[:, 4] - t0) / 100.0)
    print(f"  workgroup entries span {first.max().item():.1f} us (second-round entries from {first.sort().values[min(256, nwg - 1)].item():.1f} us), last exit {last.max().item():.1f} us")


SLOT_CYC = 32


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--alt", action="store_true", help="time the A/B twin instead of the narrower widths")
    args = ap.parse_args()
    rc = 0
    if not args.time_only:
        rc = check()
        print(f"check: {rc} mismatching case(s)", flush=True)
    if not args.check_only:
        bench(args.rounds, alt_only=args.alt)
        bench_int8(args.rounds)
    if args.probe:
        probe()
        probe(alt=True)
        probe(int8=True)
        probe(2048, 4096, 11008)
        probe(2048, 4096, 11008, grid=128)
        probe(4096, 4096, 11008, grid=128)
        probe(8192, 8192, 28672)
    sys.exit(1 if rc else 0)
