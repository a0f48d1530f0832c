"""Bring-up / A-B driver of the one-wave-per-SIMD int4 x fp16 assembly kernel (CONCH_TUNE_MIXED_KERNEL = 3) against the HIP kernels.

  python tools/try_mixed1w.py [--check-only] [--time-only] [--probe]

Check: bit equality with the column-strip / LDS-tiled HIP kernels (same dequantised weights, same MFMA, same K order per output)
on single tiles, ragged M / N, both tile widths; where they differ, WHERE.  Time: interleaved rounds, kernel-only.
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
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata, mixed_precision_gemm  # noqa: E402


def inputs(m, k, n, seed=0, zero_x=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (10 * (torch.rand((m, k), device="cuda", generator=g) - 0.3)).to(torch.float16)
    if zero_x:
        x.zero_()
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda", generator=g)
    ws = (0.1 * torch.rand((k // 128, n), device="cuda", generator=g) + 0.01).to(torch.float16)
    return x, wq, ws


def run(kernel, nt, x, wq, ws):
    _C.set_gemm_variant(_C.VARIANT_MFMA_PINGPONG2 if kernel else _C.VARIANT_AUTO)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 1 if kernel else 0)
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
    try:
        out = mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
        torch.cuda.synchronize()
        return out
    finally:
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
        _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
        _C.set_gemm_variant(_C.VARIANT_AUTO)


def where(diff, tile_n):
    rows = diff.any(dim=1).nonzero().flatten()
    cols = diff.any(dim=0).nonzero().flatten()
    tiles = {}
    for r, c in diff.nonzero()[:100000].tolist():
        key = ((r % 256) // 16, (c % tile_n) // 16)
        tiles[key] = tiles.get(key, 0) + 1
    top = sorted(tiles.items(), key=lambda kv: -kv[1])[:10]
    return f"rows {rows.min().item()}..{rows.max().item()} ({rows.numel()}), cols {cols.min().item()}..{cols.max().item()} ({cols.numel()}); (m tile, n tile) -> count {top}"


def check() -> int:
    bad = 0
    for m, k, n in [(256, 256, 192), (256, 256, 256), (256, 512, 64), (512, 1024, 384), (300, 384, 400), (1000, 2048, 1008), (1024, 4096, 11008),
                    (4096, 8192, 4096)]:
        x, wq, ws = inputs(m, k, n, seed=m + n)
        ref = run(2, 0, x, wq, ws)  # column-strip HIP kernel
        for nt in (3, 4):
            got = run(3, nt, x, wq, ws)
            diff = ref.view(torch.int16) != got.view(torch.int16)
            nd = int(diff.sum().item())
            if nd:
                bad += 1
                err = (ref.float() - got.float()).abs().max().item()
                print(f"MISMATCH {m}x{k}x{n} W={nt}: {nd} of {m * n} differ, max |diff| {err:.4g} (max |ref| {ref.float().abs().max().item():.4g}); {where(diff, 64 * nt)}", flush=True)
            else:
                print(f"ok       {m}x{k}x{n} W={nt}", flush=True)
    return bad


def timer(kernel, nt, x, wq, ws, out, iters):
    md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)
    ms = ctypes.c_float()
    _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
    _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
    try:
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms))), "time")
    finally:
        _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)
        _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)
    return ms.value


def bench(rounds: int) -> None:
    for (m, k, n), zero in (((1024, 4096, 11008), False), ((1024, 4096, 11008), True), ((4096, 8192, 4096), False), ((2048, 4096, 11008), False),
                            ((512, 4096, 11008), False), ((1024, 8192, 28672), False)):
        x, wq, ws = inputs(m, k, n, zero_x=zero)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        flops = 2.0 * m * n * k
        arms = [("auto (HIP)", 0, 0), ("asm W=3", 3, 3), ("asm W=4", 3, 4)]
        iters = max(10, int(0.1 / (flops / 1.2e15)))
        for _, kern, nt in arms:
            t_end = time.perf_counter() + 0.5
            while time.perf_counter() < t_end:
                timer(kern, nt, x, wq, ws, out, iters)
        res = {a[0]: [] for a in arms}
        for _ in range(rounds):
            for name, kern, nt in arms:
                res[name].append(timer(kern, nt, x, wq, ws, out, iters))
        for name in res:
            xs = sorted(res[name])
            med = xs[len(xs) // 2]
            print(f"{m}x{k}x{n}{' ZERO activations' if zero else ''}  {name:12s}: median {med * 1e3:8.1f} us   {flops / med / 1e9:7.1f} TFLOP/s = {flops / med / 1e9 / 2500:.3f} of 2.5 PF", flush=True)


def probe(m=1024, k=4096, n=11008, nt=3) -> None:
    lib = _C.load()
    fn = lib.conch_debug_mixed1w_probe
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    x, wq, ws = inputs(m, k, n)
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    nwg = -(-m // 256) * -(-n // (64 * nt))
    buf = torch.zeros((nwg, 4, 2), dtype=torch.int64, device="cuda")
    fn(buf.data_ptr())
    try:
        t_end = time.perf_counter() + 1.5
        while time.perf_counter() < t_end:
            timer(3, nt, x, wq, ws, out, 50)
        torch.cuda.synchronize()
    finally:
        fn(None)
    st = buf.cpu().double()
    clk, rt = st[:, :, 0], st[:, :, 1]
    steps = k // 64
    print(f"probe {m}x{k}x{n} W={nt}: {nwg} workgroups, {steps} K steps (medians)")
    for i, name in enumerate(["entry -> K loop", "K loop + epilogue until stores issued", "stores issued -> retired"]):
        us = ((rt[:, i + 1] - rt[:, i]) / 100.0).median().item()
        cyc = (clk[:, i + 1] - clk[:, i]).median().item()
        extra = f"  ~ {cyc / steps:7.0f} cycles per step ({512 * nt} are MFMA)" if i == 1 else ""
        print(f"  {name:40s} {us:8.2f} us  {cyc:10.0f} cycles  clock {cyc / max(us, 1e-9) / 1e3:5.2f} GHz{extra}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    rc = 0
    if not args.time_only:
        rc = check()
        print(f"check: {rc} mismatching case(s)", flush=True)
    if not args.check_only:
        bench(args.rounds)
    if args.probe:
        probe()
        probe(4096, 8192, 4096, 4)
    sys.exit(1 if rc else 0)
