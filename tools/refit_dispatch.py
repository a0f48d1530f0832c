"""Does every dispatcher still pick (close to) the fastest kernel?  Re-fit aid for conch_amd/csrc/dispatch_fit.hpp + gemm_asm.hip's kAsm1wFit.

  python tools/refit_dispatch.py [--quick]            (GPU box; ~1-2 min)

For each family the candidates the dispatcher chooses between are FORCED through the tuning keys and timed on the sweep shapes
(kernel-only, back-to-back launches, interleaved rounds in one process: guide rule 24), the automatic choice is timed beside
them, and the worst ratio auto / best candidate is reported per family.  A ratio above ~1.10 names the rows of the table that
need refitting and the sweep script to refit them with (their docstrings say which profile file they write).

  scaled_gemm fp8 / int8   variants 5 (256x256 tiles), 6 (128x128 tiles), 4 (split-K skinny, M <= 256), 7 (256x352 assembly tiles)
  mixed_precision_gemm     LDS-tiled kernel (tile widths 2..4, auto split), strip kernel, decode kernel (M <= 256)
"""
from __future__ import annotations

import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quick_bench import time_mixed, time_scaled  # noqa: E402

from conch_amd import _C  # noqa: E402

QUICK = "--quick" in sys.argv
ROUNDS = 3 if QUICK else 5


def med(fn, rounds=ROUNDS):
    return statistics.median(fn() for _ in range(rounds))


def scaled_family(dtype, name):
    shapes = [(16, 4096, 4096), (64, 4096, 11008), (128, 4096, 4096), (256, 4096, 11008), (512, 4096, 11008), (1024, 8192, 8192),
              (2048, 4096, 11008), (4096, 4096, 11008), (4096, 8192, 4096), (8192, 8192, 3584)]
    if not QUICK:
        shapes += [(32, 8192, 8192), (192, 8192, 8192), (768, 5120, 13824), (8192, 8192, 8192)]
    worst = (0.0, None)
    print(f"== scaled_gemm {name} (us; * = fastest candidate)")
    for m, k, n in shapes:
        cand = {}
        for v, label in ((5, "t256"), (6, "t128"), (4, "skinny"), (7, "asm352")):
            try:
                time_scaled(m, k, n, dtype, v, iters=5)
                cand[label] = med(lambda v=v: time_scaled(m, k, n, dtype, v, iters=30)) * 1e3
            except (NotImplementedError, ValueError):
                pass
        time_scaled(m, k, n, dtype, 0, iters=5)
        auto = med(lambda: time_scaled(m, k, n, dtype, 0, iters=30)) * 1e3
        best_label, best = min(cand.items(), key=lambda kv: kv[1])
        ratio = auto / best
        if ratio > worst[0]:
            worst = (ratio, (m, k, n))
        line = "  ".join(f"{lb}{'*' if lb == best_label else ' '}{t:8.1f}" for lb, t in cand.items())
        print(f"  {m:5d}x{k:5d}x{n:6d}  auto {auto:8.1f}  ratio {ratio:5.2f}   {line}", flush=True)
    print(f"  worst automatic pick: {worst[0]:.2f}x at {worst[1]}  (rows ScaledTiles / ScaledSkinny / kAsm1wFit; refit: tools/dispatch_cold_sweep.py, "
          "tools/sweep_mid_split.py, tools/try_asm1w.py)")
    return worst[0]


def mixed_family():
    shapes = [(16, 4096, 11008), (128, 4096, 4096), (256, 8192, 8192), (512, 4096, 11008), (1024, 4096, 11008), (1024, 8192, 28672),
              (4096, 8192, 4096)]
    if not QUICK:
        shapes += [(64, 4096, 11008), (768, 4096, 11008), (2048, 4096, 11008), (1024, 5120, 13824), (48, 8192, 8192), (96, 4096, 4096), (128, 8192, 28672),
                   (192, 5120, 13824)]
    worst = (0.0, None)
    print("== mixed_precision_gemm int4 x fp16 (us; * = fastest candidate)")
    for m, k, n in shapes:
        cand = {}

        def forced(kernel, nt=0, variant=2):
            _C.set_tuning(_C.TUNE_MIXED_KERNEL, kernel)
            _C.set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
            try:
                time_mixed(m, k, n, torch.float16, 4, variant, iters=5)
                return med(lambda: time_mixed(m, k, n, torch.float16, 4, variant, iters=30)) * 1e3
            finally:
                _C.set_tuning(_C.TUNE_MIXED_KERNEL, 0)
                _C.set_tuning(_C.TUNE_MIXED_TILE_NT, 0)

        for nt in (2, 3, 4):
            try:
                cand[f"tiles{64 * nt}"] = forced(1, nt)
            except (NotImplementedError, ValueError):
                pass
        try:
            cand["strip"] = forced(2)  # m > 256: the unsplit 256-row tile; else strip_plan's tile height, width and K slices
        except (NotImplementedError, ValueError):
            pass
        if m <= 256:
            try:
                cand["decode"] = forced(0, 0, variant=4)
            except (NotImplementedError, ValueError):
                pass
        time_mixed(m, k, n, torch.float16, 4, 0, iters=5)
        auto = med(lambda: time_mixed(m, k, n, torch.float16, 4, 0, iters=30)) * 1e3
        best_label, best = min(cand.items(), key=lambda kv: kv[1])
        ratio = auto / best
        if ratio > worst[0]:
            worst = (ratio, (m, k, n))
        line = "  ".join(f"{lb}{'*' if lb == best_label else ' '}{t:8.1f}" for lb, t in cand.items())
        print(f"  {m:5d}x{k:5d}x{n:6d}  auto {auto:8.1f}  ratio {ratio:5.2f}   {line}", flush=True)
    print(f"  worst automatic pick: {worst[0]:.2f}x at {worst[1]}  (rows MixedDecode / MixedSplit / MixedTileSplit / MixedStrip / MixedStripSplit; refit: "
          "tools/archive/sweep_mixed_split.py, sweep_mixed_nt_split.py, sweep_mixed_strip.py, tools/sweep_mixed_mid.py)")
    return worst[0]


if __name__ == "__main__":
    w = [scaled_family(torch.float8_e4m3fn, "fp8"), scaled_family(torch.int8, "int8"), mixed_family()]
    print(f"worst ratios: fp8 {w[0]:.2f}  int8 {w[1]:.2f}  mixed {w[2]:.2f}   (auto may beat every forced candidate where it adds a K split of its own: ratio < 1)")
