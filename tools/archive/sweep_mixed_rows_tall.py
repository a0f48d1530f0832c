"""mixed_precision_gemm (int4 x fp16) a little above one row of tiles (M = 320 .. 1024): does a SHORTER strip tile (128 / 64 rows:
fewer padded rows, more workgroups) beat the 256-row tile?  Forced tile rows x width x K slices against the automatic pick, cold weights.
usage: python tools/archive/sweep_mixed_rows_tall.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402
from sweep_mixed_mid import COLD_BYTES, LIB, force, make, ms, timed  # noqa: E402

FULL = "--full" in sys.argv  # every timing on one line (the fit's input: profiles/r05/mixed_rows_tall_sweep.txt)
DIMS = ((4096, 11008), (4096, 4096), (8192, 8192), (11008, 4096), (5120, 13824)) if FULL else ((4096, 11008), (4096, 4096), (8192, 8192))
MS = (288, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536) if FULL else (320, 384, 448, 512, 640, 768, 1024)
for arg in sys.argv[1:]:
    if arg.startswith("--ms="):  # e.g. --ms=144,160,192,224: other row counts (one row of 256-row tiles against two or three shorter ones)
        MS = tuple(int(v) for v in arg[5:].split(","))
for k, n in DIMS:
    for m in MS:
        words = k // 8
        count = max(2, -(-COLD_BYTES // (words * n * 4)))
        x, wq, ws, _ = make(m, k, n, count=count)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        md = create_mixed_precision_metadata(x, wq[0], ws, None, 4, 8, 128)

        def run(iters):
            _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq[0], ws, None, md, (iters, ctypes.byref(ms))), "time")

        LIB.conch_set_timing_rotation(count, words * n * 4)
        force()
        auto = timed(run)
        force(kernel=1, variant=5)
        tiles = timed(run)
        res = {}
        for rows in (64, 128, 256):
            for nt in (2, 3, 4):
                for split in (1, 2, 3, 4, 8):
                    force(rows, nt, split, kernel=2, variant=5)
                    try:
                        res[(rows, 64 * nt, split)] = timed(run)
                    except Exception:  # noqa: BLE001
                        pass
        force()
        LIB.conch_set_timing_rotation(1, 0)
        best = sorted(res.items(), key=lambda kv: kv[1])[:4]
        best256 = min(v for kk, v in res.items() if kk[0] == 256)
        print(f"cold {m:5d}x{k}x{n}: auto {auto:6.1f}  tiles {tiles:6.1f}  best 256-row strip {best256:6.1f}  best: " + "  ".join(f"{kk} {v:.1f}" for kk, v in best) +
              ("   all: " + " ".join(f"{kk[0]}/{kk[1]}/{kk[2]}={v:.1f}" for kk, v in res.items()) if FULL else ""), flush=True)
