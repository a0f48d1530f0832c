"""Where does a K step of the column-strip kernel go at batched-decode sizes?  Builds of the library with parts of the step compiled
out (gemm_mixed_strip.hip's CONCH_EXP_STRIP_* macros: WRONG results, timing only) against the product build, same process, same data.

    python tools/ab_strip_parts.py --build      (here: compiles the variants; the .so files travel with the tree)
    python tools/ab_strip_parts.py              (GPU box)            -> stdout (profiles/r05/mixed_mid_parts.txt)
    python tools/ab_strip_parts.py --build-decode / --decode: the decode kernel with 512-element K slices (not adopted)
(profiles/r05/mixed_mid_occupancy_ab.txt came from the same harness with a build that no longer exists as a switch: the 64-row tile's
three-stage ring and 128-register bound, and the 128-row tile's three-stage ring, are the default now.)
"""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

VARIANTS = {"nodq": "-DCONCH_EXP_STRIP_NODQ", "nomfma": "-DCONCH_EXP_STRIP_NOMFMA", "nobar": "-DCONCH_EXP_STRIP_NOBAR", "noxdma": "-DCONCH_EXP_STRIP_NOXDMA",
            "noread": "-DCONCH_EXP_STRIP_NOREAD"}

if "--build-decode" in sys.argv:
    from conch_amd import _build

    print(_build.build(variant="ms8", defines=("-DCONCH_EXP_MS_STEPS=8",), only=("gemm_mixed_skinny.hip",)))
    sys.exit(0)

if "--build" in sys.argv:
    from conch_amd import _build

    for name, macro in VARIANTS.items():
        print(_build.build(variant="strip_" + name, defines=(macro,), only=("gemm_mixed_strip.hip",)))
    sys.exit(0)

import torch  # noqa: E402

from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

LIBS = {"base": _C.load()}
if "--decode" not in sys.argv:
    for name in VARIANTS:
        LIBS[name] = _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_strip_{name}.so"))
ms = ctypes.c_float()


def case(m, k, n, rows, nt, split):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)
    res = {}
    for name, lib in LIBS.items():
        for key, val in ((_C.TUNE_GEMM_VARIANT, 5), (_C.TUNE_MIXED_KERNEL, 2), (_C.TUNE_MIXED_STRIP_ROWS, rows), (_C.TUNE_MIXED_TILE_NT, nt), (_C.TUNE_MIXED_SPLITK, split)):
            lib.conch_set_tuning(key, val)
        vals = []
        for it in (30, 300, 300, 300):
            _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (it, ctypes.byref(ms)), lib=lib), "time")
            vals.append(ms.value * 1e3)
        res[name] = sorted(vals[1:])[1]
    print(f"{m}x{k}x{n} rows {rows} tile {64 * nt} x{split}: " + "  ".join(f"{nm} {t:6.1f}" for nm, t in res.items()), flush=True)


def decode_case(m, k, n):
    """the decode-batch kernel with 512-element K slices (two workgroups per CU at 64 rows) against its 1024-element slices"""
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
    wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
    md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)
    res, outs = {}, {}
    for name, lib in (("base", LIBS["base"]), ("ms8", _C.load_library(Path(_C.LIB_PATH).with_name("libconch_amd_ms8.so")))):
        for key, val in ((_C.TUNE_GEMM_VARIANT, 4), (_C.TUNE_MIXED_KERNEL, 0), (_C.TUNE_MIXED_STRIP_ROWS, 0), (_C.TUNE_MIXED_TILE_NT, 0), (_C.TUNE_MIXED_SPLITK, 0)):
            lib.conch_set_tuning(key, val)
        out = outs.setdefault(name, torch.empty((m, n), dtype=torch.float16, device="cuda"))
        vals = []
        for it in (30, 300, 300, 300):
            _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (it, ctypes.byref(ms)), lib=lib), "time")
            vals.append(ms.value * 1e3)
        res[name] = sorted(vals[1:])[1]
    torch.cuda.synchronize()
    err = (outs["base"].float() - outs["ms8"].float()).abs().max().item()
    print(f"decode kernel {m}x{k}x{n}: 1024-k slices {res['base']:6.1f} us   512-k slices {res['ms8']:6.1f} us   max|diff| {err:.3g}", flush=True)


if __name__ == "__main__":
    if "--decode" in sys.argv:
        for m in (16, 32, 64, 128, 256):
            for k, n in ((4096, 4096), (4096, 11008), (8192, 8192)):
                decode_case(m, k, n)
        sys.exit(0)
    for m, rows in ((64, 64), (128, 128), (256, 256)):
        for nt, split in ((2, 1), (3, 1), (3, 4), (4, 4)):
            case(m, 4096, 11008, rows, nt, split)
