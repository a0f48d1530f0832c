"""Timing experiments on the column-strip mixed GEMM: variant builds of the library (python -m conch_amd._build --variant NAME --only
gemm_mixed_strip.hip -DCONCH_EXP_STRIP_...) against the product build, interleaved rounds in one process.  Most variants compute WRONG
results by construction (parts of the K loop removed): they say what the loop waits for.  usage: ab_strip_variants.py name [name ...]"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

NAMES = [a for a in sys.argv[1:] if not a.isdigit()]
ROUNDS = next((int(a) for a in sys.argv[1:] if a.isdigit()), 5)
LIBS = {"base": _C.load()}
for nm in NAMES:
    LIBS[nm] = _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{nm}.so"))


def case(m, k, n, dtype, bits, nt):
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (k * bits // 32, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, bits, 8 if bits == 4 else 128, 128)
    ms = ctypes.c_float()
    for lib in LIBS.values():
        lib.conch_set_tuning(_C.TUNE_MIXED_KERNEL, 2)
        lib.conch_set_tuning(_C.TUNE_MIXED_TILE_NT, nt)
        lib.conch_set_tuning(_C.TUNE_MIXED_SPLITK, 1)

    def run(lib, iters):
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (iters, ctypes.byref(ms)), lib=lib), "time")
        return ms.value

    iters = int(min(2000, max(30, 30.0 / max(run(LIBS["base"], 20), 1e-4))))
    for lib in LIBS.values():
        for _ in range(2):
            run(lib, iters)
    res = {nm: [] for nm in LIBS}
    for _ in range(ROUNDS):
        for nm, lib in LIBS.items():
            res[nm].append(run(lib, iters))
    base = statistics.median(res["base"])
    steps = k // 64
    # variants built with -DCONCH_CLOCK_PROBE export the K-loop stamps: clock the chip held and the loop's cycles per step
    clk = {}
    for nm, lib in LIBS.items():
        fn = getattr(lib, "conch_debug_probe_mixed_strip", None) if nm != "base" else None
        try:
            fn = lib.conch_debug_probe_mixed_strip
        except AttributeError:
            continue
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
        for _ in range(3):
            run(lib, iters)
        blocks = min(4096, ((m + 255) // 256) * ((n + 64 * nt - 1) // (64 * nt)))
        buf = (ctypes.c_ulonglong * (8 * blocks))()
        assert fn(buf, blocks) == 0
        cs, ls = [], []
        for b_ in range(blocks):
            t0, r0, t1, r1 = buf[8 * b_:8 * b_ + 4]
            if r1 > r0:
                cs.append((t1 - t0) / (r1 - r0) * 0.1)
                ls.append((t1 - t0) / steps)
        if cs:
            clk[nm] = (statistics.median(cs), statistics.median(ls))
    line = f"strip {m}x{k}x{n} int{bits} {str(dtype)[6:]} nt={nt}: " + "  ".join(
        f"{nm} {statistics.median(v) * 1e3:6.1f} us ({statistics.median(v) / base:.3f}; {statistics.median(v) * 1e6 / steps:5.0f} ns/step)" for nm, v in res.items())
    print(line, flush=True)
    if clk:
        print("      in-loop: " + "  ".join(f"{nm} {c:.2f} GHz {cyc:5.0f} cyc/step" for nm, (c, cyc) in clk.items()), flush=True)


if __name__ == "__main__":
    case(1024, 4096, 11008, torch.float16, 4, 3)
    case(4096, 8192, 4096, torch.float16, 4, 4)
