"""Does the dispatcher pick the fastest kernel when the weights are NOT cache-resident?

The cost models in capi.hip (scaled: split-K skinny / 128x128 tiles / 256x256 tiles; mixed: decode kernel / tile kernel) were
fitted on back-to-back launches of ONE weight buffer, which stays in L2 + the 256 MiB Infinity Cache.  A decode step streams every
layer's weights from HBM.  This sweep times every candidate kernel in both regimes -- "warm" = one buffer, "cold" = the timing
helpers cycling through enough weight buffers to exceed 600 MB (conch_set_timing_rotation) -- and reports, per shape, the
dispatcher's pick, the fastest candidate in each regime and what the pick costs against the cold-fastest.
usage: python tools/dispatch_cold_sweep.py [--quick]"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata, create_scaled_metadata  # noqa: E402

LIB = _C.load()
COLD_BYTES = 640 << 20
QUICK = "--quick" in sys.argv
ms = ctypes.c_float()


def timed(fn, est_us):
    iters = int(min(3000, max(40, 20000.0 / max(est_us, 1.0))))
    fn(20)
    vals = []
    for _ in range(3):
        fn(iters)
        vals.append(ms.value * 1e3)
    return sorted(vals)[1]


def scaled_case(m, k, n, dtype):
    torch.manual_seed(0)
    count = max(2, -(-COLD_BYTES // (n * k)))
    if dtype == torch.int8:
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (count, n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(dtype)
        bt = (0.25 * torch.rand((count, n, k), device="cuda")).to(dtype)
    sa, sb = 0.25 * torch.rand((m, 1), device="cuda"), 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    b = bt[0].T
    md = create_scaled_metadata(a, b, sa, sb, torch.bfloat16)

    def run(iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, b, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")

    res = {}
    for regime, rot in (("warm", 1), ("cold", count)):
        LIB.conch_set_timing_rotation(rot, n * k)
        # mid128 = the 128 x 128 tiles with their own K-split rule (round 4), mid128w = whole K per workgroup, mid128x2 = whole K, 2-stage loop
        for name, variant in (("auto", 0), ("skinny", 4), ("mid128", 6), ("mid128w", 6), ("mid128x2", 6), ("tile256", 5)):
            if variant == 4 and m > 256:
                continue
            _C.set_gemm_variant(variant)
            _C.set_tuning(_C.TUNE_MID_STAGES, 2 if name == "mid128x2" else 0)
            _C.set_tuning(_C.TUNE_MID_SPLITK, 0 if name in ("auto", "mid128") else 1)
            try:
                res[(regime, name)] = timed(run, 30.0)
            except Exception:  # noqa: BLE001 -- a forced variant whose contract the shape does not meet
                pass
    _C.set_gemm_variant(0)
    LIB.conch_set_timing_rotation(1, 0)
    _C.set_tuning(_C.TUNE_MID_STAGES, 0)
    _C.set_tuning(_C.TUNE_MID_SPLITK, 0)
    report(f"scaled {str(dtype)[6:]:13s} {m:5d}x{k}x{n}", res, ("skinny", "mid128", "mid128w", "mid128x2", "tile256"))


def mixed_case(m, k, n, dtype=torch.float16, bits=4):
    torch.manual_seed(0)
    words = k * bits // 32
    count = max(2, -(-COLD_BYTES // (words * n * 4)))
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    wq = torch.randint(-2**31, 2**31 - 1, (count, words, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq[0], ws, None, bits, 8 if bits == 4 else 128, 128)

    def run(iters):
        _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq[0], ws, None, md, (iters, ctypes.byref(ms))), "time")

    res = {}
    for regime, rot in (("warm", 1), ("cold", count)):
        LIB.conch_set_timing_rotation(rot, words * n * 4)
        for name, variant in (("auto", 0), ("decode", 4), ("tile", 5), ("tile/2", 5), ("tile/4", 5), ("tile/8", 5)):
            if variant == 4 and m > 256:
                continue
            _C.set_gemm_variant(variant)
            _C.set_tuning(_C.TUNE_MIXED_SPLITK, int(name.split("/")[1]) if "/" in name else (1 if name == "tile" else 0))
            try:
                res[(regime, name)] = timed(run, 30.0)
            except Exception:  # noqa: BLE001
                pass
    _C.set_gemm_variant(0)
    LIB.conch_set_timing_rotation(1, 0)
    _C.set_tuning(_C.TUNE_MIXED_SPLITK, 0)
    report(f"mixed  int{bits} x {str(dtype)[6:]:8s} {m:5d}x{k}x{n}", res, ("decode", "tile", "tile/2", "tile/4", "tile/8"))


def report(label, res, names):
    def best(regime):
        cand = {nm: res[(regime, nm)] for nm in names if (regime, nm) in res}
        nm = min(cand, key=cand.get)
        return nm, cand[nm], cand

    wn, wt, _ = best("warm")
    cn, ct, cc = best("cold")
    auto_w, auto_c = res[("warm", "auto")], res[("cold", "auto")]
    flag = "" if auto_c <= 1.03 * ct else f"   <-- pick costs {100 * (auto_c / ct - 1):.0f} % cold"
    cold = "  ".join(f"{nm} {t:6.1f}" for nm, t in cc.items())
    print(f"{label}: warm auto {auto_w:6.1f} us (best {wn} {wt:6.1f})   cold auto {auto_c:6.1f} us (best {cn} {ct:6.1f}; {cold}){flag}", flush=True)


MIXED_MS = [384, 512, 768, 1024, 2048] if "--mixed-mid" in sys.argv else [1, 8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024]


if __name__ == "__main__":
    ms_ = [8, 32, 64, 128, 256, 512] if QUICK else [8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024]
    shapes = [(4096, 4096), (4096, 11008)] if QUICK else [(4096, 4096), (4096, 11008), (8192, 8192), (4096, 28672), (8192, 28672), (11008, 4096),
                                                          (28672, 8192), (13824, 5120)]
    if "--mixed-mid" not in sys.argv:
        for k, n in shapes:
            for m in ms_:
                scaled_case(m, k, n, torch.int8)
        for k, n in shapes[:3]:
            for m in ms_:
                scaled_case(m, k, n, torch.float8_e4m3fn)
    for k, n in shapes:
        for m in MIXED_MS:
            mixed_case(m, k, n)
