"""Quick per-variant kernel timing on the GPU box (development aid, not the judged bench)."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402


# development aid: CONCH_LIB=<path> times another build of the library (same-box comparison of two builds)
import os
from pathlib import Path
if os.environ.get("CONCH_LIB"):
    _C.LIB_PATH = Path(os.environ["CONCH_LIB"])


def time_scaled(m, k, n, dtype, variant, iters=20):
    torch.manual_seed(0)
    if dtype == torch.int8:
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(dtype)
        bt = (0.25 * torch.rand((n, k), device="cuda")).to(dtype)
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    _C.set_gemm_variant(variant)
    ms = ctypes.c_float()
    try:
        for it in (3, iters):
            rc = kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (it, ctypes.byref(ms)))
            _C.check(rc, "time")
    finally:
        _C.set_gemm_variant(0)
    return ms.value


def time_mixed(m, k, n, dtype, bits, variant, iters=20):
    from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata
    torch.manual_seed(0)
    x = (10 * (torch.rand((m, k), device="cuda") - 0.3)).to(dtype)
    pf = 32 // bits
    wq = torch.randint(-2**31, 2**31 - 1, (k // pf, n), dtype=torch.int32, device="cuda")
    ws = (0.1 * torch.rand((k // 128, n), device="cuda") + 0.01).to(dtype)
    out = torch.empty((m, n), dtype=dtype, device="cuda")
    md = create_mixed_precision_metadata(x, wq, ws, None, bits, 8 if bits == 4 else 128, 128)
    _C.set_gemm_variant(variant)
    ms = ctypes.c_float()
    try:
        for it in (3, iters):
            rc = kg._mixed_gemm_call("conch_time_mixed_precision_gemm", out, x, wq, ws, None, md, (it, ctypes.byref(ms)))
            _C.check(rc, "time")
    finally:
        _C.set_gemm_variant(0)
    return ms.value


if __name__ == "__main__":
    if "--ab" in sys.argv:
        # interleaved A/B after a warm-up (clocks ramp for ~20 ms): median of 7 rounds of 50 launches
        import statistics
        variants = [int(v) for v in sys.argv[sys.argv.index("--ab") + 1].split(",")]
        for dtype in (torch.float8_e4m3fn, torch.int8):
            for (m, k, n) in [(4096, 4096, 11008), (8192, 8192, 8192), (8192, 8192, 3584)]:
                for _ in range(10):
                    time_scaled(m, k, n, dtype, variants[0], iters=50)
                res = {v: [] for v in variants}
                for _ in range(7):
                    for v in variants:
                        res[v].append(time_scaled(m, k, n, dtype, v, iters=50))
                line = "  ".join(f"v{v}: {statistics.median(r)*1e3:7.1f} us ({2.0*m*n*k/(statistics.median(r)*1e-3)/1e12:6.0f} TF)"
                                 for v, r in res.items())
                print(f"{str(dtype):22s} {m}x{k}x{n}  {line}", flush=True)
        sys.exit(0)
    if "--trend" in sys.argv:
        import time
        for variant in (3, 4, 3, 4):
            t0 = time.time()
            out = []
            for i in range(40):
                ms = time_scaled(4096, 4096, 11008, torch.float8_e4m3fn, variant, iters=50)
                out.append(f"{ms*1e3:.1f}")
            print(f"variant {variant} ({time.time()-t0:.2f}s):", " ".join(out), flush=True)
        sys.exit(0)
    if "--mixedab" in sys.argv:
        # interleaved A/B of the mixed-GEMM loop forms: --mixedab 0,6
        import statistics
        variants = [int(v) for v in sys.argv[sys.argv.index("--mixedab") + 1].split(",")]
        for (m, k, n) in [(1024, 4096, 11008), (4096, 8192, 4096), (2048, 4096, 11008), (256, 4096, 4096), (8192, 8192, 8192)]:
            for _ in range(10):
                time_mixed(m, k, n, torch.float16, 4, variants[0], iters=30)
            for dtype, bits in ((torch.float16, 4), (torch.bfloat16, 4), (torch.float16, 8), (torch.bfloat16, 8)):
                res = {v: [] for v in variants}
                for _ in range(5):
                    for v in variants:
                        res[v].append(time_mixed(m, k, n, dtype, bits, v, iters=40))
                print(f"mixed int{bits} {str(dtype)[6:]} {m}x{k}x{n}: " + "  ".join(
                    f"v{v}: {statistics.median(t)*1e3:.1f}us ({2.0*m*n*k/(statistics.median(t)*1e-3)/1e12:.0f} TF)" for v, t in res.items()), flush=True)
        sys.exit(0)
    if "--silu" in sys.argv:
        # FFN pair: fused scaled_gemm_silu_and_mul against the unfused pair (default scaled_gemm on [gate | up], then
        # torch's silu-and-mul on the result), back-to-back launches after a warm-up, HIP events
        import statistics
        from conch_amd.ops.quantization.gemm import scaled_gemm, scaled_gemm_silu_and_mul

        def ev(fn, iters=50):
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(iters):
                fn()
            e0.record()
            torch.cuda.synchronize()
            return s0.elapsed_time(e0) / iters * 1e3

        for dtype in (torch.float8_e4m3fn, torch.int8):
            for (m, k, d) in [(4096, 4096, 11008), (8192, 8192, 14336), (1024, 4096, 11008)]:
                torch.manual_seed(0)
                if dtype == torch.int8:
                    a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
                    bt = torch.randint(-32, 32, (2 * d, k), dtype=torch.int8, device="cuda")
                else:
                    a = (0.25 * torch.rand((m, k), device="cuda")).to(dtype)
                    bt = (0.25 * torch.rand((2 * d, k), device="cuda") - 0.125).to(dtype)
                sa = 0.01 * torch.rand((m, 1), device="cuda")
                sb = 0.01 * torch.rand((2 * d, 1), device="cuda")
                fused = lambda: scaled_gemm_silu_and_mul(a, bt.T, sa, sb, torch.bfloat16)  # noqa: E731
                gemm = lambda: scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)  # noqa: E731

                def pair():
                    g = scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
                    return torch.nn.functional.silu(g[:, :d]) * g[:, d:]

                for _ in range(200):
                    fused()
                res = {"fused": [], "gemm only": [], "gemm + torch silu*mul": []}
                for _ in range(5):
                    res["fused"].append(ev(fused))
                    res["gemm only"].append(ev(gemm))
                    res["gemm + torch silu*mul"].append(ev(pair))
                flops = 2.0 * m * (2 * d) * k
                print(f"{str(dtype)[6:]:14s} M={m} K={k} d={d}: " + "  ".join(
                    f"{name} {statistics.median(v):.1f} us ({flops / statistics.median(v) / 1e6:.0f} TFLOP/s)" for name, v in res.items()), flush=True)
        from conch_amd.ops.quantization.gemm import mixed_precision_gemm, mixed_precision_gemm_silu_and_mul
        for (m, k, d) in [(4096, 4096, 11008), (1024, 4096, 11008)]:
            torch.manual_seed(0)
            x = (0.5 * (torch.rand((m, k), device="cuda") - 0.3)).to(torch.float16)
            wq = torch.randint(-2**31, 2**31 - 1, (k // 8, 2 * d), dtype=torch.int32, device="cuda")
            ws = (0.01 * torch.rand((k // 128, 2 * d), device="cuda") + 0.001).to(torch.float16)
            fused = lambda: mixed_precision_gemm_silu_and_mul(x, wq, ws, None, 4, 8, 128)  # noqa: E731
            gemm = lambda: mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)  # noqa: E731

            def pair():
                g = mixed_precision_gemm(x, wq, ws, None, 4, 8, 128)
                return torch.nn.functional.silu(g[:, :d]) * g[:, d:]

            for _ in range(100):
                fused()
            res = {"fused": [], "gemm only": [], "gemm + torch silu*mul": []}
            for _ in range(5):
                res["fused"].append(ev(fused))
                res["gemm only"].append(ev(gemm))
                res["gemm + torch silu*mul"].append(ev(pair))
            flops = 2.0 * m * (2 * d) * k
            print(f"int4 x fp16    M={m} K={k} d={d}: " + "  ".join(
                f"{name} {statistics.median(v):.1f} us ({flops / statistics.median(v) / 1e6:.0f} TFLOP/s)" for name, v in res.items()), flush=True)
        sys.exit(0)
    if "--skinnysweep" in sys.argv:
        # skinny split-K (variant 4) against the 256x256 tiled kernel (variant 5) on small-M shapes
        import statistics
        for dtype in (torch.int8, torch.float8_e4m3fn):
            for k in (4096, 8192):
                for n in (4096, 8192, 16384, 28672):
                    for m in (8, 16, 32, 48, 64, 96, 128):
                        res = {}
                        for v in (4, 5, 6):
                            try:
                                for _ in range(3):
                                    time_scaled(m, k, n, dtype, v, iters=30)
                                res[v] = statistics.median([time_scaled(m, k, n, dtype, v, iters=40) for _ in range(3)]) * 1e3
                            except Exception:  # noqa: BLE001
                                res[v] = float("nan")
                        print(f"{str(dtype)[6:]:14s} M={m:4d} K={k} N={n:6d}: skinny {res[4]:7.1f} us   tiled {res[5]:7.1f} us   mid {res[6]:7.1f} us   {min(res, key=lambda v: res[v] if res[v] == res[v] else 1e9)}", flush=True)
        sys.exit(0)
    if "--mixeddecode" in sys.argv:
        # decode-batch mixed kernel (variant 4) against the LDS-tiled kernel (variant 5)
        import statistics
        for (k, n) in [(4096, 4096), (4096, 11008), (8192, 8192), (4096, 28672)]:
            for m in (1, 16, 32, 64):
                for dtype, bits in ((torch.float16, 4), (torch.bfloat16, 8)):
                    res = {}
                    for v in (4, 5):
                        for _ in range(3):
                            time_mixed(m, k, n, dtype, bits, v, iters=20)
                        res[v] = statistics.median([time_mixed(m, k, n, dtype, bits, v, iters=40) for _ in range(3)]) * 1e3
                    wbytes = k * n * bits / 8
                    print(f"int{bits} {str(dtype)[6:]:9s} M={m:3d} K={k} N={n:6d}: decode kernel {res[4]:7.1f} us ({wbytes / res[4] / 1e6:.2f} TB/s of weights)   tiled {res[5]:7.1f} us", flush=True)
        sys.exit(0)
    if "--mixednt" in sys.argv:
        import statistics
        lib = _C.load()
        for (m, k, n) in [(1024, 4096, 11008), (4096, 8192, 4096), (2048, 4096, 11008)]:
            for _ in range(10):
                time_mixed(m, k, n, torch.float16, 4, 0, iters=30)
            for dtype, bits in ((torch.float16, 4), (torch.bfloat16, 4), (torch.float16, 8)):
                res = {}
                for variant in (0, 3):
                    for nt in (0, 4, 3):
                        lib.conch_set_tuning(1, nt)
                        res[(variant, nt)] = statistics.median([time_mixed(m, k, n, dtype, bits, variant, iters=50) for _ in range(5)])
                lib.conch_set_tuning(1, 0)
                print(f"mixed int{bits} {str(dtype)[6:]} {m}x{k}x{n}: " + "  ".join(f"v{v}/nt{nt}: {t*1e3:.1f}us ({2.0*m*n*k/(t*1e-3)/1e12:.0f} TF)" for (v, nt), t in res.items()), flush=True)
        sys.exit(0)
    if "--mixed" in sys.argv:
        for dtype in (torch.float16, torch.bfloat16):
            for bits in (4, 8):
                for (m, k, n) in [(1024, 4096, 11008), (4096, 8192, 4096), (8192, 8192, 8192)]:
                    ms = time_mixed(m, k, n, dtype, bits, 0)
                    tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
                    print(f"mixed {str(dtype):16s} int{bits} M={m} K={k} N={n}: {ms*1e3:9.1f} us  {tf:8.1f} TFLOP/s", flush=True)
        sys.exit(0)
    shapes = [(4096, 4096, 11008), (8192, 8192, 8192), (128, 4096, 4096), (8192, 8192, 3584)]
    for dtype in (torch.float8_e4m3fn, torch.int8):
        for (m, k, n) in shapes:
            for variant in (2, 3):
                ms = time_scaled(m, k, n, dtype, variant)
                tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
                print(f"{str(dtype):24s} M={m} K={k} N={n} variant={variant}: {ms*1e3:9.1f} us  {tf:8.1f} TFLOP/s", flush=True)
