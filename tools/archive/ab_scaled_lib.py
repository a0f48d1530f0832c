"""Interleaved A/B of several BUILDS of the library on the scaled GEMM (one process, same data, rounds interleaved:
cdna_hip_programming.md rule 24).  Development aid.

    python tools/ab_scaled_lib.py <workload> <variant>[,<variant>...] [rounds] [--tune key=value ...]

workload: c3 | c3i8 | c5shard | sq8k | MxKxN[:int8|fp8];  variant = name given to
`python -m conch_amd._build --variant NAME -DMACRO ...` (libconch_amd_NAME.so); "base" = the product library, always first.
Prints median / min per build, the ratio to base and whether the outputs are bit-equal to base's.
"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

SHAPES = {"c3": (4096, 4096, 11008, "fp8"), "c3i8": (4096, 4096, 11008, "int8"), "c5shard": (8192, 8192, 3584, "fp8"),
          "sq8k": (8192, 8192, 8192, "fp8"), "c2": (128, 4096, 4096, "int8")}


def main():
    wl = sys.argv[1]
    names = [v for v in sys.argv[2].split(",") if v]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 9
    tunes = []
    if "--tune" in sys.argv:
        for kv in sys.argv[sys.argv.index("--tune") + 1:]:
            k, v = kv.split("=")
            tunes.append((int(k), int(v)))
    if wl in SHAPES:
        m, k, n, dt = SHAPES[wl]
    else:
        dims, _, dt = wl.partition(":")
        m, k, n = (int(v) for v in dims.split("x"))
        dt = dt or "fp8"
    libs = {"base": _C.load()}
    for name in names:
        libs[name] = _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{name}.so"))
    for lib in libs.values():
        for key, val in tunes:
            lib.conch_set_tuning(key, val)
    torch.manual_seed(0)
    if dt == "int8":
        a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device="cuda")
        bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device="cuda")
    else:
        a = (0.25 * torch.rand((m, k), device="cuda")).to(torch.float8_e4m3fn)
        bt = (0.25 * torch.rand((n, k), device="cuda")).to(torch.float8_e4m3fn)
    sa = 0.25 * torch.rand((m, 1), device="cuda")
    sb = 0.25 * torch.rand((n, 1), device="cuda")
    outs = {name: torch.zeros((m, n), dtype=torch.bfloat16, device="cuda") for name in libs}
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()

    def run(name, iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", outs[name], a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms)),
                                      lib=libs[name]), "time")
        return ms.value

    probe = max(run("base", 20), 1e-4)
    iters = int(min(2000, max(50, 30.0 / probe)))  # ~30 ms per sample
    for _ in range(10):  # ramp
        for name in libs:
            run(name, iters)
    torch.cuda.synchronize()
    same = {name: bool(torch.equal(outs[name], outs["base"])) for name in libs}
    res = {name: [] for name in libs}
    for _ in range(rounds):
        for name in libs:
            res[name].append(run(name, iters))
    flops = 2.0 * m * n * k
    base_med = statistics.median(res["base"])
    for name, r in res.items():
        med, mn = statistics.median(r), min(r)
        print(f"{wl:10s} {name:14s} median {med * 1e3:8.2f} us  min {mn * 1e3:8.2f} us  {flops / (med * 1e-3) / 1e12:7.1f} TFLOP/s "
              f"({flops / (med * 1e-3) / 1e12 / 5000:.3f})  x{med / base_med:.3f} of base  bits {'equal' if same[name] else 'DIFFER'}",
              flush=True)


if __name__ == "__main__":
    main()
