"""Interleaved A/B of several BUILDS of the library on the int4 decode GEMM (one process, same data, rounds interleaved).

    python tools/ab_mixed_decode_lib.py <variant>[,<variant>...] [rounds] [MxKxN ...]

variant = name given to `python -m conch_amd._build --variant NAME -DMACRO ...`; "base" = the product library, always first.
"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_mixed_precision_metadata  # noqa: E402

DEFAULT = ["1x4096x4096", "16x4096x4096", "32x4096x4096", "16x4096x11008", "32x4096x11008", "16x11008x4096", "16x8192x8192", "16x4096x28672", "64x4096x11008"]


def main():
    names = [v for v in sys.argv[1].split(",") if v]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 9
    shapes = [a for a in sys.argv[2:] if "x" in a] or DEFAULT
    libs = {"base": _C.load()}
    for name in names:
        libs[name] = _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{name}.so"))
    ms = ctypes.c_float()
    for wl in shapes:
        m, k, n = (int(v) for v in wl.split("x"))
        torch.manual_seed(0)
        x = (torch.rand((m, k), device="cuda") - 0.3).to(torch.float16)
        wq = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
        ws = (0.05 * torch.rand((k // 128, n), device="cuda") + 0.01).to(torch.float16)
        outs = {name: torch.zeros((m, n), dtype=torch.float16, device="cuda") for name in libs}
        md = create_mixed_precision_metadata(x, wq, ws, None, 4, 8, 128)

        def run(name, iters):
            _C.check(kg._mixed_gemm_call("conch_time_mixed_precision_gemm", outs[name], x, wq, ws, None, md, (iters, ctypes.byref(ms)), lib=libs[name]), "time")
            return ms.value

        iters = int(min(2000, max(100, 20.0 / max(run("base", 20), 1e-4))))
        for _ in range(5):
            for name in libs:
                run(name, iters)
        same = {name: bool(torch.equal(outs[name], outs["base"])) for name in libs}
        res = {name: [] for name in libs}
        for _ in range(rounds):
            for name in libs:
                res[name].append(run(name, iters))
        base = statistics.median(res["base"])
        print(f"int4 x fp16 {wl:16s} " + "  ".join(f"{name}: {statistics.median(r) * 1e3:6.2f} us (x{statistics.median(r) / base:.3f}{'' if same[name] else ' BITS DIFFER'})"
                                                  for name, r in res.items()), flush=True)


if __name__ == "__main__":
    main()
