"""Interleaved A/B of two BUILDS of the library on the static quantisers (one process, same data, device time of back-to-back launches).
usage: python tools/ab_quant_lib.py <variant> [rounds]   (variant = name given to `python -m conch_amd._build --variant NAME -D...`)"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402

VARIANT = sys.argv[1]
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 9
LIBS = {"base": _C.load(), VARIANT: _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{VARIANT}.so"))}


def run(lib, kind, out, x, scale, iters):
    tokens, hidden = x.shape
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        if kind.startswith("dyn"):
            fn = lib.conch_dynamic_scaled_int8_quant if kind == "dyn8" else lib.conch_dynamic_scaled_fp8_quant
            args = [ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(DSCALE.data_ptr()), ctypes.c_void_p(x.data_ptr()), tokens, hidden, hidden, hidden, _C.dtype_id(x.dtype)]
            if kind != "dyn8":
                args.append(_C.dtype_id(torch.float8_e4m3fn))
            rc = fn(*args, stream)
        elif kind == "int8":
            rc = lib.conch_static_scaled_int8_quant(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(scale.data_ptr()), tokens, hidden,
                                                    hidden, hidden, _C.dtype_id(x.dtype), stream)
        else:
            rc = lib.conch_static_scaled_fp8_quant(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(scale.data_ptr()), tokens, hidden,
                                                   hidden, hidden, _C.dtype_id(x.dtype), _C.dtype_id(torch.float8_e4m3fn), stream)
        assert rc == 0
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


DSCALE = torch.empty(16384, device="cuda")
for tokens, hidden in [(4096, 4096), (4096, 11008), (1024, 4096), (16384, 4096), (128, 4096)]:
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        for kind in ("int8", "fp8", "dyn8", "dynf8"):
            x = (torch.rand((tokens, hidden), device="cuda") - 0.5).to(dt)
            scale = torch.tensor([0.004], device="cuda")
            outs = {n: torch.empty((tokens, hidden), dtype=torch.int8, device="cuda") for n in LIBS}
            res = {n: [] for n in LIBS}
            for n, lib in LIBS.items():
                run(lib, kind, outs[n], x, scale, 200)
            for _ in range(ROUNDS):
                for n, lib in LIBS.items():
                    res[n].append(run(lib, kind, outs[n], x, scale, 200))
            b, v = statistics.median(res["base"]), statistics.median(res[VARIANT])
            byt = tokens * hidden * (x.element_size() + 1)
            print(f"{kind:4s} {tokens:6d}x{hidden:5d} {str(dt)[6:]:9s}: base {b:7.2f} us ({byt / b / 1e6:5.2f} TB/s)  {VARIANT} {v:7.2f} us ({byt / v / 1e6:5.2f} TB/s)  x{v / b:.3f}  "
                  f"bits {'equal' if torch.equal(outs['base'], outs[VARIANT]) else 'DIFFER'}", flush=True)
