"""Interleaved A/B of two BUILDS of the library on bnb dequantize_4bit / dequantize_blockwise (device time, back-to-back launches).
usage: python tools/ab_bnb_dequant.py <variant> [rounds]"""
import ctypes
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C  # noqa: E402

VARIANT = sys.argv[1]
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 9
LIBS = {"base": _C.load(), VARIANT: _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{VARIANT}.so"))}
P = ctypes.c_void_p


def run(lib, out, xq, absmax, n, blocksize, qt, iters):
    stream = P(torch.cuda.current_stream().cuda_stream)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        rc = lib.conch_bnb_dequantize_blockwise(P(out.data_ptr()), P(xq.data_ptr()), P(absmax.data_ptr()), None, n, blocksize, qt, _C.dtype_id(out.dtype),
                                                _C.dtype_id(absmax.dtype), stream)
        assert rc == 0, _C.load().conch_last_error()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for rows, cols in [(11008, 4096), (4096, 4096), (28672, 8192), (1024, 1024)]:
    for odt in (torch.float16, torch.bfloat16):
        for qt, blocksize in ((0, 64), (1, 128)):
            n = rows * cols
            xq = torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device="cuda")
            absmax = torch.rand(n // blocksize, device="cuda") + 0.1
            outs = {k: torch.empty(n, dtype=odt, device="cuda") for k in LIBS}
            res = {k: [] for k in LIBS}
            for k, lib in LIBS.items():
                run(lib, outs[k], xq, absmax, n, blocksize, qt, 50)
            for _ in range(ROUNDS):
                for k, lib in LIBS.items():
                    res[k].append(run(lib, outs[k], xq, absmax, n, blocksize, qt, 100))
            b, v = statistics.median(res["base"]), statistics.median(res[VARIANT])
            byt = n * (0.5 + outs["base"].element_size()) + absmax.numel() * 4
            print(f"{'nf4' if qt == 0 else 'fp4'} b{blocksize:<4d} {rows:5d}x{cols:5d} -> {str(odt)[6:]:8s}: base {b:7.2f} us ({byt / b / 1e6:5.2f} TB/s)  {VARIANT} {v:7.2f} us "
                  f"({byt / v / 1e6:5.2f} TB/s)  x{v / b:.3f}  bits {'equal' if torch.equal(outs['base'].view(torch.int16), outs[VARIANT].view(torch.int16)) else 'DIFFER'}", flush=True)
