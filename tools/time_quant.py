"""GPU time of the static int8 / fp8 quantiser on C1 (4096 x 4096 fp16), several builds of the library interleaved in one
process; launches are replayed from a HIP graph so that the host's per-call time is not what is measured.

    python tools/time_quant.py [variant,variant,...] [rounds]
"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402

names = [v for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else []) if v]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
libs = {"base": _C.load()}
for nm in names:
    libs[nm] = _C.load_library(Path(_C.LIB_PATH).with_name(f"libconch_amd_{nm}.so"))
tokens = hidden = 4096
x = torch.rand(tokens, hidden, dtype=torch.float16, device="cuda") * 1000
scale = torch.tensor([2.1], dtype=torch.float32, device="cuda")
outs = {nm: torch.empty((tokens, hidden), dtype=torch.int8, device="cuda") for nm in libs}
N = 50
graphs = {}
stream = torch.cuda.Stream()
for nm, lib in libs.items():
    with torch.cuda.stream(stream):
        def launch(lib=lib, nm=nm):
            rc = lib.conch_static_scaled_int8_quant(outs[nm].data_ptr(), x.data_ptr(), scale.data_ptr(), tokens, hidden, hidden, hidden,
                                                    _C.dtype_id(torch.float16), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        launch()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(N):
                launch()
        graphs[nm] = g
torch.cuda.synchronize()
same = {nm: bool(torch.equal(outs[nm], outs["base"])) for nm in libs}


def run(nm):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        graphs[nm].replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * N)


for _ in range(3):
    for nm in libs:
        run(nm)
res = {nm: [] for nm in libs}
for _ in range(rounds):
    for nm in libs:
        res[nm].append(run(nm))
nbytes = tokens * hidden * 3
for nm, r in res.items():
    med = statistics.median(r)
    print(f"c1 {nm:12s} median {med * 1e3:7.2f} us  min {min(r) * 1e3:7.2f} us  {nbytes / med / 1e6:7.1f} GB/s ({nbytes / med / 1e6 / 8000:.3f} of 8 TB/s)  "
          f"bits {'equal' if same[nm] else 'DIFFER'}", flush=True)
