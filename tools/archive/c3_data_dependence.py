"""C3's launch time and in-loop clock on the benchmark's data against operands that switch less: the same binary, the same
instruction stream -- what the board's power limit costs (MI355X_MICROARCH.md, DVFS give-back).  Uses the diagnostic twin for the
clock (thread 0 of every workgroup stamps s_memtime / s_memrealtime around its K loop).  -> stdout (profiles/r04/c3_data_dependence.txt)"""
import ctypes
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from conch_amd import _C, _build  # noqa: E402
from conch_amd.kernels.quantization import gemm as kg  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402

M, K, N = 4096, 4096, 11008
PROBE = _C.load_library(_build.PROBE_LIB)
BASE = _C.load()


def run_case(name, a, bt):
    sa, sb = 0.25 * torch.rand((M, 1), device="cuda"), 0.25 * torch.rand((N, 1), device="cuda")
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
    ms = ctypes.c_float()

    def timer(lib, iters):
        _C.check(kg._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms)), lib=lib), "time")
        return ms.value

    t_end = time.perf_counter() + 2.0
    vals = []
    while time.perf_counter() < t_end:
        vals.append(timer(BASE, 200))
    t = statistics.median(vals[len(vals) // 2:])
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        timer(PROBE, 200)
    rd = PROBE.conch_debug_probe_scaled
    rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    blocks = 688
    buf = (ctypes.c_ulonglong * (8 * blocks))()
    assert rd(buf, blocks) == 0
    clk = sorted((buf[8 * b + 2] - buf[8 * b]) / (buf[8 * b + 3] - buf[8 * b + 1]) * 0.1 for b in range(blocks) if buf[8 * b + 3] > buf[8 * b + 1])
    loop = sorted((buf[8 * b + 3] - buf[8 * b + 1]) * 0.01 for b in range(blocks) if buf[8 * b + 3] > buf[8 * b + 1])
    fl = 2.0 * M * N * K
    print(f"{name:46s}: {t * 1e3:7.1f} us  {fl / t / 1e9:7.0f} TFLOP/s ({fl / t / 1e9 / 5000:.3f} of 5 PF)   in-loop clock {statistics.median(clk):.2f} GHz, "
          f"K loop {statistics.median(loop):.1f} us", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    f8 = torch.float8_e4m3fn
    rnd = lambda r: (0.25 * torch.rand((r, K), device="cuda")).to(f8)  # noqa: E731
    run_case("benchmark data: 0.25 * rand (reference recipe)", rnd(M), rnd(N))
    run_case("A as above, B all zero", rnd(M), torch.zeros((N, K), device="cuda").to(f8))
    run_case("A and B all zero", torch.zeros((M, K), device="cuda").to(f8), torch.zeros((N, K), device="cuda").to(f8))
    run_case("A and B one constant (0.125)", torch.full((M, K), 0.125, device="cuda").to(f8), torch.full((N, K), 0.125, device="cuda").to(f8))
    run_case("full-range random bytes (signs, all exponents)", torch.randint(0, 0x7E, (M, K), device="cuda", dtype=torch.uint8).view(f8),
             (torch.randint(0, 0x7E, (N, K), device="cuda", dtype=torch.uint8) | (torch.randint(0, 2, (N, K), device="cuda", dtype=torch.uint8) << 7)).view(f8))
