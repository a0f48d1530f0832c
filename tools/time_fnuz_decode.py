"""e4m3fnuz at decode sizes (round 6): the split-K kernel with the special codes split off in registers (no scan, no expansion)
against the same shapes in e4m3fn and against round 5's path (scan + gated fp8 kernel + gated bf16 expansion;
conch_debug_fnuz_in_registers(0)); clean operands and operands with +-240 in every row / column.  Op-level, torch events around 100 calls."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kgemm  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata, scaled_gemm  # noqa: E402


def main() -> None:
    dev = torch.device("cuda")
    hook = _C.load().conch_debug_fnuz_in_registers
    hook.restype, hook.argtypes = ctypes.c_int, [ctypes.c_int]
    for m, k, n in ((16, 4096, 11008), (1, 4096, 4096), (32, 4096, 4096), (64, 4096, 11008), (128, 4096, 4096), (8, 8192, 8192), (40, 4096, 11008)):
        torch.manual_seed(0)
        base = 0.25 * torch.rand((m, k), device=dev)
        wbase = 0.25 * torch.rand((n, k), device=dev)
        sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        line = f"{m:4d} x {k:5d} x {n:5d}:"
        for name, dt, sat, inreg in (("e4m3fn", torch.float8_e4m3fn, False, 1), ("fnuz clean", torch.float8_e4m3fnuz, False, 1),
                                     ("fnuz clean, round-5 path", torch.float8_e4m3fnuz, False, 0), ("fnuz saturated", torch.float8_e4m3fnuz, True, 1),
                                     ("fnuz saturated, round-5 path", torch.float8_e4m3fnuz, True, 0),
                                     ("fnuz clean, skinny forced", torch.float8_e4m3fnuz, False, 2), ("fnuz saturated, skinny forced", torch.float8_e4m3fnuz, True, 2)):
            hook(inreg)
            a, bt = base.to(dt), wbase.to(dt)
            if sat:
                ra, rb = torch.arange(m, device=dev), torch.arange(n, device=dev)
                a.view(torch.uint8)[ra, (ra * 37) % k] = 0x7F
                bt.view(torch.uint8)[rb, (rb * 53) % k] = 0xFF
            md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)
            ms = ctypes.c_float()
            for _ in range(2):
                _C.check(kgemm._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (300, ctypes.byref(ms))), "time")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(20):
                scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
            e0.record()
            for _ in range(100):
                scaled_gemm(a, bt.T, sa, sb, torch.bfloat16)
            e1.record()
            torch.cuda.synchronize()
            line += f"   {name}: {e0.elapsed_time(e1) * 10:6.2f} us"
        hook(1)
        print(line, flush=True)


if __name__ == "__main__":
    main()
