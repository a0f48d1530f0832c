"""A/B (round 6): C3 with a bias vector on the assembly kernel's *_bias twin against the 256 x 256 HIP kernel (the only kernel that
took a bias until round 6) and against the bias-free assembly launch; interleaved, kernel events inside the library."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conch_amd import _C  # noqa: E402
from conch_amd.kernels.quantization import gemm as kgemm  # noqa: E402
from conch_amd.ops.quantization.gemm import create_scaled_metadata  # noqa: E402


def main() -> None:
    dev = torch.device("cuda")
    for kind in ("fp8", "int8"):
        m, k, n = 4096, 4096, 11008
        torch.manual_seed(0)
        if kind == "fp8":
            a = (0.25 * torch.rand((m, k), device=dev)).to(torch.float8_e4m3fn)
            bt = (0.25 * torch.rand((n, k), device=dev)).to(torch.float8_e4m3fn)
        else:
            a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device=dev)
            bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=dev)
        sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
        bias = torch.rand((n,), dtype=torch.bfloat16, device=dev)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)

        def t(variant: int, with_bias: bool, iters: int = 300) -> float:
            _C.set_gemm_variant(variant)
            ms = ctypes.c_float()
            _C.check(kgemm._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, bias if with_bias else None, (iters, ctypes.byref(ms))), "time")
            _C.set_gemm_variant(_C.VARIANT_AUTO)
            return ms.value * 1e3

        for _ in range(3):
            t(_C.VARIANT_MFMA_ASM1W, False)
        for r in range(4):
            print(f"{kind} round {r}: assembly no bias {t(_C.VARIANT_MFMA_ASM1W, False):7.2f} us | assembly + bias {t(_C.VARIANT_MFMA_ASM1W, True):7.2f} us | "
                  f"HIP 256x256 + bias {t(_C.VARIANT_MFMA_PINGPONG2, True):7.2f} us | auto + bias {t(_C.VARIANT_AUTO, True):7.2f} us", flush=True)


if __name__ == "__main__":
    main()
