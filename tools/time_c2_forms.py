"""C2 (scaled_gemm int8 128 x 4096 x 4096), the forms that exist, interleaved (round 6; closes VERDICT r5 item 6):
  auto                the split-K kernel with the in-launch reduce (two 2048-byte K slices, 512 workgroups)
  in-workgroup        K split over the four waves of ONE workgroup per 16-column strip, partial sums reduced through LDS, no
                      inter-workgroup traffic at all (CONCH_TUNE_SKINNY_NO_SPLITK = 1): the family VERDICT r5 item 6 proposes
                      (its 8-wave / 32-column / 128-workgroup member moves 640 KB per CU where this one moves 576 KB, on half the CUs)
  two launches        split-K + separate reduce kernel (CONCH_TUNE_SKINNY_MODE = 1)
kernel events inside the library (conch_time_scaled_gemm), 400 launches per sample."""
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
    m, k, n = 128, 4096, 4096
    torch.manual_seed(0)
    a = torch.randint(-32, 32, (m, k), dtype=torch.int8, device=dev)
    bt = torch.randint(-32, 32, (n, k), dtype=torch.int8, device=dev)
    sa, sb = 0.25 * torch.rand((m, 1), device=dev), 0.25 * torch.rand((n, 1), device=dev)
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    md = create_scaled_metadata(a, bt.T, sa, sb, torch.bfloat16)

    def t(key: int, value: int, iters: int = 400) -> float:
        _C.set_tuning(key, value)
        try:
            ms = ctypes.c_float()
            _C.check(kgemm._scaled_gemm_call("conch_time_scaled_gemm", out, a, bt.T, sa, sb, md, None, (iters, ctypes.byref(ms))), "time")
        finally:
            _C.set_tuning(key, 0)
        return ms.value * 1e3

    want = None
    for key, value in ((_C.TUNE_SKINNY_MODE, 0), (_C.TUNE_SKINNY_NO_SPLITK, 1), (_C.TUNE_SKINNY_MODE, 1)):
        _C.set_tuning(key, value)
        kgemm.scaled_gemm_launcher(out, a, bt.T, sa, sb, md)
        _C.set_tuning(key, 0)
        torch.cuda.synchronize()
        want = out.clone() if want is None else want
        assert torch.equal(out, want), "the forms are bit-identical (int32 sums)"
    for _ in range(3):
        t(_C.TUNE_SKINNY_MODE, 0)
    for r in range(5):
        print(f"round {r}: auto (one launch, 2 K slices) {t(_C.TUNE_SKINNY_MODE, 0):6.2f} us | K split inside the workgroup {t(_C.TUNE_SKINNY_NO_SPLITK, 1):6.2f} us | "
              f"split-K + reduce kernel {t(_C.TUNE_SKINNY_MODE, 1):6.2f} us", flush=True)


if __name__ == "__main__":
    main()
