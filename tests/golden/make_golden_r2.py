"""Round-2 golden vectors (same rules as make_golden.py: run in the authoring container only, outputs come from the REAL
reference at /root/reference, nothing from this repo is involved).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py [--only dyn|modes|bnb|gelu]

* quant_dynamic.npz -- dynamic per-token quantisation.  The reference has no dynamic path; what is
  pinned is that each ROW equals the reference's STATIC oracle (conch.reference.quantization.{int8,fp8}) called on that
  row with scale = absmax(row) / QMAX (1.0 for an all-zero row), the scale being computed here with plain torch.
* gemm_modes.npz -- the kernel modes of conch/kernels/quantization/gemm.py that conch.ops never selects (SURVEY.md 8(f)
  N2: weight-group modes NONE / SHIFT / ASYMMETRIC, channel-scale modes NONE / WEIGHT_ONLY / ACTIVATION_ONLY, 2-bit
  weights), produced by running the reference's OWN Triton kernel `_gemm_kernel` through its launchers under Triton's CPU
  interpreter (TRITON_INTERPRET=1).
"""

from __future__ import annotations

import os
import sys
from pathlib import Path
from unittest import mock

os.environ.setdefault("TRITON_INTERPRET", "1")

import numpy as np
import torch

REF = "/root/reference"
if REF not in sys.path:
    sys.path.insert(0, REF)

from conch.platforms import current_platform  # noqa: E402
from conch.reference.quantization import fp8 as ref_fp8  # noqa: E402
from conch.reference.quantization import int8 as ref_int8  # noqa: E402
from conch.third_party.vllm.utils import seed_everything  # noqa: E402

OUT = Path(__file__).resolve().parent
DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
        return t.view(torch.uint8).numpy().copy()
    return t.numpy().copy()


def fp8_quant(x: torch.Tensor, scale: torch.Tensor, fnuz: bool) -> torch.Tensor:
    with mock.patch.object(type(current_platform), "is_amd", lambda self: fnuz):
        return ref_fp8.scaled_fp8_quant(x, scale)


def row_scale(row: torch.Tensor, qmax: float) -> torch.Tensor:
    absmax = row.to(torch.float32).abs().max()
    return (absmax / qmax if absmax > 0 else torch.tensor(1.0)).reshape(1)


def gen_dynamic() -> None:
    out: dict[str, np.ndarray] = {}
    for dname, dtype in DT.items():
        for tokens, hidden in ((1, 16), (7, 67), (5, 768), (3, 2064), (2, 8200)):
            key = f"{dname}_t{tokens}_h{hidden}"
            seed_everything(0)
            x = (torch.rand(tokens, hidden, dtype=torch.float32) - 0.4) * 37.0
            x = x.to(dtype)
            if tokens >= 3:
                x[1] = 0  # an all-zero row
                x[2, hidden // 2] = -1000.0  # the maximum is negative
            rows_i8, rows_fn, rows_fnuz, s_i8, s_fn, s_fnuz = [], [], [], [], [], []
            for t in range(tokens):
                row = x[t : t + 1]
                s = row_scale(row, 127.0)
                s_i8.append(s)
                rows_i8.append(ref_int8.scaled_int8_quant(row, s))
                s = row_scale(row, 448.0)
                s_fn.append(s)
                rows_fn.append(fp8_quant(row, s, fnuz=False))
                s = row_scale(row, 240.0)
                s_fnuz.append(s)
                rows_fnuz.append(fp8_quant(row, s, fnuz=True))
            out[f"x_{key}"] = bits(x)
            out[f"qi8_{key}"] = bits(torch.cat(rows_i8))
            out[f"si8_{key}"] = torch.cat(s_i8).numpy().copy()
            out[f"qfn_{key}"] = bits(torch.cat(rows_fn))
            out[f"sfn_{key}"] = torch.cat(s_fn).numpy().copy()
            out[f"qfnuz_{key}"] = bits(torch.cat(rows_fnuz))
            out[f"sfnuz_{key}"] = torch.cat(s_fnuz).numpy().copy()
    np.savez_compressed(OUT / "quant_dynamic.npz", **out)


GENERATORS = {"dyn": gen_dynamic}

if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference checkout is only available in the authoring container"
    torch.set_num_threads(1)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for name, fn in GENERATORS.items():
        if only in (None, name):
            fn()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
