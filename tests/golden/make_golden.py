"""Generate golden vectors by running the REAL reference (conch @ /root/reference) on CPU.

Run in the authoring container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes small .npz fixtures next to this file.  Inputs are built with the same recipes and seed
as the reference's own tests (tests/scaled_gemm_test.py, tests/mixed_precision_gemm_test.py,
tests/int8_quant_kernels_test.py, tests/fp8_quant_kernels_test.py); outputs come from
conch.reference.* / conch.third_party.vllm.quant_utils -- nothing from this repo is involved.
16-bit float tensors are stored as their raw uint16 bit patterns, fp8 tensors as uint8 codes.
"""

from __future__ import annotations

import os
import sys
from pathlib import Path
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
if REF not in sys.path:
    sys.path.insert(0, REF)

from conch.platforms import current_platform  # noqa: E402
from conch.reference.quantization import fp8 as ref_fp8  # noqa: E402
from conch.reference.quantization import int8 as ref_int8  # noqa: E402
from conch.reference.quantization import scaled_gemm as ref_gemm  # noqa: E402
from conch.third_party.vllm.quant_utils import pack_rows, quantize_weights  # noqa: E402
from conch.third_party.vllm.scalar_type import scalar_types  # noqa: E402
from conch.third_party.vllm.utils import seed_everything  # noqa: E402

OUT = Path(__file__).resolve().parent
DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def bits(t: torch.Tensor) -> np.ndarray:
    """Raw storage of a tensor as a numpy array."""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
        return t.view(torch.uint8).numpy().copy()
    return t.numpy().copy()


def fp8_quant(x: torch.Tensor, scale: torch.Tensor, fnuz: bool) -> torch.Tensor:
    """Run the reference's fp8 oracle with its platform switch forced (fp8.py:14)."""
    with mock.patch.object(type(current_platform), "is_amd", lambda self: fnuz):
        return ref_fp8.scaled_fp8_quant(x, scale)


def gen_quant() -> None:
    i8: dict[str, np.ndarray] = {}
    f8: dict[str, np.ndarray] = {}
    for dname, dtype in DT.items():
        for tokens in (1, 7, 83):
            for hidden in (16, 67, 768):
                for scale in (0.1, 2.1):
                    key = f"{dname}_t{tokens}_h{hidden}_s{scale}"
                    seed_everything(0)
                    x = torch.rand(tokens, hidden, dtype=dtype) * 1000  # int8 test :42
                    s = torch.tensor([scale], dtype=torch.float32)  # :43
                    i8[f"x_{key}"] = bits(x)
                    i8[f"q_{key}"] = bits(ref_int8.scaled_int8_quant(x, s))
                    seed_everything(0)
                    x = torch.rand(tokens, hidden, dtype=dtype)  # fp8 test :49
                    f8[f"x_{key}"] = bits(x)
                    f8[f"qfn_{key}"] = bits(fp8_quant(x, s, fnuz=False))
                    f8[f"qfnuz_{key}"] = bits(fp8_quant(x, s, fnuz=True))
    # edge vectors: ties, negative fractions, overflow both ways, tiny values, both scale shapes
    edge = torch.tensor(
        [0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -1.9, 64.4, 126.5, 127.5, 128.5, -128.5, -129.5, 300.0, -300.0,
         1e-3, -1e-3, 17.0, 19.0, 21.0, 23.0, 232.0, 240.0, 248.0, 440.0, 448.0, 464.0, 465.0, 1e4, -1e4,
         2.0**-6, 2.0**-7, 2.0**-9, 2.0**-10, 1.5 * 2.0**-10, 2.0**-11, 3.3e-4],
        dtype=torch.float32,
    ).reshape(1, -1)
    for dname, dtype in DT.items():
        xe = edge.to(dtype)
        for sname, s in (("one", torch.tensor([1.0])), ("s2p1", torch.tensor([2.1])), ("zdim", torch.tensor(0.5))):
            i8[f"edge_x_{dname}_{sname}"] = bits(xe)
            i8[f"edge_s_{dname}_{sname}"] = s.numpy().copy()
            i8[f"edge_q_{dname}_{sname}"] = bits(ref_int8.scaled_int8_quant(xe, s))
            f8[f"edge_x_{dname}_{sname}"] = bits(xe)
            f8[f"edge_s_{dname}_{sname}"] = s.numpy().copy()
            f8[f"edge_qfn_{dname}_{sname}"] = bits(fp8_quant(xe, s, fnuz=False))
            f8[f"edge_qfnuz_{dname}_{sname}"] = bits(fp8_quant(xe, s, fnuz=True))
    np.savez_compressed(OUT / "quant_int8.npz", **i8)
    np.savez_compressed(OUT / "quant_fp8.npz", **f8)


def gen_scaled_gemm() -> None:
    m, k, n = 128, 256, 128
    out: dict[str, np.ndarray] = {}
    in_types = {"int8": torch.int8, "fn": torch.float8_e4m3fn, "fnuz": torch.float8_e4m3fnuz}
    for iname, idt in in_types.items():
        for oname in ("f16", "bf16"):
            odt = DT[oname]
            for sa_scalar in (True, False):
                for sb_scalar in (True, False):
                    for use_bias in (True, False):
                        seed_everything(0)
                        # recipe: tests/scaled_gemm_test.py:54-73
                        sa = torch.rand((1, 1), dtype=torch.float32) if sa_scalar else 0.25 * torch.rand((m, 1))
                        sb = torch.rand((1, 1), dtype=torch.float32) if sb_scalar else 0.25 * torch.rand((n, 1))
                        if idt.is_floating_point:
                            a = (0.25 * torch.rand((m, k), dtype=torch.float32)).to(idt)
                            b = (0.25 * torch.rand((n, k), dtype=torch.float32)).to(idt).T
                        else:
                            a = torch.randint(-32, 32, (m, k), dtype=idt)
                            b = torch.randint(-32, 32, (n, k), dtype=idt).T
                        bias = torch.rand((n,), dtype=odt) if use_bias else None
                        c = ref_gemm.scaled_gemm(a, b, sa, sb, odt, bias)
                        key = f"{iname}_{oname}_sa{int(sa_scalar)}_sb{int(sb_scalar)}_b{int(use_bias)}"
                        out[f"a_{key}"] = bits(a)
                        out[f"bt_{key}"] = bits(b.T)  # stored K-contiguous as (N, K); b = bt.T
                        out[f"sa_{key}"] = sa.numpy().copy()
                        out[f"sb_{key}"] = sb.numpy().copy()
                        if bias is not None:
                            out[f"bias_{key}"] = bits(bias)
                        out[f"c_{key}"] = bits(c)
    np.savez_compressed(OUT / "scaled_gemm.npz", **out)


def gen_mixed() -> None:
    m, k, n, group = 32, 256, 128, 128
    out: dict[str, np.ndarray] = {}
    wtypes = {
        "uint4b8": scalar_types.uint4b8,
        "uint8b128": scalar_types.uint8b128,
        "uint4": scalar_types.uint4,
        "uint8": scalar_types.uint8,
    }
    for wname, wt in wtypes.items():
        for zp in (True, False):
            for dname in ("f16", "bf16"):
                dtype = DT[dname]
                seed_everything(0)
                # recipe: tests/mixed_precision_gemm_test.py:60-70
                a = (10 * (torch.rand((m, k), dtype=torch.float32) - 0.3)).to(dtype)
                b = (10 * (torch.rand((k, n), dtype=torch.float32) - 0.3)).to(dtype)
                w_ref, w_q, w_s, w_zp = quantize_weights(
                    b, wt, group, zero_points=zp, ref_zero_points_after_scales=False
                )
                packed = pack_rows(w_q, wt.size_bits, *w_q.shape)
                c = torch.matmul(a, w_ref)
                key = f"{wname}_zp{int(zp)}_{dname}"
                out[f"a_{key}"] = bits(a)
                out[f"b_{key}"] = bits(b)
                out[f"wref_{key}"] = bits(w_ref)
                out[f"wq_{key}"] = bits(w_q.to(torch.int32))
                out[f"packed_{key}"] = bits(packed)
                out[f"ws_{key}"] = bits(w_s)
                if w_zp is not None:
                    out[f"wzp_{key}"] = bits(w_zp)
                out[f"c_{key}"] = bits(c)
                out[f"meta_{key}"] = np.array([wt.size_bits, wt.bias, group], dtype=np.int64)
    np.savez_compressed(OUT / "mixed_gemm.npz", **out)


def gen_scaled_gemm_silu() -> None:
    """the FFN pair scaled_gemm -> silu_and_mul, both from the reference.

    Input: the reference's scaled_gemm outputs already stored in scaled_gemm.npz (c_<key>, 128 x 128: 64 gate + 64 up
    columns); output: conch.reference.activation.silu_and_mul.silu_and_mul (reference/activation/silu_and_mul.py:13-16)
    applied to them.  Only the new outputs are stored (y_<key>, 128 x 64).
    """
    from conch.reference.activation.silu_and_mul import silu_and_mul as ref_silu_and_mul

    g = np.load(OUT / "scaled_gemm.npz")
    out: dict[str, np.ndarray] = {}
    for name in g.files:
        if not name.startswith("c_"):
            continue
        key = name[2:]
        odt = torch.float16 if "_f16_" in key else torch.bfloat16
        c = torch.from_numpy(g[name].view(np.int16).copy()).view(odt)
        out[f"y_{key}"] = bits(ref_silu_and_mul(c))
    np.savez_compressed(OUT / "scaled_gemm_silu.npz", **out)
    # the same for the mixed-precision GEMM outputs (c_<key> of mixed_gemm.npz, 32 x 128 -> 32 x 64)
    g = np.load(OUT / "mixed_gemm.npz")
    out = {}
    for name in g.files:
        if not name.startswith("c_"):
            continue
        key = name[2:]
        odt = torch.float16 if key.endswith("_f16") else torch.bfloat16
        c = torch.from_numpy(g[name].view(np.int16).copy()).view(odt)
        out[f"y_{key}"] = bits(ref_silu_and_mul(c))
    np.savez_compressed(OUT / "mixed_gemm_silu.npz", **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference checkout is only available in the authoring container"
    torch.set_num_threads(1)  # deterministic reduction order for the fp matmuls
    if "--only-silu" in sys.argv:
        gen_scaled_gemm_silu()
        sys.exit(0)
    gen_quant()
    gen_scaled_gemm()
    gen_mixed()
    gen_scaled_gemm_silu()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
