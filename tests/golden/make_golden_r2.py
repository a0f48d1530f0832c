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


def _pack(wq: torch.Tensor, bits: int) -> torch.Tensor:
    """conch.third_party.vllm.quant_utils.pack_rows for any power-of-two width (it accepts 1/2/4/8 already)."""
    from conch.third_party.vllm.quant_utils import pack_rows

    return pack_rows(wq, bits, *wq.shape)


def gen_modes() -> None:
    """The reference's OWN Triton kernel (`_gemm_kernel`, conch/kernels/quantization/gemm.py:219-457) run through its
    launchers on CPU by Triton's interpreter, for the kernel modes conch.ops never selects.

    fp16 only: the interpreter stores bf16 as raw uint16 and has no bf16 constants, so it cannot run the bf16 form.
    One deviation from the stock interpreter, for ASYMMETRIC only: its `create_fma` evaluates tl.fma as the UNFUSED numpy
    expression x*y+z (two roundings in fp16), which no GPU lowering of tl.fma does; it is replaced by the fused form
    (product and sum in float64, then rounded to fp32 and to fp16 -- the route torch's own double -> half cast takes).
    Everything else is the stock interpreter.

    Activations: the K x K identity (the output IS the dequantised weight matrix, so the dequantisation arithmetic is
    pinned bit for bit) and a random matrix (the accumulation, pinned to fp32-accumulate tolerance by the tests).
    """
    import dataclasses

    from triton.runtime import interpreter as tri

    from conch.kernels.quantization.gemm import (
        ChannelScaleMode,
        WeightGroupMode,
        mixed_precision_gemm_launcher,
        scaled_gemm_launcher,
    )
    from conch.ops.quantization.gemm import create_mixed_precision_metadata, create_scaled_metadata

    def fused_fma(self, x, y, z):
        r = x.data.astype(np.float64) * y.data.astype(np.float64) + z.data.astype(np.float64)
        return tri.TensorHandle(r.astype(np.float32).astype(z.data.dtype), z.dtype.scalar)

    tri.InterpreterBuilder.create_fma = fused_fma

    out: dict[str, np.ndarray] = {}
    dt = torch.float16
    k, n, group = 128, 64, 64
    seed_everything(0)
    x_rand = (2 * torch.rand(32, k) - 1).to(dt)
    x_eye = torch.eye(k, dtype=dt)
    out["x_rand"] = bits(x_rand)
    for nbits, bias in ((2, 2), (4, 8), (4, 0), (8, 128), (8, 0)):
        seed_everything(nbits * 16 + bias)
        wq = torch.randint(0, 2**nbits, (k, n), dtype=torch.int32)
        packed = _pack(wq, nbits)
        ws = (torch.rand(k // group, n) * 0.37 + 0.01).to(dt)
        zi = torch.randint(0, 2**nbits, (k // group, n), dtype=torch.int32)
        zs = torch.tensor([3], dtype=torch.int32)
        zf = (torch.rand(k // group, n) * 5 - 2.5).to(dt)
        cs = (torch.rand(1, n) * 1.5 + 0.25).to(dt)  # per-column scale for ChannelScaleMode.WEIGHT_ONLY
        tag = f"b{nbits}_bias{bias}"
        out[f"packed_{tag}"] = bits(packed)
        out[f"ws_{tag}"] = bits(ws)
        out[f"zi_{tag}"] = bits(zi)
        out[f"zf_{tag}"] = bits(zf)
        out[f"cs_{tag}"] = bits(cs)
        cases = [
            ("none", WeightGroupMode.NONE, ChannelScaleMode.NONE, ws, None),
            ("shift", WeightGroupMode.SHIFT, ChannelScaleMode.NONE, ws, zi),
            ("shift_scalar", WeightGroupMode.SHIFT, ChannelScaleMode.NONE, ws, zs),
            ("sym", WeightGroupMode.SYMMETRIC_NO_SHIFT, ChannelScaleMode.NONE, ws, None),
            ("symshift", WeightGroupMode.SYMMETRIC_WITH_SHIFT, ChannelScaleMode.NONE, ws, zi),
            ("symshift_scalar", WeightGroupMode.SYMMETRIC_WITH_SHIFT, ChannelScaleMode.NONE, ws, zs),
            ("asym", WeightGroupMode.ASYMMETRIC, ChannelScaleMode.NONE, ws, zf),
            ("none_wonly", WeightGroupMode.NONE, ChannelScaleMode.WEIGHT_ONLY, cs, None),
            ("shift_wonly", WeightGroupMode.SHIFT, ChannelScaleMode.WEIGHT_ONLY, cs, zi),
        ]
        for name, gmode, cmode, scales, zeros in cases:
            for xname, x in (("eye", x_eye), ("rand", x_rand)):
                md = create_mixed_precision_metadata(x, packed, ws, zeros, nbits, bias, group)
                md = dataclasses.replace(md, weight_group_mode=gmode, channel_scale_mode=cmode,
                                         zero_is_scalar=zeros is not None and zeros.numel() == 1)
                c = torch.zeros(x.shape[0], n, dtype=dt)
                mixed_precision_gemm_launcher(c, x, packed, scales, zeros, md)
                out[f"c_{tag}_{name}_{xname}"] = bits(c)
    # scaled launcher: the four channel-scale modes on int8 operands
    seed_everything(5)
    m2, k2, n2 = 128, 128, 64
    a = torch.randint(-32, 32, (m2, k2), dtype=torch.int8)
    b = torch.randint(-32, 32, (n2, k2), dtype=torch.int8).T
    sa = 0.25 * torch.rand((m2, 1), dtype=torch.float32)
    sb = 0.25 * torch.rand((n2, 1), dtype=torch.float32)
    out["sc_a"], out["sc_bt"], out["sc_sa"], out["sc_sb"] = bits(a), bits(b.T), sa.numpy().copy(), sb.numpy().copy()
    # fp16 outputs only: the interpreter stores bf16 as raw uint16 and casts into it with a numpy integer astype
    for odt_name, odt in (("f16", torch.float16),):
        for cname, cmode in (("none", ChannelScaleMode.NONE), ("wonly", ChannelScaleMode.WEIGHT_ONLY),
                             ("aonly", ChannelScaleMode.ACTIVATION_ONLY), ("wa", ChannelScaleMode.WEIGHT_AND_ACTIVATION)):
            md = dataclasses.replace(create_scaled_metadata(a, b, sa, sb, odt), channel_scale_mode=cmode)
            c = torch.zeros(m2, n2, dtype=odt)
            # the launcher passes an EMPTY dummy zeros tensor with zero_is_scalar=True (gemm.py:578,:331-332): a GPU reads
            # a garbage word it never uses, the CPU interpreter dereferences a null pointer -> give the dummy one element
            real_tensor = torch.tensor
            with mock.patch("conch.kernels.quantization.gemm.torch.tensor",
                            side_effect=lambda data, **kw: real_tensor([[0]] if data == [[]] else data, **kw)):
                scaled_gemm_launcher(c, a, b, sa, sb, md)
            out[f"sc_c_{odt_name}_{cname}"] = bits(c)
    np.savez_compressed(OUT / "gemm_modes.npz", **out)


def gen_bnb() -> None:
    """bitsandbytes-style blockwise NF4 / FP4 / FP8: the reference's PyTorch implementations
    (conch/reference/quantization/bitsandbytes/{quantize,dequantize}_blockwise.py) on the recipes of its tests
    (tests/quantize_blockwise_test.py:150-170, tests/dequantize_blockwise_test.py:126-140), plus the dynamic 8-bit map
    (conch/ops/quantization/bitsandbytes/functional.py:30-84) and edge inputs (an all-zero block, values on the decision
    thresholds, an odd tail)."""
    from conch.ops.quantization.bitsandbytes.functional import _create_dynamic_map
    from conch.reference.quantization.bitsandbytes.dequantize_blockwise import dequantize_blockwise_launcher as ref_dequant
    from conch.reference.quantization.bitsandbytes.quantize_blockwise import quantize_blockwise_launcher as ref_quant

    out: dict[str, np.ndarray] = {}
    code = _create_dynamic_map()
    out["dynamic_map"] = code.numpy().copy()
    for qt in ("nf4", "fp4", "fp8"):
        for blocksize in (64, 1024):
            for mult in (2.5, 6):
                for dname, dtype in DT.items():
                    key = f"{qt}_b{blocksize}_m{mult}_{dname}"
                    n = int(blocksize * mult)
                    nblocks = -(-n // blocksize)
                    # quantize: tests/quantize_blockwise_test.py:150-170 (seed 2)
                    seed_everything(2)
                    x = torch.randn((n,), dtype=dtype)
                    if qt == "fp4":
                        x = x.uniform_(-1.0, 1.0)
                    absmax = torch.empty((nblocks,), dtype=dtype)
                    q = torch.zeros((n,) if qt == "fp8" else ((n + 1) // 2, 1), dtype=torch.uint8)
                    ref_quant(x, absmax, q.view(-1), code if qt == "fp8" else None, blocksize, n, qt)
                    out[f"qx_{key}"] = bits(x)
                    out[f"qabsmax_{key}"] = bits(absmax)
                    out[f"qout_{key}"] = q.view(-1).numpy().copy()
                    # dequantize: tests/dequantize_blockwise_test.py:126-140 (seed 0)
                    seed_everything(0)
                    nin = n if qt == "fp8" else n // 2
                    xq = torch.randint(0, 255, (nin,), dtype=torch.uint8)
                    am = torch.randn((nblocks,), dtype=dtype)
                    d = torch.empty((n,), dtype=dtype)
                    ref_dequant(xq, am, d, code if qt == "fp8" else None, blocksize, n, qt)
                    out[f"dx_{key}"] = xq.numpy().copy()
                    out[f"dabsmax_{key}"] = bits(am)
                    out[f"dout_{key}"] = bits(d)
    # edges: fp32 input, blocksize 64: block 0 all zero, block 1 = the decision thresholds and their fp32 neighbours,
    # block 2 short and odd (the last element is dropped by the 4-bit packing)
    nf4_t = [-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544, -0.23460740596055984,
             -0.13791173323988914, -0.045525018125772476, 0.03979014977812767, 0.1202552504837513, 0.2035212516784668,
             0.2920137718319893, 0.3893125355243683, 0.5016634166240692, 0.6427869200706482, 0.8614784181118011]
    fp4_t = [0.00260417, 0.0859375, 0.208333334, 0.29166667, 0.4166667, 0.5833334, 0.83333334]
    for qt, ts in (("nf4", nf4_t), ("fp4", fp4_t + [-t for t in fp4_t])):
        t32 = np.array(ts, dtype=np.float32)
        probe = np.concatenate([t32, np.nextafter(t32, np.float32(2)), np.nextafter(t32, np.float32(-2)), [1.0, -1.0, 0.0, -0.0]]).astype(np.float32)
        probe = np.resize(probe, 64).astype(np.float32)
        probe[0] = 1.0  # the block's absmax: the other values are then quantised as they stand
        x = torch.from_numpy(np.concatenate([np.zeros(64, np.float32), probe, np.linspace(-0.9, 0.7, 11, dtype=np.float32)]))
        n = x.numel()
        absmax = torch.empty((3,), dtype=torch.float32)
        q = torch.zeros(((n + 1) // 2,), dtype=torch.uint8)
        ref_quant(x, absmax, q, None, 64, n, qt)
        out[f"edge_x_{qt}"] = x.numpy().copy()
        out[f"edge_absmax_{qt}"] = absmax.numpy().copy()
        out[f"edge_q_{qt}"] = q.numpy().copy()
    np.savez_compressed(OUT / "bnb_blockwise.npz", **out)


def gen_gelu() -> None:
    """SURVEY.md 8(f) N3, second activation: conch.reference.activation.gelu_tanh_and_mul applied to the reference's GEMM outputs
    already stored in scaled_gemm.npz / mixed_gemm.npz (c_<key>); only the new outputs are stored (y_<key>)."""
    from conch.reference.activation.gelu_tanh_and_mul import gelu_tanh_and_mul as ref_gelu

    for src, dst in (("scaled_gemm.npz", "scaled_gemm_gelu.npz"), ("mixed_gemm.npz", "mixed_gemm_gelu.npz")):
        gz = np.load(OUT / src)
        out: dict[str, np.ndarray] = {}
        for name in gz.files:
            if not name.startswith("c_"):
                continue
            key = name[2:]
            odt = torch.float16 if ("_f16_" in key or key.endswith("_f16")) else torch.bfloat16
            c = torch.from_numpy(gz[name].view(np.int16).copy()).view(odt)
            out[f"y_{key}"] = bits(ref_gelu(c))
        np.savez_compressed(OUT / dst, **out)


GENERATORS = {"dyn": gen_dynamic, "modes": gen_modes, "bnb": gen_bnb, "gelu": gen_gelu}

if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference checkout is only available in the authoring container"
    torch.set_num_threads(1)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for name, fn in GENERATORS.items():
        if only in (None, name):
            fn()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
