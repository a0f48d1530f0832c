"""CPU restatement of the reference's quantized-matmul algorithms (TEST INFRASTRUCTURE).

Each function cites the reference file:line (relative to /root/reference) whose arithmetic it
restates.  The reference's oracle is itself PyTorch-on-CPU, so the tensor-level functions use
torch CPU ops in the same order and dtype (that is what fixes the rounding behaviour); the
bit-level FP8 codec and the pack/unpack helpers are independent numpy integer arithmetic so that
torch's own casts are cross-checked rather than trusted.

Parity status: pinned by tests/golden/*.npz (generated from the imported reference by
tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
"""

from __future__ import annotations

import numpy as np
import torch

FP8_E4M3FN = "e4m3fn"  # OCP: bias 7, max 448, NaN = S.1111.111, has -0   (gfx950 native)
FP8_E4M3FNUZ = "e4m3fnuz"  # MI300: bias 8, max 240, NaN = 0x80, no -0

_TORCH_FP8 = {FP8_E4M3FN: torch.float8_e4m3fn, FP8_E4M3FNUZ: torch.float8_e4m3fnuz}
_FP8_MAX = {FP8_E4M3FN: 448.0, FP8_E4M3FNUZ: 240.0}


# --------------------------------------------------------------------------------------
# Bit-level FP8 E4M3 codec (numpy).  Mirrors what `tensor.to(torch.float8_e4m3*)` does:
# round-to-nearest-even, NON-saturating (overflow -> NaN), which is why the reference
# clamps first (conch/reference/quantization/fp8.py:17).
# --------------------------------------------------------------------------------------
def decode_fp8(codes: np.ndarray, flavour: str) -> np.ndarray:
    """uint8 codes -> float32 values."""
    codes = np.asarray(codes, dtype=np.uint8)
    sign = (codes >> 7).astype(np.int32)
    exp = ((codes >> 3) & 0xF).astype(np.int32)
    man = (codes & 0x7).astype(np.int32)
    bias = 7 if flavour == FP8_E4M3FN else 8
    normal = np.ldexp(1.0 + man / 8.0, exp - bias)
    subnormal = np.ldexp(man / 8.0, 1 - bias)
    val = np.where(exp == 0, subnormal, normal)
    val = np.where(sign == 1, -val, val)
    if flavour == FP8_E4M3FN:
        val = np.where((codes & 0x7F) == 0x7F, np.nan, val)
    else:
        val = np.where(codes == 0x80, np.nan, val)
    return val.astype(np.float32)


def encode_fp8(values: np.ndarray, flavour: str) -> np.ndarray:
    """float32 values -> uint8 codes, RNE, overflow -> NaN code (torch cast semantics)."""
    v = np.asarray(values, dtype=np.float32)
    bias = 7 if flavour == FP8_E4M3FN else 8
    fmax = _FP8_MAX[flavour]
    a = np.abs(v).astype(np.float64)
    sign = np.signbit(v).astype(np.uint8)
    # exponent of the value, clamped to the subnormal exponent
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0))).astype(np.int64)
    e = np.maximum(e, 1 - bias)
    # quantum at this exponent is 2^(e-3); round half to even on the integer grid
    q = a / np.ldexp(1.0, e - 3)
    r = np.rint(q)  # numpy rint is round-half-even
    # mantissa overflow (r == 16) bumps the exponent
    bump = r >= 16
    e = np.where(bump, e + 1, e)
    r = np.where(bump, 8, r)
    is_sub = r < 8
    exp_field = np.where(is_sub, 0, e + bias).astype(np.int64)
    man_field = np.where(is_sub, r, r - 8).astype(np.int64)
    code = (exp_field << 3 | man_field).astype(np.int64)
    back = np.ldexp(r, e - 3)
    nan_code = 0x7F if flavour == FP8_E4M3FN else 0x80
    overflow = back > fmax
    out = (code | (sign.astype(np.int64) << 7)).astype(np.int64)
    if flavour == FP8_E4M3FNUZ:
        # no negative zero: anything that rounds to zero is +0
        out = np.where(code == 0, 0, out)
        out = np.where(overflow | np.isnan(v), nan_code, out)
    else:
        out = np.where(overflow | np.isnan(v), nan_code | (sign.astype(np.int64) << 7), out)
    return out.astype(np.uint8)


# --------------------------------------------------------------------------------------
# Static quantisation feeders
# --------------------------------------------------------------------------------------
def scaled_int8_quant_ref(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """conch/reference/quantization/int8.py:12-18.

    (x * (1/scale)).clamp(-128, 127).to(int8).  With the (1,)-shaped fp32 scale the tests use,
    type promotion makes the product fp32; the final cast truncates toward zero.
    """
    inv = scale.reciprocal()
    prod = x * inv
    lim = torch.iinfo(torch.int8)
    return prod.clamp(min=lim.min, max=lim.max).to(torch.int8)


def scaled_fp8_quant_ref(x: torch.Tensor, scale: torch.Tensor, flavour: str = FP8_E4M3FN) -> torch.Tensor:
    """conch/reference/quantization/fp8.py:12-18 with the fp8 flavour made explicit.

    The reference picks e4m3fnuz when `is_amd()` (line 14) and e4m3fn otherwise; both are
    restated so either platform choice can be checked.  fp32 multiply by the reciprocal,
    clamp to +-finfo.max, RNE cast.
    """
    qdtype = _TORCH_FP8[flavour]
    info = torch.finfo(qdtype)
    inv = scale.reciprocal()
    scaled = (x.to(torch.float32) * inv).clamp(min=info.min, max=info.max)
    return scaled.to(qdtype)


def dynamic_quant_scale_ref(x: torch.Tensor, qmax: float) -> torch.Tensor:
    """Per-token scale of the dynamic feeders: absmax over the last dim, in fp32, divided by the
    largest quantised magnitude; 1.0 for an all-zero row (so that the row quantises to zeros instead of NaN).

    The reference has no dynamic path (conch/ops/quantization/int8.py:41-44, fp8.py:46-48 raise); THIS definition is
    the build's own.  The shape (..., 1) fp32 is what scaled_gemm takes as a per-row `scale_a`
    (conch/ops/quantization/gemm.py:199-206)."""
    absmax = x.to(torch.float32).abs().amax(dim=-1, keepdim=True)
    return torch.where(absmax > 0, absmax / qmax, torch.ones_like(absmax))


def dynamic_scaled_int8_quant_ref(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """scale[t] = absmax[t] / 127, then the STATIC op's arithmetic row by row (conch/reference/quantization/int8.py:12-18:
    `(x * scale.reciprocal()).clamp(-128, 127).to(int8)`, fp32 product because the scale is a non-0-dim fp32 tensor)."""
    scale = dynamic_quant_scale_ref(x, 127.0)
    lim = torch.iinfo(torch.int8)
    q = (x * scale.reciprocal()).clamp(min=lim.min, max=lim.max).to(torch.int8)
    return q, scale


def dynamic_scaled_fp8_quant_ref(x: torch.Tensor, flavour: str = FP8_E4M3FN) -> tuple[torch.Tensor, torch.Tensor]:
    """scale[t] = absmax[t] / finfo.max, then conch/reference/quantization/fp8.py:12-18 row by row."""
    qdtype = _TORCH_FP8[flavour]
    info = torch.finfo(qdtype)
    scale = dynamic_quant_scale_ref(x, _FP8_MAX[flavour])
    q = (x.to(torch.float32) * scale.reciprocal()).clamp(min=info.min, max=info.max).to(qdtype)
    return q, scale


# --------------------------------------------------------------------------------------
# scaled_gemm
# --------------------------------------------------------------------------------------
def scaled_gemm_ref(
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    out_dtype: torch.dtype,
    bias: torch.Tensor | None = None,
) -> torch.Tensor:
    """conch/reference/quantization/scaled_gemm.py:12-27.

    fp32 matmul -> scale_a * out -> scale_b.T * out -> cast -> + bias (in out dtype).
    The multiplication ORDER matters for bit-parity on int8 inputs.
    """
    acc = torch.matmul(a.to(torch.float32), b.to(torch.float32))
    acc = scale_a * acc
    acc = scale_b.T * acc
    res = acc.to(out_dtype)
    if bias is not None:
        res = res + bias
    return res


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """conch/reference/activation/silu_and_mul.py:13-16: silu(x[..., :d]) * x[..., d:], d = last dim // 2.

    In a 16-bit dtype torch rounds after each of the two ops: silu is evaluated in fp32 and rounded to the tensor
    dtype, the product of the two 16-bit tensors is rounded again.
    """
    d = x.shape[-1] // 2
    return torch.nn.functional.silu(x[..., :d]) * x[..., d:]


def gelu_tanh_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """conch/reference/activation/gelu_tanh_and_mul.py:13-16: F.gelu(x[..., :d], approximate="tanh") * x[..., d:] -- torch
    evaluates the gelu in fp32 and rounds it to x's dtype, then the product is rounded again."""
    d = x.shape[-1] // 2
    return torch.nn.functional.gelu(x[..., :d], approximate="tanh") * x[..., d:]


def scaled_gemm_gelu_tanh_and_mul_ref(a, b, scale_a, scale_b, out_dtype, bias=None) -> torch.Tensor:
    """The reference's two PyTorch references back to back: scaled_gemm (reference/quantization/scaled_gemm.py:12-27), then
    gelu_tanh_and_mul."""
    return gelu_tanh_and_mul_ref(scaled_gemm_ref(a, b, scale_a, scale_b, out_dtype, bias))


def mixed_precision_gemm_gelu_tanh_and_mul_ref(a: torch.Tensor, w_ref: torch.Tensor) -> torch.Tensor:
    return gelu_tanh_and_mul_ref(mixed_precision_gemm_ref(a, w_ref))


def scaled_gemm_silu_and_mul_ref(
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    out_dtype: torch.dtype,
    bias: torch.Tensor | None = None,
) -> torch.Tensor:
    """The FFN pair the fused op replaces: the reference's scaled_gemm
    (reference/quantization/scaled_gemm.py:12-27) on B = [gate | up], then its silu_and_mul."""
    return silu_and_mul_ref(scaled_gemm_ref(a, b, scale_a, scale_b, out_dtype, bias))


# --------------------------------------------------------------------------------------
# Weight quantisation / packing (the on-device int4/int8 format and the mixed oracle)
# --------------------------------------------------------------------------------------
def _qrange(bits: int, bias: int) -> tuple[int, int]:
    """Unsigned `bits`-wide storage with a bias: representable values are [-bias, 2^bits-1-bias].

    conch/third_party/vllm/scalar_type.py:153-169 (min()/max() are bias-adjusted).
    """
    return -bias, (1 << bits) - 1 - bias


def quantize_weights_ref(
    w: torch.Tensor,
    bits: int,
    bias: int,
    group_size: int,
    zero_points: bool,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor | None]:
    """conch/third_party/vllm/quant_utils.py:14-101 for unsigned integer types with a bias
    (uint4, uint8, uint4b8, uint8b128), grouped along K, ref_zero_points_after_scales=False.

    Returns (w_ref[K,N] in w.dtype, w_q[K,N] int32 incl. bias, w_s[K/G,N] in w.dtype,
    w_zp[K/G,N] int32 or None).  All arithmetic is done in w.dtype with torch CPU ops, as the
    reference does, so that the roundings agree bit-for-bit.
    """
    k, n = w.shape
    assert k % group_size == 0
    groups = k // group_size
    qmin, qmax = _qrange(bits, bias)
    # view as [group, within-group, n]; reductions run over the within-group axis
    wg = w.reshape(groups, group_size, n)
    hi = wg.amax(dim=1, keepdim=True)
    lo = wg.amin(dim=1, keepdim=True)

    zp = None
    if zero_points:
        assert qmax > 0
        s = (hi - lo).clamp(min=1e-5) / qmax  # :53
        zp = torch.round(torch.abs(lo / s)).clamp(qmin, qmax).int()  # :54
    else:
        s = torch.max(  # :58-61
            torch.abs(hi / (qmax if qmax != 0 else float("inf"))),
            torch.abs(lo / (qmin if qmin != 0 else float("inf"))),
        )

    q = torch.round(wg / s).int()  # :64
    if zp is not None:
        q = q + zp
    q = torch.clamp(q, qmin, qmax)  # :65

    centred = q - zp if zp is not None else q
    w_ref = centred.to(w.dtype) * s  # :74  -- ONE rounding multiply after an exact subtraction
    w_q = q + bias  # :76-77

    return (
        w_ref.reshape(k, n).contiguous(),
        w_q.reshape(k, n).contiguous(),
        s.reshape(groups, n).contiguous(),
        None if zp is None else zp.reshape(groups, n).contiguous(),
    )


def pack_rows_ref(w_q: np.ndarray | torch.Tensor, bits: int) -> np.ndarray:
    """conch/third_party/vllm/quant_utils.py:104-125.

    32/bits consecutive K rows go into one 32-bit word: element k sits at word k // pf,
    bit offset (k % pf) * bits.  Returns int32 [K/pf, N].
    """
    q = np.asarray(w_q).astype(np.uint32)
    pf = 32 // bits
    k, n = q.shape
    assert k % pf == 0
    lanes = q.reshape(k // pf, pf, n)
    shifts = (np.arange(pf, dtype=np.uint32) * bits).reshape(1, pf, 1)
    words = np.bitwise_or.reduce(lanes << shifts, axis=1)
    return words.astype(np.uint32).view(np.int32)


def unpack_rows_ref(packed: np.ndarray | torch.Tensor, bits: int) -> np.ndarray:
    """Inverse of pack_rows_ref: int32 [K/pf, N] -> int32 [K, N] (stored values incl. bias).

    conch/kernels/quantization/gemm.py:192-193 and :320-326 (row k//eps, shift (k%eps)*bits).
    """
    p = np.asarray(packed).view(np.uint32)
    pf = 32 // bits
    rows, n = p.shape
    shifts = (np.arange(pf, dtype=np.uint32) * bits).reshape(1, pf, 1)
    q = (p.reshape(rows, 1, n) >> shifts) & np.uint32((1 << bits) - 1)
    return q.reshape(rows * pf, n).astype(np.int32)


def dequantize_packed(
    packed: torch.Tensor,
    w_s: torch.Tensor,
    w_zp: torch.Tensor | None,
    bits: int,
    bias: int,
    group_size: int,
) -> torch.Tensor:
    """The dequantisation `_gemm_kernel` fuses into its K loop, for the two modes reachable
    from conch.ops (conch/kernels/quantization/gemm.py:176-216, ops/quantization/gemm.py:34-38):

      SYMMETRIC_NO_SHIFT   (w_zp None):  (q - bias).to(meta) * s
      SYMMETRIC_WITH_SHIFT (w_zp given): ((q - bias).to(meta) - zp.to(meta)) * s

    meta dtype = w_s.dtype (= x.dtype by default, ops gemm.py:127).  (q - bias - zp) is an exact
    small integer in fp16/bf16, so there is exactly one rounding (the multiply); the result is
    bit-identical to quantize_weights' w_ref.
    """
    q = torch.from_numpy(unpack_rows_ref(packed.cpu().numpy(), bits))
    k, n = q.shape
    groups = k // group_size
    centred = q - bias
    if w_zp is not None:
        if w_zp.numel() == 1:
            centred = centred - int(w_zp.reshape(-1)[0])
        else:
            centred = centred - w_zp.reshape(groups, 1, n).expand(groups, group_size, n).reshape(k, n)
    centred = centred.to(w_s.dtype)
    scales = w_s.reshape(groups, 1, n).expand(groups, group_size, n).reshape(k, n)
    return centred * scales


# weight-group / channel-scale modes of the reference kernel (conch/kernels/quantization/gemm.py:34-41, :68-74)
GROUP_NONE, GROUP_SHIFT, GROUP_SYMMETRIC_NO_SHIFT, GROUP_SYMMETRIC_WITH_SHIFT, GROUP_ASYMMETRIC = range(5)
CHANNEL_NONE, CHANNEL_WEIGHT_ONLY, CHANNEL_ACTIVATION_ONLY, CHANNEL_WEIGHT_AND_ACTIVATION = range(4)


def dequantize_modes_ref(
    packed: torch.Tensor,
    scales: torch.Tensor | None,
    zeros: torch.Tensor | None,
    bits: int,
    bias: int,
    group_size: int,
    group_mode: int,
    meta_dtype: torch.dtype,
) -> torch.Tensor:
    """`_dequantize` for EVERY weight-group mode (conch/kernels/quantization/gemm.py:176-216), each operation carried out
    in `meta_dtype` exactly as the kernel does (torch's fp16 / bf16 elementwise ops are correctly rounded, like the GPU's):

      unpack            b = ((w >> shift) & mask).to(meta); b -= weight_bias                          :192-194
      SHIFT             b -= zeros                                                                     :197-198
      SYMMETRIC_NO_SHIFT    b = b * scales                                                             :201-202
      SYMMETRIC_WITH_SHIFT  b = (b - zeros) * scales                                                   :205-210
      ASYMMETRIC        b = fma(b, scales, zeros)   -- fused: product and sum in float64, then float32, then meta   :213-214

    `zeros`: None | one element | [K/G, N], any dtype, converted to meta on load (:363-371); `scales`: [K/G, N] meta.
    Returns [K, N] in meta_dtype (the kernel then casts to the activation dtype, which is the same here).
    """
    q = torch.from_numpy(unpack_rows_ref(packed.cpu().numpy(), bits))
    k, n = q.shape
    b = q.to(meta_dtype) - bias  # both steps in meta arithmetic (:192-194)

    def per_row(t: torch.Tensor) -> torch.Tensor:
        t = t.to(meta_dtype)
        return t.reshape(1, 1).expand(k, n) if t.numel() == 1 else t.repeat_interleave(group_size, dim=0)

    if group_mode == GROUP_NONE:
        return b
    if group_mode == GROUP_SHIFT:
        return b - per_row(zeros)
    if group_mode == GROUP_SYMMETRIC_NO_SHIFT:
        return b * per_row(scales)
    if group_mode == GROUP_SYMMETRIC_WITH_SHIFT:
        return (b - per_row(zeros)) * per_row(scales)
    if group_mode == GROUP_ASYMMETRIC:
        return (b.double() * per_row(scales).double() + per_row(zeros).double()).float().to(meta_dtype)
    raise ValueError(f"unknown weight group mode {group_mode}")


def mixed_precision_gemm_modes_ref(
    x: torch.Tensor,
    w: torch.Tensor,
    channel_scales: torch.Tensor | None,
    channel_mode: int,
    out_dtype: torch.dtype,
) -> torch.Tensor:
    """x @ w with fp32 accumulation, then the kernel's channel scaling (conch/kernels/quantization/gemm.py:408-416) and the
    cast (:457).  WEIGHT_ONLY: `acc.to(meta) * scales_b[None, :]` -- the accumulator is ROUNDED to the meta dtype first and
    the product is rounded again.  (ACTIVATION_ONLY / WEIGHT_AND_ACTIVATION need the activation scales only the scaled
    launcher passes: scaled_gemm_modes_ref.)"""
    acc = x.float() @ w.float()
    if channel_mode == CHANNEL_NONE:
        return acc.to(out_dtype)
    if channel_mode == CHANNEL_WEIGHT_ONLY:
        meta = w.dtype
        return (acc.to(meta) * channel_scales.reshape(1, -1).to(meta)).to(out_dtype)
    raise ValueError(f"channel mode {channel_mode} is not reachable through the mixed-precision launcher")


def scaled_gemm_modes_ref(a, b, scale_a, scale_b, out_dtype, channel_mode: int) -> torch.Tensor:
    """The scaled launcher with a hand-set ChannelScaleMode (conch/kernels/quantization/gemm.py:408-440): the accumulator
    times nothing (NONE), scales_b (WEIGHT_ONLY), scales_a (ACTIVATION_ONLY) or both, in fp32 (meta dtype = scale_a's),
    then the cast.  WEIGHT_AND_ACTIVATION is scaled_gemm itself, in the PyTorch oracle's order sb * (sa * acc)."""
    acc = a.float() @ b.float()
    if channel_mode == CHANNEL_WEIGHT_ONLY:
        acc = acc * scale_b.reshape(1, -1)
    elif channel_mode == CHANNEL_ACTIVATION_ONLY:
        acc = acc * scale_a.reshape(-1, 1)
    elif channel_mode == CHANNEL_WEIGHT_AND_ACTIVATION:
        acc = scale_b.reshape(1, -1) * (scale_a.reshape(-1, 1) * acc)
    return acc.to(out_dtype)


def mixed_precision_gemm_ref(a: torch.Tensor, w_ref: torch.Tensor) -> torch.Tensor:
    """The mixed-precision oracle is `torch.matmul(a, w_ref)` in the activation dtype
    (tests/mixed_precision_gemm_test.py:70, benchmarks/mixed_precision_gemm_benchmark.py:210)."""
    return torch.matmul(a, w_ref)


def mixed_precision_gemm_silu_and_mul_ref(a: torch.Tensor, w_ref: torch.Tensor) -> torch.Tensor:
    """The int4 / int8-weight FFN pair: the mixed GEMM oracle on [gate | up] weights, then the
    reference's silu_and_mul (reference/activation/silu_and_mul.py:13-16)."""
    return silu_and_mul_ref(mixed_precision_gemm_ref(a, w_ref))
