"""Launchers and metadata for the quantized GEMMs (C-ABI seam).

Same names and argument meaning as conch/kernels/quantization/gemm.py (enums :20-74, metadata
dataclasses :460-479 and :548-561, launchers :482 and :564); the Triton kernel behind them is
replaced by the hand-written gfx950 kernels of csrc/gemm_*.hip, selected inside the C library.
"""

from __future__ import annotations

from dataclasses import dataclass
from enum import Enum

import torch

from conch_amd import _C


class DType(Enum):
    """Element-type ids shared with the C ABI (conch_dtype_t)."""

    FP32 = 0
    FP16 = 1
    BF16 = 2
    FP8 = 3
    INT8 = 4
    UINT8 = 5
    INT32 = 6
    UINT32 = 7
    FP8E5 = 8
    FP8FNUZ = 9


class WeightGroupMode(Enum):
    """How group metadata enters the dequantisation."""

    NONE = 0
    SHIFT = 1
    SYMMETRIC_NO_SHIFT = 2
    SYMMETRIC_WITH_SHIFT = 3
    ASYMMETRIC = 4


class ChannelScaleMode(Enum):
    """Which per-channel scales the epilogue applies."""

    NONE = 0
    WEIGHT_ONLY = 1
    ACTIVATION_ONLY = 2
    WEIGHT_AND_ACTIVATION = 3


@dataclass
class MixedPrecisionMatmulMetadata:
    m_dim: int
    k_dim: int
    n_dim: int
    weight_size_bits: int
    weight_bias: int
    group_size: int
    elements_per_sample: int
    zero_is_scalar: bool
    unpack_mask: int
    data_contiguous: bool
    input_dtype: torch.dtype
    output_dtype: torch.dtype
    acc_dtype: torch.dtype
    meta_dtype: torch.dtype
    channel_scale_mode: ChannelScaleMode
    weight_group_mode: WeightGroupMode


@dataclass
class ScaledMatmulMetadata:
    m_dim: int
    k_dim: int
    n_dim: int
    data_contiguous: bool
    input_dtype: torch.dtype
    output_dtype: torch.dtype
    acc_dtype: torch.dtype
    meta_dtype: torch.dtype
    channel_scale_mode: ChannelScaleMode
    weight_group_mode: WeightGroupMode


def _as_fp32_vector(scale: torch.Tensor, what: str) -> torch.Tensor:
    if scale.dtype != torch.float32:
        raise NotImplementedError(f"{what} must be float32 (got {scale.dtype})")
    flat = scale.reshape(-1)
    return flat if flat.stride(0) == 1 else flat.contiguous()


def _scaled_gemm_call(
    fn_name: str,
    output: torch.Tensor,
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    metadata: ScaledMatmulMetadata,
    bias: torch.Tensor | None,
    extra: tuple = (),
    n_out: int | None = None,
    lib=None,
    extra_before_stream: tuple = (),
) -> int:
    """`n_out`: columns of `output` when they are not metadata.n_dim (the fused gate/up form passes N / 2); `lib`: a
    library handle other than the default one (bench.py times the diagnostic twin through the same marshalling)."""
    _C.require_device(output, a, b, scale_a, scale_b, bias)
    if a.dtype != b.dtype:
        raise ValueError(f"a and b must share a dtype (a: {a.dtype}, b: {b.dtype})")
    if output.dtype != metadata.output_dtype:
        raise ValueError(f"output dtype {output.dtype} != metadata.output_dtype {metadata.output_dtype}")
    # the launcher trusts `metadata` for M, N, K (as the reference's does): a mismatch with the tensors would be an
    # out-of-bounds access on the device, so it is checked here
    if tuple(a.shape) != (metadata.m_dim, metadata.k_dim) or tuple(b.shape) != (metadata.k_dim, metadata.n_dim):
        raise ValueError(f"a {tuple(a.shape)} / b {tuple(b.shape)} do not match metadata (M, K, N) = "
                         f"({metadata.m_dim}, {metadata.k_dim}, {metadata.n_dim})")
    if tuple(output.shape) != (metadata.m_dim, metadata.n_dim if n_out is None else n_out):
        raise ValueError(f"output shape {tuple(output.shape)} does not match metadata")
    sa = _as_fp32_vector(scale_a, "scale_a")
    sb = _as_fp32_vector(scale_b, "scale_b")
    # channel-scale modes of the launcher (kernels/quantization/gemm.py:408-440): a side that is not applied is passed as
    # NULL, which the C ABI reads as the constant 1.  conch.ops always builds WEIGHT_AND_ACTIVATION (ops gemm.py:228).
    mode = metadata.channel_scale_mode
    if mode in (ChannelScaleMode.NONE, ChannelScaleMode.WEIGHT_ONLY):
        sa = None
    if mode in (ChannelScaleMode.NONE, ChannelScaleMode.ACTIVATION_ONLY):
        sb = None
    if bias is not None:
        if bias.dtype != output.dtype:
            # only the fused gate/up forms get here (no reference to match: the bias is added before the activation, in
            # fp32, after this one rounding of the bias itself to the output dtype)
            bias = bias.to(output.dtype)
        bias = bias.reshape(-1).contiguous()
        if bias.numel() != metadata.n_dim:
            raise ValueError(f"bias has {bias.numel()} elements, want N={metadata.n_dim}")
    args = (
        _C.ptr(output), _C.ptr(a), _C.ptr(b), _C.ptr(sa), _C.ptr(sb), _C.ptr(bias),
        metadata.m_dim, metadata.n_dim if n_out is None else n_out, metadata.k_dim,
        a.stride(0), a.stride(1), b.stride(0), b.stride(1), output.stride(0), output.stride(1),
        1 if sa is None else sa.numel(), 1 if sb is None else sb.numel(), _C.dtype_id(a.dtype), _C.dtype_id(output.dtype),
        *extra_before_stream, _C.current_stream_handle(a.device), *extra,
    )
    with _C.on_device_of(output, a, b, sa, sb, bias):
        return getattr(lib or _C.load(), fn_name)(*args)


def scaled_gemm_launcher(
    output: torch.Tensor,
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    metadata: ScaledMatmulMetadata,
    bias: torch.Tensor | None = None,
) -> None:
    """output = cast(scale_b * (scale_a * (a @ b))) [+ bias].

    Same positional arguments as the reference launcher (kernels/quantization/gemm.py:564-571);
    `bias` is an extension: the reference adds it with a second kernel
    (ops/quantization/gemm.py:249-250), here the epilogue fuses it with identical rounding.
    """
    # The fused add reads the bias in the OUTPUT dtype.  A bias of another dtype (the reference's `output.add_(bias)`
    # computes out + bias in the promoted type, e.g. fp32, and rounds once) is therefore added the reference's way, by a
    # second pass -- casting it first would round the bias itself.
    late_bias = bias if bias is not None and bias.dtype != output.dtype else None
    status = _scaled_gemm_call("conch_scaled_gemm", output, a, b, scale_a, scale_b, metadata, None if late_bias is not None else bias)
    _C.check(status, "scaled_gemm")
    if late_bias is not None:
        if late_bias.numel() != metadata.n_dim:
            raise ValueError(f"bias has {late_bias.numel()} elements, want N={metadata.n_dim}")
        output.add_(late_bias.reshape(-1))


def scaled_gemm_multi_launcher(outputs, a, b, scale_a, scale_b, metadata: ScaledMatmulMetadata, bias: torch.Tensor | None = None) -> None:
    """conch_scaled_gemm_multi: the product stored into every tensor of `outputs` (1 to 8 views of one shape, dtype and row
    stride: the N-sharded multi-GPU form whose epilogue does the all-gather -- outputs[0] local, the rest peer-mapped results;
    conch_amd/distributed.py, direct mode).  No reference counterpart.  Tile-kernel layouts only."""
    import ctypes

    first = outputs[0]
    if not 1 <= len(outputs) <= 8:
        raise ValueError(f"1 to 8 destinations, got {len(outputs)}")
    for t in outputs[1:]:
        if t.shape != first.shape or t.dtype != first.dtype or t.stride() != first.stride() or t.device != first.device:
            raise ValueError("every destination needs the shape, dtype, strides and (mapped) device of the first")
    more = (ctypes.c_void_p * max(1, len(outputs) - 1))(*[t.data_ptr() for t in outputs[1:]])
    status = _scaled_gemm_call("conch_scaled_gemm_multi", first, a, b, scale_a, scale_b, metadata, bias,
                               extra_before_stream=(more, len(outputs) - 1))
    _C.check(status, "scaled_gemm_multi")


def static_quant_scaled_gemm_launcher(
    output: torch.Tensor,
    x: torch.Tensor,
    b: torch.Tensor,
    scale_x: torch.Tensor,
    scale_b: torch.Tensor,
    bias: torch.Tensor | None = None,
) -> None:
    """output = scaled_gemm(static_scaled_{int8,fp8}_quant(x, scale_x), b, scale_x, scale_b) [+ bias], bit for bit, without the
    quantised copy of x where the skinny-M kernel runs the shape (the quantiser fused into the GEMM prologue).

    Replaces the pair static_scaled_int8_quant_launcher / static_scaled_fp8_quant_launcher (kernels/quantization/int8.py:63-97,
    fp8.py:65-97) + scaled_gemm_launcher (kernels/quantization/gemm.py:564-627).  `b` fixes the quantised dtype."""
    _C.require_device(output, x, b, scale_x, scale_b, bias)
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"x must be float16 or bfloat16 (got {x.dtype})")
    if x.dim() != 2 or b.dim() != 2 or x.shape[1] != b.shape[0]:
        raise ValueError(f"x {tuple(x.shape)} does not multiply b {tuple(b.shape)}")
    m, k = x.shape
    n = b.shape[1]
    if tuple(output.shape) != (m, n):
        raise ValueError(f"output shape {tuple(output.shape)} != {(m, n)}")
    if scale_x.numel() != 1 or scale_x.dtype != torch.float32:
        raise ValueError("scale_x must be one float32 (static per-tensor activation scale)")
    if scale_x.dim() == 0 and b.dtype == torch.int8:
        raise ValueError("a 0-dim scale_x changes the int8 quantiser's arithmetic (product in x's dtype, like torch's type "
                         "promotion in conch/reference/quantization/int8.py:16): pass a (1,)-shaped scale, or call the op "
                         "conch_amd.ops.quantization.gemm.static_quant_scaled_gemm, which runs the unfused pair for it")
    sb = _as_fp32_vector(scale_b, "scale_b")
    if bias is not None:
        if bias.dtype != output.dtype:
            raise ValueError("bias must have the output dtype")
        bias = bias.reshape(-1).contiguous()
        if bias.numel() != n:
            raise ValueError(f"bias has {bias.numel()} elements, want N={n}")
    with _C.on_device_of(output, x, b, scale_x, sb, bias):
        status = _C.load().conch_static_quant_scaled_gemm(
            _C.ptr(output), _C.ptr(x), _C.ptr(b), _C.ptr(scale_x), _C.ptr(sb), _C.ptr(bias), m, n, k,
            x.stride(0), x.stride(1), b.stride(0), b.stride(1), output.stride(0), output.stride(1), sb.numel(),
            _C.dtype_id(x.dtype), _C.dtype_id(b.dtype), _C.dtype_id(output.dtype), _C.current_stream_handle(x.device),
        )
    _C.check(status, "static_quant_scaled_gemm")


def _scaled_gemm_act_and_mul(fn_name: str, what: str, output, a, b, scale_a, scale_b, metadata: ScaledMatmulMetadata, bias) -> None:
    if metadata.n_dim % 2:
        raise ValueError(f"{what}: B needs an even number of columns [gate | up], got {metadata.n_dim}")
    if output.shape != (metadata.m_dim, metadata.n_dim // 2):
        raise ValueError(f"output shape {tuple(output.shape)} != ({metadata.m_dim}, {metadata.n_dim // 2})")
    if output.stride(1) != 1:
        raise ValueError(f"{what}: output needs unit column stride")
    status = _scaled_gemm_call(fn_name, output, a, b, scale_a, scale_b, metadata, bias, n_out=metadata.n_dim // 2)
    _C.check(status, what)


def scaled_gemm_silu_and_mul_launcher(
    output: torch.Tensor,
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    metadata: ScaledMatmulMetadata,
    bias: torch.Tensor | None = None,
) -> None:
    """output[:, j] = silu(G[:, j]) * G[:, N/2 + j] with G = scaled_gemm(a, b, ...), in one launch.

    The FFN pair `silu_and_mul(scaled_gemm(...))` of the reference (scaled_gemm_launcher,
    kernels/quantization/gemm.py:564-627, then silu_and_mul_launcher, ops/activation/silu_and_mul.py:11-29) with the
    same roundings.  `metadata` describes the GEMM (n_dim = 2 x output columns).
    """
    _scaled_gemm_act_and_mul("conch_scaled_gemm_silu_and_mul", "scaled_gemm_silu_and_mul", output, a, b, scale_a, scale_b, metadata, bias)


def scaled_gemm_gelu_tanh_and_mul_launcher(
    output: torch.Tensor,
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    metadata: ScaledMatmulMetadata,
    bias: torch.Tensor | None = None,
) -> None:
    """output[:, j] = gelu_tanh(G[:, j]) * G[:, N/2 + j] with G = scaled_gemm(a, b, ...), in one launch: the GeGLU pair
    `gelu_tanh_and_mul(scaled_gemm(...))` (conch/ops/activation/gelu_tanh_and_mul.py; kernel
    conch/kernels/activation/gelu_tanh_and_mul.py:60-70) with the reference's roundings."""
    _scaled_gemm_act_and_mul("conch_scaled_gemm_gelu_tanh_and_mul", "scaled_gemm_gelu_tanh_and_mul", output, a, b, scale_a, scale_b,
                             metadata, bias)


def _mixed_gemm_call(
    fn_name: str,
    output: torch.Tensor,
    x: torch.Tensor,
    w_q_packed: torch.Tensor,
    scales: torch.Tensor,
    zeros: torch.Tensor | None,
    metadata: MixedPrecisionMatmulMetadata,
    extra: tuple = (),
    n_out: int | None = None,
    lib=None,
) -> int:
    """`n_out`: columns of `output` when they are not metadata.n_dim (the fused gate/up form passes N / 2)."""
    _C.require_device(output, x, w_q_packed, scales, zeros)
    if metadata.acc_dtype != torch.float32:
        raise NotImplementedError("mixed_precision_gemm accumulates in float32 only")
    if metadata.meta_dtype != x.dtype or scales.dtype != x.dtype:
        raise NotImplementedError("mixed_precision_gemm: scales / meta dtype must equal the activation dtype")
    if w_q_packed.dtype not in (torch.int32, getattr(torch, "uint32", torch.int32)):
        raise ValueError(f"Invalid datatype for packed weights: {w_q_packed.dtype}")
    if x.stride(1) != 1:
        x = x.contiguous()
    if w_q_packed.stride(1) != 1:
        w_q_packed = w_q_packed.contiguous()
    if scales.stride(-1) != 1:
        scales = scales.contiguous()
    if tuple(x.shape) != (metadata.m_dim, metadata.k_dim) or w_q_packed.shape[0] * metadata.elements_per_sample != metadata.k_dim \
            or w_q_packed.shape[1] != metadata.n_dim:
        raise ValueError(f"x {tuple(x.shape)} / w_q_packed {tuple(w_q_packed.shape)} do not match metadata (M, K, N) = "
                         f"({metadata.m_dim}, {metadata.k_dim}, {metadata.n_dim})")
    if tuple(output.shape) != (metadata.m_dim, metadata.n_dim if n_out is None else n_out):
        raise ValueError(f"output shape {tuple(output.shape)} does not match metadata")
    gmode, cmode = metadata.weight_group_mode, metadata.channel_scale_mode
    fused = (cmode == ChannelScaleMode.NONE and metadata.weight_size_bits in (4, 8)
             and (gmode == WeightGroupMode.SYMMETRIC_NO_SHIFT
                  or (gmode == WeightGroupMode.SYMMETRIC_WITH_SHIFT and zeros is not None and not zeros.dtype.is_floating_point)))
    if not fused:
        # a mode conch.ops never produces: the explicit-modes entry point
        if fn_name not in ("conch_mixed_precision_gemm", "conch_time_mixed_precision_gemm"):
            raise NotImplementedError(f"{fn_name}: only the weight-group modes conch.ops produces (SYMMETRIC_NO_SHIFT / "
                                      "SYMMETRIC_WITH_SHIFT on 4- / 8-bit weights, no channel scale)")
        return _mixed_gemm_modes_call(output, x, w_q_packed, scales, zeros, metadata)
    if gmode == WeightGroupMode.SYMMETRIC_NO_SHIFT:
        zp_mode, zeros_t, zp_stride = _C.ZP_NONE, None, 0
    else:
        if zeros.dtype == torch.bool:
            raise ValueError("zero points must be integers")
        if zeros.dtype != torch.int32:
            zeros = zeros.to(torch.int32)  # exact: every integer dtype torch has fits the packed weights' range
        if metadata.zero_is_scalar:
            zp_mode, zeros_t, zp_stride = _C.ZP_SCALAR, zeros.reshape(-1), 0
        else:
            zeros_t = zeros if zeros.stride(-1) == 1 else zeros.contiguous()
            zp_mode, zp_stride = _C.ZP_TENSOR, zeros_t.stride(0)
    args = (
        _C.ptr(output), _C.ptr(x), _C.ptr(w_q_packed), _C.ptr(scales), _C.ptr(zeros_t),
        metadata.m_dim, metadata.n_dim if n_out is None else n_out, metadata.k_dim,
        x.stride(0), w_q_packed.stride(0), scales.stride(0) if scales.dim() == 2 else metadata.n_dim, zp_stride,
        output.stride(0), metadata.weight_size_bits, metadata.weight_bias, metadata.group_size, zp_mode,
        _C.dtype_id(x.dtype), _C.dtype_id(output.dtype), _C.current_stream_handle(x.device), *extra,
    )
    with _C.on_device_of(output, x, w_q_packed, scales, zeros_t):
        return getattr(lib or _C.load(), fn_name)(*args)


_ZEROS_NONE, _ZEROS_SCALAR_INT32, _ZEROS_TENSOR_INT32, _ZEROS_SCALAR_FLOAT, _ZEROS_TENSOR_FLOAT = range(5)


def _mixed_gemm_modes_call(output, x, w_q_packed, scales, zeros, metadata: MixedPrecisionMatmulMetadata) -> int:
    """conch_mixed_precision_gemm_modes: any WeightGroupMode, ChannelScaleMode NONE / WEIGHT_ONLY, 1- to 16-bit weights.

    Mirrors what the reference launcher feeds `_gemm_kernel` (kernels/quantization/gemm.py:482-545): `scales` is the group
    scale tensor [K/G, N] for the SYMMETRIC / ASYMMETRIC modes and, under WEIGHT_ONLY, ALSO the per-column scale the kernel
    reads at `scales_ptr + n` (:409) -- i.e. its first row.  `zeros` of any dtype is converted to the meta dtype on load
    (:363-371): integers go over as int32, floating tensors as the activation dtype.
    """
    gmode, cmode = metadata.weight_group_mode, metadata.channel_scale_mode
    if cmode not in (ChannelScaleMode.NONE, ChannelScaleMode.WEIGHT_ONLY):
        raise NotImplementedError(f"mixed_precision_gemm: {cmode} needs activation scales, which this launcher never passes "
                                  "(kernels/quantization/gemm.py:507 scales_a_ptr=None)")
    needs_zeros = gmode in (WeightGroupMode.SHIFT, WeightGroupMode.SYMMETRIC_WITH_SHIFT, WeightGroupMode.ASYMMETRIC)
    needs_scales = gmode in (WeightGroupMode.SYMMETRIC_NO_SHIFT, WeightGroupMode.SYMMETRIC_WITH_SHIFT, WeightGroupMode.ASYMMETRIC)
    if needs_zeros and zeros is None:
        raise ValueError(f"{gmode} needs zero points")
    zeros_t, zkind, zstride = None, _ZEROS_NONE, 0
    if needs_zeros:
        scalar = zeros.numel() == 1
        if zeros.dtype.is_floating_point:
            zeros_t = zeros.to(x.dtype)
            zkind = _ZEROS_SCALAR_FLOAT if scalar else _ZEROS_TENSOR_FLOAT
        else:
            zeros_t = zeros.to(torch.int32)
            zkind = _ZEROS_SCALAR_INT32 if scalar else _ZEROS_TENSOR_INT32
        zeros_t = zeros_t.reshape(-1) if scalar else (zeros_t if zeros_t.stride(-1) == 1 else zeros_t.contiguous())
        zstride = 0 if scalar else zeros_t.stride(0)
    channel = None
    if cmode == ChannelScaleMode.WEIGHT_ONLY:
        channel = scales.reshape(-1)[: metadata.n_dim].to(x.dtype).contiguous()
        if channel.numel() != metadata.n_dim:
            raise ValueError(f"WEIGHT_ONLY needs {metadata.n_dim} channel scales, got {channel.numel()}")
    group_scales = scales if needs_scales else None
    args = (
        _C.ptr(output), _C.ptr(x), _C.ptr(w_q_packed), _C.ptr(group_scales), _C.ptr(zeros_t), _C.ptr(channel),
        metadata.m_dim, metadata.n_dim, metadata.k_dim,
        x.stride(0), w_q_packed.stride(0), (scales.stride(0) if scales.dim() == 2 else metadata.n_dim) if needs_scales else 0, zstride,
        output.stride(0), metadata.weight_size_bits, metadata.weight_bias, metadata.group_size, gmode.value, zkind, cmode.value,
        _C.dtype_id(x.dtype), _C.dtype_id(output.dtype), _C.current_stream_handle(x.device),
    )
    with _C.on_device_of(output, x, w_q_packed, scales, zeros_t):
        return _C.load().conch_mixed_precision_gemm_modes(*args)


def mixed_precision_gemm_launcher(
    output: torch.Tensor,
    x: torch.Tensor,
    w_q_packed: torch.Tensor,
    scales: torch.Tensor,
    zeros: torch.Tensor | None,
    metadata: MixedPrecisionMatmulMetadata,
) -> None:
    """output = x @ dequant(w_q_packed, scales, zeros)  (reference launcher: gemm.py:482-545)."""
    if output.stride(1) != 1:
        raise ValueError("output must be contiguous in its last dimension")
    status = _mixed_gemm_call("conch_mixed_precision_gemm", output, x, w_q_packed, scales, zeros, metadata)
    _C.check(status, "mixed_precision_gemm")


def _mixed_gemm_act_and_mul(fn_name: str, what: str, output, x, w_q_packed, scales, zeros, metadata: MixedPrecisionMatmulMetadata) -> None:
    if metadata.n_dim % 2:
        raise ValueError(f"{what}: the weights need an even number of columns [gate | up], got {metadata.n_dim}")
    if output.shape != (metadata.m_dim, metadata.n_dim // 2) or output.stride(1) != 1:
        raise ValueError(f"output must be a ({metadata.m_dim}, {metadata.n_dim // 2}) tensor with unit column stride")
    status = _mixed_gemm_call(fn_name, output, x, w_q_packed, scales, zeros, metadata, n_out=metadata.n_dim // 2)
    _C.check(status, what)


def mixed_precision_gemm_silu_and_mul_launcher(
    output: torch.Tensor,
    x: torch.Tensor,
    w_q_packed: torch.Tensor,
    scales: torch.Tensor,
    zeros: torch.Tensor | None,
    metadata: MixedPrecisionMatmulMetadata,
) -> None:
    """output[:, j] = silu(G[:, j]) * G[:, N/2 + j] with G = mixed_precision_gemm(x, w_q_packed, ...), in one launch.

    The FFN pair `silu_and_mul(mixed_precision_gemm(...))` of the reference (mixed_precision_gemm_launcher,
    kernels/quantization/gemm.py:482-545, then silu_and_mul_launcher, ops/activation/silu_and_mul.py:11-29) with the same
    roundings.  `metadata` describes the GEMM (n_dim = 2 x output columns).
    """
    _mixed_gemm_act_and_mul("conch_mixed_precision_gemm_silu_and_mul", "mixed_precision_gemm_silu_and_mul", output, x, w_q_packed,
                            scales, zeros, metadata)


def mixed_precision_gemm_gelu_tanh_and_mul_launcher(
    output: torch.Tensor,
    x: torch.Tensor,
    w_q_packed: torch.Tensor,
    scales: torch.Tensor,
    zeros: torch.Tensor | None,
    metadata: MixedPrecisionMatmulMetadata,
) -> None:
    """The same pair with the gelu-tanh gate (conch/ops/activation/gelu_tanh_and_mul.py after mixed_precision_gemm)."""
    _mixed_gemm_act_and_mul("conch_mixed_precision_gemm_gelu_tanh_and_mul", "mixed_precision_gemm_gelu_tanh_and_mul", output, x,
                            w_q_packed, scales, zeros, metadata)
