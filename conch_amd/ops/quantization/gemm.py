"""Quantized GEMMs (public API): `scaled_gemm` and `mixed_precision_gemm`.

Drop-in for conch/ops/quantization/gemm.py (mixed_precision_gemm :133-168, scaled_gemm :233-252); the
metadata deduction those call lives in `_metadata.py` and is re-exported here under the reference's
names.  The C library rejects requests it cannot serve even when `strict=False`.
"""

from __future__ import annotations

import torch

from conch_amd.kernels.quantization import _fast
from conch_amd.kernels.quantization.gemm import (
    mixed_precision_gemm_gelu_tanh_and_mul_launcher,
    mixed_precision_gemm_launcher,
    mixed_precision_gemm_silu_and_mul_launcher,
    scaled_gemm_gelu_tanh_and_mul_launcher,
    scaled_gemm_launcher,
    scaled_gemm_silu_and_mul_launcher,
    static_quant_scaled_gemm_launcher,
)
from conch_amd.ops.quantization import _compile
from conch_amd.ops.quantization._metadata import create_mixed_precision_metadata, create_scaled_metadata

__all__ = ["create_mixed_precision_metadata", "create_scaled_metadata", "mixed_precision_gemm", "scaled_gemm",
           "scaled_gemm_silu_and_mul", "mixed_precision_gemm_silu_and_mul", "scaled_gemm_gelu_tanh_and_mul",
           "mixed_precision_gemm_gelu_tanh_and_mul", "static_quant_scaled_gemm"]


def mixed_precision_gemm(x: torch.Tensor, w_q_packed: torch.Tensor, w_s: torch.Tensor, w_zp: torch.Tensor | None,
                         weight_size_bits: int, weight_bias: int, group_size: int, *,
                         output_dtype: torch.dtype | None = None, acc_dtype: torch.dtype | None = None,
                         meta_dtype: torch.dtype | None = None, scaled_activations: bool = False,
                         strict: bool = False) -> torch.Tensor:
    """x @ dequant(w_q_packed): int4/int8 group-quantized packed weights, fp16/bf16 activations."""
    plain = not (strict or scaled_activations) and output_dtype is None and acc_dtype is None and meta_dtype is None
    if plain and _compile.compiling():  # torch.compile: one opaque custom op (see _compile.py)
        return torch.ops.conch_amd.mixed_precision_gemm(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size)
    if plain:
        out = _fast.mixed_precision_gemm(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size)
        if out is not None:
            return out
    meta = create_mixed_precision_metadata(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size,
                                           output_dtype=output_dtype, acc_dtype=acc_dtype, meta_dtype=meta_dtype,
                                           scaled_activations=scaled_activations, strict=strict)
    # every element is written by the kernel: no zero fill
    out = x.new_empty((meta.m_dim, meta.n_dim), dtype=meta.output_dtype)
    mixed_precision_gemm_launcher(out, x, w_q_packed, w_s, w_zp, meta)
    return out


def scaled_gemm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                output_dtype: torch.dtype, bias: torch.Tensor | None = None, strict: bool = False) -> torch.Tensor:
    """cast(scale_b * (scale_a * (a @ b))) [+ bias] for int8 / fp8 operands, fp16 / bf16 output."""
    if _compile.compiling():  # torch.compile: one opaque custom op (see _compile.py)
        return torch.ops.conch_amd.scaled_gemm(a, b, scale_a, scale_b, output_dtype, bias)
    if not strict:  # the plain case without the metadata round trip (same C entry point, same arguments: _fast.py)
        out = _fast.scaled_gemm(a, b, scale_a, scale_b, output_dtype, bias)
        if out is not None:
            return out
    meta = create_scaled_metadata(a, b, scale_a, scale_b, output_dtype, strict=strict)
    out = a.new_empty((meta.m_dim, meta.n_dim), dtype=output_dtype)
    # bias goes into the kernel epilogue (the reference runs a separate in-place add afterwards)
    scaled_gemm_launcher(out, a, b, scale_a, scale_b, meta, bias=bias)
    return out


def scaled_gemm_silu_and_mul(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                             output_dtype: torch.dtype, bias: torch.Tensor | None = None, strict: bool = False) -> torch.Tensor:
    """silu_and_mul(scaled_gemm(a, b, scale_a, scale_b, output_dtype, bias)) in one launch.

    `b` is (K, 2d) = [gate | up]; the result is (M, d): silu(G[:, :d]) * G[:, d:] where G is what `scaled_gemm` would
    return -- never written to memory.  Same arguments as `scaled_gemm`; the reference runs the two ops
    (conch.ops.quantization.gemm.scaled_gemm, conch.ops.activation.silu_and_mul) back to back.
    """
    if not strict and not _compile.compiling():  # the plain case through the C++ host path (kernels/quantization/_fast.py)
        out = _fast.scaled_gemm_act(a, b, scale_a, scale_b, output_dtype, bias, 1)
        if out is not None:
            return out
    meta = create_scaled_metadata(a, b, scale_a, scale_b, output_dtype, strict=strict)
    out = a.new_empty((meta.m_dim, meta.n_dim // 2), dtype=output_dtype)
    scaled_gemm_silu_and_mul_launcher(out, a, b, scale_a, scale_b, meta, bias=bias)
    return out


def mixed_precision_gemm_silu_and_mul(x: torch.Tensor, w_q_packed: torch.Tensor, w_s: torch.Tensor, w_zp: torch.Tensor | None,
                                      weight_size_bits: int, weight_bias: int, group_size: int, *,
                                      output_dtype: torch.dtype | None = None, strict: bool = False) -> torch.Tensor:
    """silu_and_mul(mixed_precision_gemm(x, w_q_packed, ...)) in one launch.

    The packed weights, scales and zero points have 2d columns [gate | up]; the result is (M, d).
    """
    if not strict and output_dtype is None and not _compile.compiling():  # the plain case through the C++ host path
        out = _fast.mixed_precision_gemm_act(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size, 1)
        if out is not None:
            return out
    meta = create_mixed_precision_metadata(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size,
                                           output_dtype=output_dtype, strict=strict)
    out = x.new_empty((meta.m_dim, meta.n_dim // 2), dtype=meta.output_dtype)
    mixed_precision_gemm_silu_and_mul_launcher(out, x, w_q_packed, w_s, w_zp, meta)
    return out


def scaled_gemm_gelu_tanh_and_mul(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                                  output_dtype: torch.dtype, bias: torch.Tensor | None = None, strict: bool = False) -> torch.Tensor:
    """gelu_tanh_and_mul(scaled_gemm(a, b, scale_a, scale_b, output_dtype, bias)) in one launch (the GeGLU FFN pair; SURVEY.md
    8(f) N3).  `b` is (K, 2d) = [gate | up]; the result is (M, d): gelu(G[:, :d], approximate="tanh") * G[:, d:].  The
    reference runs conch.ops.quantization.gemm.scaled_gemm and conch.ops.activation.gelu_tanh_and_mul back to back."""
    if not strict and not _compile.compiling():  # the plain case through the C++ host path (kernels/quantization/_fast.py)
        out = _fast.scaled_gemm_act(a, b, scale_a, scale_b, output_dtype, bias, 2)
        if out is not None:
            return out
    meta = create_scaled_metadata(a, b, scale_a, scale_b, output_dtype, strict=strict)
    out = a.new_empty((meta.m_dim, meta.n_dim // 2), dtype=output_dtype)
    scaled_gemm_gelu_tanh_and_mul_launcher(out, a, b, scale_a, scale_b, meta, bias=bias)
    return out


def mixed_precision_gemm_gelu_tanh_and_mul(x: torch.Tensor, w_q_packed: torch.Tensor, w_s: torch.Tensor, w_zp: torch.Tensor | None,
                                           weight_size_bits: int, weight_bias: int, group_size: int, *,
                                           output_dtype: torch.dtype | None = None, strict: bool = False) -> torch.Tensor:
    """gelu_tanh_and_mul(mixed_precision_gemm(x, w_q_packed, ...)) in one launch; packed weights, scales and zero points have 2d
    columns [gate | up], the result is (M, d)."""
    if not strict and output_dtype is None and not _compile.compiling():  # the plain case through the C++ host path
        out = _fast.mixed_precision_gemm_act(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size, 2)
        if out is not None:
            return out
    meta = create_mixed_precision_metadata(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size,
                                           output_dtype=output_dtype, strict=strict)
    out = x.new_empty((meta.m_dim, meta.n_dim // 2), dtype=meta.output_dtype)
    mixed_precision_gemm_gelu_tanh_and_mul_launcher(out, x, w_q_packed, w_s, w_zp, meta)
    return out


def static_quant_scaled_gemm(x: torch.Tensor, b: torch.Tensor, scale_x: torch.Tensor, scale_b: torch.Tensor,
                             output_dtype: torch.dtype, bias: torch.Tensor | None = None) -> torch.Tensor:
    """scaled_gemm(scaled_{int8,fp8}_quant(x, scale_x)[0], b, scale_x, scale_b, output_dtype, bias) in one call, bit for bit
    (the activation quantiser fused into the GEMM).  `x`: fp16 / bf16 activations [M, K]; `b`: int8 / fp8
    weights [K, N] (its dtype is the one x is quantised to); `scale_x`: the static per-tensor activation scale (one float32).
    At decode batch sizes the quantisation happens on the way into the GEMM kernel's LDS; larger M runs the two kernels."""
    if scale_x.dim() == 0 and b.dtype == torch.int8 and x.dtype in (torch.float16, torch.bfloat16):
        # a 0-dim scale makes the int8 quantiser round its product to x's dtype (kernels/quantization/int8.py); the fused
        # kernel multiplies in fp32, so this one case runs the pair it is defined by
        from conch_amd.ops.quantization.int8 import scaled_int8_quant

        return scaled_gemm(scaled_int8_quant(x, scale_x)[0], b, scale_x, scale_b, output_dtype, bias)
    if not _compile.compiling():  # the plain case through the C++ host path, when it is built (kernels/quantization/_fast.py)
        out = _fast.static_quant_scaled_gemm(x, b, scale_x, scale_b, output_dtype, bias)
        if out is not None:
            return out
    out = torch.empty((x.shape[0], b.shape[1]), dtype=output_dtype, device=x.device)
    static_quant_scaled_gemm_launcher(out, x, b, scale_x, scale_b, bias)
    return out


_compile.ensure_defined()
