"""Shape / dtype deduction for the quantized GEMMs (the `create_*_metadata` half of the public API).

Behaviour of conch/ops/quantization/gemm.py: create_mixed_precision_metadata (:41-130) and
create_scaled_metadata (:171-230).  `strict=False` validates nothing; `strict=True` raises ValueError
for rank / dtype / shape violations and NotImplementedError for scaled activations.
"""

from __future__ import annotations

import torch

from conch_amd.kernels.quantization.gemm import (
    ChannelScaleMode,
    MixedPrecisionMatmulMetadata,
    ScaledMatmulMetadata,
    WeightGroupMode,
)

_MATRIX_RANK = 2
_PACKED_BITS = 32


def _require_rank2(t: torch.Tensor, name: str) -> None:
    if t.dim() != _MATRIX_RANK:
        raise ValueError(f"Unexpected number of dimensions of input tensor {name}: {t.dim()}")


def _check_contiguous(
    x: torch.Tensor,
    w_q_packed: torch.Tensor,
    w_s: torch.Tensor,
    w_zp: torch.Tensor | None,
) -> bool:
    """Contiguity flag recorded in the metadata.

    The reference's expression (gemm.py:27-31) evaluates to True whenever `w_zp is None` because of
    operator precedence; the flag only selected a Triton vectorisation hint, and the HIP kernels
    read real strides instead, so that quirk is kept for metadata compatibility.
    """
    if w_zp is None:
        return True
    return x.is_contiguous() and w_q_packed.is_contiguous() and w_s.is_contiguous() and w_zp.is_contiguous()


def _deduce_weight_group_mode(w_zp: torch.Tensor | None) -> WeightGroupMode:
    return WeightGroupMode.SYMMETRIC_NO_SHIFT if w_zp is None else WeightGroupMode.SYMMETRIC_WITH_SHIFT


def create_mixed_precision_metadata(
    x: torch.Tensor,
    w_q_packed: torch.Tensor,
    w_s: torch.Tensor,
    w_zp: torch.Tensor | None,
    weight_size_bits: int,
    weight_bias: int,
    group_size: int,
    *,
    output_dtype: torch.dtype | None = None,
    acc_dtype: torch.dtype | None = None,
    meta_dtype: torch.dtype | None = None,
    scaled_activations: bool = False,
    strict: bool = False,
) -> MixedPrecisionMatmulMetadata:
    """Deduce (and with `strict` verify) the parameters of a mixed-precision GEMM."""
    if strict:
        _require_rank2(x, "x")
        _require_rank2(w_q_packed, "w_q_packed")
        _require_rank2(w_s, "w_s")
        if w_zp is not None:
            _require_rank2(w_zp, "w_zp")
        packed_ok = (torch.int32, getattr(torch, "uint32", torch.int32))
        if w_q_packed.dtype not in packed_ok:
            raise ValueError(f"Invalid datatype for packed weights: {w_q_packed.dtype}")

    m_dim, k_dim = x.shape
    n_dim = w_q_packed.shape[1]
    zero_is_scalar = w_zp is not None and w_zp.numel() == 1

    if strict:
        want = (k_dim // group_size, n_dim)
        if tuple(w_s.shape) != want:
            raise ValueError(f"Invalid w_s shape (expected: {want}, actual: {tuple(w_s.shape)})")
        if w_zp is not None and not zero_is_scalar and tuple(w_zp.shape) != want:
            raise ValueError(f"Invalid w_zp shape (expected: {want}, actual: {tuple(w_zp.shape)})")
        if scaled_activations:
            msg = "Scaled activations not yet implemented (need to deduce correct channel_scale_mode)"
            raise NotImplementedError(msg)

    return MixedPrecisionMatmulMetadata(
        m_dim=m_dim,
        k_dim=k_dim,
        n_dim=n_dim,
        weight_size_bits=weight_size_bits,
        weight_bias=weight_bias,
        group_size=group_size,
        elements_per_sample=_PACKED_BITS // weight_size_bits,
        zero_is_scalar=zero_is_scalar,
        unpack_mask=2**weight_size_bits - 1,
        data_contiguous=_check_contiguous(x, w_q_packed, w_s, w_zp),
        input_dtype=x.dtype,
        output_dtype=output_dtype if output_dtype is not None else x.dtype,
        acc_dtype=acc_dtype if acc_dtype is not None else torch.float32,
        meta_dtype=meta_dtype if meta_dtype is not None else x.dtype,
        channel_scale_mode=ChannelScaleMode.NONE,
        weight_group_mode=_deduce_weight_group_mode(w_zp),
    )


def create_scaled_metadata(
    a: torch.Tensor,
    b: torch.Tensor,
    scale_a: torch.Tensor,
    scale_b: torch.Tensor,
    output_dtype: torch.dtype,
    strict: bool = False,
) -> ScaledMatmulMetadata:
    """Deduce (and with `strict` verify) the parameters of a scaled GEMM."""
    if strict:
        _require_rank2(a, "a")
        _require_rank2(b, "b")
        if a.dtype != b.dtype:
            raise ValueError(f"Input tensors a and b must have the same datatype (a: {a.dtype}, b: {b.dtype})")

    m_dim, k_dim = a.shape
    n_dim = b.shape[1]

    if strict:
        for scale, name, extent, letter in ((scale_a, "scale_a", m_dim, "m"), (scale_b, "scale_b", n_dim, "n")):
            if scale.numel() == 1:
                continue
            _require_rank2(scale, name)
            if scale.shape[0] != extent:
                raise ValueError(f"Invalid {name} shape (expected: ({extent},), actual: {tuple(scale.shape)})")

    return ScaledMatmulMetadata(
        m_dim=m_dim,
        k_dim=k_dim,
        n_dim=n_dim,
        data_contiguous=all(t.is_contiguous() for t in (a, b, scale_a, scale_b)),
        input_dtype=a.dtype,
        output_dtype=output_dtype,
        acc_dtype=torch.float32 if a.is_floating_point() else torch.int32,
        meta_dtype=scale_a.dtype,
        channel_scale_mode=ChannelScaleMode.WEIGHT_AND_ACTIVATION,
        weight_group_mode=WeightGroupMode.NONE,
    )
