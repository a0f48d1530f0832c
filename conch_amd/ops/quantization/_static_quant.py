"""Shared body of the static per-tensor quantisation wrappers (int8.py / fp8.py).

The reference keeps two near-identical wrapper modules (conch/ops/quantization/int8.py, fp8.py); here
the common part -- argument checks, output allocation, launcher call -- lives in one place and the two
public modules only bind dtypes and launchers.
"""

from __future__ import annotations

from collections.abc import Callable

import torch

from conch_amd.kernels.quantization import _fast
from conch_amd.ops.quantization import _compile

_KIND = {torch.int8: 0, torch.float8_e4m3fn: 1, torch.float8_e4m3fnuz: 2}

Launcher = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], None]


def quantize_into(out: torch.Tensor, x: torch.Tensor, scale: torch.Tensor, launcher: Launcher,
                  allowed_out_dtypes: tuple[torch.dtype, ...]) -> None:
    """Checks of the reference's `static_scaled_*_quant` (shape equality, one-element scale, output dtype)."""
    assert out.shape == x.shape  # noqa: S101
    assert scale.numel() == 1  # noqa: S101
    assert out.dtype in allowed_out_dtypes  # noqa: S101
    launcher(out, x, scale)


DynamicLauncher = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], None]


def quantize_new(x: torch.Tensor, scale: torch.Tensor | None, out_dtype: torch.dtype, launcher: Launcher,
                 allowed_out_dtypes: tuple[torch.dtype, ...], dynamic_launcher: DynamicLauncher) -> tuple[torch.Tensor, torch.Tensor]:
    """`scaled_*_quant`: allocate, quantise, hand back (result, scale).

    Static (`scale` given): returns the SAME scale object, like the reference (int8.py:48, fp8.py:58).  The output is not
    zero-filled first (the reference uses zeros_like): the kernel writes every element.
    Dynamic (`scale=None`; the reference raises NotImplementedError, int8.py:42-44, fp8.py:46-48): one scale per token,
    scale[t] = absmax(x[t]) / QMAX, returned with shape x.shape[:-1] + (1,) -- for a 2-D input exactly the (M, 1) fp32
    `scale_a` of `scaled_gemm`.
    """
    known = out_dtype in _KIND and out_dtype in allowed_out_dtypes  # anything else: the general path below raises the proper error
    if scale is not None and known and _compile.compiling():  # torch.compile: one opaque custom op (see _compile.py)
        if out_dtype is torch.int8:
            return torch.ops.conch_amd.static_scaled_int8_quant(x, scale), scale
        return torch.ops.conch_amd.static_scaled_fp8_quant(x, scale, out_dtype is torch.float8_e4m3fnuz), scale
    if scale is not None and known:  # the plain case through the C++ host path, when it is built (kernels/quantization/_fast.py)
        out = _fast.static_quant(x, scale, _KIND[out_dtype])
        if out is not None:
            return out, scale
    if scale is None and not _compile.compiling():  # dynamic per-token, the plain case through the C++ host path
        pair = _fast.dynamic_quant(x, _KIND[out_dtype]) if known else None
        if pair is not None:
            return pair
    out = torch.empty_like(x, dtype=out_dtype)
    if scale is None:
        scales = torch.empty((*x.shape[:-1], 1), dtype=torch.float32, device=x.device)
        dynamic_launcher(out, scales, x)
        return out, scales
    quantize_into(out, x, scale, launcher, allowed_out_dtypes)
    return out, scale


_compile.ensure_defined()
