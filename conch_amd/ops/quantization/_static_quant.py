"""Shared body of the static per-tensor quantisation wrappers (int8.py / fp8.py).

The reference keeps two near-identical wrapper modules (conch/ops/quantization/int8.py, fp8.py); here
the common part -- argument checks, output allocation, launcher call -- lives in one place and the two
public modules only bind dtypes and launchers.
"""

from __future__ import annotations

from collections.abc import Callable

import torch

Launcher = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], None]


def quantize_into(out: torch.Tensor, x: torch.Tensor, scale: torch.Tensor, launcher: Launcher,
                  allowed_out_dtypes: tuple[torch.dtype, ...]) -> None:
    """Checks of the reference's `static_scaled_*_quant` (shape equality, one-element scale, output dtype)."""
    assert out.shape == x.shape  # noqa: S101
    assert scale.numel() == 1  # noqa: S101
    assert out.dtype in allowed_out_dtypes  # noqa: S101
    launcher(out, x, scale)


def quantize_new(x: torch.Tensor, scale: torch.Tensor | None, out_dtype: torch.dtype, launcher: Launcher,
                 allowed_out_dtypes: tuple[torch.dtype, ...], dynamic_message: str) -> tuple[torch.Tensor, torch.Tensor]:
    """`scaled_*_quant`: allocate, quantise, hand back (result, the SAME scale object).

    `scale=None` would mean dynamic per-token quantisation, which the reference does not implement
    either (int8.py:42-44, fp8.py:46-48): same NotImplementedError.  The output is not zero-filled first
    (the reference uses zeros_like): the kernel writes every element.
    """
    if scale is None:
        raise NotImplementedError(dynamic_message)
    out = torch.empty_like(x, dtype=out_dtype)
    quantize_into(out, x, scale, launcher, allowed_out_dtypes)
    return out, scale
