"""Int8 quantization (public API; mirrors conch/ops/quantization/int8.py:11-48)."""

import torch

from conch_amd.kernels.quantization.int8 import static_scaled_int8_quant_launcher


def static_scaled_int8_quant(
    output_tensor: torch.Tensor,
    input_tensor: torch.Tensor,
    scale: torch.Tensor,
) -> None:
    """Quantize `input_tensor` to int8 into `output_tensor` with the static per-tensor `scale`.

    Args:
        output_tensor: int8 tensor to write, shape (num_tokens, hidden_size).
        input_tensor: fp16/bf16/fp32 tensor to quantize, same shape.
        scale: fp32 tensor with one element.
    """
    assert output_tensor.shape == input_tensor.shape  # noqa: S101
    assert scale.numel() == 1  # noqa: S101

    static_scaled_int8_quant_launcher(output_tensor, input_tensor, scale)


def scaled_int8_quant(
    input_tensor: torch.Tensor,
    scale: torch.Tensor | None = None,
) -> tuple[torch.Tensor, torch.Tensor]:
    """Scaled int8 quantization; returns (quantized tensor, the same `scale` object)."""
    if scale is None:
        error_msg = "Dynamic int8 quantization not yet implemented"
        raise NotImplementedError(error_msg)

    # every element is written by the kernel, so no memset is needed (the reference zero-fills)
    output_tensor = torch.empty_like(input_tensor, dtype=torch.int8)
    static_scaled_int8_quant(output_tensor, input_tensor, scale)
    return output_tensor, scale
