"""Int8 quantization, public API (names and signatures of conch/ops/quantization/int8.py:11-48)."""

import torch

from conch_amd.kernels.quantization.int8 import dynamic_scaled_int8_quant_launcher, static_scaled_int8_quant_launcher
from conch_amd.ops.quantization._static_quant import quantize_into, quantize_new

_OUT = (torch.int8,)


def static_scaled_int8_quant(output_tensor: torch.Tensor, input_tensor: torch.Tensor, scale: torch.Tensor) -> None:
    """output_tensor[t, h] = int8(clamp(input_tensor[t, h] / scale, -128, 127)), written in place.

    `output_tensor`: int8, shape (num_tokens, hidden_size); `input_tensor`: fp16/bf16/fp32, same shape;
    `scale`: fp32 tensor holding the one static per-tensor scale.
    """
    quantize_into(output_tensor, input_tensor, scale, static_scaled_int8_quant_launcher, _OUT)


def scaled_int8_quant(input_tensor: torch.Tensor, scale: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Quantize to int8; returns (int8 tensor, scale).

    `scale` given: static per-tensor quantisation (the reference's only mode).  `scale=None`: dynamic per-token
    quantisation, scale[t] = absmax(x[t]) / 127 (the reference raises NotImplementedError here).
    """
    return quantize_new(input_tensor, scale, torch.int8, static_scaled_int8_quant_launcher, _OUT,
                        dynamic_scaled_int8_quant_launcher)
