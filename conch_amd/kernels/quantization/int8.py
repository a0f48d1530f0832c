"""Launcher for the static int8 quantisation kernel (C-ABI seam).

Mirrors `static_scaled_int8_quant_launcher` of conch/kernels/quantization/int8.py:63-97: no
allocation, writes into the caller's tensor, asynchronous on torch's current stream.
"""

import torch

from conch_amd import _C


def static_scaled_int8_quant_launcher(
    output_tensor: torch.Tensor,
    input_tensor: torch.Tensor,
    scale: torch.Tensor,
) -> None:
    """output = int8(clamp(input * (1/scale), -128, 127)) with a per-tensor static scale."""
    _C.require_device(output_tensor, input_tensor, scale)
    if output_tensor.dtype != torch.int8:
        raise ValueError(f"output tensor must be int8, got {output_tensor.dtype}")
    x = input_tensor if input_tensor.dim() == 2 else input_tensor.reshape(-1, input_tensor.shape[-1])
    out = output_tensor if output_tensor.dim() == 2 else output_tensor.view(-1, output_tensor.shape[-1])
    if x.stride(-1) != 1:
        x = x.contiguous()
    if out.stride(-1) != 1:
        raise ValueError("output tensor must be contiguous in its last dimension")
    tokens, hidden = x.shape
    status = _C.load().conch_static_scaled_int8_quant(
        _C.ptr(out),
        _C.ptr(x),
        _C.ptr(scale),
        tokens,
        hidden,
        x.stride(0) if tokens > 1 else hidden,
        out.stride(0) if tokens > 1 else hidden,
        _C.dtype_id(x.dtype),
        _C.current_stream_handle(x.device),
    )
    _C.check(status, "static_scaled_int8_quant")
