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
    # the kernel reads `scale` as ONE float: any other dtype would be silently misread (the reference's Triton kernel
    # loads whatever dtype it is given; its ops/tests only ever pass float32)
    if scale.dtype != torch.float32 or scale.numel() != 1:
        raise ValueError(f"scale must be a one-element float32 tensor (got {scale.dtype}, {scale.numel()} elements)")
    tokens, hidden = x.shape
    # a 0-dim scale does not take part in torch's type promotion: the reference oracle's `x * scale.reciprocal()`
    # (conch/reference/quantization/int8.py:16) is then rounded to x's dtype before the clamp; any other one-element
    # shape -- the (1,) of the reference's tests and benchmarks -- makes the product fp32
    x_dt = _C.dtype_id(x.dtype)
    product_dt = x_dt if scale.dim() == 0 and x.dtype in (torch.float16, torch.bfloat16) else _C.DT_FP32
    with _C.on_device_of(out, x, scale):
        status = _C.load().conch_static_scaled_int8_quant_typed(
            _C.ptr(out),
            _C.ptr(x),
            _C.ptr(scale),
            tokens,
            hidden,
            x.stride(0) if tokens > 1 else hidden,
            out.stride(0) if tokens > 1 else hidden,
            x_dt,
            product_dt,
            _C.current_stream_handle(x.device),
        )
    _C.check(status, "static_scaled_int8_quant")


def dynamic_scaled_int8_quant_launcher(
    output_tensor: torch.Tensor,
    scale_out: torch.Tensor,
    input_tensor: torch.Tensor,
) -> None:
    """Per-token dynamic quantisation (the reference raises NotImplementedError here): writes
    scale_out[t] = absmax(input[t]) / QMAX and output[t] = static_quant(input[t], scale_out[t]).  `scale_out`: contiguous
    float32 with one element per token."""
    _C.require_device(output_tensor, scale_out, input_tensor)
    if output_tensor.dtype != torch.int8:
        raise ValueError(f"output tensor must be int8, got {output_tensor.dtype}")
    x = input_tensor if input_tensor.dim() == 2 else input_tensor.reshape(-1, input_tensor.shape[-1])
    out = output_tensor if output_tensor.dim() == 2 else output_tensor.view(-1, output_tensor.shape[-1])
    if x.stride(-1) != 1:
        x = x.contiguous()
    if out.stride(-1) != 1:
        raise ValueError("output tensor must be contiguous in its last dimension")
    tokens, hidden = x.shape
    if scale_out.dtype != torch.float32 or scale_out.numel() != tokens or not scale_out.is_contiguous():
        raise ValueError(f"scale_out must be a contiguous float32 tensor with {tokens} elements")
    with _C.on_device_of(out, scale_out, x):
        status = _C.load().conch_dynamic_scaled_int8_quant(
            _C.ptr(out),
            _C.ptr(scale_out),
            _C.ptr(x),
            tokens,
            hidden,
            x.stride(0) if tokens > 1 else hidden,
            out.stride(0) if tokens > 1 else hidden,
            _C.dtype_id(x.dtype),
            _C.current_stream_handle(x.device),
        )
    _C.check(status, "dynamic_scaled_int8_quant")
