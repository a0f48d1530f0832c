"""Launcher of the blockwise NF4 / FP4 / 8-bit dequantisation kernel (C-ABI seam).

Same name and arguments as `dequantize_blockwise_launcher` of
conch/kernels/quantization/bitsandbytes/dequantize_blockwise.py:149-204.
"""

from __future__ import annotations

import torch

from conch_amd import _C
from conch_amd.kernels.quantization.bitsandbytes.quantize_blockwise import QUANT_TYPE_ID


def dequantize_blockwise_launcher(
    x: torch.Tensor,
    absmax: torch.Tensor,
    out: torch.Tensor,
    code: torch.Tensor | None,
    blocksize: int,
    output_size: int,
    quant_type: str,
) -> torch.Tensor:
    """out[e] = value(code of element e) * absmax[e // blocksize].  Returns `out`."""
    _C.require_device(x, absmax, out, code)
    if quant_type not in QUANT_TYPE_ID:
        raise NotImplementedError(f"Unsupported quant_type: {quant_type}")
    if blocksize & (blocksize - 1):
        raise ValueError("blocksize must be a power of two")
    if quant_type in ("fp4", "nf4") and output_size % 2:
        raise ValueError("4-bit inputs unpack to an even number of elements")
    xq = x.view(torch.uint8)  # storage dtypes other than uint8 hold the same bytes (dequantize_blockwise.py:191-193)
    if not (xq.is_contiguous() and out.is_contiguous() and absmax.is_contiguous()):
        raise ValueError("dequantize_blockwise: x, out and absmax must be contiguous")
    if quant_type == "fp8":
        if code is None:
            raise ValueError("quant_type 'fp8' needs its 256-entry code book")
        code = code.to(device=out.device, dtype=torch.float32).contiguous()
    else:
        code = None
    nblocks = -(-output_size // blocksize)
    need = output_size if quant_type == "fp8" else output_size // 2
    if out.numel() < output_size or absmax.numel() < nblocks or xq.numel() < need:
        raise ValueError("dequantize_blockwise: a tensor is smaller than output_size requires")
    with _C.on_device_of(xq, absmax, out, code):
        status = _C.load().conch_bnb_dequantize_blockwise(
            _C.ptr(out), _C.ptr(xq), _C.ptr(absmax), _C.ptr(code), output_size, blocksize, QUANT_TYPE_ID[quant_type],
            _C.dtype_id(out.dtype), _C.dtype_id(absmax.dtype), _C.current_stream_handle(out.device),
        )
    _C.check(status, "dequantize_blockwise")
    return out
