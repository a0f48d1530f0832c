"""Launcher of the blockwise NF4 / FP4 / 8-bit quantisation kernel (C-ABI seam).

Same name and arguments as `quantize_blockwise_launcher` of conch/kernels/quantization/bitsandbytes/quantize_blockwise.py:
no allocation, writes `out` and `absmax` in place, asynchronous on torch's current stream.
"""

from __future__ import annotations

import torch

from conch_amd import _C

QUANT_TYPE_ID = {"nf4": 0, "fp4": 1, "fp8": 2}


def quantize_blockwise_launcher(
    x: torch.Tensor,
    absmax: torch.Tensor,
    out: torch.Tensor,
    code: torch.Tensor | None,
    blocksize: int,
    input_size: int,
    quant_type: str,
) -> tuple[torch.Tensor, torch.Tensor]:
    """out = packed codes of x / absmax-per-block, absmax[b] = max |x| of block b.  Returns (out, absmax)."""
    _C.require_device(x, absmax, out, code)
    if quant_type not in QUANT_TYPE_ID:
        raise NotImplementedError(f"Unsupported quant_type: {quant_type}")
    if out.dtype != torch.uint8:
        out = out.view(torch.uint8)  # quant_storage other than uint8: the same bytes
    if not (x.is_contiguous() and out.is_contiguous() and absmax.is_contiguous()):
        raise ValueError("quantize_blockwise: x, out and absmax must be contiguous")
    if quant_type == "fp8":
        if code is None:
            raise ValueError("quant_type 'fp8' needs its 256-entry code book")
        code = code.to(device=x.device, dtype=torch.float32).contiguous()
        if code.numel() != 256:
            raise ValueError(f"the code book must have 256 entries, got {code.numel()}")
    else:
        code = None
    nblocks = -(-input_size // blocksize)
    need = input_size if quant_type == "fp8" else (input_size + 1) // 2
    if x.numel() < input_size or absmax.numel() < nblocks or out.numel() < need:
        raise ValueError("quantize_blockwise: a tensor is smaller than input_size requires")
    with _C.on_device_of(x, absmax, out, code):
        status = _C.load().conch_bnb_quantize_blockwise(
            _C.ptr(out), _C.ptr(absmax), _C.ptr(x), _C.ptr(code), input_size, blocksize, QUANT_TYPE_ID[quant_type],
            _C.dtype_id(x.dtype), _C.dtype_id(absmax.dtype), _C.current_stream_handle(x.device),
        )
    _C.check(status, "quantize_blockwise")
    return out, absmax
