"""Weight pre-pack for the mixed-precision GEMM.

The reference's benchmark pre-packs the weights for its comparator kernel (benchmarks/mixed_precision_gemm_benchmark.py:59-75,
machete_prepack_B); this is the same step for the gfx950 tile kernel: the [K * bits / 32, N] tensor `pack_rows` produced is
rewritten ONCE into the order the K loop consumes it (one contiguous record per tile column and K step, nibbles re-ordered so
that the in-register dequantisation needs no permute).  The product on a pre-packed weight is bit-identical to
`mixed_precision_gemm` on the original.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch

from conch_amd import _C


@dataclass
class PrepackedWeights:
    """A pre-packed weight: `image` (int32, flat) laid out for tiles of 64 * tile_nt output columns of a (K, N) weight."""

    image: torch.Tensor
    k: int
    n: int
    weight_size_bits: int
    tile_nt: int


def _zp_mode(w_zp: torch.Tensor | None) -> int:
    return _C.ZP_NONE if w_zp is None else _C.ZP_SCALAR if w_zp.numel() == 1 else _C.ZP_TENSOR


def prepack_mixed_weights(w_q_packed: torch.Tensor, weight_size_bits: int, *, m_hint: int = 1024, per_group_zero_points: bool = False,
                          tile_nt: int | None = None) -> PrepackedWeights:
    """Pre-pack `w_q_packed` ((K * bits / 32, N) int32, as pack_rows makes it).  The tile width is the one the dispatcher
    picks for `m_hint` activation rows (pass `tile_nt` = 2, 3 or 4 to force it)."""
    _C.require_device(w_q_packed)
    if weight_size_bits not in (4, 8):
        raise NotImplementedError("prepack_mixed_weights: 4- and 8-bit weights (the widths the tile kernel takes)")
    if w_q_packed.dtype not in (torch.int32, getattr(torch, "uint32", torch.int32)) or w_q_packed.dim() != 2:
        raise ValueError("w_q_packed must be a 2-D int32 tensor")
    if w_q_packed.stride(1) != 1:
        w_q_packed = w_q_packed.contiguous()
    k, n = w_q_packed.shape[0] * (32 // weight_size_bits), w_q_packed.shape[1]
    lib = _C.load()
    if tile_nt is None:
        tile_nt = lib.conch_mixed_gemm_tile_nt(m_hint, n, k, weight_size_bits, _C.ZP_TENSOR if per_group_zero_points else _C.ZP_NONE)
    nbytes = lib.conch_prepacked_mixed_weights_bytes(k, n, weight_size_bits, tile_nt)
    if nbytes < 0:
        _C.check(1, "prepack_mixed_weights")
    image = torch.empty(nbytes // 4, dtype=torch.int32, device=w_q_packed.device)
    with _C.on_device_of(image, w_q_packed):
        status = lib.conch_prepack_mixed_weights(_C.ptr(image), _C.ptr(w_q_packed), k, n, w_q_packed.stride(0), weight_size_bits, tile_nt,
                                                 _C.current_stream_handle(w_q_packed.device))
    _C.check(status, "prepack_mixed_weights")
    return PrepackedWeights(image, k, n, weight_size_bits, tile_nt)


def unpack_mixed_weights(pre: PrepackedWeights) -> torch.Tensor:
    """The inverse: the (K * bits / 32, N) int32 tensor the image was made from."""
    out = torch.empty((pre.k * pre.weight_size_bits // 32, pre.n), dtype=torch.int32, device=pre.image.device)
    with _C.on_device_of(out, pre.image):
        status = _C.load().conch_unpack_mixed_weights(_C.ptr(out), _C.ptr(pre.image), pre.k, pre.n, out.stride(0), pre.weight_size_bits,
                                                      pre.tile_nt, _C.current_stream_handle(out.device))
    _C.check(status, "unpack_mixed_weights")
    return out


def _prepacked_args(out, x, pre: PrepackedWeights, w_s, w_zp, weight_bias: int, group_size: int):
    _C.require_device(out, x, pre.image, w_s, w_zp)
    if x.dim() != 2 or x.shape[1] != pre.k:
        raise ValueError(f"x {tuple(x.shape)} does not multiply a ({pre.k}, {pre.n}) weight")
    if w_s.dtype != x.dtype or tuple(w_s.shape) != (pre.k // group_size, pre.n):
        raise ValueError("w_s must be a (K / group_size, N) tensor of the activation dtype")
    if x.stride(1) != 1:
        x = x.contiguous()
    if w_s.stride(-1) != 1:
        w_s = w_s.contiguous()
    zeros, zstride = None, 0
    if w_zp is not None:
        zeros = w_zp.to(torch.int32)
        zeros = zeros.reshape(-1) if zeros.numel() == 1 else (zeros if zeros.stride(-1) == 1 else zeros.contiguous())
        zstride = 0 if zeros.numel() == 1 else zeros.stride(0)
    return (_C.ptr(out), _C.ptr(x), _C.ptr(pre.image), _C.ptr(w_s), _C.ptr(zeros), x.shape[0], pre.n, pre.k, x.stride(0), w_s.stride(0),
            zstride, out.stride(0), pre.weight_size_bits, weight_bias, group_size, _zp_mode(w_zp), pre.tile_nt, _C.dtype_id(x.dtype),
            _C.dtype_id(out.dtype), _C.current_stream_handle(x.device)), (x, w_s, zeros)


def mixed_precision_gemm_prepacked(x: torch.Tensor, pre: PrepackedWeights, w_s: torch.Tensor, w_zp: torch.Tensor | None,
                                   weight_bias: int, group_size: int) -> torch.Tensor:
    """`mixed_precision_gemm(x, w_q_packed, w_s, w_zp, bits, weight_bias, group_size)` on a pre-packed weight: same result,
    bit for bit."""
    out = x.new_empty((x.shape[0], pre.n))
    args, keep = _prepacked_args(out, x, pre, w_s, w_zp, weight_bias, group_size)
    with _C.on_device_of(out, *keep):
        status = _C.load().conch_mixed_precision_gemm_prepacked(*args)
    _C.check(status, "mixed_precision_gemm_prepacked")
    return out
