"""CPU restatement of the reference's bitsandbytes-style blockwise quantisation (TEST INFRASTRUCTURE).

Follows conch/reference/quantization/bitsandbytes/{quantize,dequantize}_blockwise.py -- the reference's PyTorch-only
implementation, which walks the tensors element by element -- vectorised with numpy; every comparison and product is made
in the dtype the reference makes it in.  Parity status: pinned by tests/golden/bnb_blockwise.npz (outputs of the real
reference, tests/golden/make_golden_r2.py gen_bnb), bit for bit.
"""

from __future__ import annotations

import numpy as np
import torch

# dequantisation tables: reference/quantization/bitsandbytes/dequantize_blockwise.py:12-45 (fp4), :48-83 (nf4); code c -> value
FP4_VALUES = np.array([0.0, 0.0052083333, 0.6666666, 1.0, 0.333333, 0.5, 0.166666, 0.25,
                       -0.0, -0.0052083333, -0.666666, -1.0, -0.333333, -0.5, -0.166666, -0.25], dtype=np.float64)
NF4_VALUES = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                       -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                       0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                       0.7229568362236023, 1.0], dtype=np.float64)
# quantisation thresholds: reference/quantization/bitsandbytes/quantize_blockwise.py:12-35 (fp4: on |x|), :38-74 (nf4).
# The decision trees are binary searches with strict `>` over these ascending lists: code = f(number of thresholds below x).
NF4_THRESHOLDS = np.array([-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544,
                           -0.23460740596055984, -0.13791173323988914, -0.045525018125772476, 0.03979014977812767,
                           0.1202552504837513, 0.2035212516784668, 0.2920137718319893, 0.3893125355243683,
                           0.5016634166240692, 0.6427869200706482, 0.8614784181118011], dtype=np.float64)
FP4_THRESHOLDS = np.array([0.00260417, 0.0859375, 0.208333334, 0.29166667, 0.4166667, 0.5833334, 0.83333334], dtype=np.float64)
FP4_CODE_BY_RANK = np.array([0, 1, 6, 7, 4, 5, 2, 3], dtype=np.uint8)


def create_dynamic_map_ref(signed: bool = True, max_exponent_bits: int = 7, total_bits: int = 8) -> torch.Tensor:
    """The 256-entry "dynamic" 8-bit map (conch/ops/quantization/bitsandbytes/functional.py:30-84): for each of the
    `max_exponent_bits` decades, the midpoints of an even grid on [0.1, 1] scaled by the decade, both signs, plus 0 and 1."""
    values: list[float] = []
    non_sign = total_bits - 1
    for i in range(max_exponent_bits):
        items = 2 ** (i + non_sign - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign - max_exponent_bits + 1) + 1
        edges = torch.linspace(0.1, 1, int(items), dtype=torch.float32)
        mids = (edges[:-1] + edges[1:]) / 2.0
        scale = 10 ** (-(max_exponent_bits - 1) + i)
        values += (scale * mids).tolist()
        if signed:
            values += (-scale * mids).tolist()
    extra = 2 ** (non_sign - max_exponent_bits) - 1
    if extra > 0:
        edges = torch.linspace(0.1, 1, extra + 1, dtype=torch.float32)
        mids = (edges[:-1] + edges[1:]) / 2.0
        values += (max_exponent_bits * mids).tolist()
        if signed:
            values += (-max_exponent_bits * mids).tolist()
    values += [0, 1.0]
    assert len(values) == 2**total_bits
    values += [0] * (256 - len(values))
    values.sort()
    return torch.tensor(values, dtype=torch.float32)


def _codes_4bit(v: np.ndarray, quant_type: str) -> np.ndarray:
    """v: float32 values already divided by the block's absmax.  The reference compares a float32 tensor with Python float
    constants, i.e. in float32 (the constant is rounded to float32)."""
    if quant_type == "nf4":
        return (v[:, None] > NF4_THRESHOLDS.astype(np.float32)[None, :]).sum(axis=1).astype(np.uint8)
    rank = (np.abs(v)[:, None] > FP4_THRESHOLDS.astype(np.float32)[None, :]).sum(axis=1)
    return (FP4_CODE_BY_RANK[rank] + np.where(v < 0, 8, 0)).astype(np.uint8)


def _codes_8bit(v: np.ndarray, code: np.ndarray) -> np.ndarray:
    """quantize_blockwise.py:77-119: seven bisection steps from pivot 127, then the nearer of the pivot and its neighbour."""
    n = v.shape[0]
    pivot = np.full(n, 127, dtype=np.int64)
    upper_pivot = np.full(n, 255, dtype=np.int64)
    lower_pivot = np.zeros(n, dtype=np.int64)
    lower = np.full(n, -1.0, dtype=np.float32)
    upper = np.full(n, 1.0, dtype=np.float32)
    val = code[pivot]
    for step in (64, 32, 16, 8, 4, 2, 1):
        gt = v > val
        lower_pivot = np.where(gt, pivot, lower_pivot)
        lower = np.where(gt, val, lower)
        upper_pivot = np.where(gt, upper_pivot, pivot)
        upper = np.where(gt, upper, val)
        pivot = np.where(gt, pivot + step, pivot - step)
        val = code[pivot]
    upper = np.where(upper_pivot == 255, code[upper_pivot], upper)
    lower = np.where(lower_pivot == 0, code[lower_pivot], lower)
    gt = v > val
    mid_up = ((upper + val) * np.float32(0.5)).astype(np.float32)
    mid_lo = ((lower + val) * np.float32(0.5)).astype(np.float32)
    res = np.where(gt, np.where(v > mid_up, upper_pivot, pivot), np.where(v < mid_lo, lower_pivot, pivot))
    return res.astype(np.uint8)


def quantize_blockwise_ref(x: torch.Tensor, blocksize: int, quant_type: str, code: torch.Tensor | None = None,
                           absmax_dtype: torch.dtype | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """quantize_blockwise.py:132-181.  Returns (packed uint8, absmax in `absmax_dtype` (default x.dtype))."""
    flat = x.reshape(-1)
    n = flat.numel()
    nblocks = -(-n // blocksize)
    absmax_dtype = absmax_dtype or x.dtype
    four_bit = quant_type in ("nf4", "fp4")
    out = torch.zeros((n + 1) // 2 if four_bit else n, dtype=torch.uint8)
    absmax = torch.empty(nblocks, dtype=absmax_dtype)
    code_np = None if code is None else code.float().numpy()
    for b in range(nblocks):
        block = flat[b * blocksize : min((b + 1) * blocksize, n)]
        local = block.abs().max()                    # in x's dtype (:150)
        absmax[b] = local                            # stored in the absmax tensor's dtype (:153)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.float32(1.0) / np.float32(local.float().item())   # :156
            v = (block.float().numpy() * inv).astype(np.float32)       # :169-173
        if four_bit:
            pairs = block.numel() // 2               # an odd last element is dropped (:162)
            c = _codes_4bit(v[: 2 * pairs], quant_type)
            out[b * (blocksize // 2) : b * (blocksize // 2) + pairs] = torch.from_numpy((c[0::2] << 4) | c[1::2])
        else:
            out[b * blocksize : b * blocksize + block.numel()] = torch.from_numpy(_codes_8bit(v, code_np))
    return out, absmax


def dequantize_blockwise_ref(xq: torch.Tensor, absmax: torch.Tensor, output_size: int, blocksize: int, quant_type: str,
                             out_dtype: torch.dtype, code: torch.Tensor | None = None) -> torch.Tensor:
    """dequantize_blockwise.py:103-142: value(code) * absmax[block] -- a Python float times a 0-dim tensor, i.e. the float32
    product rounded to the ABSMAX tensor's dtype -- then stored into the output's dtype."""
    q = xq.reshape(-1).numpy()
    four_bit = quant_type in ("nf4", "fp4")
    if four_bit:
        table = (NF4_VALUES if quant_type == "nf4" else FP4_VALUES).astype(np.float32)
        codes = np.stack([q >> 4, q & 0x0F], axis=1).reshape(-1)[: output_size - output_size % 2]
    else:
        table = code.float().numpy()
        codes = q[:output_size]
    vals = torch.from_numpy(table[codes.astype(np.int64)])
    block_of = torch.arange(vals.numel()) // blocksize
    prod = (vals * absmax.float()[block_of]).to(absmax.dtype)   # float32 product, rounded to absmax's dtype
    out = torch.zeros(output_size, dtype=out_dtype)
    out[: prod.numel()] = prod.to(out_dtype)
    return out
