"""bitsandbytes-compatible blockwise NF4 / FP4 / 8-bit quantisation, public API.

Names, signatures and return types of conch/ops/quantization/bitsandbytes/functional.py (QuantState :87-104,
quantize_blockwise :126-166, quantize_4bit :169-209, dequantize_blockwise :212-252, dequantize_4bit :255-269); the Triton
launchers behind them are replaced by the HIP kernels of csrc/bnb.hip.  `matmul_4bit` is the "-> GEMM" half of the row: the
product of activations with a 4-bit weight, dequantised into library scratch and multiplied on the 16-bit MFMA tile kernel.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Final, Optional

import torch

from conch_amd import _C
from conch_amd.kernels.quantization.bitsandbytes.dequantize_blockwise import dequantize_blockwise_launcher
from conch_amd.kernels.quantization.bitsandbytes.quantize_blockwise import QUANT_TYPE_ID, quantize_blockwise_launcher

SUPPORTED_QUANT_TYPES: Final = ["nf4", "fp4", "fp8"]
SUPPORTED_BLOCKSIZES: Final = [4096, 2048, 1024, 512, 256, 128, 64]

_BYTES_PER_ELEMENT: Final = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2, torch.uint8: 1, torch.int8: 1}
_NAME_TO_QMAP: dict[str, torch.Tensor] = {}


def _create_dynamic_map(signed: bool = True, max_exponent_bits: int = 7, total_bits: int = 8) -> torch.Tensor:
    """The "dynamic" 8-bit data type of bitsandbytes (8-Bit Approximations for Parallelism in Deep Learning,
    arXiv:1511.04561): a decade exponent with a linear fraction whose resolution shrinks as the exponent grows.

    Per decade i the fraction values are the midpoints of an even grid over [0.1, 1]; both signs; 0 and 1 are added and the
    table is sorted (reference: functional.py:30-84; pinned against it by tests/golden/bnb_blockwise.npz `dynamic_map`).
    """
    non_sign = total_bits - 1
    table: list[float] = []

    def midpoints(count: int) -> torch.Tensor:
        grid = torch.linspace(0.1, 1, count, dtype=torch.float32)
        return (grid[:-1] + grid[1:]) / 2.0

    for i in range(max_exponent_bits):
        exponent_width = i + non_sign - max_exponent_bits
        count = 2**exponent_width + 1 if signed else 2 ** (exponent_width + 1) + 1
        scaled = (10 ** (i - (max_exponent_bits - 1))) * midpoints(int(count))
        table += scaled.tolist()
        if signed:
            table += (-scaled).tolist()
    leftover = 2 ** (non_sign - max_exponent_bits) - 1
    if leftover > 0:
        scaled = max_exponent_bits * midpoints(leftover + 1)
        table += scaled.tolist()
        if signed:
            table += (-scaled).tolist()
    table += [0, 1.0]
    if len(table) != 2**total_bits:
        raise AssertionError(f"dynamic map has {len(table)} entries, expected {2**total_bits}")
    table += [0] * (256 - len(table))
    return torch.tensor(sorted(table), dtype=torch.float32)


@dataclass
class QuantState:
    """Quantisation state, field for field bitsandbytes' (and the reference's, functional.py:87-104)."""

    absmax: torch.Tensor
    shape: torch.Size
    dtype: torch.dtype
    blocksize: int
    quant_type: str
    code: torch.Tensor | None = None
    offset: float | None = None
    state2: Optional["QuantState"] = None

    @property
    def nested(self) -> bool:
        return self.state2 is not None


def get_absmax_shape(input_size: int, blocksize: int) -> torch.Size:
    return torch.Size((-(-input_size // blocksize),))


def get_quantized_output_shape(input_size: int, quant_type: str, quant_storage: torch.dtype = torch.uint8) -> torch.Size:
    if quant_type == "fp8":
        return torch.Size((input_size,))
    per_storage_element = _BYTES_PER_ELEMENT[quant_storage] * 2
    return torch.Size(((input_size + 1) // per_storage_element, 1))


def _check(quant_type: str, blocksize: int) -> None:
    if quant_type not in SUPPORTED_QUANT_TYPES:
        raise NotImplementedError(f"Unsupported quant_type: {quant_type} ({SUPPORTED_QUANT_TYPES = })")
    if blocksize not in SUPPORTED_BLOCKSIZES:
        raise NotImplementedError(f"Unsupported blocksize: {blocksize} ({SUPPORTED_BLOCKSIZES = })")


def quantize_blockwise(x: torch.Tensor, absmax: torch.Tensor | None = None, out: torch.Tensor | None = None,
                       code: torch.Tensor | None = None, blocksize: int = 64, quant_type: str = "fp4",
                       quant_storage: torch.dtype = torch.uint8) -> tuple[torch.Tensor, QuantState]:
    """Quantise `x` in blocks of `blocksize` elements of its flattened form."""
    _check(quant_type, blocksize)
    n = x.numel()
    want_absmax = get_absmax_shape(n, blocksize)
    if absmax is None:
        absmax = torch.empty(want_absmax, device=x.device, dtype=torch.float32)  # every block is written by the kernel
    assert absmax.shape == want_absmax  # noqa: S101
    want_out = get_quantized_output_shape(n, quant_type, quant_storage)
    if out is None:
        out = torch.zeros(want_out, device=x.device, dtype=quant_storage)  # an odd tail leaves its last byte untouched
    assert out.shape == want_out  # noqa: S101
    quantize_blockwise_launcher(x=x.contiguous().view(-1), absmax=absmax, out=out, code=code, blocksize=blocksize, input_size=n,
                                quant_type=quant_type)
    return out, QuantState(absmax=absmax, shape=x.shape, dtype=x.dtype, blocksize=blocksize, quant_type=quant_type, code=code)


def quantize_4bit(x: torch.Tensor, absmax: torch.Tensor | None = None, out: torch.Tensor | None = None, blocksize: int = 64,
                  compress_statistics: bool = False, quant_type: str = "fp4",
                  quant_storage: torch.dtype = torch.uint8) -> tuple[torch.Tensor, QuantState]:
    """Quantise to packed 4-bit values; `compress_statistics` quantises the absmax vector itself (8-bit dynamic map, blocks
    of 256, around its mean) -- the "double quantisation" of QLoRA."""
    out, state = quantize_blockwise(x=x, absmax=absmax, out=out, code=None, blocksize=blocksize, quant_type=quant_type,
                                    quant_storage=quant_storage)
    if compress_statistics:
        stats = state.absmax
        offset = stats.mean()
        stats -= offset
        if "dynamic" not in _NAME_TO_QMAP:
            _NAME_TO_QMAP["dynamic"] = _create_dynamic_map()
        code = _NAME_TO_QMAP["dynamic"].to(x.device)
        q_stats, state2 = quantize_blockwise(x=stats, absmax=None, out=None, code=code, blocksize=256, quant_type="fp8")
        state = QuantState(absmax=q_stats, shape=state.shape, dtype=state.dtype, blocksize=blocksize, code=None,
                           quant_type=quant_type, offset=offset.item(), state2=state2)
    return out, state


def dequantize_blockwise(x: torch.Tensor, quant_state: QuantState | None = None, absmax: torch.Tensor | None = None,
                         code: torch.Tensor | None = None, out: torch.Tensor | None = None, blocksize: int = 64,
                         quant_type: str = "fp4") -> torch.Tensor:
    """Inverse of quantize_blockwise."""
    _check(quant_type, blocksize)
    if quant_state is None:
        if absmax is None:
            raise ValueError("Must pass either quant_state or absmax!")
        if out is None:
            raise ValueError("Must pass either quant_state or out!")
        if code is None and quant_type == "fp8":
            raise ValueError("Must pass either quant_state or code!")
    else:
        absmax = quant_state.absmax if absmax is None else absmax
        code = quant_state.code if code is None else code
        if out is None:
            out = torch.empty(quant_state.shape, dtype=quant_state.dtype, device=x.device)
    n = out.numel()
    assert absmax.shape == (-(-n // blocksize),)  # noqa: S101
    dequantize_blockwise_launcher(x=x, absmax=absmax, out=out.view(-1), code=code, blocksize=blocksize, output_size=n,
                                  quant_type=quant_type)
    return out


def _resolve_absmax(quant_state: QuantState) -> torch.Tensor:
    """The fp32 absmax vector of a (possibly nested) state (functional.py:258-266)."""
    if not quant_state.nested:
        return quant_state.absmax
    inner = quant_state.state2
    absmax = dequantize_blockwise(x=quant_state.absmax, quant_state=inner, blocksize=inner.blocksize, quant_type=inner.quant_type)
    absmax += quant_state.offset
    return absmax if absmax.dtype == torch.float32 else absmax.float()


def dequantize_4bit(x: torch.Tensor, quant_state: QuantState | None = None, absmax: torch.Tensor | None = None,
                    out: torch.Tensor | None = None, blocksize: int = 64, quant_type: str = "fp4") -> torch.Tensor:
    """Inverse of quantize_4bit (nested statistics are expanded first)."""
    if quant_state is not None and quant_state.nested:
        absmax = _resolve_absmax(quant_state)
    return dequantize_blockwise(x=x, quant_state=quant_state, absmax=absmax, code=None, out=out, blocksize=blocksize,
                                quant_type=quant_type)


def matmul_4bit(x: torch.Tensor, w_packed: torch.Tensor, quant_state: QuantState, output_dtype: torch.dtype | None = None) -> torch.Tensor:
    """x @ dequantize_4bit(w_packed, quant_state).T for a 4-bit weight of shape (N, K) = quant_state.shape.

    The reference has no fused op for this (its users dequantise, then call torch.matmul): here the weight is dequantised into
    library scratch in x's dtype and multiplied on the 16-bit MFMA tile kernel with fp32 accumulation -- the result equals
    `x @ dequantize_4bit(...).T` up to accumulation order."""
    if quant_state.quant_type not in ("nf4", "fp4"):
        raise NotImplementedError("matmul_4bit: nf4 / fp4 weights only")
    if len(quant_state.shape) != 2 or x.dim() != 2 or x.shape[1] != quant_state.shape[1]:
        raise ValueError(f"x {tuple(x.shape)} does not multiply a weight of shape {tuple(quant_state.shape)}")
    n, k = quant_state.shape
    m = x.shape[0]
    # the C entry takes no buffer sizes and the dequantise-first kernels read through raw pointers: a state that does
    # not describe `w_packed` must fail HERE, not read device memory out of bounds
    if quant_state.blocksize not in SUPPORTED_BLOCKSIZES:
        raise NotImplementedError(f"Unsupported blocksize: {quant_state.blocksize} ({SUPPORTED_BLOCKSIZES = })")
    want_bytes = (n * k + 1) // 2
    have_bytes = w_packed.numel() * w_packed.element_size()
    if have_bytes != want_bytes:
        raise ValueError(f"w_packed holds {have_bytes} bytes; a 4-bit weight of shape {(n, k)} needs {want_bytes}")
    want_blocks = -(-(n * k) // quant_state.blocksize)
    if quant_state.absmax.numel() != want_blocks:  # nested statistics: one 8-bit code per block, the same count
        raise ValueError(
            f"absmax has {quant_state.absmax.numel()} entries; {n * k} elements in blocks of {quant_state.blocksize} need {want_blocks}"
        )
    _C.require_device(x, w_packed, quant_state.absmax)
    absmax = _resolve_absmax(quant_state).contiguous()
    if x.stride(1) != 1:
        x = x.contiguous()
    out = torch.empty((m, n), dtype=output_dtype or x.dtype, device=x.device)
    wq = w_packed.view(torch.uint8).contiguous()
    with _C.on_device_of(x, wq, absmax, out):
        status = _C.load().conch_bnb_gemm_4bit(
            _C.ptr(out), _C.ptr(x), _C.ptr(wq), _C.ptr(absmax), m, n, k, x.stride(0), out.stride(0), quant_state.blocksize,
            QUANT_TYPE_ID[quant_state.quant_type], _C.dtype_id(absmax.dtype), _C.dtype_id(x.dtype), _C.dtype_id(out.dtype),
            _C.current_stream_handle(x.device),
        )
    _C.check(status, "matmul_4bit")
    return out
