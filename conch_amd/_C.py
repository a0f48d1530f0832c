"""ctypes binding of libconch_amd.so -- the C ABI declared in include/conch_amd.h.

PyTorch is used above this layer only for device memory and streams; nothing torch-typed crosses
the boundary (raw device pointers, sizes, strides, dtype enums, the hipStream_t as void*).
The library must exist: there is no fallback path (`ConchLibraryError` otherwise).
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CONCH_AMD_LIBRARY", _PKG / "libconch_amd.so"))

# conch_dtype_t (include/conch_amd.h; values 0..8 follow conch/kernels/quantization/gemm.py:20-31)
DT_FP32, DT_FP16, DT_BF16, DT_FP8_E4M3FN, DT_INT8, DT_UINT8, DT_INT32, DT_UINT32, DT_FP8_E5M2, DT_FP8_E4M3FNUZ = range(10)
ZP_NONE, ZP_SCALAR, ZP_TENSOR = range(3)
(TUNE_GEMM_VARIANT, TUNE_MIXED_TILE_NT, TUNE_SKINNY_NO_SPLITK, TUNE_SKINNY_MODE, TUNE_TILE_SCHEDULE, TUNE_PERSISTENT, TUNE_EPILOGUE, TUNE_DIAG,
 TUNE_MID_STAGES, TUNE_MIXED_SPLITK, TUNE_SKINNY_GATHER, TUNE_MIXED_KERNEL, TUNE_MID_SPLITK, TUNE_SKINNY_CHUNKS, TUNE_MIXED_STRIP_ROWS,
 TUNE_COUNT) = range(16)
(VARIANT_AUTO, VARIANT_GENERIC, VARIANT_MFMA_SIMPLE, VARIANT_MFMA_PINGPONG, VARIANT_MFMA_SKINNY,
 VARIANT_MFMA_PINGPONG2, VARIANT_MFMA_MID, VARIANT_MFMA_ASM1W) = range(8)

TORCH_TO_DT = {
    torch.float32: DT_FP32,
    torch.float16: DT_FP16,
    torch.bfloat16: DT_BF16,
    torch.float8_e4m3fn: DT_FP8_E4M3FN,
    torch.float8_e4m3fnuz: DT_FP8_E4M3FNUZ,
    torch.int8: DT_INT8,
    torch.uint8: DT_UINT8,
    torch.int32: DT_INT32,
    torch.float8_e5m2: DT_FP8_E5M2,
}
if hasattr(torch, "uint32"):
    TORCH_TO_DT[torch.uint32] = DT_UINT32


ABI_VERSION = 3  # CONCH_AMD_ABI_VERSION of include/conch_amd.h (checked by load_library and by tests/test_host_helpers.py)


class ConchLibraryError(RuntimeError):
    """libconch_amd.so is missing or unusable."""


class ConchError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


_I64 = c_int64
_SIGNATURES = {
    "conch_abi_version": (c_int, []),
    "conch_last_error": (c_char_p, []),
    "conch_set_tuning": (c_int, [c_int, c_int]),
    "conch_get_tuning": (c_int, [c_int]),
    "conch_set_timing_rotation": (c_int, [c_int, _I64]),
    "conch_device_count": (c_int, []),
    "conch_scaled_gemm_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "conch_mixed_precision_gemm_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "conch_bnb_gemm_4bit_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "conch_reserve_scratch": (c_int, [c_void_p, _I64]),
    "conch_reset_scratch": (c_int, [c_void_p]),
    "conch_static_scaled_int8_quant": (c_int, [c_void_p, c_void_p, c_void_p, _I64, _I64, _I64, _I64, c_int, c_void_p]),
    "conch_static_scaled_int8_quant_typed": (c_int, [c_void_p, c_void_p, c_void_p, _I64, _I64, _I64, _I64, c_int, c_int, c_void_p]),
    "conch_static_scaled_fp8_quant": (
        c_int,
        [c_void_p, c_void_p, c_void_p, _I64, _I64, _I64, _I64, c_int, c_int, c_void_p],
    ),
    "conch_dynamic_scaled_int8_quant": (c_int, [c_void_p, c_void_p, c_void_p, _I64, _I64, _I64, _I64, c_int, c_void_p]),
    "conch_dynamic_scaled_fp8_quant": (
        c_int,
        [c_void_p, c_void_p, c_void_p, _I64, _I64, _I64, _I64, c_int, c_int, c_void_p],
    ),
    "conch_scaled_gemm": (
        c_int,
        [c_void_p] * 6 + [_I64] * 11 + [c_int, c_int, c_void_p],
    ),
    "conch_scaled_gemm_silu_and_mul": (
        c_int,
        [c_void_p] * 6 + [_I64] * 11 + [c_int, c_int, c_void_p],
    ),
    "conch_scaled_gemm_gelu_tanh_and_mul": (
        c_int,
        [c_void_p] * 6 + [_I64] * 11 + [c_int, c_int, c_void_p],
    ),
    "conch_mixed_precision_gemm_gelu_tanh_and_mul": (
        c_int,
        [c_void_p] * 5 + [_I64] * 8 + [c_int] * 6 + [c_void_p],
    ),
    "conch_time_scaled_gemm_silu_and_mul": (
        c_int,
        [c_void_p] * 6 + [_I64] * 11 + [c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_float)],
    ),
    "conch_scaled_gemm_multi": (
        c_int,
        [c_void_p] * 6 + [_I64] * 11 + [c_int, c_int, ctypes.POINTER(c_void_p), c_int, c_void_p],
    ),
    "conch_static_quant_scaled_gemm": (
        c_int,
        [c_void_p] * 6 + [_I64] * 10 + [c_int, c_int, c_int, c_void_p],
    ),
    "conch_time_scaled_gemm": (
        c_int,
        [c_void_p] * 6 + [_I64] * 11 + [c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_float)],
    ),
    "conch_mixed_precision_gemm": (
        c_int,
        [c_void_p] * 5 + [_I64] * 8 + [c_int] * 6 + [c_void_p],
    ),
    "conch_prepacked_mixed_weights_bytes": (_I64, [_I64, _I64, c_int, c_int]),
    "conch_mixed_gemm_tile_nt": (c_int, [_I64, _I64, _I64, c_int, c_int]),
    "conch_prepack_mixed_weights": (c_int, [c_void_p, c_void_p, _I64, _I64, _I64, c_int, c_int, c_void_p]),
    "conch_unpack_mixed_weights": (c_int, [c_void_p, c_void_p, _I64, _I64, _I64, c_int, c_int, c_void_p]),
    "conch_mixed_precision_gemm_prepacked": (c_int, [c_void_p] * 5 + [_I64] * 7 + [c_int] * 7 + [c_void_p]),
    "conch_time_mixed_precision_gemm_prepacked": (
        c_int,
        [c_void_p] * 5 + [_I64] * 7 + [c_int] * 7 + [c_void_p, c_int, ctypes.POINTER(c_float)],
    ),
    "conch_bnb_quantize_blockwise": (c_int, [c_void_p] * 4 + [_I64, c_int, c_int, c_int, c_int, c_void_p]),
    "conch_bnb_dequantize_blockwise": (c_int, [c_void_p] * 4 + [_I64, c_int, c_int, c_int, c_int, c_void_p]),
    "conch_bnb_gemm_4bit": (c_int, [c_void_p] * 4 + [_I64] * 5 + [c_int] * 5 + [c_void_p]),
    "conch_mixed_precision_gemm_modes": (
        c_int,
        [c_void_p] * 6 + [_I64] * 8 + [c_int] * 8 + [c_void_p],
    ),
    "conch_mixed_precision_gemm_silu_and_mul": (
        c_int,
        [c_void_p] * 5 + [_I64] * 8 + [c_int] * 6 + [c_void_p],
    ),
    "conch_time_mixed_precision_gemm": (
        c_int,
        [c_void_p] * 5 + [_I64] * 8 + [c_int] * 6 + [c_void_p, c_int, ctypes.POINTER(c_float)],
    ),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: ctypes.CDLL | None = None


def load_library(path: Path) -> ctypes.CDLL:
    """dlopen `path` and declare every entry point of include/conch_amd.h on it (no caching: bench.py also opens the
    diagnostic twin of the library this way)."""
    if not Path(path).exists():
        msg = (
            f"{path} not found: build it with `python -m conch_amd._build` "
            "(conch_amd has no CPU / PyTorch fallback for its ops)"
        )
        raise ConchLibraryError(msg)
    try:
        lib = ctypes.CDLL(str(path))
    except OSError as exc:
        raise ConchLibraryError(f"cannot load {path}: {exc}") from exc
    # a stale binary (CONCH_AMD_LIBRARY override, or a build older than this file) must fail as a version mismatch, not as
    # an AttributeError on whichever symbol it happens to lack
    try:
        lib.conch_abi_version.restype = c_int
        lib.conch_abi_version.argtypes = []
        have = int(lib.conch_abi_version())
    except AttributeError:
        have = -1
    if have != ABI_VERSION:
        raise ConchLibraryError(f"{path} implements C-ABI version {have}, this package binds version {ABI_VERSION}: "
                                "rebuild it with `python -m conch_amd._build --force`")
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise ConchLibraryError(f"{path} does not export {name}: rebuild it with `python -m conch_amd._build --force`") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load() -> ctypes.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not there."""
    global _lib  # noqa: PLW0603
    if _lib is None:
        _lib = load_library(LIB_PATH)
    return _lib


def check(status: int, what: str) -> None:
    """Turn a conch_status_t into a Python exception (include/conch_amd.h error convention)."""
    if status == 0:
        return
    detail = (load().conch_last_error() or b"").decode("utf-8", "replace")
    if status == 1:
        raise ValueError(f"{what}: {detail}")
    if status == 2:
        raise NotImplementedError(f"{what}: {detail}")
    raise ConchError(f"{what} failed with status {status}: {detail}")


def dtype_id(dtype: torch.dtype) -> int:
    try:
        return TORCH_TO_DT[dtype]
    except KeyError:
        raise NotImplementedError(f"conch_amd: unsupported dtype {dtype}") from None


def require_device(*tensors: torch.Tensor | None) -> None:
    """The HIP path is the only path: refuse host tensors instead of silently computing on CPU."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            msg = "conch_amd ops need tensors on a ROCm device (cuda:N); there is no CPU fallback"
            raise RuntimeError(msg)


def on_device_of(*tensors: torch.Tensor | None):
    """Context that makes the tensors' device current for the C call (the library launches on the CURRENT device and keys
    its scratch by it); a no-op -- and no Python overhead beyond one comparison -- when it already is.  All tensors must
    live on ONE device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"conch_amd: tensors on different devices ({dev} and {t.device})")
    if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
        return _NULL_CONTEXT
    return torch.cuda.device(dev)


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CONTEXT = _NullContext()


def set_tuning(key: int, value: int) -> None:
    check(load().conch_set_tuning(key, value), "conch_set_tuning")


def reserve_scratch(nbytes: int, device: torch.device | None = None) -> None:
    """Pre-allocate the library's scratch for torch's current stream (see include/conch_amd.h, "Library scratch"): after
    this, calls whose workspace need is <= nbytes neither allocate nor synchronise (graph-capture safe)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with torch.cuda.device(device):
        check(load().conch_reserve_scratch(current_stream_handle(device), int(nbytes)), "conch_reserve_scratch")


def reset_scratch(device: torch.device | None = None) -> None:
    """Zero the split-K arrival counters of torch's current stream (recovery after a launch that did not complete)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with torch.cuda.device(device):
        check(load().conch_reset_scratch(current_stream_handle(device)), "conch_reset_scratch")


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device: torch.device) -> int:
    """The hipStream_t of torch's current stream on `device` (the raw-handle query when this torch build has it: the Stream
    object `torch.cuda.current_stream` builds costs ~4 us per op call, a fifth of a decode-size call's host time)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def set_gemm_variant(variant: int) -> None:
    check(load().conch_set_tuning(TUNE_GEMM_VARIANT, variant), "conch_set_tuning")


def get_gemm_variant() -> int:
    return load().conch_get_tuning(TUNE_GEMM_VARIANT)
