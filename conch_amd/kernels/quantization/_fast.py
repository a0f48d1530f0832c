"""Short host paths of the hot ops for the plain case (decode-size calls are host-bound: DESIGN.md 9.4).

The general launchers (gemm.py, int8.py, fp8.py) validate every legal input form, build the metadata dataclass the reference's
API hands around, normalise layouts and dispatch by name; that is 13-14 us of Python per call against 7-9 us of GPU time at decode
sizes.  The functions here take the SAME arguments as the public ops, test the few properties the C ABI relies on, allocate the
result and call one cached ctypes entry point -- or return None, and the op falls through to the general path (which raises the
proper error or handles the unusual form).  No behaviour of its own: same C entry point, same arguments, same result.
"""

from __future__ import annotations

import torch

from conch_amd import _C

_F32 = torch.float32
_fns: dict[str, object] = {}
_raw_stream = _C._raw_stream
_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)  # the bare query (no lazy-init wrapper)
_empty = torch.empty
_devices: dict[int, torch.device] = {}


def _device(index: int) -> torch.device:
    d = _devices.get(index)
    if d is None:
        d = _devices[index] = torch.device("cuda", index)
    return d


def _load_host_shim():
    """The C++ form of the functions below (conch_amd/csrc_host/host_shim.cpp, built by conch_amd._build.build_host_shim): the same
    checks, allocation and C-ABI call without the interpreter in between (~3 us less per call).  Absent or unloadable (another torch
    build): the Python forms serve."""
    try:
        from conch_amd import _conch_host  # type: ignore[attr-defined]
    except ImportError as exc:
        if (_C._PKG / "_conch_host.so").exists():  # built, but not loadable (another torch build, a stale ABI): say so once
            import warnings

            warnings.warn(f"conch_amd: the C++ host shim is present but failed to import ({exc}); the Python host paths serve "
                          "(rebuild it with `python -m conch_amd._build`)", RuntimeWarning, stacklevel=2)
        return None
    try:
        _C.load()
        _conch_host.bind_library(str(_C.LIB_PATH))
    except Exception as exc:  # noqa: BLE001 -- an optional accelerator of the HOST path; the ops themselves do not depend on it
        import warnings

        warnings.warn(f"conch_amd: the C++ host shim could not bind libconch_amd.so ({exc}); the Python host paths serve",
                      RuntimeWarning, stacklevel=2)
        return None
    return _conch_host


def _fn(name: str):
    f = _fns.get(name)
    if f is None:
        f = _fns[name] = getattr(_C.load(), name)
    return f


def scaled_gemm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor, output_dtype: torch.dtype,
                bias: torch.Tensor | None) -> torch.Tensor | None:
    """conch_scaled_gemm for 2-D a, b of one 8-bit dtype on the current device, contiguous float32 scales of 1 or M / N elements and
    a contiguous bias in the output dtype; None = not that case."""
    dev = a.get_device()  # -1 for a host tensor
    if dev < 0 or _raw_stream is None or dev != _current_device() or b.get_device() != dev or scale_a.get_device() != dev or scale_b.get_device() != dev:
        return None
    dt = a.dtype
    code = _C.TORCH_TO_DT.get(dt)
    out_code = _C.TORCH_TO_DT.get(output_dtype)
    if code is None or out_code is None or b.dtype is not dt or scale_a.dtype is not _F32 or scale_b.dtype is not _F32:
        return None
    if a.dim() != 2 or b.dim() != 2:
        return None
    m, k = a.shape
    kb, n = b.shape
    if k != kb or not scale_a.is_contiguous() or not scale_b.is_contiguous():
        return None
    bias_ptr = None
    if bias is not None:
        if bias.dtype is not output_dtype or bias.get_device() != dev or bias.numel() != n or not bias.is_contiguous():
            return None
        bias_ptr = bias.data_ptr()
    out = _empty((m, n), dtype=output_dtype, device=_device(dev))
    status = _fn("conch_scaled_gemm")(
        out.data_ptr(), a.data_ptr(), b.data_ptr(), scale_a.data_ptr(), scale_b.data_ptr(), bias_ptr, m, n, k,
        a.stride(0), a.stride(1), b.stride(0), b.stride(1), n, 1, scale_a.numel(), scale_b.numel(), code, out_code, _raw_stream(dev))
    if status:
        _C.check(status, "scaled_gemm")
    return out


def mixed_precision_gemm(x: torch.Tensor, w_q_packed: torch.Tensor, w_s: torch.Tensor, w_zp: torch.Tensor | None, weight_size_bits: int,
                         weight_bias: int, group_size: int) -> torch.Tensor | None:
    """conch_mixed_precision_gemm for the forms conch.ops produces on 4- / 8-bit weights: fp16 / bf16 activations and scales, int32
    packed words, no zero points or int32 zero points (one element, or [K / G, N]); output dtype = activation dtype."""
    dev = x.get_device()  # -1 for a host tensor
    if dev < 0 or _raw_stream is None or dev != _current_device() or w_q_packed.get_device() != dev or w_s.get_device() != dev:
        return None
    dt = x.dtype
    if (dt is not torch.float16 and dt is not torch.bfloat16) or w_s.dtype is not dt or w_q_packed.dtype is not torch.int32:
        return None
    if weight_size_bits != 4 and weight_size_bits != 8:
        return None
    if x.dim() != 2 or w_q_packed.dim() != 2 or w_s.dim() != 2 or x.stride(1) != 1 or w_q_packed.stride(1) != 1 or w_s.stride(1) != 1:
        return None
    m, k = x.shape
    rows, n = w_q_packed.shape
    if rows * (32 // weight_size_bits) != k or group_size <= 0 or k % group_size or w_s.shape[0] * group_size != k or w_s.shape[1] != n:
        return None
    zp_ptr, zp_mode, zp_stride = None, _C.ZP_NONE, 0
    if w_zp is not None:
        if w_zp.get_device() != dev or w_zp.dtype is not torch.int32:
            return None
        if w_zp.numel() == 1:
            zp_mode = _C.ZP_SCALAR
        elif w_zp.dim() == 2 and w_zp.shape[0] == w_s.shape[0] and w_zp.shape[1] == n and w_zp.stride(1) == 1:
            zp_mode, zp_stride = _C.ZP_TENSOR, w_zp.stride(0)
        else:
            return None
        zp_ptr = w_zp.data_ptr()
    out = _empty((m, n), dtype=dt, device=_device(dev))
    code = _C.TORCH_TO_DT[dt]
    status = _fn("conch_mixed_precision_gemm")(
        out.data_ptr(), x.data_ptr(), w_q_packed.data_ptr(), w_s.data_ptr(), zp_ptr, m, n, k,
        x.stride(0), w_q_packed.stride(0), w_s.stride(0), zp_stride, n, weight_size_bits, weight_bias, group_size, zp_mode, code, code,
        _raw_stream(dev))
    if status:
        _C.check(status, "mixed_precision_gemm")
    return out


# static per-tensor quantisation, contiguous input: kind 0 = int8, 1 = e4m3fn, 2 = e4m3fnuz (host shim only; None without it)
def static_quant(x: torch.Tensor, scale: torch.Tensor, kind: int) -> torch.Tensor | None:
    return None


# scaled_gemm(static quant(x, scale_x), b, scale_x, scale_b) in one C-ABI call (host shim only; None without it)
def static_quant_scaled_gemm(x: torch.Tensor, b: torch.Tensor, scale_x: torch.Tensor, scale_b: torch.Tensor, output_dtype: torch.dtype,
                             bias: torch.Tensor | None) -> torch.Tensor | None:
    return None


# dynamic per-token quantisation of a contiguous tensor -> (codes, scales [..., 1]) (host shim only; None without it)
def dynamic_quant(x: torch.Tensor, kind: int) -> tuple[torch.Tensor, torch.Tensor] | None:
    return None


# the fused gate/up FFN forms, act 1 = silu, 2 = gelu-tanh (host shim only; None without it)
def scaled_gemm_act(a, b, scale_a, scale_b, output_dtype, bias, act: int) -> torch.Tensor | None:
    return None


def mixed_precision_gemm_act(x, w_q_packed, w_s, w_zp, weight_size_bits: int, weight_bias: int, group_size: int, act: int) -> torch.Tensor | None:
    return None


_py_scaled_gemm, _py_mixed_precision_gemm = scaled_gemm, mixed_precision_gemm
_host = None if __import__("os").environ.get("CONCH_AMD_NO_HOST_SHIM") else _load_host_shim()
if _host is not None:
    scaled_gemm = _host.scaled_gemm  # noqa: F811
    mixed_precision_gemm = _host.mixed_precision_gemm  # noqa: F811
    static_quant = _host.static_quant  # noqa: F811
    if hasattr(_host, "static_quant_scaled_gemm"):  # (a shim built before round 4's last additions lacks it)
        static_quant_scaled_gemm = _host.static_quant_scaled_gemm  # noqa: F811
    if hasattr(_host, "dynamic_quant"):
        dynamic_quant = _host.dynamic_quant  # noqa: F811
    if hasattr(_host, "scaled_gemm_act"):
        scaled_gemm_act = _host.scaled_gemm_act  # noqa: F811
        mixed_precision_gemm_act = _host.mixed_precision_gemm_act  # noqa: F811
