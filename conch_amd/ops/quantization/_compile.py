"""torch.compile support: the hot ops as opaque `torch.library` custom ops.

The reference's microbenchmarks take `--compile-conch` (benchmarks/static_scaled_int8_quant_benchmark.py:73-80,130:
`torch.compile(scaled_int8_quant_conch)`).  Dynamo cannot trace a ctypes call, so while a graph is being captured the public ops
route through the custom ops below -- each one the same C-ABI call as the eager op, with a fake (meta) implementation that only
describes the result -- and a compiled function holds them as single opaque nodes (`fullgraph=True` works, nothing is decomposed,
no Triton kernel is generated for them).  Eager calls never come here.
"""

from __future__ import annotations

import torch

_DEFINED = False


def compiling() -> bool:
    fn = getattr(getattr(torch, "compiler", None), "is_compiling", None)
    return bool(fn()) if fn is not None else False


def ensure_defined() -> None:
    """Register the custom ops once.  Called when the op modules are imported -- not from inside a traced function: Dynamo
    would try to trace the registration itself."""
    global _DEFINED  # noqa: PLW0603
    if _DEFINED:
        return
    from conch_amd.kernels.quantization.fp8 import static_scaled_fp8_quant_launcher
    from conch_amd.kernels.quantization.gemm import mixed_precision_gemm_launcher, scaled_gemm_launcher
    from conch_amd.kernels.quantization.int8 import static_scaled_int8_quant_launcher
    from conch_amd.ops.quantization._metadata import create_mixed_precision_metadata, create_scaled_metadata

    @torch.library.custom_op("conch_amd::static_scaled_int8_quant", mutates_args=())
    def _int8(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x, dtype=torch.int8)
        static_scaled_int8_quant_launcher(out, x, scale)
        return out

    @_int8.register_fake
    def _(x, scale):
        return torch.empty_like(x, dtype=torch.int8)

    @torch.library.custom_op("conch_amd::static_scaled_fp8_quant", mutates_args=())
    def _fp8(x: torch.Tensor, scale: torch.Tensor, fnuz: bool) -> torch.Tensor:
        out = torch.empty_like(x, dtype=torch.float8_e4m3fnuz if fnuz else torch.float8_e4m3fn)
        static_scaled_fp8_quant_launcher(out, x, scale)
        return out

    @_fp8.register_fake
    def _(x, scale, fnuz):
        return torch.empty_like(x, dtype=torch.float8_e4m3fnuz if fnuz else torch.float8_e4m3fn)

    @torch.library.custom_op("conch_amd::scaled_gemm", mutates_args=())
    def _sg(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor, output_dtype: torch.dtype,
            bias: torch.Tensor | None) -> torch.Tensor:
        meta = create_scaled_metadata(a, b, scale_a, scale_b, output_dtype)
        out = a.new_empty((meta.m_dim, meta.n_dim), dtype=output_dtype)
        scaled_gemm_launcher(out, a, b, scale_a, scale_b, meta, bias=bias)
        return out

    @_sg.register_fake
    def _(a, b, scale_a, scale_b, output_dtype, bias):
        return a.new_empty((a.shape[0], b.shape[1]), dtype=output_dtype)

    @torch.library.custom_op("conch_amd::mixed_precision_gemm", mutates_args=())
    def _mg(x: torch.Tensor, w_q_packed: torch.Tensor, w_s: torch.Tensor, w_zp: torch.Tensor | None, weight_size_bits: int,
            weight_bias: int, group_size: int) -> torch.Tensor:
        meta = create_mixed_precision_metadata(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size)
        out = x.new_empty((meta.m_dim, meta.n_dim), dtype=meta.output_dtype)
        mixed_precision_gemm_launcher(out, x, w_q_packed, w_s, w_zp, meta)
        return out

    @_mg.register_fake
    def _(x, w_q_packed, w_s, w_zp, weight_size_bits, weight_bias, group_size):
        return x.new_empty((x.shape[0], w_q_packed.shape[1]))

    _DEFINED = True
