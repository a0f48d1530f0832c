"""CPU oracle for the quantized-matmul hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``conch_amd/`` may import this package.
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` (as the checker / the timed CPU baseline, never as the product).

Two independent restatements: `oracle/reference.py` (torch CPU ops in the reference's order + a numpy
bit-level FP8 codec) and `oracle/oracle_c.c` (plain C scalar loops with hand-written IEEE rounding, built
by `oracle/build_c.py`, no torch involved).

Parity status: PINNED.  Every function here is checked against golden vectors under
``tests/golden/`` that were produced by importing the real reference
(``/root/reference/conch``) in the authoring container with
``tests/golden/make_golden.py`` (committed).  See ``tests/test_oracle_golden.py``.
"""

from oracle.reference import (  # noqa: F401
    CHANNEL_ACTIVATION_ONLY,
    CHANNEL_NONE,
    CHANNEL_WEIGHT_AND_ACTIVATION,
    CHANNEL_WEIGHT_ONLY,
    GROUP_ASYMMETRIC,
    GROUP_NONE,
    GROUP_SHIFT,
    GROUP_SYMMETRIC_NO_SHIFT,
    GROUP_SYMMETRIC_WITH_SHIFT,
    FP8_E4M3FN,
    FP8_E4M3FNUZ,
    decode_fp8,
    dequantize_modes_ref,
    dequantize_packed,
    dynamic_quant_scale_ref,
    dynamic_scaled_fp8_quant_ref,
    dynamic_scaled_int8_quant_ref,
    encode_fp8,
    gelu_tanh_and_mul_ref,
    mixed_precision_gemm_gelu_tanh_and_mul_ref,
    mixed_precision_gemm_modes_ref,
    mixed_precision_gemm_ref,
    mixed_precision_gemm_silu_and_mul_ref,
    pack_rows_ref,
    quantize_weights_ref,
    scaled_fp8_quant_ref,
    scaled_gemm_gelu_tanh_and_mul_ref,
    scaled_gemm_modes_ref,
    scaled_gemm_ref,
    scaled_gemm_silu_and_mul_ref,
    scaled_int8_quant_ref,
    silu_and_mul_ref,
    unpack_rows_ref,
)

from oracle.bnb import (  # noqa: E402,F401
    create_dynamic_map_ref,
    dequantize_blockwise_ref,
    quantize_blockwise_ref,
)
