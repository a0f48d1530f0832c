"""FP8 quantization (public API; mirrors conch/ops/quantization/fp8.py:12-58).

fp8 flavour: the reference hard-wires float8_e4m3fnuz for AMD (fp8.py:27,54 -- MI300X).  Here the
default is `current_platform.fp8_dtype()` = float8_e4m3fn on gfx950 (the MFMA-native OCP format);
`static_scaled_fp8_quant` accepts an output tensor of either flavour and `scaled_fp8_quant` takes
an optional `output_dtype` to request float8_e4m3fnuz explicitly.
"""

import torch

from conch_amd.kernels.quantization.fp8 import static_scaled_fp8_quant_launcher
from conch_amd.platforms import current_platform

_FP8_DTYPES = (torch.float8_e4m3fn, torch.float8_e4m3fnuz)


def static_scaled_fp8_quant(
    output_tensor: torch.Tensor,
    input_tensor: torch.Tensor,
    scale: torch.Tensor,
) -> None:
    """Quantize `input_tensor` to fp8 e4m3 into `output_tensor` with the static per-tensor `scale`."""
    assert output_tensor.shape == input_tensor.shape  # noqa: S101
    assert scale.numel() == 1  # noqa: S101
    assert output_tensor.dtype in _FP8_DTYPES  # noqa: S101

    static_scaled_fp8_quant_launcher(output_tensor, input_tensor, scale)


def scaled_fp8_quant(
    input_tensor: torch.Tensor,
    scale: torch.Tensor | None = None,
    output_dtype: torch.dtype | None = None,
) -> tuple[torch.Tensor, torch.Tensor]:
    """Scaled fp8 quantization; returns (quantized tensor, the same `scale` object)."""
    if scale is None:
        error_msg = "Dynamic quantization not implemented yet"
        raise NotImplementedError(error_msg)

    if output_dtype is None:
        output_dtype = current_platform.fp8_dtype()
    output_tensor = torch.empty_like(input_tensor, dtype=output_dtype)

    static_scaled_fp8_quant(output_tensor, input_tensor, scale)
    return output_tensor, scale
