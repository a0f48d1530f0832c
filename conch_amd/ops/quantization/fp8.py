"""FP8 (e4m3) quantization, public API (names and signatures of conch/ops/quantization/fp8.py:12-58).

fp8 flavour: the reference hard-wires float8_e4m3fnuz for AMD (fp8.py:27,54 -- MI300X).  Here the default is
`current_platform.fp8_dtype()` = float8_e4m3fn on gfx950 (the MFMA-native OCP format);
`static_scaled_fp8_quant` accepts an output tensor of either flavour and `scaled_fp8_quant` takes an
optional `output_dtype` to request float8_e4m3fnuz explicitly.
"""

import torch

from conch_amd.kernels.quantization.fp8 import dynamic_scaled_fp8_quant_launcher, static_scaled_fp8_quant_launcher
from conch_amd.ops.quantization._static_quant import quantize_into, quantize_new
from conch_amd.platforms import current_platform

_OUT = (torch.float8_e4m3fn, torch.float8_e4m3fnuz)


def static_scaled_fp8_quant(output_tensor: torch.Tensor, input_tensor: torch.Tensor, scale: torch.Tensor) -> None:
    """output_tensor[t, h] = fp8(clamp(input_tensor[t, h] / scale, -max, max)), written in place."""
    quantize_into(output_tensor, input_tensor, scale, static_scaled_fp8_quant_launcher, _OUT)


def scaled_fp8_quant(input_tensor: torch.Tensor, scale: torch.Tensor | None = None,
                     output_dtype: torch.dtype | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Quantize to fp8 e4m3; returns (fp8 tensor, scale).

    `scale` given: static per-tensor quantisation (the reference's only mode).  `scale=None`: dynamic per-token
    quantisation, scale[t] = absmax(x[t]) / finfo(dtype).max (the reference raises NotImplementedError here).
    """
    dtype = output_dtype if output_dtype is not None else current_platform.fp8_dtype()
    return quantize_new(input_tensor, scale, dtype, static_scaled_fp8_quant_launcher, _OUT,
                        dynamic_scaled_fp8_quant_launcher)
