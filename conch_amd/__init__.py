"""conch_amd -- MI355X-native quantized matmul behind Conch's `conch.ops.quantization` API.

Layout mirrors the slice of the reference package that the hot path touches
(conch/ops/quantization, conch/kernels/quantization, conch/platforms, conch/envs.py,
conch/utils/benchmark.py, conch/third_party/vllm/{quant_utils,scalar_type,utils}.py) so that a
user of the reference can switch `conch.` -> `conch_amd.` in imports.  The compute path is
hand-written HIP for gfx950 in `csrc/`, reached through the C ABI of `include/conch_amd.h`
(`conch_amd._C`).  There is no CPU or PyTorch fallback: ops raise if the library is missing.
"""

__version__ = "0.1.0"
