"""conch_amd -- MI355X-native quantized matmul behind Conch's `conch.ops.quantization` API.

Layout mirrors the slice of the reference package that the hot path touches
(conch/ops/quantization, conch/kernels/quantization, conch/platforms, conch/envs.py,
conch/utils/benchmark.py, conch/third_party/vllm/{quant_utils,scalar_type,utils}.py) so that a
user of the reference can switch `conch.` -> `conch_amd.` in imports.  The compute path is
hand-written HIP for gfx950 in `csrc/`, reached through the C ABI of `include/conch_amd.h`
(`conch_amd._C`).  There is no CPU or PyTorch fallback: ops raise if the library is missing.
"""

__version__ = "0.1.0"


# the slice of the reference's package tree this repository mirrors (module paths below `conch`)
_MIRRORED = (
    "", "envs", "ops", "ops.quantization", "ops.quantization.gemm", "ops.quantization.int8", "ops.quantization.fp8",
    "ops.quantization.bitsandbytes", "ops.quantization.bitsandbytes.functional", "kernels", "kernels.quantization",
    "kernels.quantization.gemm", "kernels.quantization.int8", "kernels.quantization.fp8", "kernels.quantization.bitsandbytes",
    "platforms", "utils", "utils.benchmark", "third_party", "third_party.vllm", "third_party.vllm.quant_utils",
    "third_party.vllm.scalar_type", "third_party.vllm.utils",
)


def install_as_conch(force: bool = False) -> list[str]:
    """Register this package's modules under the reference's import names (`conch`, `conch.ops.quantization.gemm`, ...), so
    that `from conch.ops.quantization.gemm import scaled_gemm` -- what the reference's own tests and benchmarks write
    (tests/scaled_gemm_test.py:11) -- resolves to the HIP path without touching their imports:

        import conch_amd; conch_amd.install_as_conch()

    Refuses (ImportError) when a real `conch` is already imported or importable, unless `force=True`: shadowing an installed
    package silently would be worse than a renamed import.  Returns the aliases it installed."""
    import importlib
    import importlib.util
    import sys

    if not force:
        existing = sys.modules.get("conch")
        if existing is not None and getattr(existing, "__name__", "conch") != __name__:
            raise ImportError("install_as_conch: a `conch` package is already imported; pass force=True to shadow it")
        if existing is None and importlib.util.find_spec("conch") is not None:
            raise ImportError("install_as_conch: a `conch` package is importable here; pass force=True to shadow it")
    done = []
    for sub in _MIRRORED:
        mod = importlib.import_module(f"{__name__}.{sub}" if sub else __name__)
        alias = f"conch.{sub}" if sub else "conch"
        sys.modules[alias] = mod
        done.append(alias)
    return done
