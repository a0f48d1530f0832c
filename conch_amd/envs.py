"""Environment flags, evaluated lazily on attribute access (mirrors conch/envs.py:16-39).

CONCH_BENCH_ENABLE_ALL_REF switches the extra comparator legs of benchmarks/*.py on (benchmarks/_common.py: the vendor
library's GEMM where the reference times vLLM's).  The other names exist so that code written against the reference's
module still imports; the third-party baselines they gate (vLLM, bitsandbytes, the CUDA extension, torchvision) do not
exist on the MI355X path and nothing here reads them.
"""

import os
from collections.abc import Callable
from typing import Any


def _flag(name: str) -> Callable[[], bool]:
    return lambda: os.environ.get(name, "0").strip().lower() in ("1", "true")


environment_variables: dict[str, Callable[[], Any]] = {
    name: _flag(name)
    for name in (
        "CONCH_BENCH_ENABLE_ALL_REF",
        "CONCH_ENABLE_BNB",
        "CONCH_ENABLE_CUDA_EXT",
        "CONCH_ENABLE_TORCHVISION",
        "CONCH_ENABLE_VLLM",
    )
}


def __getattr__(name: str) -> Any:
    if name in environment_variables:
        return environment_variables[name]()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__() -> list[str]:
    return list(environment_variables)
