"""Environment flags, evaluated lazily on attribute access (mirrors conch/envs.py:16-39).

The optional third-party baselines the reference gates behind these flags (vLLM, bitsandbytes,
its CUDA extension, torchvision) do not exist on the MI355X path, so every flag defaults to false
and only switches harness behaviour.
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
