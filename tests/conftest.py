"""pytest configuration: markers and shared helpers."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def pytest_configure(config: pytest.Config) -> None:
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Files of the -m gpu run that start OTHER processes (two-rank runs, torch.compile workers): they run LAST, so that with -x a
# failure in them can never again hide the hot-path parity files (round 5: the driver's run died in the third of six files with
# 2 712 parity tests behind it).  Within the parity files: the BASELINE-config tests first, then the rest in file order.
_LAST_FILES = ("test_gpu_compile.py", "test_gpu_distributed.py")
_FIRST_FILES = ("test_gpu_gemm.py", "test_gpu_quant.py")


def pytest_collection_modifyitems(config: pytest.Config, items: list) -> None:
    def rank(item) -> tuple[int, int]:
        name = Path(str(item.fspath)).name
        if name in _LAST_FILES:
            return (3, _LAST_FILES.index(name))
        if name in _FIRST_FILES:
            headline = any(tag in item.name for tag in ("c1_config", "c2_config", "c3_config", "c4_config", "c5_", "readme_shape"))
            return (0 if headline else 1, _FIRST_FILES.index(name))
        return (2, 0)

    items.sort(key=rank)  # stable: file order and definition order survive inside every class


def from_bits(arr: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """Inverse of make_golden.bits(): raw storage -> torch tensor of `dtype`."""
    if dtype in (torch.float16, torch.bfloat16):
        return torch.from_numpy(arr.view(np.int16).copy()).view(dtype)
    if dtype in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
        return torch.from_numpy(arr.copy()).view(dtype)
    return torch.from_numpy(arr.copy())


def to_bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
        return t.view(torch.uint8).numpy()
    return t.numpy()


@pytest.fixture(scope="session")
def golden():
    cache: dict[str, dict[str, np.ndarray]] = {}

    def load(name: str) -> dict[str, np.ndarray]:
        if name not in cache:
            with np.load(GOLDEN / f"{name}.npz") as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load
