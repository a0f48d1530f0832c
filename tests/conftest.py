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
