"""Platform description (mirrors conch/platforms/platform.py:24-87, AMD-first).

Deliberate deviation: the reference maps "AMD" to float8_e4m3fnuz
(conch/ops/quantization/fp8.py:27), which was right for MI300X.  gfx950's MFMA and conversion
instructions implement OCP e4m3fn, so `fp8_dtype()` returns float8_e4m3fn on gfx95x parts and
float8_e4m3fnuz only on gfx94x.  Both flavours are accepted by every op.
"""

from __future__ import annotations

import enum
from dataclasses import dataclass

import torch


class PlatformEnum(enum.Enum):
    NVIDIA = enum.auto()
    AMD = enum.auto()
    XPU = enum.auto()
    CPU = enum.auto()
    UNSPECIFIED = enum.auto()


_GCN_ARCH: dict[int, str] = {}  # per Platform object (a dataclass: not hashable)


@dataclass
class Platform:
    platform_enum: PlatformEnum
    device: str

    def name(self) -> str:
        return self.platform_enum.name

    def is_nvidia(self) -> bool:
        return self.platform_enum == PlatformEnum.NVIDIA

    def is_amd(self) -> bool:
        return self.platform_enum == PlatformEnum.AMD

    def is_unspecified(self) -> bool:
        return self.platform_enum == PlatformEnum.UNSPECIFIED

    def has_cuda(self) -> bool:
        return self.is_nvidia() or self.is_amd()

    def gcn_arch(self) -> str:
        """'gfx950' style architecture name of device 0 ('' when there is no GPU).  Queried once: `fp8_dtype()` sits on the host
        path of every `scaled_fp8_quant` call, where the two torch queries cost ~2 us of a 7 us call."""
        cached = _GCN_ARCH.get(id(self))
        if cached is None:
            cached = ""
            if self.is_amd() and torch.cuda.is_available():
                cached = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
            _GCN_ARCH[id(self)] = cached
        return cached

    def is_mi355x(self) -> bool:
        return self.gcn_arch().startswith("gfx95")

    def supports_fp8(self) -> bool:
        if torch.cuda.is_available():
            major, minor = torch.cuda.get_device_capability()
            return major * 10 + minor >= 89
        return True

    def fp8_dtype(self) -> torch.dtype:
        """The fp8 e4m3 flavour the quantisers emit by default on this platform."""
        if self.is_amd() and self.gcn_arch().startswith("gfx94"):
            return torch.float8_e4m3fnuz
        return torch.float8_e4m3fn

    def get_device_name(self) -> str:
        return torch.cuda.get_device_name() if torch.cuda.is_available() else "unknown"


def detect_current_platform() -> Platform:
    if torch.cuda.is_available():
        if torch.version.hip is not None:
            return Platform(PlatformEnum.AMD, "cuda")
        if torch.version.cuda is not None:
            return Platform(PlatformEnum.NVIDIA, "cuda")
    return Platform(PlatformEnum.CPU, "cpu")
