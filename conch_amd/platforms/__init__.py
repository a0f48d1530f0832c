"""`conch_amd.platforms.current_platform`: detected on first use, then cached.

Same access pattern as the reference (`from conch.platforms import current_platform`,
conch/platforms/__init__.py:14-19); implemented with a cached factory behind the module-level
`__getattr__` hook so that importing this package never touches the GPU runtime.
"""

from __future__ import annotations

import functools
from typing import Any

from conch_amd.platforms.platform import Platform, PlatformEnum, detect_current_platform

__all__ = ["Platform", "PlatformEnum", "current_platform", "detect_current_platform"]


@functools.lru_cache(maxsize=1)
def _platform_singleton() -> Platform:
    return detect_current_platform()


_LAZY = {"current_platform": _platform_singleton}


def __getattr__(name: str) -> Any:
    try:
        return _LAZY[name]()
    except KeyError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
