"""`current_platform` is detected lazily on first access (mirrors conch/platforms/__init__.py:14-19)."""

from typing import Any

from conch_amd.platforms.platform import Platform, PlatformEnum, detect_current_platform

_current_platform = None


def __getattr__(name: str) -> Any:
    if name == "current_platform":
        global _current_platform  # noqa: PLW0603
        if _current_platform is None:
            _current_platform = detect_current_platform()
        return _current_platform
    raise AttributeError(f"No attribute named '{name}' exists in {__name__}.")


__all__ = ["Platform", "PlatformEnum", "current_platform", "detect_current_platform"]
