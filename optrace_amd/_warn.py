"""Library warnings (mirrors optrace/warnings.py:5-28)."""
import warnings as _w

from .options import global_options


class OptraceWarning(UserWarning):
    """Warning category of this package."""


def warning(text: str) -> None:
    """Emit an OptraceWarning unless warnings are globally disabled."""
    if global_options.show_warnings:
        _w.warn(text, OptraceWarning, stacklevel=3)
