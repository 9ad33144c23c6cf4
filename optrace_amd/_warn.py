"""Library warnings (mirrors optrace/warnings.py:5-28)."""
import warnings as _w

from .options import global_options


class OptraceWarning(UserWarning):
    """Warning category of this package."""


def warning(text: str):
    """Emit an OptraceWarning unless warnings are globally disabled."""
    if not global_options.show_warnings:
        return
    _w.warn(text, OptraceWarning, stacklevel=3)
