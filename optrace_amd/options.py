"""Global options singleton (mirrors optrace/global_options.py:8-97)."""
from __future__ import annotations

import contextlib


class _GlobalOptions:
    """Run-time switches shared by all objects.

    multithreading: kept for API compatibility.  On the GPU backend rays are always processed in
        parallel; the flag only controls whether ray generation stratifies over several sub-ranges
        the way the reference's threads do (ray_storage.py:147-171).
    show_progress_bar / show_warnings: as in the reference.
    wavelength_range: visible range used by Constant spectra and colour tables.
    """

    def __init__(self) -> None:
        self.multithreading = True
        self.show_progress_bar = True
        self.show_warnings = True
        self.wavelength_range = [380., 780.]

    def __setattr__(self, key, val):
        if key in ("multithreading", "show_progress_bar", "show_warnings"):
            if not isinstance(val, bool):
                raise TypeError(f"Property '{key}' needs to be of type bool, but is {type(val).__name__}.")
        elif key == "wavelength_range":
            if not isinstance(val, (list, tuple)) or len(val) != 2:
                raise TypeError("wavelength_range needs to be a two element list")
            val = [float(val[0]), float(val[1])]
            if val[0] > 380. or val[1] < 780.:
                raise ValueError("wavelength_range needs to include at least [380, 780] nm")
        else:
            raise AttributeError(f"Invalid property {key}")
        object.__setattr__(self, key, val)

    @contextlib.contextmanager
    def no_warnings(self):
        state = self.show_warnings
        self.show_warnings = False
        try:
            yield
        finally:
            self.show_warnings = state

    @contextlib.contextmanager
    def no_progress_bar(self):
        state = self.show_progress_bar
        self.show_progress_bar = False
        try:
            yield
        finally:
            self.show_progress_bar = state


global_options = _GlobalOptions()
