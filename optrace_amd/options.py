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

    _SWITCHES = ("multithreading", "show_progress_bar", "show_warnings")

    def __setattr__(self, key, val):
        if key in self._SWITCHES:
            if type(val) is not bool:
                raise TypeError(f"Property '{key}' needs to be of type bool, but is {type(val).__name__}.")
        elif key != "wavelength_range":
            raise AttributeError(f"Invalid property {key}")
        else:
            if not isinstance(val, (list, tuple)) or len(val) != 2:
                raise TypeError("wavelength_range needs to be a two element list")
            val = [float(val[0]), float(val[1])]
            if val[0] > 380. or val[1] < 780.:
                raise ValueError("wavelength_range needs to include at least [380, 780] nm")
        object.__setattr__(self, key, val)

    @contextlib.contextmanager
    def _switched_off(self, switch: str):
        before = getattr(self, switch)
        setattr(self, switch, False)
        try:
            yield
        finally:
            setattr(self, switch, before)

    def no_warnings(self):
        """Context manager: no library warnings inside."""
        return self._switched_off("show_warnings")

    def no_progress_bar(self):
        """Context manager: no progress bar inside."""
        return self._switched_off("show_progress_bar")


global_options = _GlobalOptions()
