"""RenderImage: XYZ + power histogram of detector hits.

Mirror of optrace/tracer/image/render_image.py:29-421 for the rendering path: `render` bins hits on
the GPU (`ot_render_accumulate`: binning_indices_2d + CIE observer interpolation + f64 atomics) into an
(Ny, Nx, 4) float64 image.  `_data` is the host copy the reference API exposes; the device histogram is
kept alongside so that iterative rendering and the multi-GPU reduction never leave HBM.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np
import torch

from . import _capi
from .base import BaseClass, check_type, check_above
from ._device import require_device, stream_ptr, ptr, to_dev


class RenderImage(BaseClass):

    _tracked = False  # a result container (see base.mutation_epoch)

    EPS: float = 1e-9
    K: float = 683.0  # luminous efficacy [lm/W] (scipy.constants "luminous efficacy", render_image.py:35)
    SIZES = [1, 3, 5, 7, 9, 15, 21, 27, 35, 45, 63, 105, 135, 189, 315, 945]
    MAX_IMAGE_SIDE: int = max(SIZES)   # pixels of the smaller side of the stored histogram
    MAX_IMAGE_RATIO: int = 5           # side ratio at most; a member of SIZES, so coarser views divide evenly
    image_modes = ["sRGB (Absolute RI)", "sRGB (Perceptual RI)", "Outside sRGB Gamut", "Irradiance",
                   "Illuminance", "Lightness (CIELUV)", "Hue (CIELUV)", "Chroma (CIELUV)", "Saturation (CIELUV)"]

    def __init__(self, extent, projection: str = None, **kwargs) -> None:
        self._new_lock = False
        self.extent = extent
        self._extent0 = np.array(self.extent)   # as given: `extent` itself is widened for degenerate images
        self._host = self._dev = None  # host copy of the image (made when `_data` is first read), device histogram
        self._limit, self.projection = None, projection
        BaseClass.__init__(self, **kwargs)
        self._new_lock = True

    @property
    def _data(self):
        """(Ny, Nx, 4) float64 XYZW image as the reference exposes it (render_image.py:80); the histogram lives in
        HBM (`_dev`) and is copied to the host when this attribute is first read."""
        if self._host is None and self._dev is not None:
            # from here on the host copy IS the image: the reference's users (and its tests) edit `_data` in place, which no
            # device copy could notice -- whatever needs the image on the device next uploads this array again
            self.__dict__["_host"] = self._dev.cpu().numpy()
            self.__dict__["_dev"] = None
        return self._host

    def has_image(self) -> bool:
        return self._host is not None or self._dev is not None

    def _check_for_image(self) -> None:
        if not self.has_image():
            raise RuntimeError("Image was not calculated/rendered yet.")

    @property
    def s(self) -> list:
        x0, x1, y0, y1 = (float(v) for v in self.extent)
        return [x1 - x0, y1 - y0]

    @property
    def shape(self) -> tuple:
        self._check_for_image()
        return tuple(self._dev.shape) if self._host is None else self._host.shape

    @property
    def data(self) -> np.ndarray:
        self._check_for_image()
        return self._data.copy()

    @property
    def Apx(self) -> float:
        self._check_for_image()
        (sx, sy), (ny, nx) = self.s, self.shape[:2]
        return sx * sy / (nx * ny)

    def power(self) -> float:
        self._check_for_image()
        return self._plane_sum(3)

    def luminous_power(self) -> float:
        self._check_for_image()
        return self.K * self._plane_sum(1)

    def _plane_sum(self, plane: int) -> float:
        """Sum of one of the X, Y, Z, W planes; without a host copy it is reduced on the device instead of moving
        28-143 MB."""
        source = self._dev if self._host is None else self._host
        return float(source[:, :, plane].sum())

    @property
    def limit(self) -> float:
        return self._limit

    def _fix_extent(self) -> None:
        """Give point / line / extreme-ratio images a valid 2D extent (render_image.py:224-255)."""
        sx, sy = self.s
        MR = self.MAX_IMAGE_RATIO
        outwards = np.array([-1.0, 1.0, -1.0, 1.0])
        self.extent = np.array(self._extent0)
        if max(sx, sy) < 2 * self.EPS:
            self.extent += self.EPS * outwards
        elif not sx or sy / sx > MR:
            xm = (self._extent0[0] + self._extent0[1]) / 2
            self.extent[0] = xm - sy / MR / 2
            self.extent[1] = xm + sy / MR / 2
        elif not sy or sx / sy > MR:
            ym = (self._extent0[2] + self._extent0[3]) / 2
            self.extent[2] = ym - sx / MR / 2
            self.extent[3] = ym + sx / MR / 2
        if self._limit:
            self.extent += outwards * 2.7 * self._limit / 1000.0

    def _pixel_counts(self) -> tuple[int, int]:
        """Nx, Ny: the smaller side has 945 px, the ratio snaps to 1, 3 or 5 (render_image.py:383-387)."""
        Nrs = self.MAX_IMAGE_SIDE
        nf = lambda a: min(self.MAX_IMAGE_RATIO, 1 + 2 * int(a / 2))  # noqa: E731
        Nx = Nrs if self.s[0] <= self.s[1] else Nrs * nf(self.s[0] / self.s[1])
        Ny = Nrs if self.s[0] > self.s[1] else Nrs * nf(self.s[1] / self.s[0])
        return Nx, Ny

    def render(self, p=None, w=None, wl=None, limit: float = None, _dont_filter: bool = False,
               _keep_on_device: bool = False, _into: "torch.Tensor" = None, _fill: "torch.Tensor" = None) -> None:
        """Bin hit positions into the XYZW image (render_image.py:361-421).

        p (n, 3) positions, w (n,) powers, wl (n,) wavelengths: NumPy arrays, or device tensors in the
        layout `ot_detector_hits` produces (p flat component-major).  Hits with w == 0 add nothing.
        The image stays in HBM; `_data` copies it to the host when it is first read (`_keep_on_device` is accepted
        for older callers and has no effect).  `_fill`: the device lists are a compact hit list (`ot_detector_req.fill`).
        """
        self._limit = limit
        self._fix_extent()
        Nx, Ny = self._pixel_counts()
        lib = _capi.load_library()
        dev = require_device()
        if _into is not None:  # accumulate into an existing (Ny, Nx, 4) device histogram (chunked rendering)
            if tuple(_into.shape) != (Ny, Nx, 4):
                raise ValueError("histogram to accumulate into has the wrong shape")
            hist = _into.view(-1)
        else:
            hist = torch.zeros(Ny * Nx * 4, dtype=torch.float64, device=dev)

        n = 0 if p is None else (int(w.shape[0]))
        if n:
            if isinstance(p, tuple):  # (x, y) device tensors
                (px, py), dw, dwl = p, w, wl
            elif isinstance(p, torch.Tensor):
                px, py = p[:n], p[n:2 * n]
                dw, dwl = w, wl
            else:
                p = np.asarray(p, dtype=np.float64)
                px, py = to_dev(p[:, 0], np.float64), to_dev(p[:, 1], np.float64)
                dw, dwl = to_dev(w, np.float32), to_dev(wl, np.float32)
            ext = (C.c_double * 4)(*[float(v) for v in self.extent])
            if _fill is not None:
                _capi.check(lib.ot_render_accumulate_compact(n, ptr(_fill), ptr(px), ptr(py), ptr(dw), ptr(dwl), ext,
                                                             Nx, Ny, ptr(hist), stream_ptr()))
            else:
                _capi.check(lib.ot_render_accumulate(n, ptr(px), ptr(py), ptr(dw), ptr(dwl), ext, Nx, Ny,
                                                     ptr(hist), stream_ptr()))
        self._dev = hist.view(Ny, Nx, 4)
        self._host = None
        if self._limit is not None and not _dont_filter:
            self._apply_rayleigh_filter()

    _MODES = {"Irradiance": 0, "Illuminance": 1, "sRGB (Absolute RI)": 2, "sRGB (Perceptual RI)": 3,
              "Outside sRGB Gamut": 4, "Lightness (CIELUV)": 5, "Hue (CIELUV)": 6, "Chroma (CIELUV)": 7,
              "Saturation (CIELUV)": 8}

    def get(self, mode: str, N: int = 315, L_th: float = 0, chroma_scale: float = None):
        """Converted image for display mode `mode` (render_image.py:131-222), computed by `ot_image_convert`.

        N = pixel count of the smaller side; the nearest of SIZES is used and bins are joined (no interpolation).
        Returns an RGBImage for the sRGB modes, a ScalarImage otherwise."""
        from .image import RGBImage, ScalarImage
        self._check_for_image()
        N = int(N)
        if not 1 <= N <= self.MAX_IMAGE_SIDE:
            raise ValueError(f"N needs to be between 1 and {self.MAX_IMAGE_SIDE}")
        if mode not in self._MODES:
            raise ValueError(f"Invalid display_mode {mode}, should be one of {self.image_modes}.")
        lib = _capi.load_library()
        dev = require_device()
        Ny, Nx, _ = self.shape
        Na = self.SIZES[int(np.argmin(np.abs(N - np.array(self.SIZES))))]
        fact = int(self.MAX_IMAGE_SIDE / Na)
        nx, ny = Nx // fact, Ny // fact
        hist = self._dev if self._dev is not None else to_dev(self._data, np.float64)
        hist = hist.reshape(-1)
        rgb = mode.startswith("sRGB")
        out = torch.empty(ny * nx * (3 if rgb else 1), dtype=torch.float64, device=dev)
        ws = torch.empty(4 * nx * ny + 8, dtype=torch.float64, device=dev)
        cs = float("nan") if chroma_scale is None else float(chroma_scale)
        _capi.check(lib.ot_image_convert(ptr(hist), Nx, Ny, fact, self._MODES[mode], float(self.Apx), float(self.K),
                                         float(L_th), cs, ptr(out), ptr(ws), stream_ptr()))
        iargs = dict(extent=self.extent, projection=self.projection, desc=self.desc, long_desc=self.long_desc,
                     quantity=mode, limit=self.limit)
        data = out.cpu().numpy()
        if rgb:
            return RGBImage(data.reshape(ny, nx, 3), **iargs)
        return ScalarImage(data.reshape(ny, nx), **iargs)

    def _sync_host(self) -> None:
        """The device histogram changed: the host copy is made again when `_data` is next read."""
        if self._dev is not None:
            self._host = None

    # ---- on-disk format (render_image.py:298-328): same keys, files are interchangeable with the reference ----
    def save(self, path: str) -> None:
        """Save as .npz archive (keys _data, extent, limit, desc, long_desc, proj); existing files are replaced."""
        self._check_for_image()
        if not path.endswith(".npz"):
            path += ".npz"
        np.savez_compressed(path, _data=self._data, extent=self.extent, desc=self.desc, long_desc=self.long_desc,
                            limit=np.nan if self._limit is None else self._limit, proj=str(self.projection))

    @staticmethod
    def load(path: str) -> "RenderImage":
        """Load a RenderImage archive written by `save` (or by the reference's RenderImage.save)."""
        io = np.load(path)
        im = RenderImage(io["extent"], long_desc=str(io["long_desc"][()]), desc=str(io["desc"][()]),
                         projection=str(io["proj"][()]))
        stored_limit = float(io["limit"][()])
        im._limit = None if np.isnan(stored_limit) else stored_limit
        im.projection = None if im.projection == "None" else im.projection  # None is stored as a string
        im._data = np.ascontiguousarray(io["_data"], dtype=np.float64)  # uploaded on first use (get / filter)
        return im

    def _apply_rayleigh_filter(self) -> None:
        """Resolution-limit filter: convolve every channel with an Airy disc whose first zero lies at `limit`
        micrometres (render_image.py:257-296); the convolution runs on the GPU (`ot_image_convolve`)."""
        import scipy.special
        if None not in (self._limit, self.projection):
            raise RuntimeError("Resolution limit filter is not applicable for a projected image.")
        px = self._limit / 1000.0 / (self.s[0] / self.shape[1])
        py = self._limit / 1000.0 / (self.s[1] / self.shape[0])
        ps = int(np.ceil(2.7 * max(px, py)))
        ps = ps + 1 if ps % 2 else ps
        Y, X = np.mgrid[-ps:ps:(2 * ps + 1) * 1j, -ps:ps:(2 * ps + 1) * 1j]
        R = np.sqrt((X / px) ** 2 + (Y / py) ** 2) * 3.8317
        Rnz = R[R != 0]
        psf = np.ones((2 * ps + 1, 2 * ps + 1), dtype=np.float64)
        psf[R != 0] = (2 * scipy.special.j1(Rnz) / Rnz) ** 2
        psf[R > 10.1735] = 0  # up to the third zero
        psf *= 1 / psf.sum()
        lib = _capi.load_library()
        dev = require_device()
        Ny, Nx, _ = self.shape
        src = (self._dev if self._dev is not None else to_dev(self._data, np.float64)).reshape(-1)
        dpsf = to_dev(psf, np.float64)
        if ps <= self._DIRECT_PSF_MAX:
            out = torch.empty_like(src)
            _capi.check(lib.ot_image_convolve(ptr(src), Nx, Ny, ptr(dpsf), ps, ptr(out), stream_ptr()))
        else:
            out = self._fft_convolve(src.view(Ny, Nx, 4), dpsf.view(2 * ps + 1, 2 * ps + 1), ps).reshape(-1)
        self._dev = out.view(Ny, Nx, 4)
        self._host = None

    _DIRECT_PSF_MAX = 68  # (2 * 68 + 1)^2 taps fit the LDS of the direct kernel (`ot_image_convolve`)

    @staticmethod
    def _fft_convolve(img: "torch.Tensor", psf: "torch.Tensor", ps: int) -> "torch.Tensor":
        """"same"-size zero-padded convolution of the four planes with a kernel too large for the direct kernel
        (a resolution limit of many pixels, e.g. with a tiny extent): through the FFT, like the reference's
        scipy.signal.fftconvolve (render_image.py:292); rocFFT through torch.fft.  Negative round-off is clamped."""
        Ny, Nx, _ = img.shape
        fh, fw = Ny + 2 * ps, Nx + 2 * ps  # size of the full convolution
        fh, fw = fh + fh % 2, fw + fw % 2
        A = torch.fft.rfft2(img.permute(2, 0, 1), s=(fh, fw))
        B = torch.fft.rfft2(psf, s=(fh, fw))
        full = torch.fft.irfft2(A * B, s=(fh, fw))
        return full[:, ps:ps + Ny, ps:ps + Nx].clamp_min(0).permute(1, 2, 0).contiguous()

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "extent" and val is not None:
            check_type(key, val, (list, tuple, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64)
            if val.shape[0] != 4 or val[0] > val[1] or val[2] > val[3]:
                raise ValueError("extent needs to be [x0, x1, y0, y1] with x1 >= x0, y1 >= y0")
        elif key == "_limit" and val is not None:
            check_type(key, val, (float, int))
            check_above(key, val, 0)
            val = float(val)
        elif key == "projection":
            check_type(key, val, (str, type(None)))
        elif key == "_data":  # an image assigned from outside replaces the histogram; it is uploaded when next needed
            super().__setattr__("_dev", None)
            key = "_host"
        super().__setattr__(key, val)
