"""Image containers used as light sources, and the sRGB helpers the image source needs.

Mirror of the parts of optrace/tracer/image/{base_image,rgb_image,grayscale_image}.py and
optrace/tracer/color/srgb.py that feed `RaySource.create_rays` (ray_source.py:120-146, 233-258).
Arrays or image files (decoded with Pillow; the reference uses OpenCV, which is not part of this stack).
"""
from __future__ import annotations

import numpy as np

from .base import BaseClass, check_type, check_above

# relative powers of the three sRGB primary spectra (color/srgb.py:24-26)
SRGB_PRIMARY_POWER_FACTORS = [0.885651229244, 1.000000000000, 0.775993481741]

_WL_MIN0, _WL_MAX0 = 380., 780.


def srgb_to_srgb_linear(rgb: np.ndarray) -> np.ndarray:
    """Remove the sRGB gamma (color/srgb.py:30-47)."""
    a = 0.055
    size = np.abs(rgb)
    lin = np.sign(rgb) * (1 / (1 + a) * (size + a)) ** 2.4
    toe = size <= 0.04045     # linear segment near black
    lin[toe] = 1 / 12.92 * rgb[toe]
    return lin


def srgb_linear_to_srgb(rgbl: np.ndarray) -> np.ndarray:
    """Apply the sRGB gamma (color/srgb.py:357-376); odd in its argument like the inverse above."""
    a = 0.055
    size = np.abs(rgbl)
    out = np.sign(rgbl) * ((1 + a) * size ** (1 / 2.4) - a)
    toe = size <= 0.0031308
    out[toe] = 12.92 * rgbl[toe]
    return out


# linear sRGB -> CIE XYZ (D65), the matrix of color/srgb.py:61-63
_SRGB_TO_XYZ = np.array([0.4124564, 0.3575761, 0.1804375,   # X
                         0.2126729, 0.7151522, 0.0721750,   # Y: the luminance
                         0.0193339, 0.1191920, 0.9503041]).reshape(3, 3)


def power_from_srgb_linear(rgbl: np.ndarray) -> np.ndarray:
    """Relative pixel power under the primary spectra below (color/srgb.py:556-565)."""
    f = SRGB_PRIMARY_POWER_FACTORS
    return f[0] * rgbl[:, :, 0] + f[1] * rgbl[:, :, 1] + f[2] * rgbl[:, :, 2]


def _gauss(x, mu, sig):
    return 1 / (sig * np.sqrt(2 * np.pi)) * np.exp(-0.5 / sig ** 2 * (x - mu) ** 2)


def srgb_r_primary(wl: np.ndarray) -> np.ndarray:
    """Spectrum with the chromaticity of the sRGB red primary (color/srgb.py:469-481)."""
    r = 75.1660756583 * 0.951190393 * (_gauss(wl, 639.854491, 30.0) + 0.0500907584 * _gauss(wl, 418.905848, 80.6220465))
    r[~((wl >= _WL_MIN0) & (wl <= _WL_MAX0))] = 0
    return r


def srgb_g_primary(wl: np.ndarray) -> np.ndarray:
    """color/srgb.py:484-495"""
    g = 83.4999222966 * 1 * _gauss(wl, 539.13108974, 33.31164968)
    g[~((wl >= _WL_MIN0) & (wl <= _WL_MAX0))] = 0
    return g


def srgb_b_primary(wl: np.ndarray) -> np.ndarray:
    """color/srgb.py:498-509"""
    b = 47.99521746361 * 1.16364585503 * (_gauss(wl, 454.833119, 20.1460206) + 0.184484176 * _gauss(wl, 459.658190, 71.0927568))
    b[~((wl >= _WL_MIN0) & (wl <= _WL_MAX0))] = 0
    return b


class _BaseImage(BaseClass):
    """Array plus geometric extent; element [0, 0] is the lower left corner (base_image.py:14-140)."""

    _channels = 3

    def __init__(self, data: np.ndarray, s=None, extent=None, quantity: str = "", projection: str = None,
                 limit: float = None, **kwargs) -> None:
        self._new_lock = False
        if isinstance(data, str):
            data = self._load_image(data)
        self._data = data
        self.extent = extent if extent is not None else self._centred_extent(s)
        self.quantity, self.projection, self.limit = quantity, projection, limit
        BaseClass.__init__(self, **kwargs)
        self._new_lock = True

    @staticmethod
    def _centred_extent(s) -> list:
        """Extent of an image with side lengths s = (sx, sy) around the origin."""
        if s is None:
            raise ValueError("Either s or extent need to be provided for Images")
        check_type("s", s, (list, tuple, np.ndarray))
        sides = np.asarray_chkfinite(s, dtype=np.float64)
        if sides.shape[0] != 2:
            raise ValueError("s needs to have 2 elements.")
        for i, side in enumerate(sides):
            check_above(f"s[{i}]", side, 0)
        sx, sy = sides
        return [-sx / 2, sx / 2, -sy / 2, sy / 2]

    def _load_image(self, path: str) -> np.ndarray:
        """Image file -> array in [0, 1], element [0, 0] in the lower left corner (base_image.py:67-83).
        Decoded with Pillow (the reference uses OpenCV, which is not part of this stack): RGB values are
        identical for lossless formats, the grey conversion is the same ITU-R 601 luma to within one 8-bit level."""
        from PIL import Image
        try:
            with Image.open(path) as im:
                arr = np.asarray(im.convert("RGB" if self._channels == 3 and type(self).__name__ == "RGBImage" else "L"))
        except (OSError, ValueError) as err:
            raise IOError(f"Can't find/process file {path}") from err
        return np.flipud(arr) / 255.0

    @property
    def shape(self):
        return self._data.shape

    @property
    def data(self) -> np.ndarray:
        return self._data.copy()

    @property
    def s(self) -> list:
        x0, x1, y0, y1 = (float(v) for v in self.extent)
        return [x1 - x0, y1 - y0]

    @property
    def Apx(self) -> float:
        (sx, sy), (rows, cols) = self.s, self.shape[:2]
        return float(sx * sy / (cols * rows))

    def profile(self, x: float = None, y: float = None):
        """Cut through the image at one x or one y position (base_image.py:149-186): the pixel column / row that
        contains the position, no interpolation.  -> (bin edges along the cut, [one cut per channel])."""
        if (x is None) == (y is None):
            raise ValueError("Either x or y parameter must be provided.")
        ext, img = self.extent, self._data
        rows, cols = self.shape[0], self.shape[1]
        if x is not None:
            if not ext[0] <= x <= ext[1]:
                raise ValueError(f"Position x={x} is outside the image x-extent of {ext[:2]}")
            edges = np.linspace(ext[2], ext[3], rows + 1)
            k = int((x - ext[0]) / self.s[0] * cols * (1 - 1e-12))
            cut = img[:, k]
        else:
            if not ext[2] <= y <= ext[3]:
                raise ValueError(f"Position y={y} is outside the image y-extent of {ext[2:]}")
            edges = np.linspace(ext[0], ext[1], cols + 1)
            k = int((y - ext[2]) / self.s[1] * rows * (1 - 1e-12))
            cut = img[k]
        return edges, ([cut] if cut.ndim == 1 else [cut[:, c] for c in range(cut.shape[1])])

    def _check_pixels(self, px: np.ndarray) -> None:
        """Shape and value range of the pixel array; the image classes differ in this only."""
        raise NotImplementedError

    def __setattr__(self, key, val):
        if key == "_data":
            check_type(key, val, np.ndarray)
            val = np.asarray_chkfinite(val, dtype=np.float64)
            self._check_pixels(val)
        elif key == "extent":
            check_type(key, val, (list, tuple, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64)
            if val.shape[0] != 4:
                raise ValueError("Extent needs to have 4 elements.")
            if val[0] > val[1] or val[2] > val[3]:   # (equal bounds pass here, as in base_image.py:203)
                raise ValueError("Extent needs to be an array with [x0, x1, y0, y1] with x0 < x1 and y0 < y1.")
        super().__setattr__(key, val)


def _no_negative(px: np.ndarray, wanted: str) -> None:
    low = px.min() if px.size else 0.0
    if low < 0.0:
        raise ValueError(f"There is a negative value of {low} inside the image. Make sure all image data is {wanted}.")


def _at_most_one(px: np.ndarray) -> None:
    high = px.max() if px.size else 0.0
    if high > 1.0:
        raise ValueError(f"There is a value of {high} inside the image. Make sure all image data is in the range [0, 1].")


class RGBImage(_BaseImage):
    """sRGB image with values in [0, 1] (rgb_image.py:12-75)."""
    _channels = 3

    def to_grayscale_image(self) -> "GrayscaleImage":
        """The luminance (CIE Y of the linear values) with the sRGB gamma again (rgb_image.py:41-56)."""
        linear = srgb_to_srgb_linear(self._data)
        xyz = (_SRGB_TO_XYZ @ linear.reshape(-1, 3).T).T.reshape(linear.shape)
        gray = np.clip(srgb_linear_to_srgb(xyz[:, :, 1]), 0, 1)
        return GrayscaleImage(gray, extent=self.extent, desc=self.desc, long_desc=self.long_desc, quantity=self.quantity,
                              projection=self.projection, limit=self.limit)

    def _check_pixels(self, px):
        if px.ndim != 3 or px.shape[2] != 3:
            raise ValueError("Image needs to have three dimensions with 3 elements (RGB) in the third dimension, "
                             f"but has shape {px.shape}.")
        _no_negative(px, "in the range [0, 1]")
        _at_most_one(px)


class ScalarImage(_BaseImage):
    """Single-channel image of a physical quantity, non-negative (scalar_image.py:9-60)."""
    _channels = 1

    def _check_pixels(self, px):
        if px.ndim == 3:
            raise ValueError("Image can't have color information. Either use a RGBImage or remove color information.")
        if px.ndim != 2:
            raise ValueError(f"Image needs to have two dimensions but has shape {px.shape}.")
        _no_negative(px, "non-negative")


class GrayscaleImage(ScalarImage):
    """sRGB-gamma grayscale image with values in [0, 1] (grayscale_image.py:10-60)."""

    def to_rgb_image(self) -> RGBImage:
        """The same values in three channels (grayscale_image.py:36-43)."""
        return RGBImage(np.repeat(self._data[:, :, None], 3, axis=2), extent=self.extent, desc=self.desc,
                        long_desc=self.long_desc, quantity=self.quantity, projection=self.projection, limit=self.limit)

    def _check_pixels(self, px):
        _at_most_one(px)
        ScalarImage._check_pixels(self, px)
