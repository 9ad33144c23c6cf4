"""RaySource: emitting shape + divergence + orientation + polarisation + spectrum.

Host-side mirror of optrace/tracer/geometry/ray_source.py:24-523.  The object validates and stores the
parameters; ray creation itself (`create_rays`, ray_source.py:204) runs in the device generation kernel,
which reads the `ot_source` descriptor built by `_source_fields`.
"""
from __future__ import annotations

from typing import Any, Callable

import numpy as np

from .. import _capi
from ..base import check_type, check_above, check_in
from ..spectrum import LightSpectrum, d65_illuminant
from .elements import Element
from .surfaces import (Surface, Point, Line, CircularSurface, RingSurface, RectangularSurface, SlitSurface)
from ..image import RGBImage, GrayscaleImage, srgb_to_srgb_linear, power_from_srgb_linear

#: default spectrum: CIE standard illuminant D65 (presets/light_spectrum.py)
d65_spectrum = LightSpectrum("Function", func=d65_illuminant, desc="D65", long_desc="Illuminant D65")


def _usable_pdf(f: np.ndarray) -> None:
    """What random.inverse_transform_sampling demands of a density before it samples from it (random.py:129-133)."""
    if not f.sum():
        raise RuntimeError("Cumulated probability is zero.")
    if f.min() < 0:
        raise RuntimeError("Got negative value in pdf.")


class RaySource(Element):

    divergences = ["None", "Lambertian", "Isotropic", "Function"]
    orientations = ["Constant", "Converging", "Function"]
    polarizations = ["Constant", "Uniform", "List", "Function", "x", "y", "xy"]

    abbr = "RS"
    _allow_non_2D = True
    _max_image_px = 2e6

    def __init__(self, surface, pos=None, divergence: str = "None", div_angle: float = 0.5,
                 div_2d: bool = False, div_axis_angle: float = 0, div_func: Callable = None,
                 div_args: dict = {}, spectrum: LightSpectrum = None, power: float = 1., s=None,
                 s_sph=None, orientation: str = "Constant", conv_pos=None, or_func: Callable = None,
                 or_args: dict = {}, polarization: str = "Uniform", pol_angle: float = 0.,
                 pol_angles=None, pol_probs=None, pol_func: Callable = None, pol_args: dict = {},
                 **kwargs) -> None:
        self._new_lock = False

        # an image as the emitting area: a rectangle of its size, pixels chosen with probability ~ their power
        self._image = surface if isinstance(surface, (RGBImage, GrayscaleImage)) else None
        self._pIf = None
        if self._image is not None:
            linear = srgb_to_srgb_linear(surface._data if isinstance(surface, RGBImage) else surface.data)
            weight = (power_from_srgb_linear(linear) if isinstance(surface, RGBImage) else linear).ravel()
            pdf = 1 / weight.sum() * weight
            pdf.setflags(write=False)  # (derived from the read-only image data: a writeable array of this size would switch
            self._pIf = pdf            # off the unchanged-scene shortcut of Raytracer.trace and be checksummed at every trace)
            surface = RectangularSurface(dim=surface.s)
        Element.__init__(self, surface, [0, 0, 0] if pos is None else pos, **kwargs)

        if s_sph is not None:
            check_type("s_sph", s_sph, (list, np.ndarray))
            polar, azimuth = (np.radians(angle) for angle in s_sph[:2])
            s = [np.sin(polar) * np.cos(azimuth), np.sin(polar) * np.sin(azimuth), np.cos(polar)]
        settings = dict(
            power=power, spectrum=d65_spectrum if spectrum is None else spectrum,
            polarization=polarization, pol_angle=pol_angle, pol_func=pol_func, pol_angles=pol_angles,
            pol_probs=pol_probs, pol_args=pol_args,
            divergence=divergence, div_angle=div_angle, div_axis_angle=div_axis_angle, div_func=div_func,
            div_2d=div_2d, div_args=div_args,
            orientation=orientation, s=[0, 0, 1] if s is None else s,
            conv_pos=[0, 0, 0] if conv_pos is None else conv_pos, or_func=or_func, or_args=or_args)
        for name, value in settings.items():
            setattr(self, name, value)
        self._new_lock = True

    # ---- device descriptor ----------------------------------------------------------------------
    def _source_fields(self) -> dict:
        """Fields of the `ot_source` describing this source (arrays stay NumPy; scene.py packs them)."""
        f: dict = {}
        sf = self.front
        f["pos"] = [float(v) for v in sf.pos]
        if self._image is not None:
            f["shape"] = _capi.SRC_IMAGE_RGB if isinstance(self._image, RGBImage) else _capi.SRC_IMAGE_GRAY
            f["dim"] = [float(sf.dim[0]), float(sf.dim[1])]
            f["angle"] = float(sf._angle)
            f["img_h"], f["img_w"] = int(self._image.shape[0]), int(self._image.shape[1])
            f["img_pdf"] = np.ascontiguousarray(self._pIf, dtype=np.float64)
            if isinstance(self._image, RGBImage):
                f["img_rgb"] = np.ascontiguousarray(self._image._data, dtype=np.float64).reshape(-1)
        elif isinstance(sf, Point):
            f["shape"] = _capi.SRC_POINT
        elif isinstance(sf, Line):
            f["shape"] = _capi.SRC_LINE
            f["r"], f["angle"] = float(sf.r), float(np.deg2rad(sf.angle))
        elif isinstance(sf, RingSurface):
            f["shape"] = _capi.SRC_RING
            f["r"], f["ri"] = float(sf.r), float(sf.ri)
        elif isinstance(sf, CircularSurface):
            f["shape"] = _capi.SRC_CIRCLE
            f["r"] = float(sf.r)
        elif isinstance(sf, RectangularSurface):
            f["shape"] = _capi.SRC_RECT
            f["dim"] = [float(sf.dim[0]), float(sf.dim[1])]
            f["angle"] = float(sf._angle)
        else:
            raise _capi.BackendError(f"Source shape {type(sf).__name__} is not supported.")

        # spectrum (RGB images carry their own, ray_source.py:257)
        if f["shape"] != _capi.SRC_IMAGE_RGB:
            check_type("RaySource.spectrum", self.spectrum, LightSpectrum)
            f.update(self.spectrum._source_fields())

        # orientation
        if self.orientation == "Constant":
            f["orientation"] = _capi.OR_CONSTANT
            f["s"] = [float(v) for v in self.s]
        elif self.orientation == "Converging":
            f["orientation"] = _capi.OR_CONVERGING
            f["conv_pos"] = [float(v) for v in self.conv_pos]
        else:
            # or_func is a Python callable of the start positions: RayStorage evaluates it between a position
            # pre-pass and the generation proper and hands the result to the kernel as an array
            if not callable(self.or_func):
                raise TypeError("RaySource.or_func needs to be callable")
            f["orientation"] = _capi.OR_ARRAY
            f["s"] = [float(v) for v in self.s]

        # divergence
        f["div_2d"] = int(self.div_2d)
        f["div_angle"] = float(self.div_angle)
        f["div_axis_angle"] = float(self.div_axis_angle)
        if self.divergence == "None":
            f["divergence"] = _capi.DIV_NONE
        elif self.divergence == "Lambertian":
            f["divergence"] = _capi.DIV_LAMBERTIAN
        elif self.divergence == "Isotropic":
            f["divergence"] = _capi.DIV_ISOTROPIC
        else:
            if self.div_func is None:
                raise TypeError("RaySource.div_func needs to be callable")
            x = np.linspace(0, np.radians(self.div_angle), 1000)
            pdf = np.asarray(self.div_func(x, **self.div_args), dtype=np.float64)
            if not self.div_2d:
                pdf = pdf * np.sin(x)
            _usable_pdf(pdf)
            F = np.concatenate(([0.], np.cumsum((pdf[1:] + pdf[:-1]) / 2)))
            f["divergence"] = _capi.DIV_TABLE
            f["div_tab"] = np.concatenate((x, F))
            f["n_div"] = len(x)

        # polarisation
        pol = self.polarization
        if pol in ("x", "y", "Constant"):
            f["polarization"] = _capi.POL_CONSTANT
            f["pol_angle"] = {"x": 0., "y": np.pi / 2}.get(pol, float(np.radians(self.pol_angle)))
        elif pol == "Uniform":
            f["polarization"] = _capi.POL_UNIFORM
        elif pol in ("xy", "List"):
            if pol == "xy":
                ang, probs = np.array([0, np.pi / 2]), np.ones(2)
            else:
                check_type("RaySource.pol_angles", self.pol_angles, (np.ndarray, list))
                probs = np.ones_like(self.pol_angles) if self.pol_probs is None else np.asarray(self.pol_probs)
                ang = np.radians(self.pol_angles)
            _usable_pdf(np.asarray(probs, dtype=np.float64))
            keep = probs > 0
            f["polarization"] = _capi.POL_LIST
            f["pol_tab"] = np.concatenate((np.asarray(ang, dtype=np.float64)[keep], np.cumsum(probs[keep])))
            f["n_pol"] = int(keep.sum())
        else:
            if self.pol_func is None:
                raise TypeError("RaySource.pol_func needs to be callable")
            x = np.linspace(0, 2 * np.pi, 5000)
            pdf = np.asarray(self.pol_func(x, **self.pol_args), dtype=np.float64)
            _usable_pdf(pdf)
            F = np.concatenate(([0.], np.cumsum((pdf[1:] + pdf[:-1]) / 2)))
            f["polarization"] = _capi.POL_TABLE
            # the reference converts the sampled angle with np.radians once more (ray_source.py:392)
            f["pol_tab"] = np.concatenate((np.radians(x), F))
            f["n_pol"] = len(x)
        return f

    def create_rays(self, N: int, no_pol: bool = False, power: float = None):
        """N rays of this source: (p, s, pols, weights, wavelengths), generated on the GPU
        (reference: ray_source.py:204-437)."""
        from .. import ops
        return ops.create_rays(self, N, no_pol, power)

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "divergence":
            check_type(key, val, str)
            check_in(key, val, self.divergences)
        elif key == "orientation":
            check_type(key, val, str)
            check_in(key, val, self.orientations)
        elif key == "polarization":
            check_type(key, val, str)
            check_in(key, val, self.polarizations)
        elif key in ("pol_angle", "div_axis_angle"):
            check_type(key, val, (int, float))
            val = float(val)
        elif key in ("power", "div_angle"):
            check_type(key, val, (int, float))
            check_above(key, val, 0)
            val = float(val)
        elif key == "s":
            check_type(key, val, (list, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64) / np.linalg.norm(val)
            if val.shape[0] != 3:
                raise TypeError("s needs to have 3 dimensions")
            check_above("s[2]", val[2], 0)
        elif key == "conv_pos":
            check_type(key, val, (list, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64)
            if val.shape[0] != 3:
                raise TypeError("conv_pos needs to have 3 dimensions")
        elif key == "div_2d":
            check_type(key, val, bool)
        elif key == "spectrum":
            check_type(key, val, LightSpectrum)
        elif key in ("or_func", "div_func", "pol_func"):
            if val is not None and not callable(val):
                raise TypeError(f"{key} needs to be callable or None")
        elif key in ("pol_angles", "pol_probs") and val is not None:
            check_type(key, val, (list, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64)
        elif key == "_image" and val is not None:
            if val.shape[0] * val.shape[1] > self._max_image_px:
                raise RuntimeError("For performance reasons only images with less than 2 megapixels are allowed.")
            if val._data.sum() <= 0:
                raise ValueError("Image can not be completely black")
            # the pixel probabilities are derived once from the image (ray_source.py:96-107): keep the pixels they
            # were derived from, frozen, so that the device tables always describe this copy
            val = val.copy()
            val._data.flags.writeable = False
        elif key == "front":
            ok = isinstance(val, (Point, Line)) or (isinstance(val, Surface) and val.is_flat())
            if not ok or isinstance(val, SlitSurface):
                raise ValueError("Currently only RectangularSurface, CircularSurface, Point, Line and RingSurface"
                                 " are supported for RaySources.")
        super().__setattr__(key, val)
