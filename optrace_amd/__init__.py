"""optrace_amd -- MI355X-native sequential ray-tracing core with optrace's Python API.

The names exported here are the ones optrace/tracer/__init__.py:3-63 exports for the tracing hot path;
scene scripts written against `import optrace as ot` run with `import optrace_amd as ot`.
The per-ray work happens in hand-written HIP kernels (optrace_amd/csrc) behind the C-ABI of
include/optrace_amd.h; there is no CPU fallback.
"""
from .options import global_options
from ._warn import OptraceWarning
from ._capi import BackendError

from .refraction_index import RefractionIndex
from .spectrum import Spectrum, LightSpectrum, TransmissionSpectrum
from .geometry import (Surface, CircularSurface, RingSurface, RectangularSurface, SlitSurface, ConicSurface,
                       SphericalSurface, AsphericSurface, Point, Line, Element, Lens, IdealLens, Aperture,
                       Filter, Detector, Group, RaySource, TiltedSurface, DataSurface1D, DataSurface2D,
                       FunctionSurface1D, FunctionSurface2D)
from .image import RGBImage, GrayscaleImage, ScalarImage
from .render_image import RenderImage
from .ray_storage import RayStorage
from .raytracer import Raytracer
from .convolve import convolve
from . import presets, misc

__version__ = "0.1.0"
