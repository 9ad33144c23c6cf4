"""Scene geometry: surfaces, elements, sources (host-side mirror of optrace/tracer/geometry)."""
from .surfaces import (Surface, CircularSurface, RingSurface, RectangularSurface, SlitSurface, ConicSurface,
                       SphericalSurface, AsphericSurface, Point, Line)
from .data_surfaces import TiltedSurface, DataSurface1D, DataSurface2D, FunctionSurface1D, FunctionSurface2D
from .elements import Element, Lens, IdealLens, Aperture, Filter, Detector, Group
from .ray_source import RaySource
