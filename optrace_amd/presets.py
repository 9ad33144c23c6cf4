"""Presets needed by the benchmark / example scenes: spectral lines, standard illuminants, eye model.

Data restated from optrace/tracer/presets/{spectral_lines,light_spectrum,geometry}.py (numbers from the
cited publications), not an exhaustive copy of the reference's preset catalogue.
"""
from __future__ import annotations

import types

import numpy as np

from .geometry import Group, Lens, Aperture, Detector, ConicSurface, RingSurface, SphericalSurface
from .refraction_index import RefractionIndex
from .spectrum import LightSpectrum, illuminant

# Fraunhofer lines [nm] (presets/spectral_lines.py)
spectral_lines = types.SimpleNamespace(
    h=404.6561, g=435.8343, F_=479.9914, F=486.1327, e=546.0740, d=587.5618, D=589.2938, C_=643.8469,
    C=656.272, r=706.5188, A_=768.2)
spectral_lines.FDC = [spectral_lines.F, spectral_lines.D, spectral_lines.C]
spectral_lines.FdC = [spectral_lines.F, spectral_lines.d, spectral_lines.C]
spectral_lines.FeC = [spectral_lines.F, spectral_lines.e, spectral_lines.C]
spectral_lines.F_eC_ = [spectral_lines.F_, spectral_lines.e, spectral_lines.C_]
spectral_lines.rgb = [464.3118, 549.1321, 611.2826]

light_spectrum = types.SimpleNamespace(**{
    name.lower().replace("-", "_"): LightSpectrum("Function", func=illuminant(name), desc=name,
                                                   long_desc=f"Illuminant {name}")
    for name in ["A", "C", "D50", "D55", "D65", "D75", "F2", "F7", "F11", "LED-B1", "LED-B2", "LED-B3",
                 "LED-B4", "LED-B5", "LED-BH1", "LED-RGB1", "LED-V1", "LED-V2"]})
light_spectrum.FDC = LightSpectrum("Lines", lines=spectral_lines.FDC, line_vals=[1, 1, 1], desc="FDC")
light_spectrum.FdC = LightSpectrum("Lines", lines=spectral_lines.FdC, line_vals=[1, 1, 1], desc="FdC")


def arizona_eye(adaptation: float = 0., pupil: float = 5.7, r_det: float = 8, pos: list = None) -> Group:
    """Arizona eye model (Schwiegerling, Field Guide to Visual and Ophthalmic Optics, SPIE 2004), as in
    presets/geometry.py:54-108: cornea, pupil, lens with accommodation-dependent conics, spherical retina."""
    pos0 = np.array(pos if pos is not None else [0, 0, 0])
    geom = Group(long_desc="Arizona Eye Model", desc="Eye")
    A = adaptation
    d_Aq = 2.97 - 0.04 * A
    d_Lens = 3.767 + 0.04 * A

    n_Cornea = RefractionIndex("Abbe", n=1.377, V=57.1, desc="n_Cornea")
    n_Aqueous = RefractionIndex("Abbe", n=1.337, V=61.3, desc="n_Aqueous")
    n_Lens = RefractionIndex("Abbe", n=1.42 + 0.00256 * A - 0.00022 * A ** 2, V=51.9, desc="n_Lens")
    n_Vitreous = RefractionIndex("Abbe", n=1.336, V=61.1, desc="n_Vitreous")

    front = ConicSurface(r=5.45, R=7.8, k=-0.25, long_desc="Cornea Anterior")
    back = ConicSurface(r=5.45, R=6.5, k=-0.25, long_desc="Cornea Posterior")
    L0 = Lens(front, back, d1=0, d2=0.55, pos=pos0 + [0, 0, 0], n=n_Cornea, n2=n_Aqueous, desc="Cornea")
    geom.add(L0)

    ap = RingSurface(r=5.45, ri=pupil / 2, desc="Pupil")
    geom.add(Aperture(ap, pos=pos0 + [0, 0, L0.back.pos[2] + d_Aq - 1e-9], desc="Pupil"))

    front = ConicSurface(r=5.1, R=12 - 0.4 * A, k=-7.518749 + 1.285720 * A, long_desc="Lens Anterior")
    back = ConicSurface(r=5.1, R=-5.224557 + 0.2 * A, k=-1.353971 - 0.431762 * A, long_desc="Lens Posterior")
    geom.add(Lens(front, back, d1=0, d2=d_Lens, pos=pos0 + [0, 0, d_Aq + 0.55], n=n_Lens, n2=n_Vitreous,
                  desc="Lens"))

    geom.add(Detector(SphericalSurface(r=r_det, R=-13.4, desc="Retina"), pos=pos0 + [0, 0, 24], desc="Retina"))
    return geom


geometry = types.SimpleNamespace(arizona_eye=arizona_eye)
