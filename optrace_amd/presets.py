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
    origin = np.zeros(3) if pos is None else np.array(pos, dtype=np.float64)
    A = adaptation
    gap_aqueous, gap_lens, cornea_thickness = 2.97 - 0.04 * A, 3.767 + 0.04 * A, 0.55

    def at(z: float) -> np.ndarray:
        return origin + [0, 0, z]

    # media: name -> (n at the centre line, Abbe number)
    table = dict(Cornea=(1.377, 57.1), Aqueous=(1.337, 61.3), Vitreous=(1.336, 61.1),
                 Lens=(1.42 + 0.00256 * A - 0.00022 * A ** 2, 51.9))
    n = {name: RefractionIndex("Abbe", n=nc, V=V, desc=f"n_{name}") for name, (nc, V) in table.items()}

    eye = Group(desc="Eye", long_desc="Arizona Eye Model")
    cornea = Lens(ConicSurface(r=5.45, R=7.8, k=-0.25, long_desc="Cornea Anterior"),
                  ConicSurface(r=5.45, R=6.5, k=-0.25, long_desc="Cornea Posterior"),
                  d1=0, d2=cornea_thickness, pos=at(0), n=n["Cornea"], n2=n["Aqueous"], desc="Cornea")
    eye.add(cornea)
    eye.add(Aperture(RingSurface(r=5.45, ri=pupil / 2, desc="Pupil"),
                     pos=at(cornea.back.pos[2] + gap_aqueous - 1e-9), desc="Pupil"))
    eye.add(Lens(ConicSurface(r=5.1, R=12 - 0.4 * A, k=-7.518749 + 1.285720 * A, long_desc="Lens Anterior"),
                 ConicSurface(r=5.1, R=-5.224557 + 0.2 * A, k=-1.353971 - 0.431762 * A, long_desc="Lens Posterior"),
                 d1=0, d2=gap_lens, pos=at(gap_aqueous + cornea_thickness), n=n["Lens"], n2=n["Vitreous"],
                 desc="Lens"))
    eye.add(Detector(SphericalSurface(r=r_det, R=-13.4, desc="Retina"), pos=at(24), desc="Retina"))
    return eye


geometry = types.SimpleNamespace(arizona_eye=arizona_eye)
