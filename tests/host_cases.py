"""Host-object bookkeeping cases shared by the fixture generator (run on the reference) and tests/test_host_golden.py
(run on optrace_amd): surfaces, lenses, groups, sources and spectra are built, moved, flipped and rotated through the
public API both packages share, and their numeric state is recorded.  `ot` is the package under test."""
import numpy as np


def _num(v):
    return np.nan if v is None else float(v)


def surface_state(s) -> np.ndarray:
    ext = [float(v) for v in s.extent]
    parax = _num(getattr(s, "parax_roc", None))
    out = [float(s.z_min), float(s.z_max), *ext, parax, *[float(v) for v in s.pos]]
    for k in ("ds", "dn", "dp"):
        out.append(float(getattr(s, k)))
    return np.array(out)


def element_state(el) -> np.ndarray:
    out = [*[float(v) for v in el.extent], *[float(v) for v in el.pos]]
    out += [*[float(v) for v in el.front.pos]]
    if el.has_back():
        out += [*[float(v) for v in el.back.pos], float(el.d1), float(el.d2), float(el.d), float(el.de)]
    return np.array(out)


def _cos_surface(x, y):
    return 0.05 * np.cos(x) * (1 + 0.1 * y)


def _quad(r):
    return 0.5 + r ** 2 / 40


def surfaces(ot) -> dict:
    r = np.linspace(0, 3.0, 220)
    xy = np.linspace(-2.5, 2.5, 210)
    X, Y = np.meshgrid(xy, xy)
    return {
        "circle": lambda: ot.CircularSurface(r=2.5),
        "ring": lambda: ot.RingSurface(r=3, ri=0.7),
        "rect": lambda: ot.RectangularSurface(dim=[3, 1.5]),
        "slit": lambda: ot.SlitSurface(dim=[4, 3], dimi=[0.2, 1.1]),
        "conic": lambda: ot.ConicSurface(r=3, R=-12, k=-1.8),
        "conic_pos": lambda: ot.ConicSurface(r=2, R=4.5, k=0.6),
        "sphere": lambda: ot.SphericalSurface(r=2.2, R=5),
        "asphere": lambda: ot.AsphericSurface(r=2.5, R=-9, k=0.4, coeff=[1e-3, -2e-4, 1e-5]),
        "tilted": lambda: ot.TiltedSurface(r=2, normal=[0.3, -0.2, 1]),
        "tilted_sph": lambda: ot.TiltedSurface(r=1.5, normal_sph=[25, 130]),
        "data1d": lambda: ot.DataSurface1D(r=3.0, data=20 - np.sqrt(400 - r ** 2), parax_roc=20.),
        "data2d": lambda: ot.DataSurface2D(r=2.5, data=X ** 2 / 30 - Y ** 2 / 50 + 0.02 * X, parax_roc=None),
        "func1d": lambda: ot.FunctionSurface1D(r=2.5, func=_quad, z_min=0.5, z_max=0.5 + 2.5 ** 2 / 40, parax_roc=20.),
        "func2d": lambda: ot.FunctionSurface2D(r=2, func=_cos_surface, z_min=-0.06, z_max=0.06),
    }


def surface_cases(ot) -> dict:
    out = {}
    with ot.global_options.no_warnings():
        for name, make in surfaces(ot).items():
            s = make()
            out[f"surf/{name}/new"] = surface_state(s)
            s.move_to([1.5, -2.0, 7.25])
            out[f"surf/{name}/moved"] = surface_state(s)
            s.flip()
            out[f"surf/{name}/flipped"] = surface_state(s)
            s.rotate(33.0)
            out[f"surf/{name}/rotated"] = surface_state(s)
            s.move_to([-0.5, 0.25, -3.0])
            s.flip()
            out[f"surf/{name}/again"] = surface_state(s)
            if hasattr(s, "normal"):
                out[f"surf/{name}/normal"] = np.array(s.normal, dtype=np.float64)
        for name, sh in (("point", ot.Point()), ("line", ot.Line(r=1.5, angle=20.))):
            sh.move_to([1, 2, 3])
            sh.rotate(15)
            sh.flip()
            out[f"shape/{name}"] = np.array([*sh.extent, *sh.pos], dtype=np.float64)
    return out


def element_cases(ot) -> dict:
    out = {}
    n = ot.RefractionIndex("Constant", n=1.5)
    mk = surfaces(ot)
    with ot.global_options.no_warnings():
        lenses = {
            "biconvex_de": lambda: ot.Lens(mk["sphere"](), ot.SphericalSurface(r=2.2, R=-7), de=0.3, pos=[0, 0, 10], n=n),
            "meniscus_d": lambda: ot.Lens(mk["conic_pos"](), ot.ConicSurface(r=2, R=6, k=-0.5), d=1.1, pos=[1, 0, 5], n=n),
            "d1_d2": lambda: ot.Lens(mk["circle"](), mk["conic"](), d1=0.4, d2=0.9, pos=[0, -1, 2], n=n),
            "asphere_data": lambda: ot.Lens(mk["asphere"](), mk["data1d"](), de=0.25, pos=[0, 0, -4], n=n),
            "tilted_pair": lambda: ot.Lens(mk["tilted"](), ot.TiltedSurface(r=2, normal=[-0.1, 0.25, 1]), de=0.5, pos=[0, 0, 0], n=n),
        }
        for name, make in lenses.items():
            L = make()
            out[f"lens/{name}/new"] = element_state(L)
            L.move_to([0.5, 0.75, 21.0])
            out[f"lens/{name}/moved"] = element_state(L)
            L.flip()
            out[f"lens/{name}/flipped"] = element_state(L)
            out[f"lens/{name}/flipped_front"] = surface_state(L.front)
            out[f"lens/{name}/flipped_back"] = surface_state(L.back)
            L.rotate(-40.0)
            out[f"lens/{name}/rotated"] = element_state(L)
        singles = {
            "aperture": lambda: ot.Aperture(mk["ring"](), pos=[0, 1, 3]),
            "filter": lambda: ot.Filter(mk["rect"](), pos=[1, 1, 4], spectrum=ot.TransmissionSpectrum("Constant", val=0.5)),
            "detector": lambda: ot.Detector(mk["sphere"](), pos=[0, 0, 30]),
            "ideal": lambda: ot.IdealLens(r=3, D=25, pos=[0, 0, 9]),
            "source": lambda: ot.RaySource(mk["rect"](), pos=[0.5, 0, -5], s=[0.1, 0.2, 1]),
        }
        for name, make in singles.items():
            el = make()
            el.move_to([2, -1, 6.5])
            el.flip()
            el.rotate(10)
            out[f"single/{name}"] = element_state(el)
        src = singles["source"]()
        out["single/source_s"] = np.array(src.s, dtype=np.float64)
        src2 = ot.RaySource(ot.Point(), pos=[0, 0, 0], s_sph=[20, 300])
        out["single/source_s_sph"] = np.array(src2.s, dtype=np.float64)

        # a group of everything: move, flip about an axis, rotate about a point
        G = ot.Group([lenses["biconvex_de"](), lenses["d1_d2"](), singles["aperture"](), singles["detector"]()])
        G.add(singles["filter"]())
        out["group/new"] = np.array([*G.extent, *G.pos], dtype=np.float64)
        G.move_to([1.0, 2.0, -3.0])
        out["group/moved"] = np.concatenate([element_state(e) for e in G.elements])
        G.flip(y0=0.5, z0=4.0)
        out["group/flipped"] = np.concatenate([element_state(e) for e in G.elements])
        G.flip()
        out["group/flipped_default"] = np.concatenate([element_state(e) for e in G.elements])
        G.rotate(25.0, x0=0.3, y0=-0.4)
        out["group/rotated"] = np.concatenate([element_state(e) for e in G.elements])
        out["group/extent"] = np.array([*G.extent, *G.pos], dtype=np.float64)
        out["group/n_tracing_surfaces"] = np.array([len(G.tracing_surfaces)], dtype=np.float64)
        removed = G.remove(G.lenses[0])
        out["group/after_remove"] = np.array([float(removed), len(G.elements), *G.extent], dtype=np.float64)

        eye = ot.presets.geometry.arizona_eye(adaptation=1.2, pupil=3.5, pos=[0.5, -0.25, 2])
        parts = list(eye.lenses) + list(eye.apertures) + list(eye.detectors)  # (the reference adds a plot-only volume)
        out["eye/elements"] = np.concatenate([element_state(e) for e in parts])
        out["eye/surfaces"] = np.concatenate([surface_state(s) for s in eye.tracing_surfaces])
    return out


def spectrum_cases(ot) -> dict:
    out = {}
    wl = np.linspace(381., 779., 173)
    wls = np.linspace(400., 700., 61)
    with ot.global_options.no_warnings():
        light = {
            "constant": ot.LightSpectrum("Constant", val=0.7),
            "rect": ot.LightSpectrum("Rectangle", wl0=450., wl1=620., val=2.0),
            "gauss": ot.LightSpectrum("Gaussian", mu=560., sig=35., val=1.5),
            "data": ot.LightSpectrum("Data", wls=wls, vals=1 + 0.5 * np.sin(wls / 40)),
            "blackbody": ot.LightSpectrum("Blackbody", T=4200., val=0.9),
            "func": ot.LightSpectrum("Function", func=lambda x, a: a + 0.001 * (x - 380), func_args=dict(a=0.2)),
            "d65": ot.presets.light_spectrum.d65,
            "led": ot.presets.light_spectrum.led_b3,
        }
        for name, sp in light.items():
            out[f"light/{name}"] = np.asarray(sp(wl), dtype=np.float64)
        trans = {
            "gauss_inv": ot.TransmissionSpectrum("Gaussian", mu=500., sig=40., val=0.8, inverse=True),
            "rect": ot.TransmissionSpectrum("Rectangle", wl0=500., wl1=600., val=0.6),
            "data": ot.TransmissionSpectrum("Data", wls=wls, vals=0.5 + 0.4 * np.cos(wls / 50)),
        }
        for name, sp in trans.items():
            out[f"trans/{name}"] = np.asarray(sp(wl), dtype=np.float64)
        out["light/desc_lengths"] = np.array([len(light["constant"].get_desc()), len(light["gauss"].get_desc())], dtype=np.float64)
    return out


def all_cases(ot) -> dict:
    return {**surface_cases(ot), **element_cases(ot), **spectrum_cases(ot)}
