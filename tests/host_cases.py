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
        for name in ("a", "c", "d50", "d55", "d65", "d75", "f2", "f7", "f11", "led_b1", "led_b2", "led_b3", "led_b4", "led_b5",
                     "led_bh1", "led_rgb1", "led_v1", "led_v2"):
            out[f"light/preset_{name}"] = np.asarray(getattr(ot.presets.light_spectrum, name)(wl), dtype=np.float64)
        sl = ot.presets.spectral_lines
        out["light/spectral_lines"] = np.array([sl.h, sl.g, sl.F_, sl.F, sl.e, sl.d, sl.D, sl.C_, sl.C, sl.r, sl.A_,
                                                *sl.FDC, *sl.FdC, *sl.FeC, *sl.F_eC_, *sl.rgb], dtype=np.float64)
        # figures of spectra (light_spectrum.py:232-400): power, luminous power, peak, peak / centroid wavelength, width
        hist = ot.LightSpectrum("Histogram")
        hist._wls = np.linspace(450., 650., 41)
        hist._vals = np.exp(-((hist._wls[:-1] + 2.5 - 560.) / 30.) ** 2)
        figures = dict(light, mono=ot.LightSpectrum("Monochromatic", wl=532., val=1.5), hist=hist,
                       lines=ot.LightSpectrum("Lines", lines=[450., 550., 650.], line_vals=[1., 3., 2.]))
        for name, sp in figures.items():
            out[f"light/figures_{name}"] = np.array([sp.power(), sp.luminous_power(), sp.peak(), sp.peak_wavelength(),
                                                     sp.centroid_wavelength(), sp.fwhm()], dtype=np.float64)
        out["light/desc_lengths"] = np.array([len(light["constant"].get_desc()), len(light["gauss"].get_desc())], dtype=np.float64)
    return out




def _checks(RT):
    f = getattr(RT, "_geometry_checks", None) or getattr(RT, "_Raytracer__geometry_checks")
    f()
    return float(bool(RT.geometry_error))


def geometry_cases(ot) -> dict:
    """Raytracer geometry checks (raytracer.py:510-580) and Raytracer.check_collision (:581-664) on legal and illegal
    arrangements: the flag, and for collisions their number and lowest / highest point."""
    out = {}
    n = ot.RefractionIndex("Constant", n=1.5)

    def tracer(**kw):
        return ot.Raytracer(outline=kw.pop("outline", [-5, 5, -5, 5, -10, 40]), **kw)

    def src(RT, z=-5., **kw):
        RT.add(ot.RaySource(ot.CircularSurface(r=1), pos=[0, 0, z], **kw))

    def lens(z, r=3, R1=10, R2=-10, **kw):
        return ot.Lens(ot.SphericalSurface(r=r, R=R1), ot.SphericalSurface(r=r, R=R2), pos=[0, 0, z], n=n,
                       **(kw or dict(de=0.2)))

    with ot.global_options.no_warnings():
        RT = tracer(); src(RT); RT.add(lens(5)); RT.add(lens(15))
        out["geom/ok"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(lens(5)); RT.add(lens(5.6))          # second lens starts inside the first
        out["geom/lenses_collide"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(lens(5, r=3)); RT.add(lens(5.9, r=3, R1=-10, R2=-12))   # nested menisci, close
        out["geom/close_menisci"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(ot.Lens(ot.SphericalSurface(r=3, R=10), ot.SphericalSurface(r=3, R=-10), de=0.2,
                                                pos=[3.5, 0, 5], n=n))      # sticks out of the outline in x
        out["geom/outside_outline"] = np.array([_checks(RT)])
        RT = tracer(); src(RT, z=6.); RT.add(lens(5))                        # source behind the lens front
        out["geom/source_in_lens"] = np.array([_checks(RT)])
        RT = tracer(); src(RT, z=20.); RT.add(lens(5))                       # source behind every lens
        out["geom/source_behind"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(lens(5)); RT.add(ot.Aperture(ot.RingSurface(r=3, ri=1), pos=[0, 0, 5.3]))
        out["geom/aperture_in_lens"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(lens(5)); RT.add(ot.Aperture(ot.RingSurface(r=3, ri=1), pos=[0, 0, 9.]))
        out["geom/aperture_ok"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(lens(5)); RT.add(ot.Detector(ot.RectangularSurface(dim=[30, 30]), pos=[0, 0, 50]))
        out["geom/detector_outside_is_fine"] = np.array([_checks(RT)])
        RT = tracer(); RT.add(lens(5))
        out["geom/no_source"] = np.array([_checks(RT)])
        RT = tracer(); src(RT); RT.add(lens(5)); RT.add(ot.Filter(ot.CircularSurface(r=2), pos=[0, 0, 45.],
                                                                  spectrum=ot.TransmissionSpectrum("Constant", val=1.)))
        out["geom/filter_outside"] = np.array([_checks(RT)])

        pairs = {
            "spheres_apart": (ot.SphericalSurface(r=3, R=10), [0, 0, 0], ot.SphericalSurface(r=3, R=-10), [0, 0, 2]),
            "spheres_cross": (ot.SphericalSurface(r=3, R=-10), [0, 0, 0], ot.SphericalSurface(r=3, R=10), [0, 0, 0.5]),
            "shifted": (ot.ConicSurface(r=3, R=8, k=-1), [0, 0, 0], ot.CircularSurface(r=2), [1.5, 1, 0.3]),
            "tilted_flat": (ot.TiltedSurface(r=2, normal=[0.3, 0, 1]), [0, 0, 0], ot.CircularSurface(r=2), [0, 0, 0.2]),
            "rect_ring": (ot.RectangularSurface(dim=[2, 2]), [0, 0, 1.], ot.RingSurface(r=3, ri=0.5), [0, 0, 0.9]),
            "point_front": (ot.Point(), [0.5, 0, 1.], ot.SphericalSurface(r=3, R=10), [0, 0, 0.5]),
            "line_back": (ot.SphericalSurface(r=3, R=-10), [0, 0, 0.2], ot.Line(r=2, angle=30), [0, 0, 0.0]),
        }
        for name, (a, pa, b, pb) in pairs.items():
            a.move_to(pa)
            b.move_to(pb)
            hit, x, y, z = ot.Raytracer.check_collision(a, b)
            x, y, z = np.asarray(x), np.asarray(y), np.asarray(z)
            stats = [float(bool(hit)), float(x.shape[0])]
            stats += [float(z.min()), float(z.max()), float(x.min()), float(x.max())] if x.shape[0] else [0.] * 4
            out[f"collision/{name}"] = np.array(stats)
    return out


def all_cases(ot) -> dict:
    return {**surface_cases(ot), **element_cases(ot), **spectrum_cases(ot), **geometry_cases(ot), **error_cases(ot),
            **container_cases(ot), **snapshot_cases(ot), **constant_cases(ot)}


def error_cases(ot) -> dict:
    """What invalid arguments raise: the exception class name (or "ok") per case, as a string array."""
    n = ot.RefractionIndex("Constant", n=1.5)
    sph = lambda: ot.SphericalSurface(r=2, R=10)  # noqa: E731
    wls = np.linspace(400., 700., 31)
    cases = {
        "circle_r_neg": lambda: ot.CircularSurface(r=-1),
        "circle_r_str": lambda: ot.CircularSurface(r="1"),
        "circle_r_zero": lambda: ot.CircularSurface(r=0),
        "ring_ri_large": lambda: ot.RingSurface(r=1, ri=1.5),
        "ring_ri_zero": lambda: ot.RingSurface(r=1, ri=0),
        "rect_dim_neg": lambda: ot.RectangularSurface(dim=[1, -1]),
        "rect_dim_len": lambda: ot.RectangularSurface(dim=[1, 2, 3]),
        "slit_inner_large": lambda: ot.SlitSurface(dim=[1, 1], dimi=[2, 0.5]),
        "conic_R_zero": lambda: ot.ConicSurface(r=1, R=0, k=0),
        "conic_R_inf": lambda: ot.ConicSurface(r=1, R=np.inf, k=0),
        "conic_r_too_large": lambda: ot.ConicSurface(r=5, R=4, k=0.5),
        "sphere_r_gt_R": lambda: ot.SphericalSurface(r=3, R=2),
        "asphere_no_coeff": lambda: ot.AsphericSurface(r=1, R=10, k=0, coeff=[]),
        "asphere_coeff_type": lambda: ot.AsphericSurface(r=1, R=10, k=0, coeff=3),
        "tilted_none": lambda: ot.TiltedSurface(r=1),
        "tilted_nz_neg": lambda: ot.TiltedSurface(r=1, normal=[0, 1, -0.5]),
        "data2d_small": lambda: ot.DataSurface2D(r=1, data=np.zeros((10, 10))),
        "data2d_not_square": lambda: ot.DataSurface2D(r=1, data=np.zeros((60, 80))),
        "data1d_2d": lambda: ot.DataSurface1D(r=1, data=np.zeros((60, 60))),
        "func_not_callable": lambda: ot.FunctionSurface2D(r=1, func=2),
        "func_zmin_only": lambda: ot.FunctionSurface1D(r=1, func=lambda r: r * 0.1, z_min=0.),
        "line_r_neg": lambda: ot.Line(r=-1),
        "surface_move_bad": lambda: sph().move_to([1, 2]),
        "surface_locked": lambda: setattr(sph(), "r", 3),
        "lens_no_thickness": lambda: ot.Lens(sph(), sph(), pos=[0, 0, 0], n=n),
        "lens_d_and_de": lambda: ot.Lens(sph(), sph(), d=1, de=1, pos=[0, 0, 0], n=n),
        "lens_n_type": lambda: ot.Lens(sph(), sph(), de=1, pos=[0, 0, 0], n=1.5),
        "lens_pos_len": lambda: ot.Lens(sph(), sph(), de=1, pos=[0, 0], n=n),
        "lens_d1_neg": lambda: ot.Lens(sph(), sph(), d1=-1, d2=1, pos=[0, 0, 0], n=n),
        "filter_spectrum_type": lambda: ot.Filter(ot.CircularSurface(r=1), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Constant")),
        "detector_asphere": lambda: ot.Detector(ot.AsphericSurface(r=1, R=10, k=0, coeff=[1e-3]), pos=[0, 0, 0]),
        "ideal_D_zero": lambda: ot.IdealLens(r=1, D=0, pos=[0, 0, 0]),
        "ideal_D_str": lambda: ot.IdealLens(r=1, D="5", pos=[0, 0, 0]),
        "source_div_name": lambda: ot.RaySource(ot.Point(), divergence="Cone"),
        "source_power_neg": lambda: ot.RaySource(ot.Point(), power=-1),
        "source_div_angle_zero": lambda: ot.RaySource(ot.Point(), div_angle=0),
        "source_s_back": lambda: ot.RaySource(ot.Point(), s=[0, 0, -1]),
        "source_s_len": lambda: ot.RaySource(ot.Point(), s=[0, 1]),
        "source_surface_sphere": lambda: ot.RaySource(sph()),
        "source_slit": lambda: ot.RaySource(ot.SlitSurface(dim=[1, 1], dimi=[0.1, 0.1])),
        "source_spectrum_type": lambda: ot.RaySource(ot.Point(), spectrum=ot.TransmissionSpectrum("Constant", val=1.)),
        "source_pol_name": lambda: ot.RaySource(ot.Point(), polarization="z"),
        "source_or_func": lambda: ot.RaySource(ot.Point(), orientation="Function", or_func=3),
        "source_black_image": lambda: ot.RaySource(ot.RGBImage(np.zeros((10, 10, 3)), [1, 1])),
        "spec_type": lambda: ot.LightSpectrum("Sawtooth"),
        "spec_wl_range": lambda: ot.LightSpectrum("Monochromatic", wl=100.),
        "spec_val_neg": lambda: ot.LightSpectrum("Constant", val=-1.),
        "spec_sig_zero": lambda: ot.LightSpectrum("Gaussian", sig=0.),
        "spec_lines_dup": lambda: ot.LightSpectrum("Lines", lines=[500., 500.], line_vals=[1, 1]),
        "spec_lines_empty": lambda: ot.LightSpectrum("Lines", lines=[], line_vals=[]),
        "spec_line_vals_neg": lambda: ot.LightSpectrum("Lines", lines=[500., 600.], line_vals=[1, -1]),
        "spec_wls_uneven": lambda: ot.LightSpectrum("Data", wls=[400., 500., 700.], vals=[1, 1, 1]),
        "spec_vals_neg": lambda: ot.LightSpectrum("Data", wls=wls, vals=-np.ones(31)),
        "spec_func_neg": lambda: ot.LightSpectrum("Function", func=lambda x: -x),
        "spec_T_zero": lambda: ot.LightSpectrum("Blackbody", T=0.),
        "spec_call_lines": lambda: ot.LightSpectrum("Lines", lines=[500., 600.], line_vals=[1, 1])(np.array([500.])),
        "trans_val_gt1": lambda: ot.TransmissionSpectrum("Constant", val=1.5),
        "trans_vals_gt1": lambda: ot.TransmissionSpectrum("Data", wls=wls, vals=2 * np.ones(31)),
        "trans_type": lambda: ot.TransmissionSpectrum("Lines"),
        "trans_inverse_type": lambda: ot.TransmissionSpectrum("Constant", val=0.5, inverse=1),
        "index_n_below1": lambda: ot.RefractionIndex("Constant", n=0.9),
        "index_type": lambda: ot.RefractionIndex("Glass"),
        "index_coeff_count": lambda: ot.RefractionIndex("Cauchy", coeff=[1.5, 0.01]),
        "index_coeff_type": lambda: ot.RefractionIndex("Cauchy", coeff=(1.5, 0.01, 0, 0)),
        "index_V_neg": lambda: ot.RefractionIndex("Abbe", n=1.5, V=-10),
        "index_lines_order": lambda: ot.RefractionIndex("Abbe", n=1.5, V=50, lines=[600., 500., 700.]),
        "index_lines_count": lambda: ot.RefractionIndex("Abbe", n=1.5, V=50, lines=[500., 600.]),
        "index_vals_below1": lambda: ot.RefractionIndex("Data", wls=wls, vals=0.5 * np.ones(31)),
        "index_func_below1": lambda: ot.RefractionIndex("Function", func=lambda x: 0.9 + 0 * x),
        "tracer_outline_order": lambda: ot.Raytracer(outline=[1, -1, -1, 1, 0, 10]),
        "tracer_outline_len": lambda: ot.Raytracer(outline=[-1, 1, -1, 1]),
        "tracer_n0_type": lambda: ot.Raytracer(outline=[-1, 1, -1, 1, 0, 10], n0=1.0),
        "tracer_add_bad": lambda: ot.Raytracer(outline=[-1, 1, -1, 1, 0, 10]).add(5),
        "group_add_surface": lambda: ot.Group().add(sph()),
        "image_range": lambda: ot.RGBImage(2 * np.ones((5, 5, 3)), [1, 1]),
        "image_no_size": lambda: ot.RGBImage(np.ones((5, 5, 3))),
        "image_shape": lambda: ot.RGBImage(np.ones((5, 5)), [1, 1]),
        "gray_shape": lambda: ot.GrayscaleImage(np.ones((5, 5, 3)), [1, 1]),
        "image_s_neg": lambda: ot.GrayscaleImage(np.ones((5, 5)), [1, -1]),
        "render_extent": lambda: ot.RenderImage([1, 0, 0, 1]),
        "options_bool": lambda: setattr(ot.global_options, "show_warnings", 1),
        "options_range": lambda: setattr(ot.global_options, "wavelength_range", [400., 700.]),
    }
    def emit(**kw):
        """What happens when a source with these options has to produce rays: RaySource.create_rays in the reference, the
        host half of it (the source descriptor, where the same checks live) in optrace_amd."""
        rs = ot.RaySource(kw.pop("surface", None) or ot.CircularSurface(r=1), pos=[0, 0, 0], **kw)
        return rs._source_fields() if hasattr(rs, "_source_fields") else rs.create_rays(2000)

    flat = np.linspace(400., 700., 31)
    cases.update({
        "emit_ok": lambda: emit(),
        "emit_div_func_missing": lambda: emit(divergence="Function"),
        "emit_div_func_negative": lambda: emit(divergence="Function", div_func=lambda e: -1 + 0 * e),
        "emit_div_func_zero": lambda: emit(divergence="Function", div_func=lambda e: 0 * e),
        "emit_or_func_missing": lambda: emit(orientation="Function"),
        "emit_pol_func_missing": lambda: emit(polarization="Function"),
        "emit_pol_list_missing": lambda: emit(polarization="List"),
        "emit_pol_list_probs_missing": lambda: emit(polarization="List", pol_angles=[0., 45.]),
        "emit_pol_list_lengths": lambda: emit(polarization="List", pol_angles=[0., 45.], pol_probs=[1.]),
        "emit_pol_list_zero": lambda: emit(polarization="List", pol_angles=[0., 45.], pol_probs=[0., 0.]),
        "emit_pol_list_ok": lambda: emit(polarization="List", pol_angles=[0., 45.], pol_probs=[1., 3.]),
        "emit_lines_zero": lambda: emit(spectrum=ot.LightSpectrum("Lines", lines=[500., 600.], line_vals=[0., 0.])),
        "emit_data_zero": lambda: emit(spectrum=ot.LightSpectrum("Data", wls=flat, vals=np.zeros(31))),
        "emit_lambertian_2d": lambda: emit(divergence="Lambertian", div_angle=30., div_2d=True, div_axis_angle=10.),
        "emit_image": lambda: emit(surface=ot.RGBImage(np.full((4, 4, 3), 0.5), [1, 1])),
    })
    def rt(sources=True, detector=True, lens=True):
        RT = ot.Raytracer(outline=[-5, 5, -5, 5, -10, 40])
        if sources:
            RT.add(ot.RaySource(ot.CircularSurface(r=1), pos=[0, 0, -5]))
        if lens:
            RT.add(ot.Lens(sph(), ot.SphericalSurface(r=2, R=-10), de=0.2, pos=[0, 0, 5], n=n))
        if detector:
            RT.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 30]))
        return RT

    cases.update({   # a tracer that has not traced yet: every refusal comes before any ray exists
        "rt_trace_no_source": lambda: rt(sources=False).trace(1000),
        "rt_trace_N_zero": lambda: rt().trace(0),
        "rt_trace_N_float": lambda: rt().trace(1000.5),
        "rt_image_untraced": lambda: rt().detector_image(),
        "rt_image_no_detector": lambda: rt(detector=False).detector_image(),
        "rt_spectrum_untraced": lambda: rt().detector_spectrum(),
        "rt_source_image_untraced": lambda: rt().source_image(),
        "rt_source_spectrum_no_source": lambda: rt(sources=False).source_spectrum(),
        "rt_focus_untraced": lambda: rt().focus_search("RMS Spot Size", 10.),
        "rt_focus_method": lambda: rt().focus_search("Sharpest", 10.),
        "rt_focus_outside": lambda: rt().focus_search("RMS Spot Size", 100.),
        "rt_iter_N_zero": lambda: rt().iterative_render(0),
        "rt_iter_no_detector": lambda: rt(detector=False).iterative_render(10000),
        "rt_iter_no_source": lambda: rt(sources=False).iterative_render(10000),
        "rt_iter_index_list": lambda: rt().iterative_render(10000, detector_index=[0, 0]),
        "rt_iter_extent_len": lambda: rt().iterative_render(10000, pos=[[0, 0, 30], [0, 0, 32]], extent=[None]),
        "rt_remove_missing": lambda: rt().remove(ot.Detector(ot.CircularSurface(r=1), pos=[0, 0, 1])),
        "rt_current_untraced": lambda: rt().check_if_rays_are_current(),
    })
    names, res = [], []
    with ot.global_options.no_warnings():
        for name, f in cases.items():
            try:
                f()
                r = "ok"
            except Exception as e:  # noqa: BLE001 - the class is the result
                r = type(e).__name__
            names.append(name)
            res.append(r)
    return {"errors/names": np.array(names), "errors/raised": np.array(res)}


def _image_grid(ot, extent, limit):
    """Extent and pixel counts a RenderImage settles on for a given hit extent (render_image.py:224-255, 383-387)."""
    img = ot.RenderImage(extent)
    if hasattr(img, "_pixel_counts"):   # optrace_amd: the grid is host logic, the binning itself a device call
        img._limit = limit
        img._fix_extent()
        nx, ny = img._pixel_counts()
    else:                               # the reference: render an empty hit list
        img.render(np.zeros((0, 3)), np.zeros(0), np.zeros(0), limit=limit, _dont_filter=True)
        ny, nx = img._data.shape[:2]
    return np.array([*img.extent, nx, ny, *img.s], dtype=np.float64)


def container_cases(ot) -> dict:
    """Order and bookkeeping of Raytracer / Group containers, description strings, RenderImage grids."""
    out = {}
    n = ot.RefractionIndex("Constant", n=1.5)
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-6, 6, -6, 6, -10, 60])
        mk = lambda z, **kw: ot.Lens(ot.SphericalSurface(r=2, R=9), ot.SphericalSurface(r=2, R=-9), de=0.2, pos=[0, 0, z], n=n, **kw)  # noqa: E731
        L30, L10, L20 = mk(30., desc="far"), mk(10.), mk(20., long_desc="the middle lens")
        ap = ot.Aperture(ot.RingSurface(r=2, ri=0.5), pos=[0, 0, 15.])
        flt = ot.Filter(ot.CircularSurface(r=2), pos=[0, 0, 25.], spectrum=ot.TransmissionSpectrum("Constant", val=0.5))
        det2, det1 = ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 50.]), ot.Detector(ot.CircularSurface(r=3), pos=[0, 0, 40.])
        src = ot.RaySource(ot.CircularSurface(r=1), pos=[0, 0, -5.], desc="lamp")
        RT.add([L30, ap, det2, L10])
        RT.add(flt)
        RT.add(ot.Group([L20, det1, src]))
        out["cont/lens_z"] = np.array([L.pos[2] for L in RT.lenses], dtype=np.float64)
        out["cont/detector_z"] = np.array([d.pos[2] for d in RT.detectors], dtype=np.float64)
        out["cont/surface_z"] = np.array([s.pos[2] for s in RT.tracing_surfaces], dtype=np.float64)
        out["cont/counts"] = np.array([len(RT.lenses), len(RT.apertures), len(RT.filters), len(RT.detectors),
                                       len(RT.ray_sources), len(RT.elements)], dtype=np.float64)
        out["cont/extent"] = np.array(RT.extent, dtype=np.float64)
        out["cont/pos"] = np.array(RT.pos, dtype=np.float64)
        out["cont/has"] = np.array([RT.has(L10), RT.has(mk(1.))], dtype=np.float64)
        removed = [RT.remove(L10), RT.remove(L10), RT.remove([ap, flt])]
        out["cont/removed"] = np.array([float(bool(r)) for r in removed] + [len(RT.elements)], dtype=np.float64)
        RT.clear()
        out["cont/cleared"] = np.array([len(RT.elements)], dtype=np.float64)
        texts = [L30.get_desc(), L30.get_long_desc(), L20.get_desc("fallback"), L20.get_long_desc(), L10.get_long_desc("fb"),
                 src.get_desc(), ot.LightSpectrum("Constant", val=0.25).get_desc(), ot.LightSpectrum("Gaussian").get_desc(),
                 ot.presets.light_spectrum.d65.get_long_desc(), ot.RefractionIndex("Abbe", n=1.5, V=50, desc="nX").get_desc(),
                 ot.CircularSurface(r=1).info, ot.RectangularSurface(dim=[2, 1]).info, ot.TiltedSurface(r=1, normal=[0, 0.2, 1]).info,
                 ot.ConicSurface(r=1, R=5, k=-1).info, ot.SphericalSurface(r=1, R=5).info,
                 ot.AsphericSurface(r=1, R=5, k=0, coeff=[1e-3]).info]   # (not RingSurface.info: unformatted in the reference)
        out["cont/texts"] = np.array(texts)
        for name, (ext, limit) in {"point": ([0.5, 0.5, -1, -1], None), "line_x": ([-1, 1, 0.25, 0.25], None),
                                   "line_y": ([2, 2, -3, 1], None), "ratio_10": ([-1, 1, -0.1, 0.1], None),
                                   "ratio_2": ([-1, 1, -0.5, 0.5], None), "ratio_4_tall": ([0, 1, 0, 4], None),
                                   "square_limit": ([-1, 1, -1, 1], 5.0), "point_limit": ([0, 0, 0, 0], 2.0),
                                   "tiny": ([0, 1e-12, 0, 5e-13], None)}.items():
            out[f"cont/grid_{name}"] = _image_grid(ot, ext, limit)
        # cuts through images (base_image.py:149-186) and their pixel bookkeeping
        rng = np.random.default_rng(5)
        rgb = ot.RGBImage(rng.uniform(0, 1, (7, 11, 3)), extent=[-1.1, 3.3, 0.5, 2.6])
        gray = ot.GrayscaleImage(rng.uniform(0, 1, (9, 5)), [2.5, 4.5])
        for name, img in (("rgb", rgb), ("gray", gray)):
            x0, x1, y0, y1 = img.extent
            for k, (kw, tag) in enumerate([(dict(x=x0), "x_lo"), (dict(x=x1), "x_hi"), (dict(x=(x0 + x1) / 2 + 0.01), "x_mid"),
                                           (dict(y=y0), "y_lo"), (dict(y=y1), "y_hi"), (dict(y=y0 + 0.3 * (y1 - y0)), "y_in")]):
                edges, cuts = img.profile(**kw)
                out[f"cont/profile_{name}_{tag}"] = np.concatenate([np.asarray(edges, dtype=np.float64)] +
                                                                   [np.asarray(c, dtype=np.float64) for c in cuts])
            out[f"cont/image_{name}"] = np.array([*img.s, img.Apx, *img.shape[:2], *img.extent], dtype=np.float64)
        g2 = rgb.to_grayscale_image()
        out["cont/to_gray"] = np.concatenate((g2.data.ravel(), g2.extent))
        r2 = gray.to_rgb_image()
        out["cont/to_rgb"] = np.concatenate((r2.data.ravel(), r2.extent))
    return out


def snapshot_cases(ot) -> dict:
    """Raytracer.property_snapshot / compare_property_snapshot (raytracer.py:141-205): which groups of a scene count as
    changed after an edit."""
    out = {}
    n = ot.RefractionIndex("Constant", n=1.5)

    def scene():
        RT = ot.Raytracer(outline=[-5, 5, -5, 5, -10, 60])
        RT.add(ot.RaySource(ot.CircularSurface(r=1), pos=[0, 0, -5]))
        RT.add(ot.Lens(ot.SphericalSurface(r=2, R=10), ot.SphericalSurface(r=2, R=-10), de=0.2, pos=[0, 0, 5], n=n))
        RT.add(ot.Aperture(ot.RingSurface(r=2, ri=0.5), pos=[0, 0, 12]))
        RT.add(ot.Filter(ot.CircularSurface(r=2), pos=[0, 0, 16], spectrum=ot.TransmissionSpectrum("Constant", val=0.5)))
        RT.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 30]))
        return RT

    edits = {
        "nothing": lambda RT: None,
        "move_lens": lambda RT: RT.lenses[0].move_to([0, 0, 6]),
        "source_power": lambda RT: setattr(RT.ray_sources[0], "power", 2.0),
        "source_spectrum": lambda RT: setattr(RT.ray_sources[0], "spectrum", ot.LightSpectrum("Constant")),
        "add_detector": lambda RT: RT.add(ot.Detector(ot.CircularSurface(r=1), pos=[0, 0, 40])),
        "move_detector": lambda RT: RT.detectors[0].move_to([0, 0, 31]),
        "ambient": lambda RT: setattr(RT, "n0", ot.RefractionIndex("Constant", n=1.1)),
        "lens_index": lambda RT: setattr(RT.lenses[0], "n", ot.RefractionIndex("Constant", n=1.6)),
        "remove_aperture": lambda RT: RT.remove(RT.apertures[0]),
        "filter_spectrum": lambda RT: setattr(RT.filters[0], "spectrum", ot.TransmissionSpectrum("Constant", val=0.6)),
        "outline": lambda RT: setattr(RT, "outline", [-6, 6, -6, 6, -10, 60]),
        "no_pol": lambda RT: setattr(RT, "no_pol", True),
        "flip_lens": lambda RT: RT.lenses[0].flip(),
    }
    with ot.global_options.no_warnings():
        for name, edit in edits.items():
            RT = scene()
            h1 = RT.property_snapshot()
            edit(RT)
            cmp = RT.compare_property_snapshot(h1, RT.property_snapshot())
            keys = sorted(cmp)
            out[f"snap/{name}_keys"] = np.array(keys)
            out[f"snap/{name}"] = np.array([float(bool(cmp[k])) for k in keys])
    return out


def constant_cases(ot) -> dict:
    """Class constants and option lists users (and GUIs) read."""
    S = ot.CircularSurface
    out = {
        "const/numbers": np.array([ot.Raytracer.N_EPS, ot.Raytracer.HURB_FACTOR, S.C_EPS, S.N_EPS, ot.RenderImage.EPS,
                                   ot.RenderImage.K, ot.RenderImage.MAX_IMAGE_SIDE, ot.RenderImage.MAX_IMAGE_RATIO,
                                   *ot.RenderImage.SIZES], dtype=np.float64),
        "const/infos": np.array([f"{m.name}={int(m.value)}" for m in ot.Raytracer.INFOS]),
        "const/focus_methods": np.array(ot.Raytracer.focus_search_methods),
        "const/image_modes": np.array(ot.RenderImage.image_modes),
        "const/source_options": np.array([*ot.RaySource.divergences, "|", *ot.RaySource.orientations, "|",
                                          *ot.RaySource.polarizations]),
        "const/abbr": np.array([getattr(ot, c).abbr for c in ("Lens", "Filter", "Aperture", "Detector", "IdealLens", "RaySource")]),
        "const/spectrum_types": np.array([*ot.LightSpectrum.spectrum_types, "|", *ot.TransmissionSpectrum.spectrum_types]),
        "const/n_types": np.array(ot.RefractionIndex.n_types),
        "const/coeff_count": np.array([f"{k}={v}" for k, v in sorted(ot.RefractionIndex.coeff_count.items())]),
        "const/projections": np.array(ot.SphericalSurface.sphere_projection_methods),
        "const/rotational_symmetry": np.array([float(getattr(ot, c).rotational_symmetry) for c in (
            "CircularSurface", "RingSurface", "RectangularSurface", "SlitSurface", "ConicSurface", "SphericalSurface",
            "AsphericSurface", "TiltedSurface", "DataSurface1D", "DataSurface2D", "FunctionSurface1D", "FunctionSurface2D")]),
    }
    t, l = ot.TransmissionSpectrum("Constant", val=0.5), ot.LightSpectrum("Lines", lines=[500., 600.], line_vals=[1, 1])
    c, r = ot.LightSpectrum("Constant"), ot.RefractionIndex("Constant", n=1.2)
    out["const/units"] = np.array([t.unit, t.quantity, l.unit, l.quantity, c.unit, c.quantity, r.unit, r.quantity])
    return out
