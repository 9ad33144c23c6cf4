"""CPU tests of the host-side mirror of the reference API: constructor validation and error types, lens
thickness logic, groups, geometry checks, scene flattening, storage accounting.  Modelled on the reference's
tests/test_geometry.py, tests/test_surface.py, tests/test_refraction_index.py, tests/test_misc.py:172-203."""
import doctest

import numpy as np
import pytest

import optrace_amd as ot
from optrace_amd import _capi, misc
from optrace_amd.scene import CompiledScene

import scenes


def test_misc_doctests():
    assert doctest.testmod(misc, optionflags=doctest.ELLIPSIS | doctest.NORMALIZE_WHITESPACE).failed == 0


def test_surface_validation():
    with pytest.raises(ValueError):
        ot.CircularSurface(r=0)
    with pytest.raises(TypeError):
        ot.CircularSurface(r=[1])
    with pytest.raises(ValueError):
        ot.RingSurface(r=1, ri=1)
    with pytest.raises(ValueError):
        ot.ConicSurface(r=3, R=2, k=1)  # r beyond the conic's rim
    with pytest.raises(ValueError):
        ot.SphericalSurface(r=1, R=0)
    with pytest.raises(ValueError):
        ot.SphericalSurface(r=1, R=np.inf)
    with pytest.raises(ValueError):
        ot.RectangularSurface(dim=[1, -1])
    with pytest.raises(ValueError):
        ot.SlitSurface(dim=[1, 1], dimi=[2, 0.1])
    with pytest.raises(ValueError):
        ot.AsphericSurface(r=1, R=10, k=0, coeff=[])
    s = ot.SphericalSurface(r=1, R=5)
    with pytest.raises(RuntimeError):
        s.r = 2  # locked
    with pytest.raises(AttributeError):
        ot.Aperture(ot.CircularSurface(r=1), pos=[0, 0, 0]).foo = 1


def test_surface_geometry_bookkeeping():
    s = ot.SphericalSurface(r=3, R=8)
    z1 = 8 - np.sqrt(64 - 9)
    assert s.z_min == 0 and abs(s.z_max - z1) < 1e-14
    s.move_to([1, 2, 3])
    assert np.allclose(s.extent, [-2, 4, -1, 5, 3, 3 + z1])
    assert abs(s.dp - z1) < 1e-14 and s.dn == 0 and abs(s.ds - z1) < 1e-14
    s.flip()
    assert s.R == -8 and abs(s.z_min - (3 - z1)) < 1e-14 and s.z_max == 3
    r = ot.RectangularSurface(dim=[4, 2])
    r.rotate(90)
    assert np.allclose(r.extent[:4], [-1, 1, -2, 2])
    c = s.copy()
    assert c is not s and c.R == s.R


def test_lens_thickness_logic():
    f, b = ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8)
    n = ot.RefractionIndex("Constant", n=1.5)
    L = ot.Lens(f, b, n=n, pos=[0, 0, 10], de=0.1)
    assert abs(L.de - 0.1) < 1e-12
    assert abs(L.d - (0.1 + f.dp + b.dn)) < 1e-12
    L2 = ot.Lens(f, b, n=n, pos=[0, 0, 0], d=2.0)
    assert abs(L2.d - 2.0) < 1e-12
    L3 = ot.Lens(f, b, n=n, pos=[0, 0, 0], d1=0.3, d2=0.9)
    assert L3.front.pos[2] == -0.3 and L3.back.pos[2] == 0.9
    with pytest.raises(ValueError):
        ot.Lens(f, b, n=n, pos=[0, 0, 0], de=None, d1=1)
    with pytest.raises(TypeError):
        ot.Lens(f, b, n=1.5, pos=[0, 0, 0])
    with pytest.raises(RuntimeError):
        L.front = f  # geometry lock
    assert L.front is not f  # elements own copies of their surfaces


def test_refraction_index_validation_and_equality():
    with pytest.raises(ValueError):
        ot.RefractionIndex("Constant", n=0.9)
    with pytest.raises(ValueError):
        ot.RefractionIndex("Cauchy", coeff=[1.5, 0.1])
    with pytest.raises(ValueError):
        ot.RefractionIndex("Nope")
    with pytest.raises(ValueError):
        ot.RefractionIndex("Abbe", n=1.5, V=-3)
    a, b = ot.RefractionIndex("Abbe", n=1.5, V=40), ot.RefractionIndex("Abbe", n=1.5, V=40)
    assert a == b and a != ot.RefractionIndex("Abbe", n=1.5, V=41)
    # Abbe coefficients are solved in the reference's float32 line arithmetic
    A, B, d = a._abbe_AB()
    assert A == float(np.float32(A)) and B == float(np.float32(B)) and d == 0.014


def test_ray_source_validation():
    with pytest.raises(ValueError):
        ot.RaySource(ot.Point(), divergence="Nope")
    with pytest.raises(ValueError):
        ot.RaySource(ot.Point(), s=[0, 0, -1])
    with pytest.raises(ValueError):
        ot.RaySource(ot.SphericalSurface(r=1, R=5))
    with pytest.raises(ValueError):
        ot.RaySource(ot.Point(), power=0)
    with pytest.raises(TypeError):
        ot.RaySource(ot.Point(), spectrum=ot.TransmissionSpectrum("Constant", val=0.5))
    rs = ot.RaySource(ot.Point(), s_sph=[90 - 1e-9, 0])
    assert rs.s[2] > 0
    f = ot.RaySource(ot.RectangularSurface(dim=[2, 1]), polarization="List", pol_angles=[0, 90], pol_probs=[1, 3])._source_fields()
    assert f["polarization"] == _capi.POL_LIST and f["n_pol"] == 2 and np.allclose(f["pol_tab"], [0, np.pi / 2, 1, 4])


def test_group_and_tracing_surfaces():
    RT = scenes.mixed_geometry(ot)
    assert len(RT.lenses) == 5 and len(RT.apertures) == 1 and len(RT.filters) == 1 and len(RT.detectors) == 2
    z = [el.pos[2] for el in RT.elements]
    assert z == sorted(z)
    assert len(RT.tracing_surfaces) == 11  # 4 lenses x 2 + aperture + filter + ideal lens
    L = RT.lenses[0]
    assert RT.has(L)
    assert RT.remove(L) and not RT.has(L) and not RT.remove(L)
    RT.clear()
    assert RT.elements == []
    G = ot.presets.geometry.arizona_eye()
    RT2 = ot.Raytracer(outline=[-10, 10, -10, 10, -10, 30])
    with pytest.warns(ot.OptraceWarning):
        RT2.n0 = ot.RefractionIndex("Constant", n=1.2)
        RT2.add(G)  # different ambient index in the group: overwritten with a warning
    assert len(RT2.lenses) == 2 and len(RT2.apertures) == 1 and len(RT2.detectors) == 1


def test_scene_flattening_tables():
    RT = scenes.mixed_geometry(ot)
    RT._geometry_checks()
    assert not RT.geometry_error
    sc = CompiledScene(RT)
    assert sc.nt == 13 and sc.desc.n_elements == 8 and sc.desc.n_surfaces == 12
    kinds = [sc.elements[i].kind for i in range(sc.desc.n_elements)]
    assert kinds == [_capi.EL_LENS, _capi.EL_LENS, _capi.EL_APERTURE, _capi.EL_LENS, _capi.EL_FILTER, _capi.EL_LENS,
                     _capi.EL_IDEAL_LENS, _capi.EL_APERTURE]
    assert sc.desc.filters[0].type == _capi.T_DATA  # Function spectrum, continuous source -> host table
    # the same medium object is uploaded once
    RT2 = scenes.double_gauss(ot)
    sc2 = CompiledScene(RT2)
    assert sc2.desc.n_media == 1 + 7 and sc2.desc.n_surfaces == 16
    assert all(sc2.media[i].model == _capi.N_ABBE for i in range(1, 8)) and sc2.media[0].model == _capi.N_CONSTANT


def test_geometry_checks_report_errors_as_warnings():
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 5])
    with pytest.warns(ot.OptraceWarning, match="RaySource Missing"):
        RT._geometry_checks()
    assert RT.geometry_error
    RT.add(ot.RaySource(ot.Point(), pos=[0, 0, -2]))
    RT.add(ot.Aperture(ot.CircularSurface(r=1), pos=[0, 0, 20]))
    with pytest.warns(ot.OptraceWarning, match="outside outline"):
        RT._geometry_checks()
    assert RT.geometry_error
    # colliding surfaces
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 15])
    RT.add(ot.RaySource(ot.Point(), pos=[0, 0, -2]))
    n = ot.RefractionIndex("Constant", n=1.5)
    RT.add(ot.Lens(ot.SphericalSurface(r=3, R=4), ot.SphericalSurface(r=3, R=-4), n=n, pos=[0, 0, 2], de=0.1))
    RT.add(ot.Lens(ot.SphericalSurface(r=3, R=-4), ot.SphericalSurface(r=3, R=4), n=n, pos=[0, 0, 3.5], d=0.4))
    with pytest.warns(ot.OptraceWarning, match="collision"):
        RT._geometry_checks()
    assert RT.geometry_error and RT.fault_pos.shape[1] == 3
    with pytest.raises(ValueError):
        RT.trace(0)
    with pytest.raises(TypeError):
        RT.trace(10.5)
    with pytest.raises(ValueError):
        ot.Raytracer(outline=[0, 0, 1, 2, 3, 4])
    # HURB only for ring / slit apertures
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 15], use_hurb=True)
    RT.add(ot.RaySource(ot.Point(), pos=[0, 0, -2]))
    RT.add(ot.Aperture(ot.CircularSurface(r=1), pos=[0, 0, 2]))
    with pytest.warns(ot.OptraceWarning, match="Ray bending"):
        RT._geometry_checks()
    assert RT.geometry_error


def test_storage_size_accounting():
    RS = ot.RayStorage
    for nt in [2, 3, 8, 17]:
        for no_pol in [False, True]:
            for N in [1, 101, 2000, 321455]:
                size = RS.storage_size(N, nt, no_pol)
                expect = N * nt * 24 + N * 24 + N * nt * 4 + N * nt * 8 + N * 4 + (8 if no_pol else N * nt * 12)
                assert size == expect
            for max_size in [1000, 46570689, 6000000000]:
                Nmax = RS.max_rays_for_size(max_size, nt, no_pol)
                sz = RS.storage_size(Nmax, nt, no_pol)
                assert 0 <= max_size - sz < sz / max(Nmax, 1) + 1
    # the survey's per-unit byte figures: N*[(M+2)*48+28] with pol
    assert RS.storage_size(10_000_000, 17, False) == 10_000_000 * (17 * 48 + 28)
    assert RS.storage_size(10_000_000, 17, True) == 10_000_000 * (17 * 36 + 28) + 8


def test_render_image_extent_logic():
    img = ot.RenderImage(extent=[0, 0, 0, 0])
    img._fix_extent()
    assert np.allclose(img.extent, [-1e-9, 1e-9, -1e-9, 1e-9])
    img = ot.RenderImage(extent=[0, 1, 0, 100])
    img._fix_extent()
    assert np.isclose(img.extent[1] - img.extent[0], 20)  # ratio limited to 5
    assert img._pixel_counts() == (945, 945 * 5)
    img = ot.RenderImage(extent=[-1.0, 2.0, 0.5, 1.5])
    img._fix_extent()
    assert img._pixel_counts() == (945 * 3, 945)
    with pytest.raises(ValueError):
        ot.RenderImage(extent=[1, 0, 0, 1])


def test_spectrum_descriptors():
    f = ot.LightSpectrum("Lines", lines=[486.1327, 589.2938, 656.272], line_vals=[1, 2, 0.5])._source_fields()
    assert f["spectrum"] == _capi.SPEC_LINES and f["n_spec"] == 3
    assert np.array_equal(f["spec_tab"][:3], np.float32([486.1327, 589.2938, 656.272]).astype(np.float64))
    assert np.allclose(f["spec_tab"][3:], [1, 3, 3.5])
    f = ot.LightSpectrum("Blackbody", T=5000)._source_fields()
    assert f["spectrum"] == _capi.SPEC_TABLE and f["n_spec"] == 4000 and f["spec_tab"][4000] == 0
    assert np.all(np.diff(f["spec_tab"][4000:]) > 0)
    with pytest.raises(ValueError):
        ot.LightSpectrum("Lines", lines=[300.], line_vals=[1])
    with pytest.raises(ValueError):
        ot.TransmissionSpectrum("Constant", val=1.5)


def test_render_image_archive_round_trip(tmp_path):
    """On-disk format of RenderImage (render_image.py:298-328): read a file the reference wrote, write it back,
    read again -- keys and contents are preserved, so archives are interchangeable."""
    from helpers import GOLDEN
    ref = np.load(GOLDEN / "render_image_ref.npz")
    im = ot.RenderImage.load(str(GOLDEN / "render_image_ref.npz"))
    assert im._data.shape == (945, 945, 4) and im._data.dtype == np.float64
    assert np.array_equal(im._data, ref["_data"]) and np.array_equal(im.extent, ref["extent"])
    assert im.limit == 3.0 and im.projection is None
    assert im.long_desc == str(ref["long_desc"][()]) and "DET0" in im.long_desc
    assert abs(im.power() - ref["_data"][:, :, 3].sum()) < 1e-12
    im.save(str(tmp_path / "again"))  # extension is appended
    back = np.load(tmp_path / "again.npz")
    assert sorted(back.files) == sorted(ref.files)
    for k in ref.files:
        assert np.array_equal(back[k], ref[k]), k
    im2 = ot.RenderImage.load(str(tmp_path / "again.npz"))
    assert np.array_equal(im2._data, im._data) and im2.limit == im.limit
    with pytest.raises(RuntimeError):
        ot.RenderImage([-1, 1, -1, 1]).save(str(tmp_path / "empty"))


def test_image_from_file(tmp_path):
    """Image sources from files (base_image.py:67-83): element [0, 0] is the lower left corner, values in [0, 1]."""
    from PIL import Image
    arr = (np.arange(4 * 6 * 3).reshape(4, 6, 3) * 3).astype(np.uint8)
    Image.fromarray(arr, "RGB").save(tmp_path / "card.png")
    img = ot.RGBImage(str(tmp_path / "card.png"), [3, 2])
    assert img.shape == (4, 6, 3) and np.array_equal(img.data, np.flipud(arr) / 255.0)
    g = ot.GrayscaleImage(str(tmp_path / "card.png"), [3, 2])
    luma = np.flipud(arr) @ np.array([0.299, 0.587, 0.114]) / 255.0
    assert g.shape == (4, 6) and np.abs(g.data - luma).max() <= 1.0 / 255
    with pytest.raises(IOError):
        ot.RGBImage(str(tmp_path / "missing.png"), [3, 2])


def test_tilted_data_function_surfaces_host_logic():
    """Constructors, validation and bookkeeping of the rank-4 surfaces (tilted_surface.py, data_surface_2d.py,
    function_surface_2d.py); the device descriptors carry consistent spline tables."""
    from optrace_amd import _capi
    with ot.global_options.no_warnings():
        t = ot.TiltedSurface(r=3, normal=[0, -0.45, np.sqrt(1 - 0.45 ** 2)])
        assert t.z_max == -t.z_min and abs(t.z_max - 3 * 0.45 / np.sqrt(1 - 0.45 ** 2)) < 1e-14
        t2 = t.copy()
        t2.rotate(180)
        assert np.allclose(t2.normal, [0, 0.45, np.sqrt(1 - 0.45 ** 2)])
        t2.flip()
        assert np.allclose(t2.normal[0], 0)
        with pytest.raises(ValueError):
            ot.TiltedSurface(r=3, normal=[0, 1, 0])  # normal[2] must be above 0
        with pytest.raises(RuntimeError):
            ot.TiltedSurface(r=3)
        d = t._desc()
        assert d.kind == _capi.SURF_TILTED and abs(d.normal[1] + 0.45) < 1e-15

        with pytest.raises(ValueError):
            ot.DataSurface1D(r=2, data=np.zeros(20))  # too few values
        with pytest.raises(ValueError):
            ot.DataSurface2D(r=2, data=np.zeros((60, 70)))  # not square
        r = np.linspace(0, 3, 200)
        d1 = ot.DataSurface1D(r=3, data=10 - np.sqrt(100 - r ** 2), parax_roc=10.)
        assert abs(d1.z_min) < 1e-12 and abs(d1.z_max - (10 - np.sqrt(91))) < 1e-9 and d1.rotational_symmetry
        dd = d1._desc()
        n = dd.nknots
        assert dd.kind == _capi.SURF_DATA1D and dd.tab_len == 3 * n and n == 2 * 200 - 1 + _capi.SPL_K + 1
        zmin, zmax = d1.z_min, d1.z_max
        d1.flip()
        assert d1._sign == -1 and d1.parax_roc == -10. and abs(d1.z_max + zmin) < 1e-15 and abs(d1.z_min + zmax) < 1e-15

        xy = np.linspace(-2, 2, 80)
        X, Y = np.meshgrid(xy, xy)
        d2 = ot.DataSurface2D(r=2, data=X ** 2 / 20 + Y ** 2 / 30)
        d2.rotate(30)
        dd = d2._desc()
        nc = dd.nknots - _capi.SPL_K - 1
        assert dd.kind == _capi.SURF_DATA2D and dd.tab_len == dd.nknots + nc * nc + 2 * (nc - 1) * nc
        assert abs(dd.angle - np.deg2rad(30)) < 1e-15 and not d2.rotational_symmetry

        with pytest.raises(TypeError):
            ot.FunctionSurface2D(r=2, func=3)
        with pytest.raises(RuntimeError):
            ot.FunctionSurface2D(r=2, func=lambda x, y: 1.0)  # must return an array
        with pytest.raises(RuntimeError):   # mask_func must return booleans (function_surface_2d.py:186)
            ot.FunctionSurface2D(r=2, func=lambda x, y: x * 0.1, mask_func=lambda x, y: x * 1.0)
        # mask_func travels as a bitmap behind the spline tables (include/optrace_amd.h, OT_SURF_FLAG_MASK_TABLE)
        fm = ot.FunctionSurface2D(r=2, func=lambda x, y: x * 0.1, mask_func=lambda x, y: (x > 0) & (np.abs(y) <= 1))
        dm, n = fm._desc(), ot.FunctionSurface2D.N_MASK
        nc = dm.nknots - _capi.SPL_K - 1
        spline_len = dm.nknots + nc * nc + 2 * (nc - 1) * nc
        assert dm.flags == _capi.SURF_FLAG_MASK_TABLE and dm.tab_len == spline_len + 1 + n * n // 64
        assert fm._tab[spline_len] == n
        bits = np.unpackbits(fm._tab[spline_len + 1:].view(np.uint8), bitorder="little").reshape(n, n)  # [iy, ix]
        assert bits[n // 2, n // 2 + 5] and not bits[n // 2, n // 2 - 5] and not bits[n - 10, n // 2 + 5]
        assert bits.sum() == (n // 2) * (n // 2)       # x > 0: half the columns; |y| <= 1: half the rows
        assert 0 <= fm.z_min < 1e-4 and abs(fm.z_max - 0.2) < 1e-4     # z range over the masked part only
        assert fm._mask_host(np.array([0.5, -0.5, 0.5]), np.array([0.5, 0.5, 1.5])).tolist() == [True, False, False]
        fm.rotate(90)   # the frame turns with the surface: the kept half is now y > 0
        assert fm._mask_host(np.array([0.5, -0.5, 0.5]), np.array([0.5, 0.5, -0.5])).tolist() == [True, True, False]
        fm1 = ot.FunctionSurface1D(r=2, func=lambda r: r ** 2 / 10, mask_func=lambda r: r <= 1.0)
        d1m = fm1._desc()
        assert d1m.flags == _capi.SURF_FLAG_MASK_TABLE and d1m.tab_len == 3 * d1m.nknots + 1 + fm1.N_MASK_1D // 64
        assert abs(fm1.z_max - 0.1) < 1e-4 and fm1.z_min == 0
        with pytest.raises(ValueError):
            ot.FunctionSurface1D(r=2, func=lambda r: r ** 2 / 10, z_min=0.)  # z_min and z_max only together
        f1 = ot.FunctionSurface1D(r=2, func=lambda r: 0.5 + r ** 2 / 10, z_min=0.5, z_max=0.9)
        assert abs(f1.z_min) < 1e-12 and abs(f1.z_max - 0.4) < 1e-12 and f1._tab_residual < 1e-12
        f2 = ot.FunctionSurface2D(r=2, func=lambda x, y: x ** 2 / 10 + y ** 2 / 25)
        # a quadratic is exact on the coarsest grid tried
        assert f2._nknots == 17 + _capi.SPL_K + 1 and f2._tab_residual < 1e-12 and f2._grad_residual < 1e-9
        f2.rotate(45)
        f2.flip()
        assert f2._sign == -1 and abs(f2._angle - np.pi / 4) < 1e-15 and f2._desc().flags == 0
        f3 = ot.FunctionSurface2D(r=2, func=lambda x, y: x ** 2 / 10, deriv_func=lambda x, y: (x / 5, 0 * y))
        assert f3._desc().flags == _capi.SURF_FLAG_DERIV_UNROTATED


@pytest.mark.parametrize("N_list", [[200_001], [1 << 20], [5, 0, 70_000, 1_000_003], [10_000_000 // 5] * 5, [3] * 70,
                                    [2_000_000_000]])
def test_source_ranges_are_power_of_two_blocks(N_list):
    """Stratification blocks of a launch (RayStorage._source_ranges): they tile [0, N) source by source, all but the
    last block of a source are powers of two in descending order, at most 64 blocks travel as kernel arguments
    unless there are more sources than that, and every block of a source carries the same ray power."""
    from optrace_amd.ray_storage import RayStorage
    st = RayStorage()
    st.N_list = np.array(N_list)
    st.B_list = np.concatenate(([0], np.cumsum(N_list)))
    st.ray_source_list = [ot.RaySource(ot.Point(), spectrum=ot.LightSpectrum("Monochromatic", wl=550.)) for _ in N_list]
    st._powers = [1.5 + k for k in range(len(N_list))]
    rng = st._source_ranges()
    assert len(rng) <= max(64, len(N_list))
    pos = 0
    for i, n in enumerate(N_list):
        blocks = [r for r in rng if r.source == i]
        assert sum(b.count for b in blocks) == n
        for b in blocks:
            assert b.first == pos
            pos += b.count
            assert b.ray_power == (st._powers[i] / n if n else 0.)
        sizes = [b.count for b in blocks[:-1]]
        assert all(s & (s - 1) == 0 and s >= (1 << 16) for s in sizes) and sizes == sorted(sizes, reverse=True)
        if len(blocks) > 1:  # the ragged rest is smaller than every power-of-two block before it or the cap was hit
            assert blocks[-1].count < sizes[-1] or len(blocks) == max(1, 64 // len(N_list))
    assert pos == sum(N_list)


def test_image_profile_cuts():
    """BaseImage.profile (base_image.py:149-186): nearest pixel column / row, bin edges along the cut."""
    data = np.arange(12, dtype=float).reshape(3, 4) / 12
    im = ot.ScalarImage(data, extent=[0, 4, 10, 13])
    edges, (cut,) = im.profile(y=11.5)
    np.testing.assert_array_equal(edges, [0, 1, 2, 3, 4])
    np.testing.assert_array_equal(cut, data[1])
    edges, (cut,) = im.profile(x=4)           # the upper edge belongs to the last column
    np.testing.assert_array_equal(edges, [10, 11, 12, 13])
    np.testing.assert_array_equal(cut, data[:, 3])
    rgb = ot.RGBImage(np.stack([data, data / 2, data / 3], axis=2), extent=[0, 4, 10, 13])
    _, cuts = rgb.profile(x=0.5)
    assert len(cuts) == 3
    np.testing.assert_array_equal(cuts[1], data[:, 0] / 2)
    with pytest.raises(ValueError):
        im.profile()
    with pytest.raises(ValueError):
        im.profile(x=5)


def test_function_surface_z_bounds_like_the_reference():
    """After the reference's test_surface_zmin_zmax_cases (tests/test_surface.py:508-542): user z bounds of a function
    surface are taken when they enclose the measured range (also generously), and ignored with a warning when they are
    too narrow or shifted."""
    with ot.global_options.no_warnings():
        lin = lambda x, y: x  # noqa: E731
        for kw in (dict(z_max=5), dict(z_min=5)):
            with pytest.raises(ValueError):
                ot.FunctionSurface2D(r=2, func=lin, **kw)
        sfunc = lambda x, y: x ** 2 + y ** 2 / 10  # noqa: E731
        z_min, z_max = ot.FunctionSurface2D(func=sfunc, r=3).extent[4:]
        assert abs(z_min) < 1e-9 and abs(z_max - 9) < 1e-6

        def bounds(dz0, dz1):
            sf = ot.FunctionSurface2D(func=sfunc, r=3, z_min=z_min + dz0, z_max=z_max + dz1)
            return sf.z_min, sf.z_max

        a = bounds(0, 0)
        assert abs(a[0] - z_min) < 1e-7 and abs(a[1] - z_max) < 1e-7
        a = bounds(-1e-9, 1e-9)
        assert abs(a[0] - (z_min - 1e-9)) < 1e-12 and abs(a[1] - (z_max + 1e-9)) < 1e-12          # taken
        for dz in ((1e-3, -1e-3), (1e-3, 1e-3), (-1e-3, -1e-3)):                                  # too narrow / shifted
            a = bounds(*dz)
            assert abs(a[0] - (z_min + dz[0])) > 1e-4 and abs(a[1] - (z_max + dz[1])) > 1e-4
        a = bounds(0, 5)                                                                          # generous: taken
        assert abs(a[0] - z_min) < 1e-7 and abs(a[1] - (z_max + 5)) < 1e-12


def test_no_large_writeable_arrays_on_tracked_objects():
    """The unchanged-scene shortcut of `Raytracer.trace` relies on large arrays of tracked objects being read-only (a
    writeable one switches the shortcut off and is checksummed at every trace: 0.45 ms per trace for the pixel pdf of a
    256 x 256 image source before it was locked).  Image sources, data surfaces, tabulated spectra and media included."""
    import scenes
    from optrace_amd.base import BaseClass

    def writeable_arrays(root):
        seen, out = set(), []

        def walk(o, path):
            if id(o) in seen:
                return
            seen.add(id(o))
            for k, v in vars(o).items():
                if isinstance(v, np.ndarray) and v.size >= 20 and v.flags.writeable:
                    out.append(f"{path}.{k} {v.shape}")
                elif isinstance(v, BaseClass):
                    walk(v, f"{path}.{k}")
                elif isinstance(v, list):
                    for i, e in enumerate(v):
                        if isinstance(e, BaseClass):
                            walk(e, f"{path}.{k}[{i}]")
        walk(root, "RT")
        return out

    with ot.global_options.no_warnings():
        builders = {**scenes.SCENES, **scenes.SCENES2, **scenes.SCENES3}
        for name, (build, _) in builders.items():
            assert writeable_arrays(build(ot)) == [], name
        RT = ot.Raytracer(outline=[-8, 8, -8, 8, 0, 40])
        RT.add(ot.RaySource(ot.RGBImage(scenes.synthetic_rgb_image(), [4, 3]), pos=[0, 0, 0]))
        RT.add(ot.RaySource(ot.GrayscaleImage(scenes.synthetic_gray_image(), [4, 3]), pos=[0, 0, 1],
                            spectrum=ot.LightSpectrum("Data", wls=np.linspace(400., 700., 50), vals=np.linspace(1., 2., 50))))
        assert writeable_arrays(RT) == []


def test_chunk_plan_of_an_iterative_render():
    """`Raytracer._chunk_plan`: with ITER_RAYS_STEP the reference's rule (raytracer.py:1216-1217, 1238-1239: N // step
    iterations, the last one takes the remainder); otherwise by storage -- render-only chunks as large as the tail storage
    and one launch allow, then ONE stored chunk of ITER_LAST_RAYS; without render-only chunks equal chunks that fit
    ITER_STORAGE_BYTES of ray storage.  Every plan sums to N, chunks before the last are multiples of 1024 rays."""
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, 0, 10])
    old = (ot.Raytracer.ITER_RAYS_STEP, ot.Raytracer.ITER_LAST_RAYS, ot.Raytracer.ITER_STORAGE_BYTES)
    try:
        ot.Raytracer.ITER_RAYS_STEP = 1_000_000
        assert RT._chunk_plan(3_500_000, 4, True) == [1_000_000, 1_000_000, 1_500_000]
        assert RT._chunk_plan(999, 4, True) == [999]
        ot.Raytracer.ITER_RAYS_STEP = None
        last = ot.Raytracer.ITER_LAST_RAYS
        assert RT._chunk_plan(last, 4, True) == [last]
        assert RT._chunk_plan(last + 5, 4, True) == [5, last]
        plan = RT._chunk_plan(200_000_000, 4, True)
        assert plan == [200_000_000 - last, last]
        for N in (10 ** 9, 3 * 10 ** 8 + 17, (1 << 28) + last + 1):
            plan = RT._chunk_plan(N, 4, True)
            assert sum(plan) == N and plan[-1] == last and len(plan) >= 2
            assert all(0 < n <= min(1 << 28, ot.Raytracer.ITER_STORAGE_BYTES // 60) for n in plan[:-1])
            assert all(n % 1024 == 0 for n in plan[:-2])
        # through the ray storage: equal chunks that fit the budget
        RT.no_pol = True
        plan = RT._chunk_plan(200_000_000, 4, False)
        assert sum(plan) == 200_000_000 and len(plan) == 3 and all(n % 1024 == 0 for n in plan[:-1])
        assert max(plan) * (4 * 36 + 28) <= ot.Raytracer.ITER_STORAGE_BYTES * 1.01
        assert RT._chunk_plan(500_000, 4, False) == [500_000]
    finally:
        ot.Raytracer.ITER_RAYS_STEP, ot.Raytracer.ITER_LAST_RAYS, ot.Raytracer.ITER_STORAGE_BYTES = old
