"""CPU-only checks of the C-ABI boundary: the HIP library loads, exports every symbol that
include/optrace_amd.h declares, the ctypes structures mirror the header, and the product path refuses to
run without a device instead of falling back to the CPU."""
import ctypes as C
import pathlib
import re
import subprocess

import pytest

from optrace_amd import _capi

ROOT = pathlib.Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "optrace_amd.h"


@pytest.fixture(scope="module")
def lib():
    if not _capi.library_path().exists():
        import __graft_entry__ as g
        g.build()
    return C.CDLL(str(_capi.library_path()))


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ot_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _capi.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_capi.SIGNATURES) == names


def test_abi_version(lib):
    lib.ot_abi_version.restype = C.c_int
    assert lib.ot_abi_version() == _capi.ABI_VERSION == 9


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    names = ["ot_surface", "ot_medium", "ot_filter", "ot_element", "ot_scene_desc", "ot_source",
             "ot_source_range", "ot_rays", "ot_detector_req", "ot_detector_image_req"]
    body = "\n".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names)
    extra = 'printf("off_coeff %zu\\n", offsetof(ot_surface, coeff)); printf("off_spec %zu\\n", offsetof(ot_source, spec_tab));'
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(){{{body}{extra}return 0;}}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    py = {"ot_surface": _capi.Surface, "ot_medium": _capi.Medium, "ot_filter": _capi.Filter,
          "ot_element": _capi.Element, "ot_scene_desc": _capi.SceneDesc, "ot_source": _capi.Source,
          "ot_source_range": _capi.SourceRange, "ot_rays": _capi.Rays, "ot_detector_req": _capi.DetectorReq,
          "ot_detector_image_req": _capi.DetectorImageReq}
    for n, cls in py.items():
        assert C.sizeof(cls) == int(out[n]), n
    assert _capi.Surface.coeff.offset == int(out["off_coeff"])
    assert _capi.Source.spec_tab.offset == int(out["off_spec"])


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    import numpy as np
    import optrace_amd as ot
    sf = ot.SphericalSurface(r=1, R=5)
    with pytest.raises(ot.BackendError):
        sf.find_hit(np.zeros((2, 3)), np.array([[0., 0., 1.], [0., 0., 1.]]))
    RT = ot.Raytracer(outline=[-1, 1, -1, 1, -1, 10])
    RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0]))
    with pytest.raises(ot.BackendError):
        RT.trace(100)


def test_product_package_does_not_reference_the_oracle():
    """optrace_amd must not import, load or call anything under oracle/ (task rule 3)."""
    for path in (ROOT / "optrace_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".hpp", ".h", ".inc") and path.is_file():
            text = path.read_text()
            assert "liboracle" not in text and "oracle_bridge" not in text and "orc_" not in text, path


def test_automatic_extent_entry_points_check_their_arguments_before_anything_else():
    """`ot_detector_extent_sample` / `ot_detector_image_auto_*` (include/optrace_amd.h): null arguments are refused with
    OT_ERR_INVALID before a device is looked for; cancelling nothing is harmless; a handle comes back null on failure."""
    lib = _capi.load_library()  # (bound signatures: ot_last_error returns bytes)
    st = C.c_void_p()
    assert lib.ot_detector_extent_sample(None, 0, 1, None, 0, 128, None, st) == -1  # OT_ERR_INVALID
    assert b"null argument" in lib.ot_last_error()
    handle = C.c_void_p(12345)
    rc = lib.ot_detector_image_auto_begin(None, 0, 1, None, 0, None, None, None, None, C.byref(handle), st)
    assert rc == -1 and not handle.value
    assert lib.ot_detector_image_auto_finish(None, None, 1, 1, None, st) == -1
    lib.ot_detector_image_auto_cancel(None)
    # a ray storage without buffers
    rays = _capi.Rays()
    sd = _capi.Surface()
    ext = (C.c_double * 4)()
    assert lib.ot_detector_extent_sample(C.byref(rays), 0, 1, C.byref(sd), 0, 128, ext, st) == -1
    assert b"null buffers" in lib.ot_last_error()


def test_tail_append_checks_its_arguments_before_anything_else():
    """`ot_tail_append` (include/optrace_amd.h, ABI 9): null arguments, storages without buffers, ranges outside the storage,
    a tail that cannot hold both chunks and a weight scale that is not a positive number are refused with OT_ERR_INVALID
    before a device is looked for."""
    lib = _capi.load_library()
    st = C.c_void_p()
    assert lib.ot_tail_append(None, 0, 1, 1.0, 0, None, None, None, st) == -1
    assert b"null argument" in lib.ot_last_error()
    buf = (C.c_double * 8)()
    fill = (C.c_uint32 * 1024)()
    res = (C.c_int64 * 2)()
    addr = C.addressof(buf)

    def storage(N, nt, with_buffers=True):
        r = _capi.Rays()
        r.N, r.nt = N, nt
        if with_buffers:  # (never dereferenced: every call below fails in the checks)
            r.p = r.w = r.wl = addr
        return r

    rays, tail = storage(1000, 4), storage(65536, 2)
    call = lambda rays, first, count, scale, before, tail: lib.ot_tail_append(
        C.byref(rays), first, count, scale, before, C.byref(tail), fill, res, st)
    assert call(storage(1000, 4, False), 0, 10, 1.0, 0, tail) == -1 and b"needs p, w, wl" in lib.ot_last_error()
    assert call(storage(1000, 1), 0, 10, 1.0, 0, tail) == -1
    assert call(rays, 0, 1001, 1.0, 0, tail) == -1 and b"range outside" in lib.ot_last_error()
    assert call(rays, -1, 10, 1.0, 0, tail) == -1
    assert call(rays, 0, 10, 1.0, -5, tail) == -1
    assert call(rays, 0, 10, 1.0, 0, storage(65536, 3)) == -1 and b"two sections" in lib.ot_last_error()
    assert call(rays, 0, 10, 1.0, 0, storage(65536, 2, False)) == -1
    for bad in (0.0, -1.0, float("nan"), float("inf")):
        assert call(rays, 0, 10, bad, 0, tail) == -1 and b"weight_scale" in lib.ot_last_error()
    # 1024 pieces of 64 slots hold 1024 waves: one wave more (before + appended) does not fit 65536 slots
    assert call(storage(70000, 4), 0, 64, 1.0, 65536, tail) == -1 and b"smaller than" in lib.ot_last_error()
    assert call(rays, 0, 10, 1.0, 0, storage(65537, 2)) == -1
