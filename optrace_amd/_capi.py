"""ctypes binding of the C-ABI declared in include/optrace_amd.h.

The structures below mirror the header field by field; `tests/test_capi.py` checks their sizes
against the compiled library and that every declared symbol is exported.  There is deliberately no
CPU fallback: if the HIP library is missing or no device is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "liboptrace_hip.so"

OT_MAX_ASPH = 12
OT_MAX_LINES = 8

# --- enums (kept in sync with the header) -------------------------------------------------------
SURF_CIRCLE, SURF_RING, SURF_RECT, SURF_SLIT, SURF_CONIC, SURF_ASPHERE, SURF_TILTED, SURF_DATA1D, SURF_DATA2D = range(9)
SPL_K = 4  # OT_SPL_K
SURF_FLAG_DERIV_UNROTATED = 1
SURF_FLAG_MASK_TABLE = 2
(N_CONSTANT, N_ABBE, N_CAUCHY, N_CONRADY, N_SELLMEIER1, N_SELLMEIER2, N_SELLMEIER3, N_SELLMEIER4,
 N_SELLMEIER5, N_SCHOTT, N_HERZBERGER, N_HOO1, N_HOO2, N_EXTENDED, N_EXTENDED2, N_EXTENDED3,
 N_DATA, N_LINES) = range(18)
T_CONSTANT, T_DATA, T_RECTANGLE, T_GAUSSIAN, T_LINES = range(5)
EL_LENS, EL_IDEAL_LENS, EL_FILTER, EL_APERTURE = range(4)
SRC_POINT, SRC_LINE, SRC_CIRCLE, SRC_RING, SRC_RECT, SRC_IMAGE_RGB, SRC_IMAGE_GRAY = range(7)
DIV_NONE, DIV_LAMBERTIAN, DIV_ISOTROPIC, DIV_TABLE = range(4)
OR_CONSTANT, OR_CONVERGING, OR_ARRAY = range(3)
POL_CONSTANT, POL_UNIFORM, POL_LIST, POL_TABLE = range(4)
SPEC_MONO, SPEC_UNIFORM, SPEC_LINES, SPEC_GAUSSIAN, SPEC_TABLE = range(5)
PROJ_NONE, PROJ_EQUIDISTANT, PROJ_ORTHOGRAPHIC, PROJ_EQUAL_AREA, PROJ_STEREOGRAPHIC = range(5)
N_INFOS = 5

PROJECTIONS = {None: PROJ_NONE, "Equidistant": PROJ_EQUIDISTANT, "Orthographic": PROJ_ORTHOGRAPHIC,
               "Equal-Area": PROJ_EQUAL_AREA, "Stereographic": PROJ_STEREOGRAPHIC}

d3 = C.c_double * 3
d2 = C.c_double * 2


class Surface(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ncoeff", C.c_int32), ("pos", d3), ("r", C.c_double),
                ("ri", C.c_double), ("dim", d2), ("dimi", d2), ("angle", C.c_double),
                ("R", C.c_double), ("k", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("coeff", C.c_double * OT_MAX_ASPH),
                ("normal", d3), ("sign", C.c_double), ("offset", C.c_double),
                ("tab", C.POINTER(C.c_double)), ("tab_len", C.c_int64), ("nknots", C.c_int32), ("flags", C.c_int32)]


class Medium(C.Structure):
    _fields_ = [("model", C.c_int32), ("tab_len", C.c_int32), ("tab_off", C.c_int64),
                ("c", C.c_double * 10)]


class Filter(C.Structure):
    _fields_ = [("type", C.c_int32), ("inverse", C.c_int32), ("tab_len", C.c_int32),
                ("_pad", C.c_int32), ("tab_off", C.c_int64), ("val", C.c_double), ("wl0", C.c_double),
                ("wl1", C.c_double), ("mu", C.c_double), ("sig", C.c_double)]


class Element(C.Structure):
    _fields_ = [("kind", C.c_int32), ("front", C.c_int32), ("back", C.c_int32), ("n_lens", C.c_int32),
                ("n_after", C.c_int32), ("filter", C.c_int32), ("hurb", C.c_int32), ("_pad", C.c_int32),
                ("D", C.c_double)]


class SceneDesc(C.Structure):
    _fields_ = [("outline", C.c_double * 6),
                ("n_surfaces", C.c_int32), ("n_elements", C.c_int32), ("n_media", C.c_int32),
                ("n_filters", C.c_int32),
                ("surfaces", C.POINTER(Surface)), ("elements", C.POINTER(Element)),
                ("media", C.POINTER(Medium)), ("filters", C.POINTER(Filter)),
                ("table_pool", C.POINTER(C.c_double)), ("table_pool_len", C.c_int64),
                ("n0", C.c_int32), ("no_pol", C.c_int32), ("use_hurb", C.c_int32), ("n_lines", C.c_int32),
                ("hurb_factor", C.c_double), ("lines", C.POINTER(C.c_double))]


class Source(C.Structure):
    _fields_ = [("shape", C.c_int32), ("divergence", C.c_int32), ("div_2d", C.c_int32),
                ("orientation", C.c_int32), ("polarization", C.c_int32), ("spectrum", C.c_int32),
                ("img_w", C.c_int32), ("img_h", C.c_int32),
                ("pos", d3), ("r", C.c_double), ("ri", C.c_double), ("dim", d2), ("angle", C.c_double),
                ("div_angle", C.c_double), ("div_axis_angle", C.c_double), ("s", d3), ("conv_pos", d3),
                ("pol_angle", C.c_double), ("wl", C.c_double), ("wl0", C.c_double), ("wl1", C.c_double),
                ("mu", C.c_double), ("sig", C.c_double), ("power", C.c_double),
                ("spec_tab", C.POINTER(C.c_double)), ("n_spec", C.c_int64),
                ("pol_tab", C.POINTER(C.c_double)), ("n_pol", C.c_int64),
                ("div_tab", C.POINTER(C.c_double)), ("n_div", C.c_int64),
                ("img_pdf", C.POINTER(C.c_double)), ("img_rgb", C.POINTER(C.c_double)),
                ("s_or", C.c_void_p), ("n_or", C.c_int64)]


class DetectorReq(C.Structure):
    _fields_ = [("detector", C.c_void_p), ("projection", C.c_int32), ("xy_only", C.c_int32),
                ("crop4", C.c_void_p), ("ph", C.c_void_p), ("hw", C.c_void_p), ("extent4", C.c_void_p),
                ("ill_count", C.c_void_p), ("wl_out", C.c_void_p), ("fill", C.c_void_p)]


class DetectorImageReq(C.Structure):
    _fields_ = [("detector", C.c_void_p), ("projection", C.c_int32), ("Nx", C.c_int32), ("Ny", C.c_int32),
                ("_pad", C.c_int32), ("crop4", C.c_void_p), ("extent", C.c_double * 4), ("hist", C.c_void_p),
                ("ill_count", C.c_void_p), ("weight_scale", C.c_double)]


class SourceRange(C.Structure):
    _fields_ = [("source", C.c_int32), ("_pad", C.c_int32), ("first", C.c_int64), ("count", C.c_int64),
                ("ray_power", C.c_double)]


class Rays(C.Structure):
    _fields_ = [("N", C.c_int64), ("nt", C.c_int32), ("_pad", C.c_int32),
                ("p", C.c_void_p), ("s", C.c_void_p), ("w", C.c_void_p), ("n", C.c_void_p),
                ("wl", C.c_void_p), ("pol", C.c_void_p)]


vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int32
u64 = C.c_uint64

# name -> (restype, argtypes); every symbol of include/optrace_amd.h
SIGNATURES = {
    "ot_abi_version": (C.c_int, []),
    "ot_last_error": (C.c_char_p, []),
    "ot_device_count": (C.c_int, []),
    "ot_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(vp)]),
    "ot_scene_destroy": (None, [vp]),
    "ot_scene_sections": (C.c_int, [vp]),
    "ot_sources_create": (C.c_int, [C.POINTER(Source), i32, C.POINTER(vp)]),
    "ot_sources_destroy": (None, [vp]),
    "ot_rays_generate": (C.c_int, [vp, C.POINTER(SourceRange), i32, u64, i32, C.POINTER(Rays), vp]),
    "ot_trace": (C.c_int, [vp, C.POINTER(Rays), vp, u64, vp, vp]),
    "ot_generate_and_trace": (C.c_int, [vp, vp, C.POINTER(SourceRange), i32, u64, C.POINTER(Rays), vp, vp]),
    "ot_generate_and_trace_host": (C.c_int, [vp, vp, C.POINTER(SourceRange), i32, u64, C.POINTER(Rays), vp, vp]),
    "ot_tail_capacity": (i64, [i64]),
    "ot_scene_tail_supported": (C.c_int, [vp]),
    "ot_generate_and_trace_tail": (C.c_int, [vp, vp, C.POINTER(SourceRange), i32, u64, i64, C.POINTER(Rays), vp, vp, vp, vp]),
    "ot_tail_append": (C.c_int, [C.POINTER(Rays), i64, i64, C.c_double, i64, C.POINTER(Rays), vp, vp, vp]),
    "ot_scene_set_timing": (C.c_int, [vp, i32]),
    "ot_scene_last_trace_ms": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "ot_surface_find_hit": (C.c_int, [C.POINTER(Surface), i64, vp, vp, vp, vp, vp, vp]),
    "ot_surface_normals": (C.c_int, [C.POINTER(Surface), i64, vp, vp, vp, vp]),
    "ot_surface_mask": (C.c_int, [C.POINTER(Surface), i64, vp, vp, vp, vp]),
    "ot_surface_values": (C.c_int, [C.POINTER(Surface), i64, vp, vp, vp, vp]),
    "ot_surface_hurb_props": (C.c_int, [C.POINTER(Surface), i64, vp, vp, vp, vp, vp, vp, vp]),
    "ot_refraction_index": (C.c_int, [C.POINTER(Medium), vp, i64, i64, vp, vp, vp]),
    "ot_detector_hits": (C.c_int, [C.POINTER(Rays), i64, i64, C.POINTER(Surface), i32, C.POINTER(C.c_double), vp, vp, vp, vp, vp]),
    "ot_detector_hits_multi": (C.c_int, [C.POINTER(Rays), i64, i64, C.POINTER(DetectorReq), i32, vp]),
    "ot_detector_images": (C.c_int, [C.POINTER(Rays), i64, i64, C.POINTER(DetectorImageReq), i32, vp]),
    "ot_detector_extent_sample": (C.c_int, [C.POINTER(Rays), i64, i64, C.POINTER(Surface), i32, i32, vp, vp]),
    "ot_detector_image_auto_begin": (C.c_int, [C.POINTER(Rays), i64, i64, C.POINTER(Surface), i32, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.POINTER(i32), vp, C.POINTER(vp), vp]),
    "ot_detector_image_auto_finish": (C.c_int, [vp, C.POINTER(C.c_double), i32, i32, vp, vp]),
    "ot_detector_image_auto_cancel": (None, [vp]),
    "ot_scratch_trim": (C.c_int, []),
    "ot_scratch_set_cap": (C.c_int, [i64]),
    "ot_scratch_stats": (C.c_int, [C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]),
    "ot_sphere_projection": (C.c_int, [C.POINTER(Surface), i32, i64, vp, vp, vp]),
    "ot_image_convert": (C.c_int, [vp, i32, i32, i32, i32, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, vp]),
    "ot_image_convolve": (C.c_int, [vp, i32, i32, vp, i32, vp, vp]),
    "ot_render_accumulate": (C.c_int, [i64, vp, vp, vp, vp, C.POINTER(C.c_double), i32, i32, vp, vp]),
    "ot_render_accumulate_compact": (C.c_int, [i64, vp, vp, vp, vp, vp, C.POINTER(C.c_double), i32, i32, vp, vp]),
    "ot_hit_piece_len": (i64, [i64]),
    "ot_spectrum_range": (C.c_int, [i64, vp, vp, vp, vp, vp]),
    "ot_spectrum_histogram": (C.c_int, [i64, vp, vp, vp, i32, vp, vp]),
    "ot_spectrum_range_compact": (C.c_int, [i64, vp, vp, vp, vp, vp, vp]),
    "ot_spectrum_histogram_compact": (C.c_int, [i64, vp, vp, vp, vp, i32, vp, vp]),
    "ot_focus_prepare": (C.c_int, [C.POINTER(Rays), i64, i64, C.c_double, vp, vp, vp, vp]),
    "ot_focus_cost": (C.c_int, [i64, vp, vp, i32, C.POINTER(C.c_double), i32, i32, vp, vp, vp]),
    "ot_focus_moments": (C.c_int, [i64, vp, vp, C.c_double, C.c_double, vp, vp]),
    "ot_selftest_arith": (C.c_int, [i32, i32, i64, u64, C.POINTER(i64), C.POINTER(C.c_double), vp]),
    "ot_selftest_eval": (C.c_int, [i32, i64, vp, vp, vp, vp, vp, vp]),
}

FOCUS_WS = 16  # OT_FOCUS_WS
HIT_PIECES = 1024  # OT_HIT_PIECES
ABI_VERSION = 9  # OT_ABI_VERSION
ERR_UNSUPPORTED = -3  # OT_ERR_UNSUPPORTED

_lib = None


class BackendError(RuntimeError):
    """Raised when the HIP library is missing, fails to load or reports an error."""


def library_path() -> pathlib.Path:
    return pathlib.Path(os.environ.get("OPTRACE_AMD_LIB", LIB_PATH))


def load_library() -> C.CDLL:
    """Load liboptrace_hip.so (built in-tree by __graft_entry__.build() / `make -C optrace_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise BackendError(f"HIP library {path} not found. Build it with `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` or `make -C optrace_amd/csrc`. There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7; importing it first makes both share one HIP runtime
    import torch  # noqa: F401
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ot_abi_version() != ABI_VERSION:
        raise BackendError(f"ABI version mismatch: library reports {lib.ot_abi_version()}, "
                           f"binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load_library().ot_last_error()
        err = BackendError(f"optrace_amd backend error {status}: {msg.decode() if msg else '?'}")
        err.status = status  # OT_ERR_*
        raise err
