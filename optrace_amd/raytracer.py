"""Raytracer: sequential tracing of a scene on one MI355X (or one per rank).

Mirror of optrace/tracer/raytracer.py:28-1279 for the hot path named in BASELINE.json:
`trace` (ray generation, all surface interactions, section stores, counters), `detector_image`
(detector hit search + binning) and `iterative_render`.  The object keeps the reference's API; the work
happens in HIP kernels reached through the C-ABI (include/optrace_amd.h):

    trace(N)          -> ot_generate_and_trace   (one fused launch: generation + every surface)
    detector_image()  -> ot_detector_hits + ot_render_accumulate

Geometry validation (element order, collisions) is one-off host work per trace and stays in NumPy.
"""
from __future__ import annotations

import ctypes as C
from enum import IntEnum
from typing import Any

import numpy as np
import torch

from . import _capi
from . import base as _base
from .base import check_type
from .geometry.elements import Group, Aperture, Detector
from .geometry.surfaces import Surface, Point, Line, SphericalSurface, RingSurface, SlitSurface
from .options import global_options
from .ray_storage import RayStorage, TailStorage
from .refraction_index import RefractionIndex
from .render_image import RenderImage
from .spectrum import LightSpectrum
from .geometry.ray_source import RaySource
from .scene import CompiledScene, tracing_elements
from ._device import require_device, stream_ptr, ptr, alloc_retry, mailbox
from ._warn import warning
from . import detector as _detector
from . import checks as _checks


class Raytracer(Group):

    N_EPS: float = 1e-11
    HURB_FACTOR: float = 2 ** 0.5
    MAX_RAY_STORAGE_RAM: int = 200_000_000_000
    """Upper bound for the ray storage of one trace.  The reference guards host RAM with 6 GB
    (raytracer.py:37); here the storage lives in the 288 GB of HBM3E of one MI355X."""
    ITER_RAYS_STEP: int = None
    """Rays per chunk of `iterative_render` (the reference: 1 M, to bound host RAM, raytracer.py:40).  None: as many
    as ITER_STORAGE_BYTES of ray storage hold, in equal chunks."""
    COMPACT_HITS_FROM: int = 1 << 21
    """`detector_image` with an automatic extent: from this many rays on the hit list between hit search and binning
    holds the valid hits only (`ot_detector_req.fill`)."""
    AUTO_ONE_PASS_FROM: int = 1 << 24
    """... and from this many rays on, where the detector allows it, the sections are read once and the hits sorted on a
    provisional tile grid laid over the extent of a sample: every AUTO_SAMPLE_STRIDE-th wave of 64 rays
    (`_auto_image_one_pass`)."""
    AUTO_SAMPLE_STRIDE: int = 128
    AUTO_MARGINS: tuple = ((0.05, 361), (0.02, 361), (0.5, 1024), (0.3, 1024), (0.15, 1024), (0.3, 2048), (0.15, 2048), (0.05, 2048))
    """(margin around the sample's extent as a fraction of its sides, most tiles) in order of preference."""
    ITER_EXTENT_RAYS: int = 1_000_000
    """`iterative_render` with automatic extents: the extents are those of the hits of the reference's FIRST ITERATION, 1 M
    rays (ITER_RAYS_STEP, raytracer.py:40, 1212-1262); everything after is cropped to them.  A chunk here is tens of
    millions of rays: an evenly spread sample of about this many of the first chunk's rays plays the first iteration's part
    (every k-th wave of 64 rays, `ot_detector_extent_sample`), the rest of the chunk is cropped to that extent like the
    later ones (`_render_detectors` passes the automatic extent as the crop)."""
    ITER_RENDER_ONLY: bool = True
    """`iterative_render`: every chunk but the last is traced render-only -- no section is stored, the last section of
    every ray alive behind the last surface goes to a compact `TailStorage` (56 B per LIVING ray instead of 36-48 B per
    ray and section), which the detector passes read -- wherever that gives the stored path's images: no
    orientation="Function" source, every detector position behind the last tracing surface (every feature level of the
    trace kernel has its render-only form).  False: every chunk through the ray storage."""
    ITER_LAST_RAYS: int = 1 << 20
    """... and the last chunk, whose rays stay in `self.rays` afterwards like those of the reference's last iteration
    (ITER_RAYS_STEP = 1 M there, raytracer.py:40, 1235-1267), then has this many rays."""
    ITER_MERGE_LAST: bool = True
    """... and is binned together with the render-only chunk before it: the last sections of its living rays join that chunk's
    tail (`TailStorage.append_living`), one binning pass less per render.  False: a pass of its own."""
    ITER_GROUP: int = 8
    """Detector positions per pass over a chunk's rays (at most 8: `ot_detector_images`)."""
    ITER_STORAGE_BYTES: int = 16_000_000_000
    """Ray storage of one chunk of `iterative_render` when ITER_RAYS_STEP is None: 16 GB of the 288 GB of HBM are
    19 M rays of a 15-surface scene with polarisation or 93 M rays of a two-surface scene without."""

    class INFOS(IntEnum):
        ABSORB_MISSING = 0
        TIR = 1
        ILL_COND = 2
        OUTLINE_INTERSECTION = 3
        HURB_NEG_DIR = 4

    def __init__(self, outline, n0: RefractionIndex = None, no_pol: bool = False, use_hurb: bool = False,
                 seed: int = None, **kwargs) -> None:
        self.outline = outline
        self.no_pol = no_pol
        self.use_hurb = use_hurb
        self.seed = seed  #: base seed of the device RNG (None: drawn from numpy's global RNG per trace)
        self.rays = RayStorage()
        self._msgs = np.array([])
        self.geometry_error = self._ignore_geometry_error = False   # found by the checks / tracing goes on regardless
        self.fault_pos = np.array([])        # where the geometry checks found surfaces colliding
        self._last_trace_snapshot = None     # tracing_snapshot() of the last trace
        self._scene = None
        self._scene_handle = None
        self._scene_key = None
        self._checked_key = None
        self._rays_known_current = False
        self._source_cache = None  # (key, SourceTable): device copy of the source records, reused while unchanged
        self._fast = None  # what the last full trace established, valid while base.mutation_epoch() stands still
        self._msgs_host = None
        self._kernel_ms_log = None  # a list: every trace appends its kernel's duration (HIP events of the library; bench.py)
        self._tail_book = None      # RayStorage without buffers: split, ranges and powers of render-only traces
        super().__init__(None, n0, **kwargs)
        self._new_lock = True

    # bookkeeping of the tracer itself: writing these is not a scene change (base.mutation_epoch)
    _INTERNAL = frozenset(("_msgs", "_last_trace_snapshot", "_scene", "_scene_handle", "_scene_key", "_checked_key",
                           "_rays_known_current", "_source_cache", "geometry_error", "fault_pos", "_fast",
                           "_msgs_host", "_kernel_ms_log", "_tail_book", "_lock", "_new_lock",
                           "seed"))  # (the RNG seed is read at every trace and part of nothing that is compiled or checked)

    def __setattr__(self, key: str, val: Any) -> None:
        if key in self._INTERNAL:
            self.__dict__[key] = val
            return
        if key == "outline":
            check_type(key, val, (list, np.ndarray))
            box = np.asarray_chkfinite(val, dtype=np.float64)
            if box.shape[0] != 6 or np.any(box[0::2] >= box[1::2]):
                raise ValueError("Outline needs to be specified as [x1, x2, y1, y2, z1, z2] "
                                 "with x2 > x1, y2 > y1, z2 > z1.")
            val = box
        elif key in ("no_pol", "use_hurb"):
            check_type(key, val, bool)
        super().__setattr__(key, val)

    @property
    def extent(self):
        """The outline box as a tuple."""
        return tuple(self.outline.tolist())

    @property
    def pos(self):
        """Centre of the outline's front face."""
        x0, x1, y0, y1, z0, _ = self.outline
        return 0.5 * (x0 + x1), 0.5 * (y0 + y1), z0

    def clear(self) -> None:
        super().clear()
        self.rays.__init__()

    # ---- change detection (raytracer.py:129-179) ------------------------------------------------------
    _SNAP_LISTS = (("Lenses", "lenses"), ("Filters", "filters"), ("Apertures", "apertures"), ("RaySources", "ray_sources"))

    def tracing_snapshot(self) -> dict:
        """State of everything a trace depends on, as nested lists / tuples of plain values (compared, not hashed)."""
        rays = self.rays
        snap = {"Rays": [rays.N, rays.Nt, rays.no_pol],
                "Ambient": [tuple(self.outline), self.n0.crepr()],
                "TraceSettings": [self.no_pol, self.use_hurb, self.HURB_FACTOR]}
        for key, attr in self._SNAP_LISTS:
            snap[key] = [el.crepr() for el in getattr(self, attr)]
        return snap

    def property_snapshot(self) -> dict:
        """The tracing snapshot plus the detectors (they matter to images, not to the rays)."""
        snap = self.tracing_snapshot()
        snap["Detectors"] = [det.crepr() for det in self.detectors]
        snap["Markers"], snap["Volumes"] = [], []  # plot-only elements of the reference: none here, the keys for its callers
        return snap

    def compare_property_snapshot(self, h1: dict, h2: dict) -> dict:
        """Which groups differ between two snapshots; "Any" sums up, and new lenses also count as a change of the
        ambient (their n2 takes part in it)."""
        changed = {key: h1[key] != h2[key] for key in h1}
        changed["Ambient"] |= changed["Lenses"]
        changed["Any"] = any(changed.values())
        return changed

    def check_if_rays_are_current(self) -> bool:
        last = self._last_trace_snapshot
        return last is not None and not self.compare_property_snapshot(last, self.tracing_snapshot())["Any"]

    # ---- messages (raytracer.py:181-244) ----------------------------------------------------------------
    _EVENT_TEXT = {
        "TIR": "with total inner reflection at surface {s} ({n}), treating as absorbed.",
        "ABSORB_MISSING": "missing lens surface {s} ({n}), set to absorbed",
        "ILL_COND": "are ill-conditioned for numerical hit finding at surface {s} ({n}). "
                    "Where and whether they intersect might be wrong.",
        "OUTLINE_INTERSECTION": "hitting outline after surface {s} ({n}), set to absorbed.",
        "HURB_NEG_DIR": "have negative z-direction after ray bending at surface {s} ({n}), set to absorbed.",
    }

    def _surface_names(self) -> list:
        """Names of the ray sections in tracing order: the source, every tracing surface by z, the outline."""
        labelled = []
        for kind, group in (("Lens", self.lenses), ("Aperture", self.apertures), ("Filter", self.filters)):
            for k, el in enumerate(group):
                tag = f"{kind} {el.abbr}{k}"
                if el.has_back():
                    labelled += [(el.front.pos[2], f"front surface of {tag}"), (el.back.pos[2], f"back surface of {tag}")]
                else:
                    labelled.append((el.pos[2], f"surface of {tag}"))
        labelled.sort(key=lambda item: item[0])
        return ["RaySource"] + [name for _, name in labelled] + ["Outline"]

    def _show_messages(self, N) -> None:
        """One warning per non-zero event counter (raytracer.py:207-244)."""
        if not global_options.show_warnings or not self._msgs.any():
            return
        names = self._surface_names()
        for kind, sec in zip(*np.nonzero(self._msgs)):
            count = self._msgs[kind, sec]
            where = names[sec] if sec < len(names) else "?"
            warning(f"{count} rays ({100*count/N:.3g}% of all rays) "
                    + self._EVENT_TEXT[self.INFOS(kind).name].format(s=sec, n=where))

    # ---- geometry checks (raytracer.py:510-664) ---------------------------------------------------------
    def _pretrace_check(self, N: int, snap: dict = None) -> bool:
        check_type("N", N, int)
        if N < 1:
            raise ValueError(f"Ray number N needs to be at least 1, but is {N}.")
        # the checks sample every surface pair on a 100 x 100 grid (host NumPy); skip them while nothing that
        # they depend on changed since they last passed (chunked rendering calls trace() many times)
        key = self._geometry_key(snap)
        if self._checked_key is None or key != self._checked_key or self.geometry_error:
            self._geometry_checks()
            self._checked_key = key if not self.geometry_error else None
        if self.geometry_error and not self._ignore_geometry_error:  # (flag of the reference's test suite)
            warning("ABORTED TRACING")
            return True
        return False

    def _geometry_key(self, snap: dict = None):
        snap = dict(self.tracing_snapshot() if snap is None else snap)
        snap.pop("Rays", None)
        return repr(snap)

    def _geometry_checks(self) -> None:
        """Scene validation (checks.py): sets `geometry_error` / `fault_pos` and warns about the first finding."""
        finding = _checks.find_geometry_error(self, tracing_elements(self))
        self.geometry_error = finding is not None
        if finding is not None:
            text, where = finding
            warning(text)
            if where is not None:
                self.fault_pos = where

    @staticmethod
    def check_collision(front, back, res: int = 100):
        """Is `front` anywhere behind `back` where both are defined?  -> (collision, x, y, z) of the offending sample
        positions on a res x res grid (raytracer.py:580-664); see checks.collision_points."""
        return _checks.collision_points(front, back, res)

    # ---- scene upload --------------------------------------------------------------------------------------
    def _compile(self, snap: dict = None) -> CompiledScene:
        """Flatten the scene and upload its tables; reused while the tracing-relevant state is unchanged."""
        lib = _capi.load_library()
        key = self._geometry_key(snap)
        if self._scene is not None and self._scene_handle is not None and key == self._scene_key:
            return self._scene
        self._release_scene()
        self._scene = CompiledScene(self)
        self._check_media()
        handle = C.c_void_p()
        _capi.check(lib.ot_scene_create(C.byref(self._scene.desc), C.byref(handle)))
        self._scene_handle = handle
        self._scene_key = key
        return self._scene

    def _check_media(self) -> None:
        """RefractionIndex.__call__ raises for n < 1 (refraction_index.py:165-167) and for wavelengths outside a
        tabulated range (:152-156); the tracing kernel cannot raise per ray, so every medium is evaluated once per
        scene over the wavelengths the sources can emit (device evaluation through RefractionIndex.__call__)."""
        from .scene import discrete_lines
        lines = discrete_lines(self.ray_sources)
        wl = lines.astype(np.float64) if lines is not None else np.linspace(*global_options.wavelength_range, 201)
        seen = set()
        for n in [self.n0] + [L.n for L in self.lenses] + [L.n2 for L in self.lenses if L.n2 is not None]:
            if id(n) not in seen:
                seen.add(id(n))
                n(wl)

    def _release_scene(self) -> None:
        if self._scene_handle is not None and self._scene_handle.value:
            _capi.load_library().ot_scene_destroy(self._scene_handle)
        self._scene_handle = None
        self._scene_key = None

    def __del__(self):
        try:
            self._release_scene()
        except Exception:
            pass

    # ---- tracing (raytracer.py:262-415) -----------------------------------------------------------------
    def _structure(self) -> tuple:
        """Identity of everything in the tracing lists (list edits do not pass through __setattr__)."""
        return (tuple(map(id, self.lenses)), tuple(map(id, self.apertures)), tuple(map(id, self.filters)),
                tuple(map(id, self.ray_sources)))

    def _scene_unchanged(self) -> bool:
        """Is everything the last full trace established (checks passed, compiled scene, source table) still valid?  One
        integer comparison, the identities of the list members and the bytes of the few small writeable arrays."""
        fast = self._fast
        return (fast is not None and fast[0] == _base.mutation_epoch() and fast[1] == self._structure()
                and not self.geometry_error and all(a.tobytes() == b for a, b in fast[5]))

    def trace(self, N: int, _initial_rays: tuple = None, _hurb_normals: np.ndarray = None, _N_list=None,
              _chunk: int = 0, _power_scale: float = 1.0, _tail: TailStorage = None, _tail_room: int = 0) -> None:
        """Trace N rays through the current geometry.

        Geometry errors are reported as warnings and set `geometry_error` instead of raising, like the
        reference.  `_initial_rays` = (p, s, pols, w, wl), `_N_list` (rays per source) and `_hurb_normals`
        (2*n_hurb, N) inject recorded inputs for parity tests; normally rays are generated inside the
        tracing kernel and the remainder of the per-source split is drawn at random like the reference does.
        `_power_scale`: share of the source powers these rays carry (a rank's shard, distributed.py).
        `_tail`: render-only trace (`iterative_render`, every chunk but the last): no section is stored and `self.rays`
        stays as it is; the last section of every ray still alive behind the last surface goes to that `TailStorage`
        (`ot_generate_and_trace_tail`).  Counters and warnings as for a stored trace.  `_tail_room`: slots for that many
        further rays (`TailStorage.append_living`).

        The reference re-reads the whole object graph on every call (raytracer.py:246-278).  Here everything
        derived from it -- geometry checks, the compiled scene, the source table, the snapshot -- is kept while
        nothing changed: every tracked object reports assignments to a global counter (base.mutation_epoch), so
        an unchanged scene is recognised by one integer comparison, the identities of the list members and the bytes of
        the few small arrays that are still writeable (large ones are read-only or switch the shortcut off).
        """
        if _tail is not None and _initial_rays is not None:
            raise ValueError("a render-only trace generates its rays on the device")
        fast = self._fast
        if self._scene_unchanged():
            check_type("N", N, int)
            if N < 1:
                raise ValueError(f"Ray number N needs to be at least 1, but is {N}.")
            _, _, snap, scene, splits, _ = fast
            snap = dict(snap)
            writeable = False
        else:
            self._fast = fast = None
            _base._SAW_WRITEABLE[0] = False
            _base._WATCH[0] = [self.outline] if self.outline.flags.writeable else []
            snap = self.tracing_snapshot()  # taken once: geometry-check key, scene key and the post-trace record
            writeable = _base._SAW_WRITEABLE[0]  # large arrays that can still change in place: no shortcut next time
            # small arrays that can: their bytes are compared before every shortcut (assignments move the counter, an
            # edit in place like `RT.outline[5] += 1` or `RS.s[0] = 0.1` does not)
            watch = [(a, a.tobytes()) for a in _base._WATCH[0]]
            _base._WATCH[0] = None
            if self._pretrace_check(N, snap):
                return
            scene = self._compile(snap)
            splits = {}
        lib = _capi.load_library()
        dev = require_device()

        nt = scene.nt
        if _tail is None and self.rays.storage_size(N, nt, self.no_pol) > self.MAX_RAY_STORAGE_RAM:
            raise RuntimeError(f"More than {self.MAX_RAY_STORAGE_RAM*1e-9:.1f} GB RAM requested. Either decrease"
                               " the number of rays, surfaces or do an iterative render. If your system can handle"
                               " more RAM usage, increase the Raytracer.MAX_RAY_STORAGE_RAM parameter.")

        split = splits.get(N)
        if split is None:
            split = splits[N] = RayStorage.split_rays(self.ray_sources, N)
        rng = None  # draws the remainder of the split: a seeded tracer repeats it with the seed
        if self.seed is not None and split[1]:
            rng = np.random.RandomState((int(self.seed) + 1000003 * int(_chunk)) % 2 ** 32)
        rays_obj = self.rays
        if _tail is not None:  # the split, the ranges and the powers live in a storage object without buffers
            if self._tail_book is None:
                self._tail_book = RayStorage()
            rays_obj = self._tail_book
        # unchanged sources and the same deterministic split as last time: the same range records
        rays_obj.init(self.ray_sources, N, nt, self.no_pol, _N_list=_N_list, _rng=rng, _power_scale=_power_scale,
                      _split=split, _keep_ranges=fast is not None, _alloc=_tail is None)
        rays = rays_obj._rays_struct() if _tail is None else None
        # a seeded tracer repeats itself call for call; the chunks of one iterative render must differ
        seed = int(np.random.randint(0, 2 ** 31 - 1)) if self.seed is None else int(self.seed) + 1000003 * int(_chunk)

        n_msgs = len(self.INFOS) * nt + 1
        if self._msgs_host is None or self._msgs_host.shape[0] != n_msgs:
            self._msgs_host = np.zeros(n_msgs, dtype=np.int64)
        msgs_h = self._msgs_host
        if _initial_rays is None:
            if rays_obj._has_function_orientation:  # per-trace orientation arrays: nothing to reuse
                tab = rays_obj._source_table(seed)
            else:
                cache = self._source_cache
                if fast is None or cache is None or cache[2] != _power_scale:
                    skey = (repr(snap["RaySources"]), tuple(rays_obj._powers))
                    if cache is None or cache[0] != skey:
                        self._source_cache = cache = (skey, rays_obj._source_table(), _power_scale)
                tab = cache[1]
            rng_c = rays_obj._source_ranges()
            # one synchronous call: launch, wait, counters in host memory (no device-to-host copy)
            if _tail is not None:
                _tail.ensure(int(lib.ot_tail_capacity(N + _tail_room)))
                mb_t, mb = mailbox()
                _capi.check(lib.ot_generate_and_trace_tail(self._scene_handle, tab.handle, rng_c, len(rng_c), seed, N,
                                                           C.byref(_tail._rays_struct()), ptr(_tail._dev["fill"]),
                                                           C.c_void_p(mb_t.data_ptr()), msgs_h.ctypes.data, stream_ptr()))
                _tail.N, _tail.alive, _tail.traced = int(mb[0]), int(mb[1]), N
            else:
                _capi.check(lib.ot_generate_and_trace_host(self._scene_handle, tab.handle, rng_c, len(rng_c), seed,
                                                           C.byref(rays), msgs_h.ctypes.data, stream_ptr()))
        else:
            rays_obj.set_initial_rays(*_initial_rays)
            hn = None
            if _hurb_normals is not None:
                rows = np.ascontiguousarray(_hurb_normals, dtype=np.float64).reshape(-1, N)
                if rays_obj._Np > N:  # one row per (aperture, component), laid out with the storage's plane stride
                    rows = np.pad(rows, ((0, 0), (0, rays_obj._Np - N)))
                hn = torch.from_numpy(np.ascontiguousarray(rows).reshape(-1)).to(dev)
            msgs = torch.zeros(n_msgs, dtype=torch.int64, device=dev)
            _capi.check(lib.ot_trace(self._scene_handle, C.byref(rays), ptr(hn), seed, ptr(msgs), stream_ptr()))
            msgs_h = msgs.cpu().numpy()  # (synchronises the stream)

        if _tail is None:
            rays_obj.lock()
        if self._kernel_ms_log is not None:  # (the caller switched the scene's event pair on: ot_scene_set_timing)
            ms = C.c_double()
            _capi.check(lib.ot_scene_last_trace_ms(self._scene_handle, C.byref(ms)))
            self._kernel_ms_log.append(ms.value)
        if msgs_h[-1]:
            raise TimeoutError("Timeout after 200 iterations in hit finding. Try retracing.")
        self._msgs = msgs_h[:-1].reshape(len(self.INFOS), nt).astype(int)
        if self._msgs[self.INFOS.HURB_NEG_DIR, 0]:  # generation reports directions with s_z <= 0 in this cell
            self._msgs[self.INFOS.HURB_NEG_DIR, 0] = 0
            raise RuntimeError("All ray divergences s need to be in positive z-divergence")
        self._show_messages(N)
        if _tail is None:  # (a render-only trace leaves `self.rays` and what is known about them alone)
            snap["Rays"] = [rays_obj.N, rays_obj.Nt, rays_obj.no_pol]
            self._last_trace_snapshot = snap
        if fast is None and not writeable and _initial_rays is None and not rays_obj._has_function_orientation:
            # read the counter last: objects this call created itself (the end aperture of the element list) count too
            self._fast = (_base.mutation_epoch(), self._structure(), snap, scene, splits, watch)

    # ---- detector (raytracer.py:881-1098) ----------------------------------------------------------------
    # ---- shared argument checks of the post-processing entry points ------------------------------------------
    def _need_rays(self) -> None:
        if not self.rays.N:
            raise RuntimeError("No rays traced.")

    def _need_current(self, hint: str = "") -> None:
        """The stored rays must belong to the present geometry (raytracer.py:903-913)."""
        if self._rays_known_current or self.check_if_rays_are_current():
            return
        raise RuntimeError(f"Tracing geometry/properties changed{hint}. Please retrace first.")

    def _ray_range(self, source_index) -> tuple:
        """[first, end) of the rays of one source, or of all rays for None."""
        if source_index is None:
            return 0, int(self.rays.N)
        if not 0 <= source_index < len(self.ray_sources):
            raise IndexError("Invalid source_index.")
        return int(self.rays.B_list[source_index]), int(self.rays.B_list[source_index + 1])

    def _detector_label(self, detector_index: int) -> str:
        det = self.detectors[detector_index]
        title = f": {det.desc}" if det.desc else ""
        return f"{Detector.abbr}{detector_index}{title} at z = {det.pos[2]:.5g} mm"

    def _warn_ill(self, ill_count: int, detector_index: int) -> None:
        if ill_count:
            warning(f"{ill_count} rays ({100*ill_count/self.rays.N:.3g}% of all rays) were ill-conditioned for "
                    f"numerical hit finding at detector {detector_index}. "
                    "Where and whether they intersect might be wrong.")

    def _detector_requests(self, specs: list, rays=None, no_rays: bool = False) -> list:
        """Checks of `_hit_detector` (raytracer.py:897-920) and, per spec, everything the device calls need:
        dicts with Ns, Ne (ray range), surf_desc, projection (name), crop (user extent or None), desc, centre.
        `rays`: a `TailStorage` to take the rays from instead of `self.rays` (all its slots, no per-source ranges);
        `no_rays`: no storage yet (`_plan_renders` before the first trace): nothing about rays is checked or recorded."""
        if not self.detectors:
            raise RuntimeError("Detector Missing")
        if rays is None and not no_rays:
            self._need_rays()
        ranges = []
        for sp in specs:  # all indices first: nothing is moved or launched for a bad request
            if no_rays:  # (a plan made before the rays exist: it ranges over the whole of whatever storage it is launched on)
                ranges.append((0, 0))
            else:
                ranges.append(self._ray_range(sp.get("source_index")) if rays is None else (0, int(rays.N)))
            if not 0 <= sp.get("detector_index", 0) < len(self.detectors):
                raise IndexError("Invalid detector_index.")
        if rays is None and not no_rays:
            self._need_current()

        out = []
        for sp, (Ns, Ne) in zip(specs, ranges):
            k = sp.get("detector_index", 0)
            det = self.detectors[k]
            if sp.get("pos") is not None:
                # moving a detector changes nothing a trace depends on: the change counter (base.mutation_epoch) stays
                # where it was, so that the next trace of an iterative render still takes its shortcut -- unless the
                # detector shares its surface object with a tracing element (then the move IS a change of the scene)
                epoch = _base._EPOCH[0]
                det.move_to(sp["pos"])
                if all(det.surface is not ts for ts in self.tracing_surfaces):
                    _base._EPOCH[0] = epoch
            dsurf = det.surface

            method = sp.get("projection_method", "Equidistant")
            projection = None
            if isinstance(dsurf, SphericalSurface) and method is not None:
                if method not in SphericalSurface.sphere_projection_methods:
                    raise ValueError(f"Invalid projection_method {method}, "
                                     f"must be one of {SphericalSurface.sphere_projection_methods}.")
                projection = method

            extent = sp.get("extent")
            if not (extent is None or isinstance(extent, (list, np.ndarray))):
                raise ValueError(f"Invalid extent '{extent}'.")
            crop = None if extent is None else np.asarray_chkfinite(np.array(extent, dtype=np.float64), dtype=np.float64)
            out.append(dict(Ns=Ns, Ne=Ne, surf_desc=dsurf._desc(), projection=projection, crop=crop,
                            desc=self._detector_label(k), centre=det.pos[:2].repeat(2),
                            want_z=bool(sp.get("want_z", False))))
        return out

    def _hit_detectors(self, info: str, specs: list, _reqs: list = None) -> list:
        """Device hit search for several (detector, position) pairs in one pass over the ray sections.

        specs: dicts with detector_index, source_index, extent, projection_method and optionally pos (the detector is
        moved there first, as `iterative_render` does position by position, raytracer.py:1244).
        -> per spec (ph, hw, wl, extent_out, projection, ill_count, desc): device tensors of the selected ray range; ph
        holds the x and y planes, the z plane as well with want_z
        (dense: rays without a valid hit carry weight 0), the extent actually used, the projection name, the
        ill-conditioned count and the image description at that position."""
        reqs = self._detector_requests(specs) if _reqs is None else _reqs  # (`_reqs`: formed by the caller already)
        groups: dict = {}  # ray range -> requests (one launch per range and at most 8 detectors)
        for n, rq in enumerate(reqs):
            # rays outside a user extent are dropped (raytracer.py:1036-1040): the hit kernel gives them weight 0
            groups.setdefault((rq["Ns"], rq["Ne"]), []).append(
                (n, dict(surf_desc=rq["surf_desc"], want_extent=rq["crop"] is None,
                         projection=_capi.PROJECTIONS[rq["projection"]], crop=rq["crop"], want_z=rq["want_z"],
                         compact=bool(specs[n].get("compact", False)),
                         weights_only=bool(specs[n].get("weights_only", False)))))

        out = [None] * len(specs)
        for (Ns, Ne), part_all in groups.items():
            for b in range(0, len(part_all), 8):
                part = part_all[b:b + 8]
                res = _detector.detector_hits_multi(self.rays, Ns, Ne - Ns, [r for _, r in part])
                for (n, _), one in zip(part, res):
                    ph, hw, ext4, ill_count = one[:4]
                    rq = reqs[n]
                    extent_out = rq["crop"]
                    if extent_out is None:
                        extent_out = ext4.copy() if np.all(np.isfinite(ext4)) else rq["centre"]
                    wl = self.rays._dev["wl"][Ns:Ne]
                    if len(one) > 4:  # compact list: its own wavelengths, and the fill counts ride along with them
                        wl = one[4]
                    out[n] = (ph, hw, wl, extent_out, rq["projection"], ill_count, rq["desc"])
        return out

    def _render_detectors(self, specs: list, limits: list, into: list = None, rays=None, weight_scale: float = 1.0,
                          **kwargs) -> list:
        """Detector images whose extents are known beforehand (every spec carries a user extent, or the automatic
        one of `_auto_extents` as "auto_extent"): hit search and
        binning in ONE pass over the ray sections (`ot_detector_images`), up to 8 images per pass; the hit positions
        are never written to memory.  into: per spec a (Ny, Nx, 4) device histogram to add to, or None.
        `rays`: a `TailStorage` (render-only chunk) instead of `self.rays`; `weight_scale`: every hit's weight times this (in
        f64) before it is added -- the chunks of an iterative render bin straight into one image with rays_step / N each.
        -> RenderImages (raytracer.py:1053-1098 for each spec)."""
        plan = self._plan_renders(specs, limits, into=into, rays=rays)
        self._launch_renders(plan, rays=rays, weight_scale=weight_scale)
        images = plan["images"]
        if not kwargs.get("_dont_filter", False):
            for img in images:
                if img._limit is not None:
                    img._apply_rayleigh_filter()
        return images

    def _plan_renders(self, specs: list, limits: list, into: list = None, rays=None, whole_storage: bool = False) -> dict:
        """Everything `_render_detectors` does on the host before the launch: checks, the detectors moved to their positions,
        one `RenderImage` per spec with its fixed extent and pixel grid, the histograms (one zero-filled allocation for all
        images that have none yet) and the request records of `ot_detector_images`.  `whole_storage`: the requests range over
        whatever storage `_launch_renders` is given (no source_index) -- `iterative_render` plans once, before its first trace
        where the extents are known, and launches the plan chunk after chunk."""
        reqs = self._detector_requests(specs, rays, no_rays=whole_storage)
        images, calls, shapes = [], {}, []
        dev = require_device()
        for n, (sp, rq, limit) in enumerate(zip(specs, reqs, limits)):
            label = rq["desc"]
            if sp.get("source_index") is not None:
                label = f"Rays from RS{sp['source_index']} at {label}"
            # the image extent: the user extent, or an automatic one (`_auto_extents`); the hits are cropped to it either
            # way.  An automatic extent of ALL the hits loses none by that; one taken from a sample of the rays or agreed with
            # other ranks then treats the first chunk of an iterative render like the later ones, which are cropped to this
            # very extent (`_extent0`, raytracer.py:1262) -- without the crop the margin `_fix_extent` adds around it (and the
            # band a line-like image is widened to) would collect hits of the first chunk only
            if rq["crop"] is None:
                rq["crop"] = np.asarray(sp["auto_extent"], dtype=np.float64)
            img = RenderImage(extent=rq["crop"], projection=rq["projection"], long_desc=label)
            img._limit = limit
            img._fix_extent()
            Nx, Ny = img._pixel_counts()
            tgt = None if into is None else into[n]
            if tgt is not None and tuple(tgt.shape) != (Ny, Nx, 4):
                raise ValueError("histogram to accumulate into has the wrong shape")
            images.append(img)
            shapes.append((Ny, Nx, tgt))
        # the new histograms: ONE zero-filled allocation, sliced (six images of an iterative render: one fill kernel)
        sizes = [Ny * Nx * 4 if tgt is None else 0 for Ny, Nx, tgt in shapes]
        pool = alloc_retry(lambda: torch.zeros(sum(sizes), dtype=torch.float64, device=dev)) if sum(sizes) else None
        off = 0
        for n, (img, rq, (Ny, Nx, tgt)) in enumerate(zip(images, reqs, shapes)):
            if tgt is not None:
                hist = tgt.view(-1)
            else:
                hist = pool[off:off + sizes[n]]
                off += sizes[n]
            img._dev = hist.view(Ny, Nx, 4)
            img._host = None
            calls.setdefault(None if whole_storage else (rq["Ns"], rq["Ne"]), []).append(
                (n, dict(surf_desc=rq["surf_desc"], projection=_capi.PROJECTIONS[rq["projection"]], crop=rq["crop"],
                         extent=img.extent, Nx=Nx, Ny=Ny, hist=hist)))
        return dict(images=images, calls=calls, detector_indices=[sp.get("detector_index", 0) for sp in specs])

    def _launch_renders(self, plan: dict, rays=None, weight_scale: float = 1.0) -> None:
        """`ot_detector_images` for a plan of `_plan_renders`, on `rays` (a `TailStorage`) or `self.rays`."""
        src_rays = self.rays if rays is None else rays
        for key, part_all in plan["calls"].items():
            Ns, Ne = (0, int(src_rays.N)) if key is None else key
            if Ne <= Ns:  # (a render-only chunk none of whose rays survived)
                continue
            for b in range(0, len(part_all), 8):
                part = part_all[b:b + 8]
                for _, r in part:
                    r["weight_scale"] = weight_scale
                ills = _detector.detector_images(src_rays, Ns, Ne - Ns, [r for _, r in part])
                for (n, _), ill_count in zip(part, ills):
                    self._warn_ill(ill_count, plan["detector_indices"][n])

    def _auto_extents(self, specs: list, agree=None, sample_rays: int = None, rays=None) -> list:
        """Automatic extents (raytracer.py:1042-1049) of the specs without a user extent, from an extent-only pass over
        the ray sections (no hit list: `detector.detector_extents`, up to 8 detectors per pass).  `agree`: callable
        mapping the (n, 4) array of raw extents (+-inf where no ray hits) to the one every rank uses
        (distributed.py); an extent no ray reaches collapses to the detector centre.  `sample_rays`: the extents of
        about that many evenly spread rays instead of all (ITER_EXTENT_RAYS).  -> specs with "auto_extent" set."""
        todo = [n for n, sp in enumerate(specs) if sp.get("extent") is None]
        if not todo:
            return specs
        reqs = self._detector_requests(specs, rays)
        src_rays = self.rays if rays is None else rays
        # (a render-only chunk holds the living rays of `traced` generated ones: the sample keeps the stride that
        # `sample_rays` of the GENERATED rays would have)
        traced = None if rays is None else int(rays.traced)
        raw = np.empty((len(todo), 4), dtype=np.float64)
        raw[:] = [np.inf, -np.inf, np.inf, -np.inf]
        groups: dict = {}
        for m, n in enumerate(todo):
            rq = reqs[n]
            count, proj = rq["Ne"] - rq["Ns"], _capi.PROJECTIONS[rq["projection"]]
            if count < 1:
                continue
            n_gen = count if traced is None else traced
            if sample_rays and n_gen >= 2 * sample_rays and _detector.auto_image_supported(rq["surf_desc"], proj):
                raw[m] = _detector.detector_extent_sample(src_rays, rq["Ns"], count, rq["surf_desc"], proj,
                                                          n_gen // sample_rays)
                continue
            groups.setdefault((rq["Ns"], rq["Ne"]), []).append(m)
        for (Ns, Ne), ms in groups.items():
            for b in range(0, len(ms), 8):
                part = ms[b:b + 8]
                res = _detector.detector_extents(src_rays, Ns, Ne - Ns, [
                    dict(surf_desc=reqs[todo[m]]["surf_desc"], projection=_capi.PROJECTIONS[reqs[todo[m]]["projection"]])
                    for m in part])
                for m, (ext4, _) in zip(part, res):
                    raw[m] = ext4
        if agree is not None:
            raw = np.asarray(agree(raw), dtype=np.float64).reshape(len(todo), 4)
        out = [dict(sp) for sp in specs]
        for m, n in enumerate(todo):
            out[n]["auto_extent"] = raw[m].copy() if np.all(np.isfinite(raw[m])) else reqs[n]["centre"]
        return out

    def _hit_detector(self, info: str, detector_index: int = 0, source_index: int = None, extent=None,
                      projection_method: str = "Equidistant"):
        """One detector: (ph, hw, wl, extent_out, projection, ill_count), see `_hit_detectors`."""
        return self._hit_detectors(info, [dict(detector_index=detector_index, source_index=source_index, extent=extent,
                                               projection_method=projection_method, want_z=True)])[0][:6]

    def _image_from_hits(self, hits: tuple, detector_index: int, source_index, limit, **kwargs) -> RenderImage:
        """Bin one hit list of `_hit_detectors` into a RenderImage."""
        xy, weights, wavelengths, extent_out, projection, ill_count, label = hits
        if source_index is not None:
            label = f"Rays from RS{source_index} at {label}"
        image = RenderImage(extent=extent_out, projection=projection, long_desc=label)
        if isinstance(wavelengths, tuple):  # compact hit list: (wavelengths, fill counts)
            wavelengths, kwargs["_fill"] = wavelengths
        image.render(xy, weights, wavelengths, limit=limit, **kwargs)
        self._warn_ill(ill_count, detector_index)
        return image

    def detector_image(self, detector_index: int = 0, source_index: int = None, extent=None,
                       limit: float = None, projection_method: str = "Equidistant", **kwargs) -> RenderImage:
        """Render the image on a detector for the traced rays (raytracer.py:1053-1098)."""
        if limit is not None and extent is not None and "_dont_filter" not in kwargs:
            warning("Using the limit parameter in combination with a user defined extent"
                    " will produce an incorrect detector image, as the rays outside the extent"
                    " are not included in the convolution calculation.")
        spec = dict(detector_index=detector_index, source_index=source_index, extent=extent,
                    projection_method=projection_method)
        # extent known: one pass, no hit positions in memory -- except for long bundles on a spherical detector whose
        # projection has a transcendental (the fused kernels would take their general form, numeric hit search included:
        # the chain over a compact list of the hits inside the extent is faster, C3 1.55 -> 1.4 ms)
        projected = (0 <= detector_index < len(self.detectors) and self.rays.N >= self.COMPACT_HITS_FROM
                     and isinstance(self.detectors[detector_index].surface, SphericalSurface)
                     and projection_method in ("Equidistant", "Equal-Area", "Stereographic"))
        if extent is not None and not kwargs.get("_unfused", False) and not projected:
            return self._render_detectors([spec], [limit], **kwargs)[0]
        # Automatic extent: hit list first, then the binning.  (Measured against an extent-only pass followed by the
        # fused kernels, `_auto_extents` + `_render_detectors`, profiles/r3/detector_full_size.txt: C4 5.7 against 5.8 ms,
        # C5 3.7 / 3.6, and slower where the fused entry point falls back to this chain anyway -- spherical detectors, C3
        # 3.0 against 1.9 ms -- or the image is point-like, C2 1.3 against 0.9 ms: the sections are read twice either
        # way.  The extent-only pass serves where hit lists would have to be kept or exchanged: the first chunk of
        # `iterative_render` and the sharded forms in distributed.py.)
        unfused = kwargs.pop("_unfused", False)
        # Long bundles, detectors with a closed-form hit: the sections are read once, the hits are sorted on a provisional
        # tile grid while their extent is found (`_auto_image_one_pass`; None: not applicable, the chain below runs)
        reqs = None
        if not unfused and self.rays.N >= self.AUTO_ONE_PASS_FROM:
            reqs = self._detector_requests([spec])  # (formed once: the chain below uses them where the one-pass form declines)
            img = self._auto_image_one_pass(spec, limit, _rq=reqs[0], **kwargs)
            if img is not None:
                return img
        # (long bundles: the hit list holds the valid hits only, gathered piece-wise -- a third of the bytes for C4)
        spec["compact"] = (extent is None or projected) and self.rays.N >= self.COMPACT_HITS_FROM
        hits = self._hit_detectors("Detector Image", [spec], _reqs=reqs)[0]
        return self._image_from_hits(hits, detector_index, source_index, limit, **kwargs)

    @staticmethod
    def _auto_grid(e0, limit, projection, margins):
        """Provisional tile grid for an image whose final extent E contains the sample extent e0 (`_auto_image_one_pass`).
        Tiles are 60 x 60 pixels of e0's OWN image (after `RenderImage._fix_extent` with `limit`); E's image has pixels at
        least that large, because its sides are at least e0's and its pixel counts at most those assumed here (the long
        side's count is taken for a side ratio 20 % beyond e0's, `RenderImage._pixel_counts` snaps at 2 and 4).
        -> ((X0, Y0, tile_w, tile_h, tiles_x, tiles_y), tile_w, tile_h), or None for point- / line-like and very
        elongated sample extents and where no margin of `margins` ((fraction of e0's sides, most tiles), ...) fits."""
        sx0, sy0 = e0[1] - e0[0], e0[3] - e0[2]
        MR, side = RenderImage.MAX_IMAGE_RATIO, RenderImage.MAX_IMAGE_SIDE
        if min(sx0, sy0) <= 0 or max(sx0, sy0) / min(sx0, sy0) > MR / 1.2:
            return None
        probe = RenderImage(extent=np.array(e0, dtype=np.float64), projection=projection)
        probe._limit = limit
        probe._fix_extent()
        sx, sy = probe.s
        n_long = side * min(MR, 1 + 2 * int(1.2 * max(sx, sy) / min(sx, sy) / 2))
        Nx0, Ny0 = (n_long, side) if sx > sy else (side, n_long)
        tw, th = 60 * sx / Nx0, 60 * sy / Ny0
        for margin, most in margins:
            tx = int(np.ceil((1 + 2 * margin) * sx0 / tw)) + 1
            ty = int(np.ceil((1 + 2 * margin) * sy0 / th)) + 1
            if tx * ty <= most:  # (up to 1024 tiles: the tile kernel's faster form, two rays per thread)
                return (e0[0] - margin * sx0, e0[2] - margin * sy0, tw, th, tx, ty), tw, th
        return None

    def _auto_image_one_pass(self, spec: dict, limit, _rq: dict = None, **kwargs):
        """Image with an automatic extent (raytracer.py:1042-1049, 1053-1098) in one pass over the ray sections
        (`detector.AutoImage`, csrc/ot_detector_fused.hpp last section).  The extent E0 of the hits of a sample of the
        rays lies inside the final extent E, so the pixels of E's image are at least as large as those of E0's own
        image; tiles of 60 such pixels therefore fit 64 x 64 windows of the final grid.  -> RenderImage, or None where
        this form does not apply (detector with a numeric hit search or a sphere projection, no hit in the sample,
        point- / line-like or very elongated sample extent, too many hits outside the provisional grid): the caller
        takes the hit-list chain."""
        rq = self._detector_requests([spec])[0] if _rq is None else _rq
        Ns, count = rq["Ns"], rq["Ne"] - rq["Ns"]
        sd, proj = rq["surf_desc"], _capi.PROJECTIONS[rq["projection"]]
        if count < 1 or not _detector.auto_image_supported(sd, proj):
            return None
        e0 = _detector.detector_extent_sample(self.rays, Ns, count, sd, proj, self.AUTO_SAMPLE_STRIDE)
        if not np.all(np.isfinite(e0)):
            return None
        plan = self._auto_grid(e0, limit, rq["projection"], self.AUTO_MARGINS)
        if plan is None:
            return None
        grid, tw, th = plan

        try:
            auto = _detector.AutoImage(self.rays, Ns, count, sd, proj, grid)
        except _capi.BackendError as err:
            if getattr(err, "status", 0) == _capi.ERR_UNSUPPORTED:  # e.g. no room for the records: the chain needs less
                return None
            raise
        if auto.escaped > auto.escape_capacity or not np.all(np.isfinite(auto.extent)):
            auto.cancel()
            return None
        label = rq["desc"]
        if spec.get("source_index") is not None:
            label = f"Rays from RS{spec['source_index']} at {label}"
        img = RenderImage(extent=auto.extent.copy(), projection=rq["projection"], long_desc=label)
        img._limit = limit
        img._fix_extent()
        Nx, Ny = img._pixel_counts()
        if tw * Nx / img.s[0] > 61 or th * Ny / img.s[1] > 61:  # a tile would not fit its window (ratio snapped further)
            auto.cancel()
            return None
        hist = alloc_retry(lambda: torch.zeros(Ny * Nx * 4, dtype=torch.float64, device=require_device()))
        auto.finish(img.extent, Nx, Ny, hist)
        img._dev = hist.view(Ny, Nx, 4)
        img._host = None
        if limit is not None and not kwargs.get("_dont_filter", False):
            img._apply_rayleigh_filter()
        return img

    def detector_spectrum(self, detector_index: int = 0, source_index: int = None, extent=None,
                          **kwargs) -> LightSpectrum:
        """Spectrum of the light hitting a detector (raytracer.py:1100-1132); hit search and histogram on the GPU."""
        # long bundles: a compact list of the valid hits' weights and wavelengths (no positions: 8 B per valid hit written
        # instead of 28 B per ray, and the two histogram passes read those alone)
        compact = self.rays.N >= self.COMPACT_HITS_FROM
        spec_rq = dict(detector_index=detector_index, source_index=source_index, extent=extent,
                       projection_method="Equidistant", want_z=not compact, compact=compact, weights_only=compact)
        _, w, wl, _, _, ill_count = self._hit_detectors("Detector Spectrum", [spec_rq])[0][:6]
        if isinstance(wl, tuple):
            wl, kwargs["_fill"] = wl

        prefix = "Spectrum at " if source_index is None else f"Spectrum of RS{source_index} at "
        spec = LightSpectrum.render(wl, w, long_desc=prefix + self._detector_label(detector_index), **kwargs)
        self._warn_ill(ill_count, detector_index)
        return spec

    # ---- source side (raytracer.py:1281-1352) ---------------------------------------------------------------
    def _hit_source(self, info: str, source_index: int = 0):
        """Section-0 device views of one source's rays: ((x, y), w, wl, extent)  (raytracer.py:1281-1309)."""
        if not self.ray_sources:
            raise RuntimeError("Ray Sources Missing.")
        self._need_rays()
        Ns, Ne = self._ray_range(int(source_index))
        self._need_current()
        N, nt, d = self.rays._Np, self.rays.Nt, self.rays._dev  # (N: the plane stride of the storage)
        # element (ray r, section i, component c) of p lives at r + N * (i + nt * c); section 0 here
        xy = d["p"][Ns:Ne], d["p"][N * nt + Ns:N * nt + Ne]
        return xy, d["w"][Ns:Ne], d["wl"][Ns:Ne], self.ray_sources[source_index].extent[:4]

    def _source_label(self, source_index: int) -> str:
        source = self.ray_sources[source_index]
        title = f": {source.desc}" if source.desc else ""
        return f"{RaySource.abbr}{source_index}{title} at z = {source.pos[2]:.5g} mm"

    def source_spectrum(self, source_index: int = 0, **kwargs) -> LightSpectrum:
        """Spectrum emitted by a source, from its traced rays (raytracer.py:1311-1329)."""
        _, w, wl, _ = self._hit_source("Source Spectrum", source_index)
        return LightSpectrum.render(wl, w, long_desc="Spectrum of " + self._source_label(source_index), **kwargs)

    def source_image(self, source_index: int = 0, limit: float = None, **kwargs) -> RenderImage:
        """Image of a source's emitting area, from its traced rays (raytracer.py:1331-1352)."""
        xy, w, wl, extent = self._hit_source("Source Image", source_index)
        image = RenderImage(long_desc=self._source_label(source_index), extent=extent, projection=None)
        image.render(xy, w, wl, limit=limit, **kwargs)
        return image

    # ---- focus search (raytracer.py:1354-1640) ----------------------------------------------------------------
    focus_search_methods = ['RMS Spot Size', 'Irradiance Variance', 'Image Sharpness', 'Image Center Sharpness']

    def focus_search(self, method: str, z_start: float, source_index: int = None, return_cost: bool = False,
                     _z_samples: np.ndarray = None):
        """Find the focal position around `z_start` (raytracer.py:1463-1640).

        The search region is the gap between the tracing surfaces (or sources / outline) around z_start.
        Per ray the section crossing that gap is turned into the line ph(z) = pa + sb * z on the device
        (`ot_focus_prepare`); every cost evaluation -- the 320 samples of the cost curve as well as each step
        of SciPy's scalar optimisers -- is a chain of streaming kernels over those lines (`ot_focus_cost`).
        `_z_samples` injects the sample positions of the cost curve (parity runs against recorded reference
        samples); by default they are stratified draws from NumPy's global RNG like the reference's.
        Returns (scipy OptimizeResult, dict(pos, bounds, z, cost, N))."""
        import scipy.optimize

        z_lo, z_hi = self.outline[4:]
        if z_start < z_lo or z_start > z_hi:
            raise ValueError(f"Starting position z_start={z_start} outside raytracer"
                             f" z-outline range {self.outline[4:]}.")
        if method not in self.focus_search_methods:
            raise ValueError(f"Invalid method '{method}', should be one of {self.focus_search_methods}.")
        self._need_rays()
        n_src = len(self.rays.N_list)
        if source_index is not None and source_index < 0:
            raise IndexError(f"source_index needs to be >= 0, but is {source_index}")
        if n_src == 0 or (source_index is not None and source_index >= n_src):
            raise IndexError(f"source_index={source_index} larger than number of simulated sources "
                             f"({len(self.rays.N_list)}. "
                             "Either the source was not added or the new geometry was not traced.")
        self._need_current(" or last trace had errors")

        # the search runs in the free gap around z_start: from the last surface (or source end) before it to the
        # first surface (or the outline's far face) behind it
        gap = [self.N_EPS + max(rs.extent[5] for rs in self.ray_sources), self.outline[5] - self.N_EPS]
        for surface in self.tracing_surfaces:
            if surface.z_max > z_start:
                gap[1] = surface.z_min
                break
            gap[0] = surface.z_max
        bounds = [float(gap[0]), float(gap[1])]

        Nt = 320  # cost function sampling points
        Ns, Ne = self._ray_range(source_index)
        n = Ne - Ns

        lib = _capi.load_library()
        dev = require_device()
        pasb = torch.empty(4 * n, dtype=torch.float64, device=dev)
        w = torch.empty(n, dtype=torch.float32, device=dev)
        n_use_d = torch.empty(1, dtype=torch.int64, device=dev)
        rays = self.rays._rays_struct()
        _capi.check(lib.ot_focus_prepare(C.byref(rays), Ns, n, bounds[0] + self.N_EPS, ptr(pasb), ptr(w), ptr(n_use_d),
                                         stream_ptr()))
        N_use = int(n_use_d.item())
        if N_use < 1000:
            warning(f"WARNING: Less than 1000 rays for focus_search ({N_use}).")
        if N_use <= 1:  # nothing to focus: an empty result with the reference's keys
            nan_curve = np.full(Nt, np.nan)
            return scipy.optimize.OptimizeResult(), dict(pos=[np.nan] * 3, bounds=bounds, z=nan_curve,
                                                         cost=nan_curve.copy(), N=N_use)

        mode = self.focus_search_methods.index(method)
        # image side grows with sqrt(N) (raytracer.py:1381-1385), odd
        N_px = 100 * int(1 + np.sqrt(N_use) / 1500)
        N_px = N_px if N_px % 2 else N_px + 1
        ws = torch.empty(_capi.FOCUS_WS + N_px * N_px, dtype=torch.float64, device=dev)

        # weighted moments of the hit lines: mean line, direct RMS solution, and the whole RMS cost curve
        sums = torch.empty(16, dtype=torch.float64, device=dev)
        _capi.check(lib.ot_focus_moments(n, ptr(pasb), ptr(w), bounds[0], bounds[1], ptr(sums), stream_ptr()))
        sm = sums.cpu().numpy()

        def cost_at(zs: np.ndarray) -> np.ndarray:
            zs = np.ascontiguousarray(zs, dtype=np.float64)
            if mode == 0:
                # RMS spot size sqrt(var_x + var_y) with np.cov(aweights) normalisation (raytracer.py:1376-1379):
                # the centred second moments are quadratic in z, one pass over the rays serves every z
                dz = zs - 0.5 * (bounds[0] + bounds[1])
                fact = sm[0] - sm[7] / sm[0]
                var = (sm[8] + sm[11] + 2 * dz * (sm[9] + sm[12]) + dz ** 2 * (sm[10] + sm[13])) / fact
                return np.sqrt(np.maximum(var, 0.0))
            out = torch.empty(zs.shape[0], dtype=torch.float64, device=dev)
            _capi.check(lib.ot_focus_cost(n, ptr(pasb), ptr(w), mode,
                                          zs.ctypes.data_as(C.POINTER(C.c_double)), zs.shape[0], N_px, ptr(ws), ptr(out),
                                          stream_ptr()))
            return out.cpu().numpy()

        r = vals = None
        needs_curve = mode >= 2  # the sharpness methods start their optimiser at the best sample of the curve
        if return_cost or needs_curve:
            if _z_samples is not None:
                r = np.asarray(_z_samples, dtype=np.float64)
            else:  # random.stratified_interval_sampling(b0, b1, Nt, shuffle=False), random.py:48-67
                dba = (bounds[1] - bounds[0]) / Nt
                r = np.linspace(bounds[0], bounds[1] - dba, Nt) + np.random.uniform(0., dba, Nt)
            vals = cost_at(r)

        if mode == 0:
            # RMS spot size: direct solution, extended by ray weights (raytracer.py:1420-1460)
            z_best = -sm[6] / sm[5] if sm[5] else 0.5 * (bounds[0] + bounds[1])
            z_best = float(min(max(z_best, bounds[0]), bounds[1]))
            res = scipy.optimize.OptimizeResult(x=z_best, fun=float(cost_at(np.array([z_best]))[0]))
        else:
            def scalar_cost(z, *_):
                return float(cost_at(np.array([z[0]]))[0])

            # same optimisers and limits as the reference: Nelder-Mead from the middle of the gap for the irradiance
            # variance, COBYLA from the best curve sample for the sharpness methods
            if mode == 1:
                start, solver, steps = 0.5 * (bounds[0] + bounds[1]), "Nelder-Mead", 100
            else:
                start, solver, steps = r[int(np.argmin(vals))], "COBYLA", 30
            res = scipy.optimize.minimize(scalar_cost, start, method=solver, bounds=[bounds], tol=None, callback=None,
                                          options=dict(maxiter=steps))
            res.x = res.x[0]

        rrl = (res.x - bounds[0]) < 10 * (bounds[1] - bounds[0]) / Nt
        rrr = (bounds[1] - res.x) < 10 * (bounds[1] - bounds[0]) / Nt
        if rrl or rrr:
            warning("Found minimum near search bounds, "
                    "this can mean the focus is outside of the search range.")

        # weighted mean ray position at the focus; the z component of pa + sb * z is z itself
        pos = (float((sm[1] + sm[3] * res.x) / sm[0]), float((sm[2] + sm[4] * res.x) / sm[0]), float(res.x))
        if not return_cost:
            r = vals = None
        return res, {"pos": pos, "bounds": bounds, "z": r, "cost": vals, "N": N_use}

    # ---- iterative rendering (raytracer.py:1134-1279) -------------------------------------------------------
    def _render_only_applies(self, detector_index: list, pos: list) -> bool:
        """May the chunks of an iterative render be traced without storing their sections (`ITER_RENDER_ONLY`)?  A
        detector sees a ray through the section that crosses it (raytracer.py:929-985); behind the last tracing surface
        that is the ray's last section, which is all a render-only trace keeps."""
        if any(rs.orientation == "Function" for rs in self.ray_sources):  # (their generation needs a position pre-pass)
            return False
        z_last = max([surf.z_max for surf in self.tracing_surfaces] + [rs.extent[5] for rs in self.ray_sources])
        for k, p in zip(detector_index, pos):
            if not 0 <= k < len(self.detectors):  # (reported by the detector pass, after the trace like the reference)
                return False
            surf = self.detectors[k].surface
            if float(p[2]) + (surf.z_min - surf.pos[2]) <= z_last + self.N_EPS:
                return False
        return True

    def _chunk_plan(self, N: int, n_sec: int, render_only: bool) -> list:
        """Ray counts of the chunks of an iterative render of N rays (raytracer.py:1216-1217, 1238-1239): with
        ITER_RAYS_STEP set, the reference's rule; otherwise chunks sized by storage -- with render-only chunks: as large as
        ITER_STORAGE_BYTES of tail storage and one launch allow, then one stored chunk of ITER_LAST_RAYS."""
        rays_step = self.ITER_RAYS_STEP  # None: sized by storage
        if rays_step is not None:  # the reference's rule (raytracer.py:1216-1217, 1238-1239)
            iterations = max(N // rays_step, 1)
            chunks = [rays_step] * (iterations - 1) + [N - (iterations - 1) * rays_step]
        elif render_only and N > self.ITER_LAST_RAYS:
            # one stored chunk of ITER_LAST_RAYS at the end; before it render-only chunks as large as ITER_STORAGE_BYTES
            # of tail storage (60 B per ray at most) and one launch (2^28 rays) allow
            rest = N - self.ITER_LAST_RAYS
            most = min(1 << 28, self.ITER_STORAGE_BYTES // 60) // 1024 * 1024
            k = -(-rest // most)
            step = -(-(-(-rest // k)) // 1024) * 1024
            chunks = [step] * (k - 1) + [rest - (k - 1) * step] + [self.ITER_LAST_RAYS]
        else:
            # chunk = what ITER_STORAGE_BYTES of ray storage hold (the reference's 1 M rays are sized for a few GB of
            # host RAM, raytracer.py:40); chunks of equal size, at least 1 M rays
            step_max = max(1_000_000, self.rays.max_rays_for_size(self.ITER_STORAGE_BYTES, n_sec, self.no_pol))
            iterations = -(-N // step_max)
            rays_step = -(-N // iterations)
            if iterations > 1:
                # a multiple of 1024 rays: the planes of the storage then start on 128-byte lines.  With an odd count
                # every wave's store straddles a line it shares with its neighbour (C4: 66 666 667 rays traced in 4.8
                # ms, 66 666 688 in 3.4 ms); the last chunk takes what is left
                rays_step = -(-rays_step // 1024) * 1024
            chunks = [rays_step] * (iterations - 1) + [N - (iterations - 1) * rays_step]
        return chunks

    def iterative_render(self, N, detector_index=0, limit=None, projection_method="Equidistant", pos=None,
                         extent=None, _power_scale: float = 1.0, _agree_extents=None, _finish: bool = True) -> list:
        """Render detector images from N rays traced in chunks of ITER_RAYS_STEP; images of all chunks
        are summed (the extent of the first chunk fixes later ones)."""
        if not self.ray_sources:
            raise RuntimeError("Ray Source(s) Missing.")
        if not self.detectors:
            raise RuntimeError("Detector(s) Missing.")
        if (N := int(N)) <= 0:
            raise ValueError(f"Ray number N_rays needs to be a positive int, but is {N}.")

        # one image per detector position: scalars are repeated, lists must have one entry per position
        if pos is None:
            if isinstance(detector_index, list):
                raise ValueError("detector_index list needs to have the same length as pos list")
            pos = [np.array(self.detectors[detector_index].pos)]
        elif not isinstance(pos[0], (list, np.ndarray)):
            pos = [pos]
        n_img = len(pos)

        def per_image(value, name, is_list):
            if not is_list:
                return [value] * n_img
            if len(value) != n_img:
                raise ValueError(f"{name} list needs to have the same length as pos list")
            return list(value)

        detector_index = per_image(detector_index, "detector_index", isinstance(detector_index, list))
        limit = per_image(limit, "limit", isinstance(limit, list))
        projection_method = per_image(projection_method, "projection_method", isinstance(projection_method, list))
        # (an extent is itself a list of numbers: only a list of extents counts as one)
        extentc = per_image(extent, "extent", isinstance(extent, list) and not isinstance(extent[0], (int, float)))

        n_sec = len(self.tracing_surfaces) + 2
        if not self._scene_unchanged() and self._pretrace_check(min(N, 1000)):
            raise RuntimeError("Geometry checks failed. Tracing aborted. Check the warnings.")
        # Only the rays of the LAST chunk stay in the tracer (raytracer.py:1235-1267).  Every chunk before it is traced
        # render-only where the scene and the detector positions allow it: no section is stored, the living rays' last
        # sections go to a compact storage the detector passes read (`trace(_tail=...)`, TailStorage)
        render_only = self.ITER_RENDER_ONLY and self._render_only_applies(detector_index, pos)
        chunks = self._chunk_plan(N, n_sec, render_only)
        tail = TailStorage() if render_only and len(chunks) > 1 else None

        nt = n_sec
        msgs_cum = np.zeros((len(self.INFOS), n_sec), dtype=int)

        # Up to 8 positions are intersected in one pass over the sections, hit search and binning fused (`ot_detector_images`);
        # the extents are given by the caller or fixed by the first chunk (raytracer.py:1262).  The reference scales every
        # chunk's image by rays_step / N and adds it (raytracer.py:1257-1264); here the factor rides with the weights into the
        # binning (`weight_scale`, applied in f64), so that every chunk is binned straight into the one image of its position:
        # no second set of histograms, no passes over them.  The host side of a group of positions -- detectors moved, images
        # with their grids, one zero-filled allocation, the request records -- is a PLAN made once (`_plan_renders`): before
        # the first trace where the caller gave the extents, right after it otherwise, and launched chunk after chunk (the
        # gap between a trace and its binning was 0.15-0.3 ms of host work per chunk: a fifth of a rank's time when 2e8 rays
        # are sharded over eight GPUs).
        groups = [list(range(j0, min(j0 + self.ITER_GROUP, len(pos)))) for j0 in range(0, len(pos), self.ITER_GROUP)]
        plans = [None] * len(groups)
        images = [None] * len(pos)

        def specs_of(group):
            return [dict(detector_index=detector_index[j], extent=extentc[j], projection_method=projection_method[j],
                         pos=pos[j]) for j in group]

        def make_plan(gi, specs):
            plans[gi] = self._plan_renders(specs, [limit[j] for j in groups[gi]], whole_storage=True)
            for g, j in enumerate(groups[gi]):
                images[j] = plans[gi]["images"][g]
                extentc[j] = images[j]._extent0

        for gi, group in enumerate(groups):
            if all(extentc[j] is not None for j in group):
                make_plan(gi, specs_of(group))

        # The LAST chunk goes through the ray storage (its rays stay in the tracer), but not through a binning pass of its
        # own: it is traced FIRST, the last sections of its living rays later join the tail of the final render-only chunk
        # (weights scaled to that chunk's rays: x n_last / n_i), and the two are binned together -- for 2^20 rays the fixed
        # costs of the tile chain were most of the pass (0.3-0.4 ms per render).  Automatic extents are those of this
        # stored chunk's hits (the reference takes its first iteration's, 1 M rays as well, raytracer.py:1262): its rays
        # have the order of generation, so a seeded render finds the same extents every time -- the slots of a tail
        # storage are filled in the order the waves happen to finish
        merged = tail is not None and self.ITER_MERGE_LAST

        def bin_chunk(n_i, src):
            self._rays_known_current = True  # traced a moment ago: skip the snapshot comparison per image
            try:
                for gi, group in enumerate(groups):
                    if plans[gi] is None:  # automatic extents: those of this, the first chunk (an extent-only pass or a sample)
                        make_plan(gi, self._auto_extents(specs_of(group), agree=_agree_extents,
                                                         sample_rays=self.ITER_EXTENT_RAYS, rays=src))
                    if n_i:
                        self._launch_renders(plans[gi], rays=src, weight_scale=n_i / N)
            finally:
                self._rays_known_current = False

        if merged:
            n_last = chunks[-1]
            with global_options.no_warnings():
                self.trace(N=n_last, _chunk=len(chunks) - 1, _power_scale=_power_scale)
                msgs_cum += self._msgs
            bin_chunk(0, None)  # (plans with automatic extents only; nothing is binned yet)
        for i, n_i in enumerate(chunks):  # one chunk of rays per iteration (raytracer.py:1235-1267)
            last = i == len(chunks) - 1
            if merged and last:
                break  # (traced at the start, binned with the chunk before it)
            src = tail if (tail is not None and not last) else None
            with global_options.no_warnings():
                if merged and i == len(chunks) - 2:
                    self.trace(N=n_i, _chunk=i, _power_scale=_power_scale, _tail=src, _tail_room=n_last + 64)
                    msgs_cum += self._msgs
                    src.append_living(self.rays, n_last / n_i)
                else:
                    self.trace(N=n_i, _chunk=i, _power_scale=_power_scale, _tail=src)
                    msgs_cum += self._msgs
            bin_chunk(n_i, src)
        if tail is not None:
            tail.release()

        self._msgs = msgs_cum
        if not _finish:  # distributed.sharded_iterative_render: histograms still on the device, summed over the ranks first
            return images
        for i, img in enumerate(images):
            img._sync_host()
            if limit[i] is not None:
                img._apply_rayleigh_filter()

        self._show_messages(N)
        return images
