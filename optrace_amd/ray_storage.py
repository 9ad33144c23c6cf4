"""RayStorage: all ray sections of a trace, resident in HBM.

Mirror of optrace/tracer/ray_storage.py:11-293.  The arrays keep the reference's shapes, dtypes and
Fortran (struct-of-arrays) order -- p_list (N, nt, 3) f64, s0_list (N, 3) f64, w_list (N, nt) f32,
n_list (N, nt) f64, wl_list (N,) f32, pol_list (N, nt, 3) f32 -- but live in device memory
(torch tensors); the NumPy attributes of the reference API are read-only host copies made on first use.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from .base import BaseClass
from ._device import require_device, stream_ptr, alloc_retry, ptr, mailbox
from ._warn import warning
from . import misc


class SourceTable:
    """Device copy of the `ot_source` table of a list of RaySources (handle from ot_sources_create)."""

    _ptr_fields = ("spec_tab", "pol_tab", "div_tab", "img_pdf", "img_rgb")

    def __init__(self, sources: list, powers: list, s_or: dict = None) -> None:
        """`s_or`: source index -> flat device tensor x[n] | y[n] | z[n] of base orientations (orientation
        "Function"); sources of that kind without an entry emit along their constant `s` (position pre-pass)."""
        lib = _capi.load_library()
        require_device()
        self._keep = [s_or]
        arr = (_capi.Source * len(sources))()
        for j, (rs, power) in enumerate(zip(sources, powers)):
            f = rs._source_fields()
            s = arr[j]
            for key, val in f.items():
                if key in self._ptr_fields:
                    a = np.ascontiguousarray(val, dtype=np.float64)
                    self._keep.append(a)
                    setattr(s, key, a.ctypes.data_as(C.POINTER(C.c_double)))
                elif isinstance(val, list):
                    getattr(s, key)[:] = val
                else:
                    setattr(s, key, val)
            s.power = float(power)
            if s_or and j in s_or:
                s.s_or, s.n_or = s_or[j].data_ptr(), s_or[j].shape[0] // 3
        self.handle = C.c_void_p()
        _capi.check(lib.ot_sources_create(arr, len(sources), C.byref(self.handle)))
        self._lib = lib

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            self._lib.ot_sources_destroy(self.handle)
            self.handle = C.c_void_p()


class TailStorage:
    """Render-only chunks of `Raytracer.iterative_render` (raytracer.py:1235-1267 keeps the rays of the last chunk only):
    the LAST SECTION of every ray that is alive behind the last surface -- positions at sections nt - 2 and nt - 1, the
    weight at nt - 2, the wavelength --, gathered by `ot_generate_and_trace_tail` into a compact device storage with the
    layout of a two-section `RayStorage` (include/optrace_amd.h).  The detector entry points read it like any other
    storage (`_rays_struct`, `N` slots in use; slots without a ray carry weight 0); the order of the rays is not the
    order of generation and there are no per-source ranges."""

    def __init__(self) -> None:
        self._dev = None   # name -> flat device tensor
        self._cap = 0      # plane stride (slots)
        self._rays_c = None
        self.N = 0         # leading slots in use after the last trace
        self.alive = 0     # living rays among them
        self.traced = 0    # rays the last trace generated

    def ensure(self, cap: int) -> None:
        """Buffers for `cap` slots (ot_tail_capacity of the chunk); kept while large enough."""
        dev = require_device()
        if self._dev is not None and self._cap >= cap and self._dev["p"].device == dev:
            return
        self._dev = self._rays_c = None
        self._dev = alloc_retry(lambda: {
            "p": torch.empty(6 * cap, dtype=torch.float64, device=dev),
            "w": torch.empty(2 * cap, dtype=torch.float32, device=dev),  # (section 1 is written as 0 for every slot in use)
            "wl": torch.empty(cap, dtype=torch.float32, device=dev),
            "fill": torch.zeros(1024, dtype=torch.int32, device=dev),
        })
        self._cap = int(cap)

    def _rays_struct(self) -> _capi.Rays:
        if self._rays_c is None:
            d = self._dev
            r = _capi.Rays()
            r.N, r.nt = self._cap, 2
            r.p, r.w, r.wl = d["p"].data_ptr(), d["w"].data_ptr(), d["wl"].data_ptr()
            r.s = r.n = r.pol = None
            self._rays_c = r
        return self._rays_c

    def append_living(self, rays: "RayStorage", weight_scale: float) -> None:
        """The last sections of the living rays of a stored chunk join the tail (`ot_tail_append`), weights times
        `weight_scale`; `N`, `alive` and `traced` then count both.  The buffers must hold both (`ensure` with
        ot_tail_capacity(tail rays + chunk rays + 64) before the tail was traced)."""
        lib = _capi.load_library()
        mb_t, mb = mailbox()
        _capi.check(lib.ot_tail_append(C.byref(rays._rays_struct()), 0, int(rays.N), float(weight_scale), int(self.traced),
                                       C.byref(self._rays_struct()), ptr(self._dev["fill"]), C.c_void_p(mb_t.data_ptr()),
                                       stream_ptr()))
        self.N, self.alive, self.traced = int(mb[0]), int(mb[1]), self.traced + int(rays.N)

    def release(self) -> None:
        self._dev = self._rays_c = None
        self._cap = self.N = self.alive = 0


class RayStorage(BaseClass):

    _tracked = False  # a result container: filling it must not look like a scene change

    PAD_FROM: int = 1 << 20
    """From this many rays on the planes of the device buffers are PAD_TO elements apart at least: with a ray count that
    is not a multiple of 32 every plane would start off the 128-byte lines, every wave's store would share a line with
    its neighbour's and the trace would run 30-50 % slower (10 000 001 rays: 0.83 against 0.63 ms).  The stride is
    `_Np` >= N; the C-ABI takes it as `ot_rays.N` (counts travel separately: source ranges, first / count arguments),
    the entries behind ray N - 1 of a plane are never read.  Host views and the reference's (N, nt) shapes are
    unaffected."""
    PAD_TO: int = 128

    def __init__(self, **kwargs) -> None:
        self._lock = False
        self.N_list = np.array([], dtype=int)
        self.B_list = np.array([], dtype=int)
        self.no_pol = False
        self.ray_source_list = []
        self._N = 0
        self._Np = 0      # plane stride of the device buffers (= _N unless padded, see PAD_FROM)
        self._nt = 0
        self._dev = {}    # name -> torch tensor (flat, component-major)
        self._host = {}   # name -> cached read-only numpy view
        self._powers = []
        self._ranges = None   # ot_source_range array of the current split
        self._split_key = None
        self._rays_c = None   # ot_rays of the current buffers
        super().__init__(**kwargs)

    # ---- allocation (ray_storage.py:35-90) ---------------------------------------------------------
    def init(self, ray_source_list: list, N: int, nt: int, no_pol: bool, _single_power: float = None,
             _N_list=None, _rng=None, _power_scale: float = 1.0, _split=None, _keep_ranges: bool = False,
             _alloc: bool = True) -> None:
        """`_alloc=False`: the split, the ranges and the source powers only, no section buffers (the book-keeping of a
        render-only trace, `TailStorage`).
        `_power_scale`: the share of the sources' power this storage carries (one rank's shard of a sharded trace);
        `_split`: (N_list, dN, p) precomputed by `split_rays` for exactly these sources and N; `_keep_ranges`: the
        caller knows the sources did not change since the previous init (the range records are reused if the split is
        the same deterministic one)."""
        d = self.__dict__  # plain dict writes: this runs once per trace
        d["_lock"] = False
        d["no_pol"] = no_pol
        assert N >= 0 and nt >= 0 and len(ray_source_list)
        dev = require_device()

        # rays per source proportional to power, remainder drawn with the powers as probabilities
        N_list, dN, prob = self.split_rays(ray_source_list, N) if _split is None else _split
        if dN:
            # (a seeded tracer passes its own generator, so that the split repeats with the seed)
            index_add = (np.random if _rng is None else _rng).choice(N_list.shape[0], size=dN, p=prob)
            N_list = N_list.copy()
            np.add.at(N_list, index_add, np.ones(index_add.shape))
        if _N_list is not None:  # parity runs: the split the recorded rays were created with
            N_list = np.asarray(_N_list).astype(int)
            assert N_list.shape[0] == len(ray_source_list) and N_list.sum() == N
        if not N_list.all():
            warning("There are RaySources that have no rays assigned. "
                    "Change the power ratio or raise the overall ray number")
        d["N_list"] = N_list
        d["B_list"] = np.concatenate(([0], np.cumsum(N_list))).astype(int)
        d["ray_source_list"] = ray_source_list
        d["_powers"] = [float(_single_power or RS.power) * _power_scale for RS in ray_source_list]
        split_key = (int(N), float(_power_scale), dN == 0 and _N_list is None and _single_power is None)
        if not (_keep_ranges and split_key[2] and split_key == self._split_key):
            d["_ranges"] = None
        d["_split_key"] = split_key

        N, nt = int(N), int(nt)
        if not _alloc:
            d["_dev"], d["_rays_c"], d["_host"] = {}, None, {}
            d["_N"], d["_Np"], d["_nt"] = N, N, nt
            return
        Np = -(-N // self.PAD_TO) * self.PAD_TO if N >= self.PAD_FROM else N
        old = self._dev
        reuse = bool(old and self._N == N and self._Np == Np and self._nt == nt and (old["pol"] is None) == bool(no_pol)
                     and old["p"].device == dev)
        del old
        if not reuse:
            # (a trace with the shape of the previous one writes into the same buffers: host views already handed
            # out are copies, and nothing on the device outlives the trace that produced it)
            d["_dev"] = {}  # the previous storage goes back to the allocator before the new one is requested

            def alloc() -> dict:
                return {
                    "p": torch.empty(3 * nt * Np, dtype=torch.float64, device=dev),
                    "s": torch.empty(3 * Np, dtype=torch.float64, device=dev),
                    "w": torch.empty(nt * Np, dtype=torch.float32, device=dev),
                    "n": torch.empty(nt * Np, dtype=torch.float64, device=dev),
                    "wl": torch.empty(Np, dtype=torch.float32, device=dev),
                    "pol": None if no_pol else torch.empty(3 * nt * Np, dtype=torch.float32, device=dev),
                }
            d["_dev"] = alloc_retry(alloc)  # (out of memory: the library's kept binning scratch goes back first)
            d["_rays_c"] = None
        d["_N"], d["_Np"], d["_nt"] = N, Np, nt
        d["_host"] = {}

    @staticmethod
    def split_rays(ray_source_list: list, N: int):
        """(N_list, dN, p): floor(N * P_i / sum P) rays per source, dN rays left to be drawn with probabilities p
        (ray_storage.py:59-68)."""
        P_list = np.array([RS.power for RS in ray_source_list])
        P_all = np.sum(P_list)
        N_list = (N * P_list / P_all).astype(int)
        return N_list, int(N - np.sum(N_list)), P_list / P_all

    @staticmethod
    def storage_size(N: int, nt: int, no_pol: bool) -> int:
        """Bytes needed for N rays with nt sections (ray_storage.py:92-104)."""
        f32, f64 = 4, 8
        fpol = f32 * N * nt * 3 if not no_pol else f64
        return N * nt * 3 * f64 + N * 3 * f64 + fpol + N * nt * f32 + N * nt * f64 + N * f32

    @staticmethod
    def max_rays_for_size(size: int, nt: int, no_pol: bool) -> int:
        f32, f64 = 4, 8
        if no_pol:
            return (size - f64) // (nt * 3 * f64 + 3 * f64 + nt * f32 + nt * f64 + f32)
        return size // (nt * 3 * f64 + 3 * f64 + f32 * nt * 3 + nt * f32 + nt * f64 + f32)

    @property
    def N(self) -> int:
        return self._N if self.N_list.shape[0] else 0

    @property
    def Nt(self) -> int:
        return self._nt if self.N_list.shape[0] else 0

    # ---- device side --------------------------------------------------------------------------------
    def _rays_struct(self) -> _capi.Rays:
        if self._rays_c is not None:
            return self._rays_c
        d = self._dev
        r = _capi.Rays()
        r.N, r.nt = self._Np, self._nt  # the plane stride; how many rays there are, the ranges / counts of each call say
        r.p, r.s, r.w, r.n, r.wl = (d["p"].data_ptr(), d["s"].data_ptr(), d["w"].data_ptr(),
                                    d["n"].data_ptr(), d["wl"].data_ptr())
        r.pol = d["pol"].data_ptr() if d["pol"] is not None else None
        self.__dict__["_rays_c"] = r
        return r

    # blocks below this size are not cut further (their share of the rays, and of the time, is negligible)
    _MIN_BLOCK = 1 << 16

    def _source_ranges(self):
        """Stratification ranges of the launch.  The reference cuts every source's rays among its threads and each
        thread stratifies its own share (ray_storage.py:147-166); here a source's rays are cut into power-of-two
        blocks, largest first, plus one ragged rest: a power-of-two block's stratum permutation needs no rejection
        step (ot_generate.hpp::permute_index), and a wave runs as long as its slowest lane.  Every ray of a source
        carries power / N_source, whatever its block."""
        if self._ranges is not None:
            return self._ranges
        per_source = max(1, 64 // max(len(self.N_list), 1))  # at most 64 ranges travel as kernel arguments
        recs = []
        for i, n in enumerate(int(v) for v in self.N_list):
            first, rest = int(self.B_list[i]), n
            ray_power = self._powers[i] / n if n else 0.
            if self.ray_source_list[i].orientation != "Function":  # those read one orientation array per range
                while rest >= self._MIN_BLOCK and rest & (rest - 1) and sum(r[0] == i for r in recs) < per_source - 1:
                    blk = 1 << (rest.bit_length() - 1)
                    recs.append((i, first, blk, ray_power))
                    first, rest = first + blk, rest - blk
            recs.append((i, first, rest, ray_power))
        rng = (_capi.SourceRange * len(recs))()
        for r, (i, first, count, ray_power) in zip(rng, recs):
            r.source, r.first, r.count, r.ray_power = i, first, count, ray_power
        self.__dict__["_ranges"] = rng
        return rng

    @property
    def _has_function_orientation(self) -> bool:
        return any(rs.orientation == "Function" for rs in self.ray_source_list)

    def _function_orientations(self, seed: int) -> dict:
        """orientation="Function" (ray_source.py:272-274): `or_func` is a Python callable of the start positions.
        The start position of a ray depends on (seed, ray index, source shape) only, not on its direction, so a
        pre-pass generates the rays once with a constant orientation, the callable is evaluated on the host at
        the positions of its source's rays, and the result goes to the device as an array the generation proper
        reads its base orientations from.  -> {source index: device tensor x[n] | y[n] | z[n]}"""
        idx = [i for i, rs in enumerate(self.ray_source_list) if rs.orientation == "Function" and self.N_list[i] > 0]
        if not idx:
            return {}
        lib, dev, N = _capi.load_library(), require_device(), self._N
        pre = _capi.Rays()
        buf = {k: torch.empty(n * N, dtype=dt, device=dev) for k, n, dt in
               (("p", 3, torch.float64), ("s", 3, torch.float64), ("w", 1, torch.float32), ("n", 1, torch.float64),
                ("wl", 1, torch.float32))}
        pre.N, pre.nt = N, 1
        pre.p, pre.s, pre.w, pre.n, pre.wl = (buf[k].data_ptr() for k in ("p", "s", "w", "n", "wl"))
        pre.pol = None
        tab = SourceTable(self.ray_source_list, self._powers)
        rng = self._source_ranges()
        _capi.check(lib.ot_rays_generate(tab.handle, rng, len(rng), int(seed), 1, C.byref(pre), stream_ptr()))
        out = {}
        for i in idx:
            rs, Ns, Ne = self.ray_source_list[i], int(self.B_list[i]), int(self.B_list[i + 1])
            x, y = buf["p"][Ns:Ne].cpu().numpy(), buf["p"][N + Ns:N + Ne].cpu().numpy()
            s_or = rs.or_func(x, y, **rs.or_args)
            if not isinstance(s_or, np.ndarray) or s_or.shape != (Ne - Ns, 3):
                raise RuntimeError("or_func must return a np.ndarray of shape (N, 3).")
            out[i] = torch.from_numpy(np.ascontiguousarray(s_or.T, dtype=np.float64).reshape(-1)).to(dev)
        return out

    def _source_table(self, seed: int = None) -> SourceTable:
        """`seed`: the seed of the generation this table is for (needed by orientation="Function" sources)."""
        s_or = self._function_orientations(seed) if self._has_function_orientation else None
        return SourceTable(self.ray_source_list, self._powers, s_or)

    def generate(self, seed: int | None) -> None:
        """Fill section 0 from the sources (RaySource.create_rays on the device)."""
        lib = _capi.load_library()
        seed = int(np.random.randint(0, 2**31 - 1)) if seed is None else int(seed)
        tab = self._source_table(seed)
        rng = self._source_ranges()
        rays = self._rays_struct()
        _capi.check(lib.ot_rays_generate(tab.handle, rng, len(rng), seed, int(self.no_pol), C.byref(rays),
                                         stream_ptr()))
        torch.cuda.current_stream().synchronize()
        self._host.clear()

    def set_initial_rays(self, p, s, pols, w, wl) -> None:
        """Inject section 0 from host arrays (used for parity runs against recorded reference rays)."""
        N, Np, nt, dev = self._N, self._Np, self._nt, require_device()
        d = self._dev
        p = np.asarray(p, dtype=np.float64)
        if Np > N:  # ot_trace walks the whole stride: the padding carries dead rays (weight 0, a valid direction)
            for name in ("p", "s", "w", "wl") + (("pol",) if d["pol"] is not None else ()):
                d[name].zero_()
            d["s"][2 * Np:] = 1.0
        for c in range(3):
            d["p"][c * nt * Np: c * nt * Np + N] = torch.from_numpy(np.ascontiguousarray(p[:, c])).to(dev)
            d["s"][c * Np:c * Np + N] = torch.from_numpy(np.ascontiguousarray(np.asarray(s, dtype=np.float64)[:, c])).to(dev)
            if d["pol"] is not None:
                d["pol"][c * nt * Np: c * nt * Np + N] = torch.from_numpy(
                    np.ascontiguousarray(np.asarray(pols)[:, c], dtype=np.float32)).to(dev)
        d["w"][:N] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dev)
        d["wl"][:N] = torch.from_numpy(np.ascontiguousarray(wl, dtype=np.float32)).to(dev)
        self._host.clear()

    # ---- host views (reference attribute API) --------------------------------------------------------
    def _view(self, name: str) -> np.ndarray:
        if name not in self._host:
            N, Np, nt = self._N, self._Np, self._nt
            t = self._dev.get(name)
            if name == "pol" and t is None:
                a = np.broadcast_to(np.nan, (N, nt, 3))
            elif t is None:
                a = np.array([])
            else:
                h = t.cpu().numpy()
                shape = {"p": (3, nt, Np), "pol": (3, nt, Np), "s": (3, Np), "w": (nt, Np), "n": (nt, Np), "wl": (Np,)}[name]
                a = h.reshape(shape).transpose(*reversed(range(len(shape))))  # F-ordered (N, nt, 3) view
                if Np > N:  # padded planes: the rays' part, packed
                    a = np.asfortranarray(a[:N])
                a.flags.writeable = False
            self._host[name] = a
        return self._host[name]

    p_list = property(lambda self: self._view("p"))
    s0_list = property(lambda self: self._view("s"))
    w_list = property(lambda self: self._view("w"))
    n_list = property(lambda self: self._view("n"))
    wl_list = property(lambda self: self._view("wl"))
    pol_list = property(lambda self: self._view("pol"))

    def lock(self) -> None:
        d = self.__dict__
        d["_lock"] = True
        d["_new_lock"] = True

    def source_sections(self, index: int = None):
        """(p, s, pol, w, wl) of the first section of one source's rays, or of all (ray_storage.py:212-233); only these
        rows are copied from the device."""
        assert self.N, "ray_source_list has no rays stored."
        assert index is None or 0 <= index < len(self.N_list)
        Ns, Ne = self.B_list[index:index + 2] if index is not None else (0, self.N)
        ind = np.arange(int(Ns), int(Ne))
        first = np.zeros(ind.shape[0], dtype=np.int64)
        pol = self.pol_list[ind, 0] if self._dev.get("pol") is None else self._select("pol", ind, first)
        if "s" in self._host:
            s0 = self.s0_list[ind]
        else:
            idx = torch.from_numpy(ind).to(self._dev["s"].device)
            s0 = self._dev["s"].view(3, self._Np)[:, idx].t().cpu().numpy()
        return (self._select("p", ind, first), s0, pol, self._select("w", ind, first), self._select("wl", ind, None))

    def ray_lengths(self, ch=None, ch2=None) -> np.ndarray:
        _, s, _, _, _, _, _ = self.rays_by_mask(ch, ch2, ret=[0, 1, 0, 0, 0, 0, 0], normalize=False)
        return np.linalg.norm(s, axis=s.ndim - 1)

    def optical_lengths(self, ch=None, ch2=None) -> np.ndarray:
        _, s, _, _, _, _, n = self.rays_by_mask(ch, ch2, ret=[0, 1, 0, 0, 0, 0, 1], normalize=False)
        return np.linalg.norm(s, axis=s.ndim - 1) * n

    def source_numbers(self) -> np.ndarray:
        return self.rays_by_mask(ret=[0, 0, 0, 0, 0, 1, 0])[5]

    def direction_vectors(self, normalize: bool = True) -> np.ndarray:
        return self.rays_by_mask(ret=[0, 1, 0, 0, 0, 0, 0], normalize=normalize)[1]

    def _select(self, name: str, ind: np.ndarray, ch2) -> np.ndarray:
        """Rows `ind` (ray indices) of the list `name`, all sections (ch2 = slice) or one section per ray (ch2 = index
        array): what `list[ch, ch2]` gives on the host copy, gathered on the device unless the host copy exists already
        (a selection of a few thousand rays must not pull gigabytes over PCIe)."""
        N, nt = self._Np, self._nt  # (plane stride)
        t = self._dev.get(name)
        if name in self._host or t is None:
            a = self._view(name)
            return a[ind] if name == "wl" else a[ind, ch2]
        idx = torch.from_numpy(np.ascontiguousarray(ind, dtype=np.int64)).to(t.device)
        if name == "wl":
            return t[idx].cpu().numpy()
        vec = name in ("p", "pol")
        v = t.view(3, nt, N) if vec else t.view(nt, N)  # element (ray, section, component) at ray + N*(section + nt*comp)
        if isinstance(ch2, slice):
            g = v[..., idx][..., ch2, :]                                  # ([3,] sections, n)
            out = g.permute(2, 1, 0) if vec else g.permute(1, 0)            # (n, sections[, 3])
        else:
            sec = torch.from_numpy(np.ascontiguousarray(ch2, dtype=np.int64)).to(t.device)
            g = v[..., sec, idx]                                          # ([3,] n)
            out = g.permute(1, 0) if vec else g
        return np.asfortranarray(out.cpu().numpy()) if vec and isinstance(ch2, slice) else out.cpu().numpy()

    def rays_by_mask(self, ch=None, ch2=None, ret=None, normalize: bool = True):
        """Properties of selected rays / sections (ray_storage.py:235-293): (p, s, pol, w, wl, snum, n).
        Post-processing helper; the selection is gathered on the device, only the selected rows reach the host."""
        assert self.N, "ray_source_list has no rays stored."
        ret = [1, 1, 1, 1, 1, 1, 1] if ret is None else ret
        ch = np.ones(self.N, dtype=bool) if ch is None else ch
        ch2 = slice(None) if ch2 is None else ch2
        assert ch.shape[0] == self.N
        ind = np.nonzero(ch)[0]

        snums = s = None
        if ret[5]:
            snums = np.zeros_like(ind, dtype=int)
            for i, _ in enumerate(self.N_list):
                Ns, Ne = self.B_list[i:i + 2]
                snums[(Ns <= ind) & (ind < Ne)] = i
        if ret[1]:
            if not isinstance(ch2, slice):
                ch21 = np.where(ch2 < self.Nt - 1, ch2 + 1, ch2)
                s = self._select("p", ind, ch21) - self._select("p", ind, ch2)
                if normalize:
                    s = misc.normalize(s)
            else:
                pa = self._select("p", ind, slice(None))
                s = pa[:, 1:] - pa[:, :-1]
                s = np.hstack((s, np.zeros((s.shape[0], 1, 3), order='F', dtype=np.float64)))
                if normalize:
                    s = misc.normalize(s.reshape((s.shape[0] * s.shape[1], 3))).reshape(s.shape)
        pol = None
        if ret[2]:
            if self._dev.get("pol") is None:
                pol = self.pol_list[ind, ch2]
            else:
                pol = self._select("pol", ind, ch2)
        return (self._select("p", ind, ch2) if ret[0] else None, s if ret[1] else None, pol,
                self._select("w", ind, ch2) if ret[3] else None, self._select("wl", ind, None) if ret[4] else None,
                snums, self._select("n", ind, ch2) if ret[6] else None)
