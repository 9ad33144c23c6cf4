"""Multi-GPU composition: rays shard across ranks, one exchange for the detector histogram.

The reference has no distributed layer; its own composition rules are the template:
  * threads own contiguous ray ranges and generate + trace them independently
    (RayStorage.thread_rays ray_storage.py:147-171),
  * iterative_render adds the images of independent chunks (raytracer.py:1235-1267).
Here a rank = one process = one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  The ray path needs no collective; the reductions below run once per image:
  1. min/max of 4 doubles when the image extent is automatic (raytracer.py:1042-1046),
  2. sum of the (Ny, Nx, 4) float64 histograms (28.6 MB each at 945 x 945; only the window of lit pixels travels,
     `allreduce_images`),
  3. sum of the 5 x nt event counters and the ill-conditioned count.
A ring all-reduce of 28.6 MB over 7 xGMI links is ~0.3 ms, so no custom collective is warranted.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from ._device import require_device


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(N: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous global ray range [first, end) of `rank`: N // world rays each, the last rank takes the
    remainder -- exactly the reference's per-thread split (ray_storage.py:147-149)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("invalid rank / world size")
    Np = int(N / world_size)
    first = rank * Np
    end = first + Np if rank != world_size - 1 else N
    return first, end


def shard_source_powers(N: int, powers: list[float], rank: int, world_size: int, N_list=None) -> tuple[np.ndarray, np.ndarray]:
    """Per-source ray counts and powers of this rank's shard.

    The global rays are assigned to sources by power (RayStorage.init ray_storage.py:59-74; `N_list` can be
    passed to reuse the global split including its random remainder), then the rank's contiguous range is
    intersected with the per-source ranges; each piece carries `n_piece / N_source * P_source` of power
    (ray_storage.py:160).  Returns (counts, powers) with one entry per source (count 0 = source not in shard).
    """
    P = np.asarray(powers, dtype=np.float64)
    if N_list is None:
        N_list = (N * P / P.sum()).astype(int)
        N_list[: N - N_list.sum()] += 1  # deterministic remainder: the first sources get one more ray
    N_list = np.asarray(N_list, dtype=np.int64)
    B = np.concatenate(([0], np.cumsum(N_list)))
    first, end = shard_range(N, rank, world_size)
    counts = np.zeros(len(P), dtype=np.int64)
    shard_p = np.zeros(len(P), dtype=np.float64)
    for i in range(len(P)):
        lo, hi = max(first, B[i]), min(end, B[i + 1])
        if hi > lo:
            counts[i] = hi - lo
            shard_p[i] = (hi - lo) / N_list[i] * P[i]
    return counts, shard_p


def _device_collectives() -> bool:
    """True when the process group reduces device tensors in place (nccl = RCCL).  Any other backend (gloo in the
    tests and in rehearsals where several ranks share a device) reduces host copies."""
    return dist.get_backend() == "nccl"


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over all ranks (no-op without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if t.is_cuda and not _device_collectives():
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_extent(ext4: np.ndarray, device=None) -> np.ndarray:
    """Common automatic extent [xmin, xmax, ymin, ymax] of all ranks (raytracer.py:1042-1046).
    Ranks without hits pass [+inf, -inf, +inf, -inf]."""
    if world()[1] == 1:
        return ext4
    device = device if _device_collectives() else None
    lo = torch.tensor([ext4[0], ext4[2]], dtype=torch.float64, device=device)
    hi = torch.tensor([ext4[1], ext4[3]], dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    lo, hi = lo.cpu().numpy(), hi.cpu().numpy()
    return np.array([lo[0], hi[0], lo[1], hi[1]])


def allreduce_image(hist: torch.Tensor) -> torch.Tensor:
    """Sum the (Ny, Nx, 4) float64 detector histograms of all ranks in place (the one data exchange)."""
    if hist.dtype != torch.float64:
        raise TypeError("detector histograms are float64")
    return allreduce_sum_(hist)


def _as_one_block(hists: list):
    """The (K, Ny, Nx, 4) view of K equally shaped histograms that lie back to back in one allocation (`_plan_renders`
    slices them from one), or None."""
    h0 = hists[0]
    if any(h.shape != h0.shape or not h.is_contiguous() or h.device != h0.device for h in hists):
        return None
    step = h0.numel() * h0.element_size()
    if any(h.data_ptr() != h0.data_ptr() + k * step for k, h in enumerate(hists)):
        return None
    if h0.untyped_storage().data_ptr() != hists[-1].untyped_storage().data_ptr():
        return None
    return torch.as_strided(h0, (len(hists),) + tuple(h0.shape), (h0.numel(),) + tuple(h0.stride()))


def lit_window(block: torch.Tensor) -> torch.Tensor:
    """Per histogram of the (K, Ny, Nx, 4) block: [y0, -y1, x0, -x1] (int64 (K, 4), on the block's device) of the pixel window
    [y0, y1) x [x0, x1) outside of which the histogram is zero; an all-dark one gives [Ny, 0, Nx, 0].  (The negated upper ends
    let ONE MIN all-reduce form the union of the ranks' windows.)"""
    K, Ny, Nx, _ = block.shape
    lit = (block != 0).any(dim=3)
    rows, cols = lit.any(dim=2), lit.any(dim=1)  # (K, Ny), (K, Nx)
    iy = torch.arange(Ny, device=block.device).expand(K, Ny)
    ix = torch.arange(Nx, device=block.device).expand(K, Nx)
    return torch.stack([torch.where(rows, iy, Ny).amin(dim=1), -(torch.where(rows, iy, -1).amax(dim=1) + 1),
                        torch.where(cols, ix, Nx).amin(dim=1), -(torch.where(cols, ix, -1).amax(dim=1) + 1)], dim=1)


#: what the last `allreduce_images` of this process sent (bench.py reports it)
last_exchange: dict = dict(bytes=0, window=None)

#: the lit windows are exchanged instead of the whole histograms when they hold at most this share of their pixels
WINDOW_EXCHANGE_BELOW = 0.7


def _sent(n_bytes: int, window) -> dict:
    last_exchange.update(bytes=int(n_bytes), window=window)
    return dict(last_exchange)


def allreduce_images(hists: list) -> dict:
    """Sum K detector histograms ((Ny, Nx, 4) float64 device tensors) over all ranks in place -- the one data exchange of
    a sharded render.  Equally shaped histograms go as ONE message; and of every histogram only the WINDOW of pixels that
    any rank has lit travels (four integers per image are agreed first): BASELINE config 4 renders a 4 x 3 mm picture onto
    a 16 x 16 mm detector at six positions behind the focus -- the windows around its pictures are 21 to 80 % of their
    histograms, half of the 6 x 28.6 MB together --, a point-spread image sends next to nothing.  Outside its window every
    rank holds zeros, so the result is the sum of the whole histograms.
    -> {"bytes": what travelled, "window": per image [y0, y1, x0, x1], or None where the histograms travelled whole}."""
    if any(h.dtype != torch.float64 or h.dim() != 3 or h.shape[2] != 4 for h in hists):
        raise TypeError("detector histograms are (Ny, Nx, 4) float64")
    full = int(sum(h.numel() * 8 for h in hists))
    if world()[1] == 1 or not hists:
        return _sent(0, None)
    if len({tuple(h.shape) for h in hists}) != 1:
        for h in hists:
            allreduce_sum_(h)
        return _sent(full, None)
    block = _as_one_block(hists)
    stacked = block is None
    if stacked:
        block = torch.stack(hists)
    win = lit_window(block)
    if not _device_collectives():
        win = win.cpu()
    dist.all_reduce(win, op=dist.ReduceOp.MIN)
    K, Ny, Nx, _ = block.shape
    wins = [[int(w[0]), -int(w[1]), int(w[2]), -int(w[3])] for w in win.tolist()]  # [y0, y1, x0, x1]
    wins = [[0, 0, 0, 0] if (y1 <= y0 or x1 <= x0) else [y0, y1, x0, x1] for y0, y1, x0, x1 in wins]  # (no hit anywhere)
    pixels = sum((y1 - y0) * (x1 - x0) for y0, y1, x0, x1 in wins)
    if pixels == 0:
        return _sent(0, wins)
    if pixels > WINDOW_EXCHANGE_BELOW * K * Ny * Nx:
        allreduce_sum_(block)
        window, sent = None, full
    else:
        part = torch.cat([block[k, y0:y1, x0:x1, :].reshape(-1) for k, (y0, y1, x0, x1) in enumerate(wins)])
        allreduce_sum_(part)
        off = 0
        for k, (y0, y1, x0, x1) in enumerate(wins):
            n = (y1 - y0) * (x1 - x0) * 4
            block[k, y0:y1, x0:x1, :] = part[off:off + n].view(y1 - y0, x1 - x0, 4)
            off += n
        window, sent = wins, int(part.numel() * 8)
    if stacked:
        for k, h in enumerate(hists):
            h.copy_(block[k])
    return _sent(sent, window)


def allreduce_counters(msgs: np.ndarray, device=None) -> np.ndarray:
    """Sum the (5, nt) event counters of all ranks (Raytracer._set_messages raytracer.py:181-190)."""
    if world()[1] == 1:
        return msgs
    device = device if _device_collectives() else None
    t = torch.as_tensor(np.ascontiguousarray(msgs, dtype=np.int64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(int)


def sharded_detector_image(RT, N: int, detector_index: int = 0, extent=None, projection_method="Equidistant",
                           base_seed: int = 0):
    """Trace this rank's shard of N rays and return the all-reduced detector image (identical on every rank).

    Each rank traces `shard_range(N, rank, world)` rays with seed `base_seed + rank`; its rays carry the shard's share
    of the source powers (`ray_storage.py:160`: a thread's rays carry `n_thread / N_source` of the power), so the
    summed image carries the full source power.  `RT._msgs` becomes the sum over all ranks.  With an automatic extent
    the ranks first agree on the common one (raytracer.py:1042-1046); if no ray of any rank reaches the detector it
    collapses to the detector centre like the reference's (raytracer.py:1048-1049).
    """
    from . import global_options
    rank, ws = world()
    first, end = shard_range(N, rank, ws)
    n_local = end - first
    seed0 = RT.seed
    RT.seed = base_seed + rank
    try:
        RT.trace(n_local, _power_scale=n_local / N)
    finally:
        RT.seed = seed0
    dev = RT.rays._dev["p"].device
    spec = dict(detector_index=detector_index, source_index=None, extent=extent, projection_method=projection_method)
    if extent is None:
        # agree on the automatic extent first: extent-only pass over the sections, min / max exchange, then the binning
        spec = RT._auto_extents([spec], agree=lambda raw: allreduce_extents(raw, device=dev))[0]
    with global_options.no_warnings():
        img = RT._render_detectors([spec], [None])[0]
    allreduce_images([img._dev])
    img._sync_host()
    RT._msgs = allreduce_counters(RT._msgs, device=dev)
    return img


def allreduce_extents(raw: np.ndarray, device=None) -> np.ndarray:
    """Common automatic extents of all ranks for K images at once: rows [xmin, xmax, ymin, ymax], +-inf where a rank has
    no hit.  One MIN and one MAX all-reduce of 2 K doubles each."""
    raw = np.asarray(raw, dtype=np.float64).reshape(-1, 4)
    if world()[1] == 1:
        return raw
    device = device if _device_collectives() else None
    lo = torch.tensor(raw[:, [0, 2]].ravel(), dtype=torch.float64, device=device)
    hi = torch.tensor(raw[:, [1, 3]].ravel(), dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    lo, hi = lo.cpu().numpy().reshape(-1, 2), hi.cpu().numpy().reshape(-1, 2)
    return np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1]], axis=1)


def sharded_iterative_render(RT, N: int, detector_index=0, limit=None, projection_method="Equidistant", pos=None,
                             extent=None, base_seed: int = 0) -> list:
    """`Raytracer.iterative_render` (raytracer.py:1134-1279) with the N rays sharded over the ranks -- BASELINE config 4
    as stated: every rank traces its shard ONCE (in chunks sized by its storage) and bins each chunk into all K detector
    positions in one pass over the sections; the ranks exchange
      * the automatic extents of the first chunk (one MIN and one MAX all-reduce of 2 K doubles; nothing with user
        extents), so that all ranks bin into the same pixel grids -- the reference fixes the extents with its first
        chunk in the same way (raytracer.py:1262),
      * the K histograms, summed once at the end (`allreduce_images`: one message for equally shaped images, and only the
        window of pixels that some rank has lit),
      * the event counters.
    The rank's rays carry its share of the source powers (ray_storage.py:160) and the seed `base_seed + rank`
    (chunks advance it by 1000003 like the single-process form, so rank and chunk streams do not meet).
    Returns the K images, identical on every rank; `RT._msgs` becomes the sum over all ranks."""
    rank, ws = world()
    first, end = shard_range(int(N), rank, ws)
    n_local = end - first
    if n_local <= 0:
        raise ValueError("fewer rays than ranks")
    seed0 = RT.seed
    RT.seed = base_seed + rank
    dev = None
    try:
        images = RT.iterative_render(n_local, detector_index=detector_index, limit=limit,
                                     projection_method=projection_method, pos=pos, extent=extent,
                                     _power_scale=n_local / N, _finish=False,
                                     _agree_extents=lambda raw: allreduce_extents(raw, device=require_device()))
    finally:
        RT.seed = seed0
    dev = require_device()
    allreduce_images([img._dev for img in images])
    RT._msgs = allreduce_counters(RT._msgs, device=dev)
    for img in images:
        img._sync_host()
        if img._limit is not None:
            img._apply_rayleigh_filter()
    return images
