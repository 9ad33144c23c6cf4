"""Multi-GPU composition: rays shard across ranks, one exchange for the detector histogram.

The reference has no distributed layer; its own composition rules are the template:
  * threads own contiguous ray ranges and generate + trace them independently
    (RayStorage.thread_rays ray_storage.py:147-171),
  * iterative_render adds the images of independent chunks (raytracer.py:1235-1267).
Here a rank = one process = one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  The ray path needs no collective; the reductions below run once per image:
  1. min/max of 4 doubles when the image extent is automatic (raytracer.py:1042-1046),
  2. sum of the (Ny, Nx, 4) float64 histogram (28.6 MB at 945 x 945),
  3. sum of the 5 x nt event counters and the ill-conditioned count.
A ring all-reduce of 28.6 MB over 7 xGMI links is ~0.3 ms, so no custom collective is warranted.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


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
    det = RT.detectors[detector_index]
    if extent is None:
        # agree on the automatic extent first (two-pass: hit search, min/max exchange, then binning)
        ph, hw, wl, ext, proj, ill = RT._hit_detector("Detector Image", detector_index, None, None, projection_method)
        has = bool((hw > 0).any().item())
        e = ext if has else np.array([np.inf, -np.inf, np.inf, -np.inf])
        ext = allreduce_extent(np.asarray(e, dtype=np.float64), device=dev)
        extent = list(ext) if np.all(np.isfinite(ext)) else list(det.pos[:2].repeat(2))
    with global_options.no_warnings():
        img = RT.detector_image(detector_index=detector_index, extent=extent, projection_method=projection_method,
                                _keep_on_device=True)
    allreduce_image(img._dev)
    img._sync_host()
    RT._msgs = allreduce_counters(RT._msgs, device=dev)
    return img
