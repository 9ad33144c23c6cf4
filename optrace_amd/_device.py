"""Device plumbing: torch owns HBM allocations and streams, the C-ABI gets raw pointers."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi


_devices: dict = {}  # index -> torch.device, filled once a device has been seen (these run once per trace)


def require_device() -> torch.device:
    """cuda:<current> or a loud failure -- the product path has no CPU fallback."""
    if not _devices:
        if not torch.cuda.is_available():
            raise _capi.BackendError("No HIP device available (torch.cuda.is_available() is False); "
                                     "optrace_amd has no CPU fallback.")
        torch.cuda.current_device()  # initialises torch's HIP state
    i = torch._C._cuda_getDevice()
    d = _devices.get(i)
    if d is None:
        d = _devices[i] = torch.device("cuda", i)
    return d


def stream_ptr() -> C.c_void_p:
    """The raw hipStream_t of torch's current stream on the current device."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def to_dev(a: np.ndarray, dtype) -> torch.Tensor:
    """Copy a host array to the device as a flat contiguous tensor."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype).reshape(-1)).to(require_device())


def f_order_flat(a: np.ndarray, dtype=np.float64) -> np.ndarray:
    """(n, 3) array -> flat component-major buffer (the reference's Fortran order)."""
    return np.ascontiguousarray(np.asarray(a, dtype=dtype).T).reshape(-1)


def from_f_order(t: torch.Tensor, n: int, ncomp: int) -> np.ndarray:
    """flat component-major device buffer -> (n, ncomp) F-ordered host array."""
    return t.cpu().numpy().reshape(ncomp, n).T


def alloc_retry(make):
    """make() -> device tensor(s).  The library keeps its binning scratch between calls in a pool of its own
    (csrc/ot_scratch.hpp) that torch's allocator cannot see: when a torch allocation runs out of memory the idle part of
    that pool and torch's cached blocks go back to the driver and the allocation is tried once more."""
    try:
        return make()
    except torch.OutOfMemoryError:
        _capi.check(_capi.load_library().ot_scratch_trim())
        torch.cuda.empty_cache()
        return make()


_mailbox: dict = {}


def mailbox(n: int = 128) -> tuple[torch.Tensor, np.ndarray]:
    """A pinned, device-mapped host buffer of n 8-byte words (one per thread of the caller, reused): small results a
    kernel writes for the host with plain stores -- the extents of the detector hits -- arrive with the stream synchronisation, without a
    device-to-host copy and without a host-to-device copy for their start values (two tiny DMA transfers and a
    device allocation less per call).
    -> (torch view as int64 for data_ptr(), numpy int64 view of the same memory).  The caller must synchronise the stream
    before reading and must not have two calls in flight."""
    import threading
    key = (threading.get_ident(), n)
    mb = _mailbox.get(key)
    if mb is None:
        t = torch.zeros(n, dtype=torch.int64).pin_memory()
        mb = _mailbox[key] = (t, t.numpy())
    return mb


def sync_stream() -> None:
    """Wait for torch's current stream (compute queue signal: not affected by busy DMA engines)."""
    torch.cuda.current_stream().synchronize()
