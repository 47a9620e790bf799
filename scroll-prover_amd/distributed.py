"""
distributed.py -- the multi-GPU leg of the MSM: one process per GPU, point-range shards, one tiny
exchange.  best_multiexp on the CPU already splits the point range across threads and folds the partial sums with `+`
[halo2_proofs arithmetic.rs, EXT-recalled]; here the threads are ranks, each owns bases[lo:hi] resident in its HBM, and
the fold input is gathered with ONE all_gather of 96-byte partials (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  There is no elliptic-curve reduction op in RCCL, hence gather + local fold.

Nothing here touches the data path: the compute callables are injected (the HIP path in production / bench.py, the CPU
oracle in tests/test_multi_gpu_gloo.py).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """point range [lo, hi) owned by `rank`: contiguous, sizes differ by at most one, covers [0, n) exactly."""
    assert 0 <= rank < world and n >= 0
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_BUFFERS = {}   # (world, device) -> (send, recv, pinned host copy): the exchange is 96 B per rank, so allocation would dominate it


def allgather_partials(partial: np.ndarray, device=None) -> np.ndarray:
    """every rank contributes one G1 (12 x u64 = 96 B); returns [world, 12] on every rank."""
    import torch
    import torch.distributed as dist
    partial = np.ascontiguousarray(partial, dtype=np.uint64).reshape(12)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return partial.reshape(1, 12).copy()
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    key = (world, str(dev))
    if key not in _BUFFERS:
        _BUFFERS[key] = (torch.empty(96, dtype=torch.uint8, device=dev), torch.empty(world * 96, dtype=torch.uint8, device=dev))
    mine, out = _BUFFERS[key]
    mine.copy_(torch.from_numpy(partial.view(np.uint8)))
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().view(np.uint64).reshape(world, 12).copy()


def sharded_multiexp(local_msm: Callable[[], np.ndarray], fold: Callable[[np.ndarray], np.ndarray], device=None) -> np.ndarray:
    """local_msm() -> this rank's partial sum over its own shard; fold([world,12]) -> their sum.  Same result on every rank."""
    return fold(allgather_partials(local_msm(), device))


def sharded_multiexp_device(capi, srs_handle: int, scalars_dev, n: int, base_offset: int = 0) -> np.ndarray:
    """the production form of the exchange (backend "nccl" = RCCL): this rank's partial sum never leaves the GPU --
    mi355_msm_g1_dev_async leaves 96 bytes in device memory in stream order, ONE all_gather_into_tensor moves world x 96 bytes over
    xGMI, mi355_g1_sum_dev folds and normalises them on the device; the only host synchronisation is the final 96-byte read-back.
    The per-rank Horner tail skips its inversion (the partial is requested un-normalised for this call only: the option is per calling
    thread and is restored before returning).  Same result on every rank."""
    import torch
    import torch.distributed as dist
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if grouped else 1
    key = ("dev", world, str(scalars_dev.device))
    if key not in _BUFFERS:
        _BUFFERS[key] = (torch.empty(96, dtype=torch.uint8, device=scalars_dev.device), torch.empty(world * 96, dtype=torch.uint8, device=scalars_dev.device))
    mine, out = _BUFFERS[key]
    lib = capi.lib()
    capi.check(lib.mi355_msm_set_normalise(0))
    try:
        capi.check(lib.mi355_msm_g1_dev_async(srs_handle, base_offset, capi.ptr(scalars_dev), n, capi.ptr(mine)))
    finally:
        capi.check(lib.mi355_msm_set_normalise(1))
    if grouped:
        dist.all_gather_into_tensor(out, mine)     # also with one rank: the same RCCL call, so a 1-GPU box exercises the real path
    else:
        out = mine
    result = np.zeros(12, dtype=np.uint64)
    capi.check(lib.mi355_g1_sum_dev(capi.ptr(out), world, capi.ptr(result)))
    return result
