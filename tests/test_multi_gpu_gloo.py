"""world_size 2 over gloo: the N > 1 orchestration of the one-process-per-GPU sharded MSM (scroll-prover_amd/distributed.py) -- shard
ranges, the all_gather of 96-byte partials and the fold.
  * CPU (no marker): the oracle is injected as the compute callables, so the control flow is covered in the GPU-less container;
  * -m gpu: the SAME control flow with the HIP library as the compute on both ranks (two processes share the box's one MI355X; the
    exchange stays on gloo because RCCL cannot put two ranks on one device): mi355_msm_g1_host per shard, mi355_g1_sum_host fold."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import cref
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    zk = ge.load_package()
    rng = np.random.default_rng(123)                 # same inputs on every rank
    G = cref.g1_generator()
    sc = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
    ks = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); ks[:, 3] &= np.uint64((1 << 60) - 1)
    bases = cref.g1_to_affine(np.stack([cref.g1_mul(G, ks[i]) for i in range(n)]))
    lo, hi = zk.distributed.shard_range(n, rank, world)

    def fold(parts):
        acc = parts[0]
        for p in parts[1:]:
            acc = cref.g1_add(acc, p)
        return acc

    got = zk.distributed.sharded_multiexp(lambda: cref.best_multiexp(sc[lo:hi], bases[lo:hi], threads=2) if hi > lo else np.zeros(12, dtype=np.uint64), fold)
    want = cref.best_multiexp(sc, bases, threads=2)
    ok = bool((cref.g1_to_affine(got) == cref.g1_to_affine(want)).all())
    parts = zk.distributed.allgather_partials(np.full(12, rank + 1, dtype=np.uint64))
    ok = ok and parts.shape == (world, 12) and all((parts[r] == r + 1).all() for r in range(world))
    q.put((rank, ok))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (2, 1)])
def test_sharded_msm_over_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res)


def _worker_hip(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import cref
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    zk = ge.load_package()
    zk.init(0)                                       # both ranks on the one visible GPU
    h2, capi = zk.halo2, zk._capi
    rng = np.random.default_rng(321)                 # same inputs on every rank
    sc = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
    ks = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); ks[:, 3] &= np.uint64((1 << 60) - 1)
    bases = cref.g1_mul_generator_vec(ks, threads=4)
    lo, hi = zk.distributed.shard_range(n, rank, world)
    handle = C.c_uint64()
    capi.check(capi.lib().mi355_srs_register_host(capi.ptr(np.ascontiguousarray(bases[lo:hi])), hi - lo, C.byref(handle)))
    capi.check(capi.lib().mi355_msm_set_normalise(0))          # partials are folded afterwards

    def local():
        out = np.zeros(12, dtype=np.uint64)
        capi.check(capi.lib().mi355_msm_g1_host(handle.value, 0, capi.ptr(np.ascontiguousarray(sc[lo:hi])), hi - lo, capi.ptr(out)))
        return out

    got = zk.distributed.sharded_multiexp(local, h2.g1_sum)   # fold = mi355_g1_sum_host (normalises)
    want = cref.g1_to_affine(cref.best_multiexp(sc, bases, threads=4))
    ok = bool((np.asarray(got)[:8] == want).all())
    capi.check(capi.lib().mi355_srs_release(handle.value))
    q.put((rank, ok))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n", [(2, 1 << 15), (2, 37)])
def test_sharded_msm_over_gloo_with_hip_compute(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_hip, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs: p.join(timeout=120)
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res)
