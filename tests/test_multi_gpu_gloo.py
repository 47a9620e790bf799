"""CPU-only, world_size 2 over gloo: the N > 1 orchestration of the sharded MSM (scroll-prover_amd/distributed.py) -- shard
ranges, the all_gather of 96-byte partials and the fold -- with the CPU oracle injected as the compute callables.
The HIP path uses exactly this function with mi355_msm_g1_dev / mi355_g1_sum_host (bench.py)."""
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
