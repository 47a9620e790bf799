"""
-m gpu: several devices behind ONE process (mi355_init_multi, SURVEY 8e "one communicator per process") -- the shape of the reference,
where one prover process holds one params_map [REF integration/src/prove.rs:11-21].

The GPU box has one MI355X, so the N = 2 control flow runs with the same physical device bound to two slots (test mode,
MI355_ALLOW_DUP_DEVICES=1): two contexts, two streams, two shards, two host worker threads, real kernels on both; the exchange is a
device copy because no RCCL communicator can span one device twice.  The RCCL leg itself is exercised with one rank
(MI355_MULTI_FORCE=1: partial -> ncclAllGather -> fold).  The 8-GPU run is the driver's (bench.py).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import affine_of
from tests.test_gpu_properties import dev_scalars, field_commit
from tests.test_gpu_headline import last_run

pytestmark = pytest.mark.gpu
TAU = 0x5343524F4C4C0009


def _reinit(pkg, ids, env):
    pkg.shutdown()
    for k in ("MI355_ALLOW_DUP_DEVICES", "MI355_MULTI_FORCE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    pkg.init(ids)


@pytest.fixture(scope="module")
def zk2():
    pkg = ge.load_package()
    pkg.init(0)
    _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})
    yield pkg
    _reinit(pkg, 0, {})


def test_init_multi_contract(zk2):
    lib = zk2._capi.lib()
    n = C.c_int()
    zk2._capi.check(lib.mi355_device_count(C.byref(n)))
    assert n.value == 2
    ids = (C.c_int * 2)(0, 0)
    assert lib.mi355_init_multi(ids, 2) == zk2._capi.OK                 # same list again: no-op
    assert lib.mi355_init(0) == zk2._capi.EBADARG                        # a different list while bound
    assert lib.mi355_init_multi(None, 2) == zk2._capi.EBADARG


@pytest.mark.parametrize("k", [16, 20])
def test_sharded_msm_two_slots_matches_field_and_single_device(zk2, k):
    h2 = zk2.halo2
    n = 1 << k
    params = h2.ParamsKZG.setup(k, TAU)                 # register_dev: shard 0 aliases the tensor, shard 1 is a copy
    sc = dev_scalars(n, 700 + k)
    want = field_commit(sc, TAU)
    got = affine_of(params.commit(sc))                   # device-resident scalars
    run = last_run(zk2)
    assert run["devices"] == 2 and run["exchange"] == "device_copy", run
    assert (got == want).all()
    sc_host = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert (affine_of(params.commit(sc_host)) == want).all()            # host scalars: one worker thread per slot
    params.precompute(lagrange=False)                    # window tables per shard
    assert (affine_of(params.commit(sc_host)) == want).all()
    assert (affine_of(params.commit(sc)) == want).all()
    # a slice that crosses the shard boundary, and one that lies inside shard 1
    lo, m = n // 2 - 1000, 5000
    w = cref.g1_to_affine(cref.best_multiexp(sc_host[:m], params.read_g()[lo:lo + m], threads=4))
    assert (affine_of(h2.best_multiexp(sc_host[:m].copy(), params.g_slice(lo, m))) == w).all()
    lo2 = n // 2 + 17
    w2 = cref.g1_to_affine(cref.best_multiexp(sc_host[:m], params.read_g()[lo2:lo2 + m], threads=4))
    assert (affine_of(h2.best_multiexp(sc_host[:m].copy(), params.g_slice(lo2, m))) == w2).all()
    # batched commitments: M partials per slot, one exchange
    polys = [dev_scalars(n, 800 + i) for i in range(3)]
    outs = params.commit_many(polys)
    for i in range(3):
        assert (affine_of(outs[i]) == field_commit(polys[i], TAU)).all()
    outs_h = params.commit_many([p.cpu().numpy().view(np.uint64).reshape(n, 4) for p in polys])
    assert (outs_h == outs).all()
    # un-normalised representative on request (per-thread option), still the same point
    zk2._capi.check(zk2._capi.lib().mi355_msm_set_normalise(0))
    raw = params.commit(sc)
    zk2._capi.check(zk2._capi.lib().mi355_msm_set_normalise(1))
    assert (cref.g1_to_affine(raw) == want).all()
    params.release()


def test_small_basis_stays_on_primary_and_empty_inputs(zk2):
    h2 = zk2.halo2
    params = h2.ParamsKZG.setup(10, TAU)
    sc = dev_scalars(1 << 10, 3)
    assert (affine_of(params.commit(sc)) == field_commit(sc, TAU)).all()
    assert last_run(zk2)["devices"] == 1
    out = h2.best_multiexp(np.zeros((0, 4), dtype=np.uint64), params.g_slice(0, 0))
    assert (out == 0).all()
    params.release()


def test_downsize_and_read_back_on_sharded_basis(zk2):
    h2 = zk2.halo2
    k = 16
    big = h2.ParamsKZG.setup(k, TAU + 1)
    small = h2.ParamsKZG.setup(k - 1, TAU + 1)
    g_all = big.read_g()
    assert (g_all[: 1 << (k - 1)] == small.read_g()).all()
    big.downsize(k - 1)                                  # gathers g[..2^15] from both shards, inverse G1 DFT, re-sharded result
    assert (big.read_g(lagrange=True) == small.read_g(lagrange=True)).all()
    ev = dev_scalars(1 << (k - 1), 11)
    assert (affine_of(big.commit_lagrange(ev)) == affine_of(small.commit_lagrange(ev))).all()
    big.release(); small.release()


def test_rccl_allgather_leg_with_one_rank():
    """MI355_MULTI_FORCE=1: even with one device the MSM runs as partial -> ncclAllGather (a real RCCL communicator created by
    ncclCommInitAll inside the library) -> fold."""
    pkg = ge.load_package()
    _reinit(pkg, [0], {"MI355_MULTI_FORCE": "1"})
    try:
        h2 = pkg.halo2
        k = 18
        params = h2.ParamsKZG.setup(k, TAU + 2)
        sc = dev_scalars(1 << k, 99)
        got = affine_of(params.commit(sc))
        run = last_run(pkg)
        assert run["exchange"] == "rccl_allgather" and run["devices"] == 1, run
        assert (got == field_commit(sc, TAU + 2)).all()
        got_h = affine_of(params.commit(sc.cpu().numpy().view(np.uint64).reshape(1 << k, 4)))
        assert (got_h == got).all()
        params.release()
    finally:
        _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})
