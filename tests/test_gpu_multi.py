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
    for k in ("MI355_ALLOW_DUP_DEVICES", "MI355_MULTI_FORCE", "MI355_SHARD_MIN_LOG"):
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


def test_three_slots_ragged_length_and_params_file(tmp_path):
    """three device slots, a length that does not divide evenly, bases uploaded from the host and streamed from a params file."""
    pkg = ge.load_package()
    _reinit(pkg, [0, 0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})
    try:
        h2 = pkg.halo2
        k = 17
        src = h2.ParamsKZG.setup(k, TAU + 3)
        g, gl = src.read_g(), src.read_g(lagrange=True)
        n = (1 << k) - 77
        sc = dev_scalars(n, 41).cpu().numpy().view(np.uint64).reshape(n, 4)
        want = cref.g1_to_affine(cref.best_multiexp(sc, g[:n], threads=8))
        assert (affine_of(h2.best_multiexp(sc.copy(), src.g_slice(0, n))) == want).all()
        run = last_run(pkg)
        assert run["devices"] == 3 and run["exchange"] == "device_copy", run
        up = h2.ParamsKZG.from_host(k, g, gl)                         # mi355_srs_register_host: three uploads
        assert (affine_of(h2.best_multiexp(sc.copy(), up.g_slice(0, n))) == want).all()
        path = str(tmp_path / "params17")
        src.write(path)
        ld = h2.params_from_file(path, validate=True)                 # streamed shard by shard
        assert (ld.read_g() == g).all() and (ld.read_g(lagrange=True) == gl).all()
        assert (affine_of(h2.best_multiexp(sc.copy(), ld.g_slice(0, n))) == want).all()
        assert ld.s_g2 == src.s_g2
        for p in (src, up, ld):
            p.release()
    finally:
        _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})


def test_eight_slots_the_target_machines_device_count(tmp_path):
    """D = 8 is what the target node has (BASELINE configs[4]) and what no one-GPU box can bind distinctly: eight duplicate slots run every D-sized array,
    the 2^14-points-per-device rule, the replica dealing and the per-slot worker threads with real kernels.  Sharded commitments at 2^20 / 2^22
    incl. slices that cross ALL seven shard boundaries, ragged ones, a basis just under 8 x 2^14 points (stays on fewer devices), window tables per shard, batched
    commitments, and 20 transforms dealt over the 8 slots."""
    pkg = ge.load_package()
    _reinit(pkg, [0] * 8, {"MI355_ALLOW_DUP_DEVICES": "1"})
    try:
        h2 = pkg.halo2
        lib, check = pkg._capi.lib(), pkg._capi.check
        cnt = C.c_int(); check(lib.mi355_device_count(C.byref(cnt))); assert cnt.value == 8
        for k in (20, 22):
            n = 1 << k
            params = h2.ParamsKZG.setup(k, TAU + 6)
            sc = dev_scalars(n, 900 + k)
            want = field_commit(sc, TAU + 6)
            assert (affine_of(params.commit(sc)) == want).all()
            run = last_run(pkg)
            assert run["devices"] == 8 and run["exchange"] == "device_copy", run
            sc_host = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
            assert (affine_of(params.commit(sc_host)) == want).all()                 # host scalars: eight worker threads, eight uploads
            if k == 20:
                g = params.read_g()
                # [n/8 - 5, 7n/8 + 5): touches all eight shards, crosses all seven boundaries; and a ragged slice inside shards 2..5
                for lo, m in ((n // 8 - 5, 6 * n // 8 + 10), (2 * n // 8 + 3, 3 * n // 8 + 1001)):
                    w = cref.g1_to_affine(cref.best_multiexp(sc_host[:m], g[lo:lo + m], threads=8))
                    assert (affine_of(h2.best_multiexp(sc_host[:m].copy(), params.g_slice(lo, m))) == w).all()
                    assert last_run(pkg)["devices"] >= 4
                params.precompute(lagrange=False)                                      # eight per-shard window tables
                assert (affine_of(params.commit(sc)) == want).all() and last_run(pkg)["devices"] == 8
                polys = [dev_scalars(n, 950 + i) for i in range(3)]
                outs = params.commit_many(polys)
                for i in range(3):
                    assert (affine_of(outs[i]) == field_commit(polys[i], TAU + 6)).all()
            params.release()
        # just under 8 x 2^14 points: the per-device minimum keeps the basis on fewer than eight devices, results unchanged
        k = 17
        n = (1 << k) - 1
        src = h2.ParamsKZG.setup(k, TAU + 7)
        g = src.read_g()
        sc = dev_scalars(n, 77).cpu().numpy().view(np.uint64).reshape(n, 4)
        want = cref.g1_to_affine(cref.best_multiexp(sc, g[:n], threads=8))
        assert (affine_of(h2.best_multiexp(sc.copy(), src.g_slice(0, n))) == want).all()
        assert 1 <= last_run(pkg)["devices"] <= 8
        src.release()
        # replicas: 20 independent transforms dealt over the 8 slots (host pointers round-robin, device pointers on their owners) equal the serial loop
        dom = h2.EvaluationDomain(2, 14)
        polys = [dev_scalars(1 << 14, 60 + i).cpu().numpy().view(np.uint64).reshape(1 << 14, 4) for i in range(20)]
        serial = [p.copy() for p in polys]
        for p in serial:
            dom.coeff_to_lagrange(p)
        h2.best_fft_many(polys, dom.omega, 14)
        for a, b in zip(polys, serial):
            assert (a == b).all()
        bufs = [h2.DeviceBuffer.from_host(serial[i], slot=i % 8) for i in range(20)]    # resident on all eight slots
        h2.best_fft_many(bufs, dom.omega_inv, 14, divisor=dom.ifft_divisor)
        for i, b in enumerate(bufs):
            want_i = serial[i].copy(); dom.lagrange_to_coeff(want_i)
            assert (b.fr() == want_i).all()
            b.free()
    finally:
        _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})


def test_prefix_view_gets_private_tables_when_much_smaller(zk2):
    """a 2^12 prefix of a 2^20 basis: the inherited table (built for 2^19-point shards) is far from the best width for 2^12 points, so
    mi355_srs_precompute builds a private one for the view; the parent keeps its own."""
    h2 = zk2.halo2
    lib, check = zk2._capi.lib(), zk2._capi.check
    big = h2.ParamsKZG.setup(20, TAU + 4)
    big.precompute(lagrange=False)
    view = big.clone_downsized(12)
    pa, pb = C.c_void_p(), C.c_void_p(); ca, cb = C.c_int(), C.c_int()
    check(lib.mi355_srs_pre_dev_ptr(big._g, C.byref(pa), C.byref(ca), None))
    check(lib.mi355_srs_pre_dev_ptr(view._g, C.byref(pb), C.byref(cb), None))
    assert pa.value == pb.value                                       # inherited
    check(lib.mi355_srs_precompute(view._g, 0, 0))
    check(lib.mi355_srs_pre_dev_ptr(view._g, C.byref(pb), C.byref(cb), None))
    assert pb.value != pa.value and cb.value < ca.value               # private, narrower windows
    sc = dev_scalars(1 << 12, 5)
    assert (affine_of(view.commit(sc)) == field_commit(sc, TAU + 4)).all()
    sc_big = dev_scalars(1 << 20, 6)
    assert (affine_of(big.commit(sc_big)) == field_commit(sc_big, TAU + 4)).all()
    check(lib.mi355_srs_pre_dev_ptr(big._g, C.byref(pb), C.byref(cb), None))
    assert pb.value == pa.value and cb.value == ca.value
    view.release(); big.release()


def test_distinct_devices_rccl_allgather_over_xgmi():
    """Arms itself on the first box with >= 2 GPUs (VERDICT r2 next #1 iii): mi355_init_multi over DISTINCT devices, one communicator
    (ncclCommInitAll), the grouped ncclAllGather of the 96-byte partials over D > 1 ranks, fold on the primary -- the path that duplicate
    slots cannot execute.  Skips only when the box has a single GPU."""
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("one GPU visible: the D > 1 RCCL exchange needs distinct devices (runs unedited on any multi-GPU box)")
    pkg = ge.load_package()
    D = min(ngpu, 8)
    _reinit(pkg, list(range(D)), {})
    try:
        h2 = pkg.halo2
        k = 20
        n = 1 << k
        params = h2.ParamsKZG.setup(k, TAU + 5)
        sc = dev_scalars(n, 1234)
        want = field_commit(sc, TAU + 5)
        got = affine_of(params.commit(sc))                       # scalars on the primary, slices cross xGMI to their shard's device
        run = last_run(pkg)
        assert run["exchange"] == "rccl_allgather" and run["devices"] == D, run
        assert (got == want).all()
        sc_host = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
        assert (affine_of(params.commit(sc_host)) == want).all()  # one worker thread and PCIe link per device
        params.precompute(lagrange=False)
        assert (affine_of(params.commit(sc)) == want).all() and last_run(pkg)["exchange"] == "rccl_allgather"
        outs = params.commit_many([sc, dev_scalars(n, 1235)])
        assert (affine_of(outs[0]) == want).all()
        # replicas: a batch of independent transforms dealt over the D devices equals the serial loop on one
        dom = h2.EvaluationDomain(2, 16)
        polys = [dev_scalars(1 << 16, 50 + i).cpu().numpy().view(np.uint64).reshape(1 << 16, 4) for i in range(2 * D + 1)]
        serial = [p.copy() for p in polys]
        for p in serial:
            dom.coeff_to_lagrange(p)
        h2.best_fft_many(polys, dom.omega, 16)
        for a, b in zip(polys, serial):
            assert (a == b).all()
        params.release()
    finally:
        _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})
