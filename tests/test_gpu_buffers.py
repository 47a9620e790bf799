"""
-m gpu: the resident-buffer half of the C-ABI (mi355_buf_*: what rust_shim/mi355zk.rs wraps as `DevicePoly`), the batched transforms and
the per-device locks (VERDICT r2 missing #1 / #2, weak #10).

  * a polynomial uploaded once goes through iNTT -> coset NTT -> element-wise work -> commitment -> evaluation without touching host memory,
    every step equal to the oracle's;
  * pool semantics: a freed block is handed out again (no hipMalloc / hipFree between proofs), an upload into a recycled block waits for the
    work that was queued on it, wrong pointers are refused;
  * several device slots behind one process (the same physical GPU twice on this box): buffers on slot 1, transforms there, commitments
    whose scalars live on slot 1 while the basis is sharded over both, peer copies, mi355_ntt_fr_batch_{host,dev} / mi355_coset_ntt_fr_batch_dev
    against the serial loop, four host threads calling into the library at once.
"""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import affine_of, rand_fr
from tests.test_gpu_properties import dev_scalars, field_commit
from tests.test_gpu_headline import last_run

pytestmark = pytest.mark.gpu
TAU = 0x5343524F4C4C000B
R = pyref.R_MOD


def _reinit(pkg, ids, env):
    pkg.shutdown()
    for k in ("MI355_ALLOW_DUP_DEVICES", "MI355_MULTI_FORCE", "MI355_SHARD_MIN_LOG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    pkg.init(ids)


@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    return pkg


def test_resident_polynomial_pipeline_matches_oracle(zk):
    h2 = zk.halo2
    k = 12
    n = 1 << k
    rng = np.random.default_rng(5)
    evals = rand_fr(rng, n)
    dom = h2.EvaluationDomain(3, k)
    params = h2.ParamsKZG.setup(k, TAU)
    a = h2.DeviceBuffer.from_host(evals)                      # witness upload: the only bulk host -> device traffic
    c_l = affine_of(params.commit_lagrange(a))
    dom.lagrange_to_coeff(a)                                  # iNTT in place, resident
    coeffs = cref.ifft(evals, dom.omega_inv, k, dom.ifft_divisor, threads=4)
    assert (a.fr() == coeffs).all()
    assert (affine_of(params.commit(a)) == c_l).all()         # commit(coeff(a)) == commit_lagrange(a)
    assert (c_l == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(coeffs, cref.fr_mont(TAU))))).all()
    ext = h2.DeviceBuffer(32 << dom.extended_k)
    dom.coeff_to_extended(a, out=ext)
    want_ext = cref.coeff_to_extended(coeffs, k, dom.extended_k, dom.g_coset, dom.g_coset_inv, dom.extended_omega, threads=4)
    assert (ext.fr() == want_ext).all()
    h2.fr_vec_op("mul", ext, ext, ext)                        # element-wise, resident
    sq = cref.f_mul_vec(cref.FR, want_ext, want_ext)
    assert (ext.fr() == sq).all()
    x = h2.fr(0x1234567)
    assert (h2.eval_polynomial(a, x) == cref.eval_polynomial(coeffs, x)).all()
    # offsets into a block are device pointers like any other
    lib, capi = zk._capi.lib(), zk._capi
    half = C.c_void_p(a.data_ptr() + 32 * (n // 2))
    out = np.zeros(4, dtype=np.uint64)
    capi.check(lib.mi355_eval_polynomial_dev(half, n // 2, capi.ptr(x), capi.ptr(out)))
    assert (out == cref.eval_polynomial(coeffs[n // 2:], x)).all()
    a.free(); ext.free(); params.release()


def test_pool_reuse_and_argument_checks(zk):
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    k = 14
    n = 1 << k
    rng = np.random.default_rng(6)
    dom = h2.EvaluationDomain(2, k)
    v1, v2 = rand_fr(rng, n), rand_fr(rng, n)
    b = h2.DeviceBuffer.from_host(v1)
    p1 = b.data_ptr()
    dom.coeff_to_lagrange(b)                                  # queued on the block ...
    b.free()                                                  # ... returned to the pool while that work may still be running
    b2 = h2.DeviceBuffer(32 * n)
    assert b2.data_ptr() == p1                                # recycled, no hipMalloc
    b2.upload(v2)                                             # must wait for the transform queued on the block before its free
    assert (b2.fr() == v2).all()
    dom.coeff_to_lagrange(b2)
    assert (b2.fr() == cref.best_fft(v2, dom.omega, k, threads=4)).all()
    slot = C.c_int(-1)
    capi.check(lib.mi355_buf_slot(C.c_void_p(b2.data_ptr() + 64), C.byref(slot)))
    assert slot.value == 0
    assert lib.mi355_buf_free(C.c_void_p(b2.data_ptr() + 32)) == capi.EBADARG      # not a base pointer
    assert lib.mi355_buf_upload(C.c_void_p(b2.data_ptr() + 32), capi.ptr(v1), 32 * n) == capi.EBADARG   # runs past the end of the block
    p = C.c_void_p()
    assert lib.mi355_buf_alloc(0, 0, C.byref(p)) == capi.EBADARG
    assert lib.mi355_buf_alloc(64, 5, C.byref(p)) == capi.ENODEVICE               # slot 5 is not bound
    b2.free()
    assert lib.mi355_buf_free(C.c_void_p(p1)) == capi.EBADARG                     # double free
    capi.check(lib.mi355_buf_trim())
    b3 = h2.DeviceBuffer.from_host(v1)                        # after a trim the pool is empty: a fresh allocation
    assert (b3.fr() == v1).all()
    b3.free()


def test_batched_transforms_equal_the_serial_loop_one_device(zk):
    h2 = zk.halo2
    k = 13
    n = 1 << k
    rng = np.random.default_rng(7)
    dom = h2.EvaluationDomain(2, k)
    polys = [rand_fr(rng, n) for _ in range(5)]
    want_f = [cref.best_fft(p, dom.omega, k, threads=4) for p in polys]
    hp = [p.copy() for p in polys]
    h2.best_fft_many(hp, dom.omega, k)
    for a, w in zip(hp, want_f):
        assert (a == w).all()
    h2.best_fft_many(hp, dom.omega_inv, k, divisor=dom.ifft_divisor)     # EvaluationDomain::ifft x 5
    for a, p in zip(hp, polys):
        assert (a == p).all()
    bufs = [h2.DeviceBuffer.from_host(p) for p in polys]
    h2.best_fft_many(bufs, dom.omega, k)
    for b, w in zip(bufs, want_f):
        assert (b.fr() == w).all()
    h2.best_fft_many([], dom.omega, k)
    # the same buffer listed twice: the transform is applied twice, one after the other, as the serial loop did (the batched passes must not race on it)
    twice = h2.DeviceBuffer.from_host(polys[0])
    h2.best_fft_many([twice, bufs[1], twice], dom.omega, k)
    assert (twice.fr() == cref.best_fft(want_f[0], dom.omega, k, threads=4)).all() and (bufs[1].fr() == cref.best_fft(want_f[1], dom.omega, k, threads=4)).all()
    h2.best_fft_many([bufs[1]], dom.omega_inv, k, divisor=dom.ifft_divisor); twice.free()
    for b in bufs:
        b.free()


def test_host_batch_overlaps_copies_and_matches_the_serial_loop(zk):
    """mi355_ntt_fr_batch_host pipelines upload | transform | download over two staging buffers with a helper thread: seven polynomials of
    2^19 (three reuses of each staging buffer) must equal seven single calls, and the oracle on the first."""
    h2 = zk.halo2
    k = 19
    n = 1 << k
    rng = np.random.default_rng(11)
    dom = h2.EvaluationDomain(2, k)
    polys = [rand_fr(rng, n) for _ in range(7)]
    batch = [p.copy() for p in polys]
    h2.best_fft_many(batch, dom.omega, k)
    for a, p in zip(batch, polys):
        single = p.copy()
        h2.best_fft(single, dom.omega, k)
        assert (a == single).all()
    assert (batch[0] == cref.best_fft(polys[0], dom.omega, k, threads=8)).all()
    h2.best_fft_many(batch, dom.omega_inv, k, divisor=dom.ifft_divisor)
    for a, p in zip(batch, polys):
        assert (a == p).all()


def test_buffers_and_batches_over_two_device_slots():
    pkg = ge.load_package()
    pkg.init(0)
    _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "8"})
    try:
        h2 = pkg.halo2
        lib, capi = pkg._capi.lib(), pkg._capi
        k = 14
        n = 1 << k
        rng = np.random.default_rng(8)
        dom = h2.EvaluationDomain(2, k)
        params = h2.ParamsKZG.setup(k, TAU + 1)                # sharded over both slots
        vals = [rand_fr(rng, n) for _ in range(4)]
        bufs = [h2.DeviceBuffer.from_host(v, slot=i % 2) for i, v in enumerate(vals)]
        for i, b in enumerate(bufs):
            s = C.c_int(-1); capi.check(lib.mi355_buf_slot(C.c_void_p(b.data_ptr()), C.byref(s))); assert s.value == i % 2
        # a commitment whose scalars live on slot 1 while the basis is sharded over both slots
        got = affine_of(params.commit(bufs[1]))
        run = last_run(pkg)
        assert run["devices"] == 2, run
        assert (got == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(vals[1], cref.fr_mont(TAU + 1))))).all()
        outs = params.commit_many(bufs)                        # scalars on alternating slots in one batch
        for o, v in zip(outs, vals):
            assert (affine_of(o) == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(v, cref.fr_mont(TAU + 1))))).all()
        # transforms run where the buffers live; the batch spreads over both slots
        h2.best_fft_many(bufs, dom.omega, k)
        for b, v in zip(bufs, vals):
            assert (b.fr() == cref.best_fft(v, dom.omega, k, threads=4)).all()
        dom.lagrange_to_coeff(bufs[3])                         # single call on slot 1
        assert (bufs[3].fr() == vals[3]).all()
        # coset batch (coeff_to_extended_part of the scroll fork) on both slots
        factor = h2.fr(h2.FR_ZETA)
        dsts = [h2.DeviceBuffer(32 * n, slot=i % 2) for i in range(4)]
        srcs = [h2.DeviceBuffer.from_host(v, slot=i % 2) for i, v in enumerate(vals)]
        capi.check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * 4)(*[d.data_ptr() for d in dsts]), (C.c_void_p * 4)(*[s.data_ptr() for s in srcs]), 4, k, capi.ptr(factor), capi.ptr(dom.omega)))
        zeta = h2.FR_ZETA
        for d, v in zip(dsts, vals):
            scaled = np.stack([cref.f_mul(cref.FR, v[i], cref.fr_mont(pow(zeta, i, R))) for i in range(0, n, 997)])
            full = v.copy()
            pw = np.stack([cref.fr_mont(pow(zeta, i, R)) for i in range(n)])
            want = cref.best_fft(cref.f_mul_vec(cref.FR, full, pw), dom.omega, k, threads=4)
            assert (d.fr() == want).all()
            _ = scaled
        # device-to-device copy between slots, then an element-wise operation on the destination's slot
        cp = h2.DeviceBuffer(32 * n, slot=0)
        capi.check(lib.mi355_buf_copy(C.c_void_p(cp.data_ptr()), C.c_void_p(srcs[1].data_ptr()), 32 * n))
        assert (cp.fr() == vals[1]).all()
        # host batches are dealt over the slots; four threads call into the library at once (per-device locks)
        hp = [v.copy() for v in vals * 2]
        h2.best_fft_many(hp, dom.omega, k)
        for a, v in zip(hp, vals * 2):
            assert (a == cref.best_fft(v, dom.omega, k, threads=2)).all()
        errs = []

        def worker(seed):
            try:
                r = np.random.default_rng(seed)
                for _ in range(6):
                    v = rand_fr(r, n)
                    w = v.copy()
                    dom.coeff_to_lagrange(w)                   # mi355_ntt_fr_host: lands on whichever slot is free
                    dom.lagrange_to_coeff(w)
                    if not (w == v).all():
                        errs.append("round trip")
                    x = h2.fr(int(r.integers(1, 1 << 60)))
                    if not (h2.eval_polynomial(v, x) == cref.eval_polynomial(v, x)).all():
                        errs.append("eval")
                    if not (affine_of(params.commit(v)) == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(v, cref.fr_mont(TAU + 1))))).all():
                        errs.append("commit")
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=worker, args=(100 + i,)) for i in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for b in bufs + dsts + srcs + [cp]:
            b.free()
        params.release()
    finally:
        _reinit(pkg, 0, {})


def test_gate_eval_matches_oracle_on_random_term_lists(zk):
    """mi355_fr_gate_eval_dev (one launch: rotated operands, sums of products) against the oracle's restatement of the evaluate_h operand
    shape: random term lists, negative and wrapping rotations, a constant term, accumulation, the per-launch limits."""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    rng = np.random.default_rng(91)
    for k in (6, 11, 14):
        n = 1 << k
        polys_h = [rand_fr(rng, n) for _ in range(6)]
        polys_d = [h2.DeviceBuffer.from_host(p) for p in polys_h]
        dst = h2.DeviceBuffer(32 * n)
        for trial in range(4):
            nt = [1, 5, 16, 9][trial]
            terms = []
            for j in range(nt):
                ln = 0 if (trial == 1 and j == 2) else int(rng.integers(1, 4 if nt > 9 else 6))
                terms.append((rand_fr(rng, 1, full=False)[0], [(int(rng.integers(0, 6)), int(rng.integers(-3 * n, 3 * n)) if j % 3 else int(rng.integers(-2, 3))) for _ in range(ln)]))
            # unit coefficients take the multiplication-free path of the kernel (1: factor as it is, -1: negated first factor), also on one-factor and constant terms
            for j in range(0, nt, 3):
                terms[j] = (cref.fr_mont(1) if j % 2 == 0 else cref.fr_mont(R - 1), terms[j][1])
            coeffs = np.stack([c for c, _ in terms])
            tl = [len(f) for _, f in terms]; fp = [p for _, f in terms for p, _ in f]; fr_ = [r for _, f in terms for _, r in f]
            want = cref.gate_eval(polys_h, coeffs, tl, fp, fr_, n)
            h2.gate_eval(dst, polys_d, terms, n)
            got = dst.fr()
            assert (got == want).all(), (k, trial)
            want2 = cref.gate_eval(polys_h, coeffs, tl, fp, fr_, n, dst=want)
            h2.gate_eval(dst, polys_d, terms, n, accumulate=True)
            assert (dst.fr() == want2).all()
        # a permutation-argument-shaped expression: z(wX) * prod(a_i + beta s_i + gamma) - z(X) * prod(a_i + beta d^i X + gamma) has degree-3
        # products of rotated columns; here: z[i+1] * a[i] * b[i] - z[i] * c[i] * d[i]   (rotation scaled as on a coset part: rot_scale = 1)
        one, minus_one = cref.fr_mont(1), cref.fr_mont(R - 1)
        terms = [(one, [(0, 1), (1, 0), (2, 0)]), (minus_one, [(0, 0), (3, 0), (4, 0)])]
        h2.gate_eval(dst, polys_d, terms, n)
        z, a, b, c, d = polys_h[:5]
        lhs = cref.f_mul_vec(cref.FR, cref.f_mul_vec(cref.FR, np.roll(z, -1, axis=0), a), b)
        rhs = cref.f_mul_vec(cref.FR, cref.f_mul_vec(cref.FR, z, c), d)
        want = np.stack([cref.f_sub(cref.FR, lhs[i], rhs[i]) for i in range(n)]) if n <= 2048 else None
        if want is not None:
            assert (dst.fr() == want).all()
        for b_ in polys_d + [dst]:
            b_.free()
    # limits and argument checks
    d0 = h2.DeviceBuffer(32 * 8)
    arr = (C.c_void_p * 1)(d0.data_ptr())
    one = np.ascontiguousarray(np.stack([cref.fr_mont(1)] * 17))
    tl17 = (C.c_uint32 * 17)(*([0] * 17)); fp1 = (C.c_uint32 * 1)(0); fr1 = (C.c_int32 * 1)(0)
    assert lib.mi355_fr_gate_eval_dev(C.c_void_p(d0.data_ptr()), arr, 1, capi.ptr(one), tl17, 17, fp1, fr1, 8, 0) == capi.EBADARG    # 17 terms
    assert lib.mi355_fr_gate_eval_dev(C.c_void_p(d0.data_ptr()), arr, 1, capi.ptr(one), tl17, 1, fp1, fr1, 12, 0) == capi.EBADARG    # n not a power of two
    tl1 = (C.c_uint32 * 1)(1); fp_bad = (C.c_uint32 * 1)(3)
    assert lib.mi355_fr_gate_eval_dev(C.c_void_p(d0.data_ptr()), arr, 1, capi.ptr(one), tl1, 1, fp_bad, fr1, 8, 0) == capi.EBADARG   # factor outside the list
    d0.free()


def test_uploads_allocations_and_frees_race_the_compute_of_other_threads(zk):
    """round 4: mi355_buf_alloc / _free / _upload take no device lock and an upload no longer orders the compute stream behind itself.  Three threads
    recycle pool blocks (alloc -> upload -> consume on the device -> download -> free, contents checked every time, sizes drawn from a small set so that
    blocks are handed from thread to thread) while a fourth keeps the device busy with commitments and transforms whose results are checked too."""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    k = 16
    n = 1 << k
    dom = h2.EvaluationDomain(2, k)
    params = h2.ParamsKZG.setup(k, TAU + 77)
    base = rand_fr(np.random.default_rng(70), n)
    want_commit = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(base, cref.fr_mont(TAU + 77))))
    want_fft = cref.best_fft(base, dom.omega, k, threads=4)
    errs, stop = [], threading.Event()

    def compute():
        try:
            buf = h2.DeviceBuffer.from_host(base)
            while not stop.is_set():
                if not (affine_of(params.commit(buf)) == want_commit).all():
                    errs.append("commit")
                w = h2.DeviceBuffer.from_host(base)
                dom.coeff_to_lagrange(w)
                if not (w.fr() == want_fft).all():
                    errs.append("fft")
                w.free()
            buf.free()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    def recycler(seed):
        try:
            r = np.random.default_rng(seed)
            for it in range(60):
                m = int(r.choice([1 << 12, 1 << 14, 1 << 16]))
                v = np.ascontiguousarray(r.integers(0, 2**62, size=(m, 4), dtype=np.uint64))
                b = h2.DeviceBuffer.from_host(v)                                   # alloc (pool hit or hipMalloc) + upload, no device lock
                d = h2.DeviceBuffer(32 * m)
                capi.check(lib.mi355_fr_vec_op_dev(0, C.c_void_p(d.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(b.data_ptr()), m))   # consumer queued right after the upload
                got = d.fr()
                if not (got[::97] == np.stack([cref.f_add(cref.FR, x, x) for x in v[::97]])).all():
                    errs.append(f"recycler {seed} iteration {it}: doubled values differ")
                if it % 3 == 0:
                    b.upload(v[::-1].copy())                                        # second upload into a block in use: must wait for the add above
                    if not (b.fr() == v[::-1]).all():
                        errs.append(f"recycler {seed} iteration {it}: re-upload")
                b.free(); d.free()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=compute)] + [threading.Thread(target=recycler, args=(200 + i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th[1:]:
        t.join()
    stop.set(); th[0].join()
    assert not errs, errs[:5]
    params.release()


def test_narrow_uploads_equal_the_plain_upload(zk):
    """mi355_buf_upload_packed (1 / 2 / 4 / 8-byte canonical cells expanded to Montgomery words on the device) and mi355_buf_upload_sparse (non-zero cells only, any order)
    against the plain 32-byte upload of the same column; SURVEY 8d's witness-like mix through the sparse form; offsets into a block; error paths"""
    h2 = zk.halo2
    rng = np.random.default_rng(77)
    lib, check, capi = zk._capi.lib(), zk._capi.check, zk._capi
    n = (1 << 16) + 77
    for dt, hi in ((np.uint8, 1 << 8), (np.uint16, 1 << 16), (np.uint32, 1 << 32), (np.uint64, 1 << 64)):
        vals = rng.integers(0, hi, size=n, dtype=np.uint64).astype(dt)
        vals[:5] = [0, 1, hi - 1, 2, 0]
        want = cref.f_from_canonical_vec(cref.FR, np.stack([vals.astype(np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)], axis=1))
        b = h2.DeviceBuffer.from_packed(vals)
        assert (b.fr() == want).all()
        b.free()
    col = rand_fr(rng, n)
    u = rng.random(n)
    col[u < 0.6] = 0
    small = u >= 0.8
    col[small] = cref.f_from_canonical_vec(cref.FR, np.stack([rng.integers(0, 256, size=int(small.sum()), dtype=np.uint64)] + [np.zeros(int(small.sum()), np.uint64)] * 3, axis=1))
    b = h2.DeviceBuffer.from_sparse(col, threads=5)
    assert (b.fr() == col).all()
    # pairs in any order, into the middle of a block, and the all-zero column
    idx, vals = h2.compact_nonzero(col)
    perm = rng.permutation(idx.shape[0])
    big = h2.DeviceBuffer(32 * (n + 64))
    check(lib.mi355_buf_upload_sparse(C.c_void_p(big.data_ptr() + 32 * 64), n, capi.ptr(np.ascontiguousarray(idx[perm])), capi.ptr(np.ascontiguousarray(vals[perm])), idx.shape[0]))
    assert (big.fr()[64:] == col).all()
    check(lib.mi355_buf_upload_sparse(C.c_void_p(big.data_ptr()), n, None, None, 0))
    assert not big.fr()[:n].any()
    # a pair whose index lies outside the column is dropped on the device: the 64 cells after the column keep their contents
    guard = rand_fr(rng, 64)
    big.upload(np.concatenate([np.zeros((n, 4), np.uint64), guard]))
    bad_idx = np.array([3, n, n + 5, 0xFFFFFFFF, 7], np.uint32)
    bad_vals = rand_fr(rng, 5)
    check(lib.mi355_buf_upload_sparse(C.c_void_p(big.data_ptr()), n, capi.ptr(bad_idx), capi.ptr(bad_vals), 5))
    got = big.fr()
    assert (got[n:] == guard).all() and (got[3] == bad_vals[0]).all() and (got[7] == bad_vals[4]).all() and np.count_nonzero(got[:n].any(axis=1)) == 2
    assert lib.mi355_buf_upload_packed(C.c_void_p(big.data_ptr()), capi.ptr(col), n, 3) == capi.EBADARG
    assert lib.mi355_buf_upload_sparse(C.c_void_p(big.data_ptr()), n, capi.ptr(idx), capi.ptr(vals), n + 1) == capi.EBADARG
    # a column longer than the block that receives it is refused (blocks are carved next to each other inside slabs, an overrun would land in a neighbouring live
    # polynomial).  Block sizes are rounded up to 256 bytes = 8 cells, so "too long" starts 8 cells past the n + 64 cells `big` was asked for
    packed = np.zeros(n + 80, np.uint8)
    check(lib.mi355_buf_upload_packed(C.c_void_p(big.data_ptr()), capi.ptr(packed), n + 64, 1))
    assert lib.mi355_buf_upload_packed(C.c_void_p(big.data_ptr()), capi.ptr(packed), n + 72, 1) == capi.EBADARG and b"exceeds the block" in lib.mi355_last_error()
    assert lib.mi355_buf_upload_packed(C.c_void_p(big.data_ptr() + 32 * 64), capi.ptr(packed), n + 8, 1) == capi.EBADARG
    assert lib.mi355_buf_upload_sparse(C.c_void_p(big.data_ptr()), n + 72, capi.ptr(idx), capi.ptr(vals), idx.shape[0]) == capi.EBADARG and b"exceeds the block" in lib.mi355_last_error()
    assert lib.mi355_buf_upload_sparse(C.c_void_p(big.data_ptr() + 32 * 72), n, None, None, 0) == capi.EBADARG
    assert (big.fr()[:n + 64] == 0).all()                                                # the refused calls wrote nothing; the accepted one zeroed the block
    b.free(); big.free()


def test_slabs_serve_another_block_size_without_going_back_to_hip(zk):
    """blocks of one size, freed, then blocks of ANOTHER size: the library carves them out of the same slabs (lib_core.hip "Slabs") -- HIP's free memory does not move, the new
    blocks lie inside the address ranges of the old ones, their contents are what was uploaded; mi355_buf_trim gives whole slabs back"""
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    check(lib.mi355_buf_trim())
    rng = np.random.default_rng(5)
    small, big = 32 << 20, 96 << 20
    a = [h2.DeviceBuffer(small) for _ in range(48)]                     # 1.5 GiB in 32 MiB blocks: two 1 GiB slabs
    lo, hi = min(b.data_ptr() for b in a), max(b.data_ptr() + small for b in a)
    for b in a:
        b.free()
    before = h2.mem_info()
    assert before["pooled"] >= 48 * small
    data = [rng.integers(0, 1 << 63, size=(big // 32, 4), dtype=np.uint64) for _ in range(3)]
    c = [h2.DeviceBuffer.from_host(d) for d in data]                    # 96 MiB blocks: no pooled block has this size
    after = h2.mem_info()
    assert abs(int(after["free"]) - int(before["free"])) < (8 << 20), (before, after)      # nothing was hipMalloc'd or hipFree'd
    assert all(lo <= b.data_ptr() and b.data_ptr() + big <= hi for b in c)
    assert len({b.data_ptr() for b in c}) == 3 and all((b.fr() == d).all() for b, d in zip(c, data))
    small_again = h2.DeviceBuffer.from_host(data[0][: small // 32])
    assert (small_again.fr() == data[0][: small // 32]).all() and lo <= small_again.data_ptr() < hi
    for b in c + [small_again]:
        b.free()
    check(lib.mi355_buf_trim())
    assert int(h2.mem_info()["free"]) >= int(after["free"]) + (1 << 30)                        # the slabs went back to HIP
