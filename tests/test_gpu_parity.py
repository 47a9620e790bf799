"""
-m gpu parity tests: the HIP path (through the C-ABI, via the halo2 mirror) against the CPU oracle on the same
seeded inputs -- bit-exact (raw Montgomery bytes for NTT, normalised affine for MSM).  Sizes are kept where the
oracle finishes in seconds; the full BASELINE sizes are covered by size-independent properties in
tests/test_gpu_properties.py.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import R, affine_of, rand_fr, rand_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    yield pkg
    pkg._capi.check(pkg._capi.lib().mi355_msm_set_window_bits(0))


# ------------------------------------------------------------------------------------------------ NTT
@pytest.mark.parametrize("k", list(range(0, 15)) + [16, 18, 19, 20])
def test_best_fft_matches_oracle(zk, k):
    h2 = zk.halo2
    rng = np.random.default_rng(1000 + k)
    n = 1 << k
    a = rand_fr(rng, n)
    w = h2.fr(pyref.omega(k))
    want = cref.best_fft(a, w, k)
    got = a.copy()
    h2.best_fft(got, w, k)
    assert (got == want).all()


@pytest.mark.parametrize("k", [1, 5, 9, 13])
def test_ntt_patterns(zk, k):
    h2 = zk.halo2
    n = 1 << k
    w = h2.fr(pyref.omega(k))
    one, zero = cref.fr_mont(1), cref.fr_mont(0)
    delta = np.tile(zero, (n, 1)); delta[0] = one
    a = delta.copy(); h2.best_fft(a, w, k)
    assert (a == np.tile(one, (n, 1))).all()                       # NTT(delta_0) = all ones
    a = np.tile(one, (n, 1)); h2.best_fft(a, w, k)
    want = np.tile(zero, (n, 1)); want[0] = cref.fr_mont(n)
    assert (a == want).all()                                       # NTT(all ones) = n * delta_0
    a = np.tile(zero, (n, 1)); h2.best_fft(a, w, k)
    assert (a == 0).all()
    d1 = np.tile(zero, (n, 1)); d1[1] = one; h2.best_fft(d1, w, k)  # NTT(delta_1)[i] = omega^i (transpose-detecting)
    assert (d1 == cref.best_fft(np.vstack([zero, one] + [zero] * (n - 2)), w, k)).all()


@pytest.mark.parametrize("k", [3, 8, 12, 17])
def test_domain_roundtrip_and_ifft(zk, k):
    h2 = zk.halo2
    rng = np.random.default_rng(2000 + k)
    dom = h2.EvaluationDomain(4, k)
    a = rand_fr(rng, 1 << k)
    b = a.copy()
    dom.coeff_to_lagrange(b)
    c = b.copy()
    dom.lagrange_to_coeff(c)
    assert (c == a).all()
    assert (c == cref.ifft(b, dom.omega_inv, k, dom.ifft_divisor)).all()
    # KAT A1/A2 consistency: the domain constants the mirror derives equal the fixture's for k = 25/26 (host-only arithmetic)


@pytest.mark.parametrize("k,j", [(4, 4), (7, 3), (10, 5), (13, 4)])
def test_coset_extension_matches_oracle(zk, k, j):
    h2 = zk.halo2
    rng = np.random.default_rng(3000 + k)
    dom = h2.EvaluationDomain(j, k)
    coeffs = rand_fr(rng, 1 << k)
    ext = dom.coeff_to_extended(coeffs)
    want = cref.coeff_to_extended(coeffs, k, dom.extended_k, dom.g_coset, dom.g_coset_inv, dom.extended_omega)
    assert ext.shape == want.shape and (ext == want).all()
    back = dom.extended_to_coeff(ext)
    wantb = cref.extended_to_coeff(want, dom.extended_k, dom.g_coset, dom.g_coset_inv, dom.extended_omega_inv, dom.extended_ifft_divisor)
    assert (back == wantb[: back.shape[0]]).all()
    assert (back[: 1 << k] == coeffs).all() and (back[1 << k:] == 0).all()


# ------------------------------------------------------------------------------------------------ MSM
@pytest.fixture(scope="module")
def points(zk):
    rng = np.random.default_rng(42)
    return rand_points(rng, 2048)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 31, 32, 33, 100, 255, 256, 257, 1000, 2048])
def test_best_multiexp_matches_oracle(zk, points, n):
    h2 = zk.halo2
    rng = np.random.default_rng(4000 + n)
    sc = rand_fr(rng, n)
    got = affine_of(h2.best_multiexp(sc, points[:n]))
    want = cref.g1_to_affine(cref.best_multiexp(sc, points[:n]))
    assert (got == want).all()
    if n <= 33:
        assert (got == cref.g1_to_affine(cref.msm_naive(sc, points[:n]))).all()


@pytest.mark.parametrize("c", [2, 5, 8, 11, 13, 16])
def test_msm_window_bits_do_not_change_the_result(zk, points, c):
    h2 = zk.halo2
    rng = np.random.default_rng(77)
    n = 1500
    sc = rand_fr(rng, n)
    zk._capi.check(zk._capi.lib().mi355_msm_set_window_bits(c))
    try:
        got = affine_of(h2.best_multiexp(sc, points[:n]))
    finally:
        zk._capi.check(zk._capi.lib().mi355_msm_set_window_bits(0))
    assert (got == cref.g1_to_affine(cref.best_multiexp(sc, points[:n]))).all()


def test_msm_edge_cases(zk, points):
    h2 = zk.halo2
    rng = np.random.default_rng(5)
    n = 512
    pts = points[:n]
    zero, one = cref.fr_mont(0), cref.fr_mont(1)

    def gpu(sc, bases=pts):
        return affine_of(h2.best_multiexp(np.ascontiguousarray(sc), np.ascontiguousarray(bases)))

    def cpu(sc, bases=pts):
        return cref.g1_to_affine(cref.best_multiexp(np.ascontiguousarray(sc), np.ascontiguousarray(bases)))

    assert (gpu(np.tile(zero, (n, 1))) == 0).all()                                  # all-zero scalars -> identity
    assert (gpu(np.tile(one, (n, 1))) == cpu(np.tile(one, (n, 1)))).all()            # all ones = sum of the points (one giant bucket)
    e = np.tile(zero, (n, 1)); e[137] = one
    assert (gpu(e) == pts[137]).all()                                               # unit vector
    neg1 = np.tile(cref.fr_mont(R - 1), (n, 1))
    assert (gpu(neg1) == cpu(neg1)).all()                                           # r - 1 everywhere (negative digits, top window)
    sc = rand_fr(rng, n)
    b2 = pts.copy(); b2[::3] = 0                                                    # identity bases contribute nothing
    assert (gpu(sc, b2) == cpu(sc, b2)).all()
    rep = np.tile(pts[5], (n, 1))                                                   # one point repeated: exercises P + P inside buckets
    assert (gpu(sc, rep) == cpu(sc, rep)).all()
    assert (gpu(np.tile(cref.fr_mont(7), (n, 1)), rep) == cpu(np.tile(cref.fr_mont(7), (n, 1)), rep)).all()
    pm = pts.copy(); negp = pts[0].copy()
    negy = cref.f_sub(cref.FQ, np.zeros(4, dtype=np.uint64), pts[0][4:]); negp[4:] = negy
    pm[0::2] = pts[0]; pm[1::2] = negp                                              # P, -P, P, -P ... with equal scalars -> identity
    assert (gpu(np.tile(cref.fr_mont(5), (n, 1)), pm) == 0).all()
    assert (gpu(sc, pm) == cpu(sc, pm)).all()
    # linearity: (s + t).P == s.P + t.P
    t = rand_fr(rng, n)
    st = np.stack([cref.f_add(cref.FR, sc[i], t[i]) for i in range(n)])
    lhs = gpu(st)
    rhs = cref.g1_to_affine(cref.g1_add(cref.best_multiexp(sc, pts), cref.best_multiexp(t, pts)))
    assert (lhs == rhs).all()


def test_msm_witness_like_distribution(zk, points):
    """60 % zero, 20 % < 256, 10 % < 2^64, 10 % uniform: heavily skewed buckets."""
    h2 = zk.halo2
    rng = np.random.default_rng(6)
    n = 2048
    kind = rng.random(n)
    vals = []
    for i in range(n):
        if kind[i] < 0.6: vals.append(0)
        elif kind[i] < 0.8: vals.append(int(rng.integers(1, 256)))
        elif kind[i] < 0.9: vals.append(int(rng.integers(0, 2**63)))
        else: vals.append(int(rng.integers(0, 2**63)) ** 4 % R)
    sc = np.stack([h2.fr(v) for v in vals])
    got = affine_of(h2.best_multiexp(sc, points[:n]))
    assert (got == cref.g1_to_affine(cref.best_multiexp(sc, points[:n]))).all()


def test_g1_sum_and_errors(zk, points):
    h2 = zk.halo2
    G = cref.g1_generator()
    parts = np.stack([cref.g1_mul(points[i], cref.fr_mont(i + 3)) for i in range(8)])   # non-normalised Jacobians
    want = parts[0]
    for p in parts[1:]:
        want = cref.g1_add(want, p)
    assert (affine_of(h2.g1_sum(parts)) == cref.g1_to_affine(want)).all()
    assert (affine_of(h2.g1_sum(np.zeros((3, 12), dtype=np.uint64))) == 0).all()
    with pytest.raises(AssertionError):            # best_multiexp asserts equal lengths
        h2.best_multiexp(rand_fr(np.random.default_rng(0), 4), points[:5])
    with pytest.raises(zk.Mi355Error):             # unknown SRS handle
        h2.best_multiexp(rand_fr(np.random.default_rng(0), 4), h2.SrsSlice(987654, 0, 4))
    _ = G


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 1000, 2048])
def test_batch_normalize_matches_oracle(zk, points, n):
    """group::Curve::batch_normalize: random Jacobian representatives (x z^2, y z^3, z), identities in between, ragged tile sizes;
    host pointers and device-resident buffers against the oracle's per-point to_affine."""
    import torch
    h2 = zk.halo2
    rng = np.random.default_rng(900 + n)
    zs = rand_fr(rng, max(n, 1), full=False)[:n]
    jac = np.zeros((n, 12), dtype=np.uint64)
    for i in range(n):
        z = cref.f_mul(cref.FQ, zs[i], zs[i]) if i % 3 else zs[i]          # any non-zero field element (read as an Fq Montgomery residue)
        z2 = cref.f_mul(cref.FQ, z, z)
        jac[i, 0:4] = cref.f_mul(cref.FQ, points[i, 0:4], z2)
        jac[i, 4:8] = cref.f_mul(cref.FQ, points[i, 4:8], cref.f_mul(cref.FQ, z2, z))
        jac[i, 8:12] = z
    for i in range(0, n, 7):
        jac[i, 8:12] = 0                                                    # identity: z = 0 with arbitrary x, y
    if n > 300:
        jac[256:300, 8:12] = 0                                              # a run of identities across a tile boundary
    want = cref.g1_to_affine(jac) if n else np.zeros((0, 8), dtype=np.uint64)
    got = h2.batch_normalize(jac)
    assert got.shape == (n, 8) and (got == want).all()
    if n:
        assert (got[0] == 0).all()
        d_in = torch.from_numpy(jac.view(np.int64)).cuda()
        d_out = torch.empty((n, 8), dtype=torch.int64, device="cuda")
        h2.batch_normalize(d_in, d_out)
        zk._capi.check(zk._capi.lib().mi355_synchronize())
        assert (d_out.cpu().numpy().view(np.uint64) == want).all()
        with pytest.raises(zk.Mi355Error):                                  # overlapping buffers are refused
            zk._capi.check(zk._capi.lib().mi355_g1_batch_normalize_dev(zk._capi.ptr(d_in), zk._capi.ptr(d_in), n))


def test_registered_srs_offsets(zk, points):
    h2 = zk.halo2
    params = h2.ParamsKZG.from_host(11, points, points[::-1].copy())
    rng = np.random.default_rng(9)
    sc = rand_fr(rng, 700)
    got = affine_of(h2.best_multiexp(sc, params.g_slice(300, 700)))
    assert (got == cref.g1_to_affine(cref.best_multiexp(sc, points[300:1000]))).all()
    got = affine_of(h2.best_multiexp(sc, params.g_lagrange_slice(0, 700)))
    assert (got == cref.g1_to_affine(cref.best_multiexp(sc, points[::-1][:700].copy()))).all()
    with pytest.raises(zk.Mi355Error):
        h2.best_multiexp(sc, params.g_slice(1500, 700))   # runs past the basis
    params.release()


def test_fixed_base_and_synthetic_srs(zk):
    """ParamsKZG::setup restated on the device vs the oracle's setup; upstream test_commit_lagrange property."""
    import torch
    h2 = zk.halo2
    k, n = 6, 64
    tau = 0x1F2E3D4C5B6A79880123456789ABCDEF
    params = h2.ParamsKZG.setup(k, tau)
    g = params._owner[0].cpu().numpy().view(np.uint64).reshape(n, 8)
    gl = params._owner[1].cpu().numpy().view(np.uint64).reshape(n, 8)
    og, ogl, _, _ = cref.srs_setup(k, h2.fr(tau), h2.fr(pyref.omega(k)))
    assert (g == og).all() and (gl == ogl).all()
    rng = np.random.default_rng(10)
    evals = rand_fr(rng, n)
    dom = h2.EvaluationDomain(4, k)
    coeffs = evals.copy(); dom.lagrange_to_coeff(coeffs)
    c1 = affine_of(params.commit(coeffs)); c2 = affine_of(params.commit_lagrange(evals))
    assert (c1 == c2).all()
    p_tau = cref.eval_polynomial(coeffs, h2.fr(tau))
    assert (c1 == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), p_tau))).all()
    params.release()
    _ = torch


def test_giant_buckets_take_the_parallel_fixup(zk):
    """all-equal scalars put every point of a window into ONE bucket (spanning thousands of accumulate threads): the queued,
    workgroup-parallel fix-up (k_msm_fixup_big) must give the same answer as n * P arithmetic in the oracle."""
    import torch
    h2 = zk.halo2
    k = 17
    params = h2.ParamsKZG.setup(k, 0xABCDEF0123)
    n = 1 << k
    for tables in (False, True):          # per-window bucket sets, then ONE shared set (window tables): buckets of 2^17 entries span
        if tables:                        # thousands of accumulate threads -> the multi-workgroup fix-up (k_msm_fixup_huge) as well
            params.precompute()
        for val in (1, 2**40 + 3, pyref.R_MOD - 1, pyref.R_MOD - 2**70 - 9, 0x1234567890abcdef1234567890abcdef1234567890abcdef):
            sc = np.tile(h2.fr(val), (n, 1))
            got = affine_of(params.commit(sc))
            # sum_i val * tau^i G = val * (tau^n - 1)/(tau - 1) G
            tau = 0xABCDEF0123
            s = val * (pow(tau, n, R) - 1) * pow(tau - 1, -1, R) % R
            want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), h2.fr(s)))
            assert (got == want).all(), (tables, hex(val))
    params.release()
    _ = torch


@pytest.mark.parametrize("distinct,zero_frac", [(4, 0.0), (24, 0.0), (3, 0.9), (200, 0.5)])
def test_many_giant_buckets_and_sparse_columns(zk, distinct, zero_frac):
    """columns of a few distinct values (every (value, window) pair is ONE bucket of n / distinct entries: dozens to hundreds of giant buckets
    whose partial sums are cut into 1 .. 64 slices, indexed slice * count + bucket on the device) and mostly-zero columns (the accumulate
    segment is derived from the actual entry count): commit(p) = p(tau) G in the field, window tables on and off."""
    import torch
    h2 = zk.halo2
    k, tau = 18, 0x5EED5EED
    n = 1 << k
    params = h2.ParamsKZG.setup(k, tau)
    rng = np.random.default_rng(distinct * 1000 + int(zero_frac * 10))
    vals = np.stack([h2.fr(int.from_bytes(rng.bytes(31), "little") % R) for _ in range(distinct)] + [h2.fr(0)])
    pick = rng.integers(0, distinct, size=n)
    pick[rng.random(n) < zero_frac] = distinct
    sc = np.ascontiguousarray(vals[pick])
    want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(sc, h2.fr(tau))))
    for tables in (False, True):
        if tables:
            params.precompute()
        assert (affine_of(params.commit(sc)) == want).all(), (distinct, zero_frac, tables)
        assert (affine_of(params.commit(torch.from_numpy(sc.view(np.int64)).cuda())) == want).all()
    params.release()


@pytest.mark.parametrize("c", [0, 5, 13])
def test_precomputed_window_tables_and_shared_bucket_msm(zk, points, c):
    """mi355_srs_precompute: rows T[w][i] = 2^(c w) P_i against the oracle, then MSMs (whole basis, slices, edge scalars)
    through the shared-bucket schedule against best_multiexp."""
    import torch
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    n = 1024
    pts = points[:n].copy(); pts[7] = 0                      # an identity base inside the basis
    params = h2.ParamsKZG.from_host(10, pts, pts[::-1].copy())
    params.precompute(c=c)
    ptr_, c_, w_ = C.c_void_p(), C.c_int(), C.c_int()
    check(lib.mi355_srs_pre_dev_ptr(params._g, C.byref(ptr_), C.byref(c_), C.byref(w_)))
    cc, W = c_.value, w_.value
    assert W == (255 + cc - 1) // cc and (c == 0 or cc == c)
    # read the table back through torch (plumbing) and check a few rows/points against the oracle
    import ctypes
    tbl = torch.empty(W * n * 64, dtype=torch.uint8, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(tbl.data_ptr()), ptr_, C.c_size_t(W * n * 64), 3) == 0   # device to device
    t = tbl.cpu().numpy().view(np.uint64).reshape(W, n, 8)
    assert (t[0] == pts).all()
    for w in (1, W // 2, W - 1):
        for i in (0, 7, 513, n - 1):
            want = cref.g1_to_affine(cref.g1_mul(pts[i], h2.fr(pow(2, cc * w, R))))
            assert (t[w][i] == want).all(), (w, i)
    rng = np.random.default_rng(1234 + c)
    for off, m in ((0, n), (0, 1000), (100, 700), (1023, 1), (5, 64)):
        sc = rand_fr(rng, m)
        got = affine_of(h2.best_multiexp(sc, params.g_slice(off, m)))
        assert (got == cref.g1_to_affine(cref.best_multiexp(sc, pts[off:off + m]))).all(), (off, m)
    one = np.tile(cref.fr_mont(1), (n, 1)); neg1 = np.tile(cref.fr_mont(R - 1), (n, 1)); zero = np.tile(cref.fr_mont(0), (n, 1))
    for sc in (one, neg1, zero):
        got = affine_of(h2.best_multiexp(sc, params.g_lagrange_slice(0, n)))
        assert (got == cref.g1_to_affine(cref.best_multiexp(sc, pts[::-1].copy()))).all()
    params.release()


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 255, 4096, 4097, 100003, 1 << 20])
def test_eval_polynomial_matches_oracle(zk, n):
    h2 = zk.halo2
    rng = np.random.default_rng(500 + n)
    poly = rand_fr(rng, n) if n else np.zeros((0, 4), dtype=np.uint64)
    for x in (0, 1, 0x1234567, R - 1, pyref.omega(20)):
        pt = h2.fr(x)
        got = h2.eval_polynomial(poly, pt)
        want = cref.eval_polynomial(poly, pt) if n else cref.fr_mont(0)
        assert (got == want).all(), (n, x)


def test_concurrent_callers_are_serialised_correctly(zk, points):
    """SURVEY 8b threading: rayon workers may issue commits concurrently -> entry points must be re-entrant."""
    import threading
    h2 = zk.halo2
    rng = np.random.default_rng(99)
    jobs = []
    for t in range(6):
        n = 200 + 37 * t
        sc = rand_fr(rng, n)
        k = 6 + t
        a = rand_fr(rng, 1 << k)
        jobs.append((sc, points[:n], a, k))
    out = [None] * len(jobs)

    def work(i):
        sc, pts, a, k = jobs[i]
        res = []
        for _ in range(3):
            m = affine_of(h2.best_multiexp(sc, pts))
            b = a.copy(); h2.best_fft(b, h2.fr(pyref.omega(k)), k)
            res.append((m, b))
        out[i] = res

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th: t.start()
    for t in th: t.join()
    for i, (sc, pts, a, k) in enumerate(jobs):
        want_m = cref.g1_to_affine(cref.best_multiexp(sc, pts)); want_f = cref.best_fft(a, h2.fr(pyref.omega(k)), k)
        for m, b in out[i]:
            assert (m == want_m).all() and (b == want_f).all()


def test_error_paths_do_not_crash(zk, points):
    lib, capi = zk._capi.lib(), zk._capi
    out = np.zeros(12, dtype=np.uint64); sc = rand_fr(np.random.default_rng(1), 8)
    assert lib.mi355_msm_g1_host(424242, 0, capi.ptr(sc), 8, capi.ptr(out)) == capi.EBADARG and b"handle" in lib.mi355_last_error()
    assert lib.mi355_msm_g1_adhoc_host(None, capi.ptr(sc), 8, capi.ptr(out)) == capi.EBADARG
    assert lib.mi355_ntt_fr_host(capi.ptr(sc), 29, capi.ptr(sc[0])) == capi.EBADARG            # beyond the two-adicity of Fr
    assert lib.mi355_ntt_fr_host(None, 3, capi.ptr(sc[0])) == capi.EBADARG
    assert lib.mi355_msm_set_window_bits(40) == capi.EBADARG
    assert lib.mi355_srs_release(424242) == capi.EBADARG
    assert lib.mi355_srs_precompute(424242, 0, 0) == capi.EBADARG
    # n == 0: identity / no-op, as best_multiexp on empty slices
    assert lib.mi355_msm_g1_adhoc_host(capi.ptr(points[:1]), capi.ptr(sc), 0, capi.ptr(out)) == capi.OK and (out == 0).all()
    # entry points added later in the round: same contract (bad argument -> EBADARG, never a crash)
    h = C.c_uint64()
    assert lib.mi355_srs_downsize(424242, 3, capi.ptr(sc[0]), capi.ptr(sc[1]), C.byref(h)) == capi.EBADARG
    assert lib.mi355_srs_read_host(424242, 0, 1, capi.ptr(out)) == capi.EBADARG
    assert lib.mi355_g1_fft_host(capi.ptr(out), 29, capi.ptr(sc[0])) == capi.EBADARG
    assert lib.mi355_g1_fft_host(None, 2, capi.ptr(sc[0])) == capi.EBADARG
    assert lib.mi355_g_to_lagrange_dev(None, None, 2, capi.ptr(sc[0]), capi.ptr(sc[1])) == capi.EBADARG
    assert lib.mi355_fr_batch_invert_dev(None, 5) == capi.EBADARG and lib.mi355_fr_batch_invert_dev(None, 0) == capi.OK
    assert lib.mi355_fr_prefix_product_dev(None, None, 5, None) == capi.EBADARG
    assert lib.mi355_fr_prefix_sum_dev(None, None, 5, None) == capi.EBADARG
    assert lib.mi355_msm_set_pipeline(99, 0) == capi.EBADARG
    assert lib.mi355_msm_g1_batch_host(424242, 0, None, 0, 8, capi.ptr(out)) == capi.EBADARG
    assert lib.mi355_g1_sum_dev(None, 3, capi.ptr(out)) == capi.EBADARG
    h2 = zk.halo2
    small = h2.ParamsKZG.from_host(3, points[:8], points[:8])
    assert lib.mi355_srs_downsize(small._g, 4, capi.ptr(sc[0]), capi.ptr(sc[1]), C.byref(h)) == capi.EBADARG   # 2^k exceeds the basis
    assert lib.mi355_srs_read_host(small._g, 4, 5, capi.ptr(np.zeros((5, 8), dtype=np.uint64))) == capi.EBADARG
    small.release()
    # a working call still works afterwards
    got = affine_of(zk.halo2.best_multiexp(sc, points[:8]))
    assert (got == cref.g1_to_affine(cref.best_multiexp(sc, points[:8]))).all()


@pytest.mark.parametrize("k", [3, 9, 14, 17])
def test_coset_parts_equal_the_extended_domain(zk, k):
    """distribute_powers + NTT on the coset (zeta * w_ext^j) H must reproduce the interleaved rows of coeff_to_extended:
    ext[q * j_stride ...]: evaluation at zeta * w_ext^(j + Q i) for the i-th point of part j (w_ext^Q = w)."""
    import torch
    h2 = zk.halo2
    rng = np.random.default_rng(800 + k)
    dom = h2.EvaluationDomain(5, k)            # extended_k = k + 2, Q = 4 parts
    Q = 1 << (dom.extended_k - k)
    coeffs = rand_fr(rng, 1 << k)
    ext = dom.coeff_to_extended(coeffs)         # oracle-checked elsewhere
    cd = torch.from_numpy(coeffs.view(np.int64)).cuda()
    out = torch.empty_like(cd)
    for j in range(Q):
        dom.coeff_to_extended_part(cd, j, out)
        got = out.cpu().numpy().view(np.uint64)
        assert (got == ext[j::Q]).all(), j
    # distribute_powers alone against the oracle
    f = 0x9876543210FEDCBA
    d = torch.from_numpy(coeffs.view(np.int64)).cuda()
    zk._capi.check(zk._capi.lib().mi355_distribute_powers_fr_dev(zk._capi.ptr(d), 1 << k, zk._capi.ptr(h2.fr(f))))
    got = d.cpu().numpy().view(np.uint64)
    pw = np.stack([h2.fr(pow(f, i, R)) for i in range(min(1 << k, 600))])
    assert (got[: pw.shape[0]] == cref.f_mul_vec(cref.FR, coeffs[: pw.shape[0]], pw)).all()


def test_unnormalised_partials_fold_to_the_same_point(zk, points):
    """mi355_msm_set_normalise(0): per-GPU partial sums as arbitrary Jacobian representatives, folded by g1_sum (the multi-GPU path)."""
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    rng = np.random.default_rng(2024)
    n = 1536
    sc = rand_fr(rng, n)
    check(lib.mi355_msm_set_normalise(0))
    try:
        parts = np.stack([h2.best_multiexp(sc[lo:hi], points[lo:hi]) for lo, hi in ((0, 500), (500, 1100), (1100, n))])
        zero_part = h2.best_multiexp(np.tile(cref.fr_mont(0), (4, 1)), points[:4])
    finally:
        check(lib.mi355_msm_set_normalise(1))
    one_q = np.array(pyref.to_limbs(pyref.MONT_R % pyref.P_MOD), dtype=np.uint64)
    assert not all((p[8:] == one_q).all() for p in parts), "partials should not be normalised in this mode"
    assert (zero_part[8:] == 0).all()
    for p, (lo, hi) in zip(parts, ((0, 500), (500, 1100), (1100, n))):
        assert (cref.g1_to_affine(p) == cref.g1_to_affine(cref.best_multiexp(sc[lo:hi], points[lo:hi]))).all()
    total = affine_of(h2.g1_sum(np.vstack([parts, zero_part[None]])))
    assert (total == cref.g1_to_affine(cref.best_multiexp(sc, points[:n]))).all()


@pytest.mark.parametrize("pre_c", [None, 0, 6])
def test_batched_commitments_equal_separate_commitments(zk, points, pre_c):
    """mi355_msm_g1_batch_dev (ParamsKZG.commit_many): M polynomials over one basis in one pass == M x best_multiexp,
    with per-window buckets (no tables) and with the shared-bucket schedule (window tables)."""
    import torch
    h2 = zk.halo2
    n = 1000
    pts = points[:1024].copy(); pts[11] = 0
    params = h2.ParamsKZG.from_host(10, pts, pts[::-1].copy())
    if pre_c is not None:
        params.precompute(c=pre_c)
    rng = np.random.default_rng(9100 + (pre_c or 0))
    for M in (1, 2, 5, 16):
        polys = [rand_fr(rng, n) for _ in range(M)]
        if M >= 5:
            polys[2] = np.tile(cref.fr_mont(0), (n, 1))             # a zero column: the identity in the middle of the batch
            polys[3] = np.tile(cref.fr_mont(R - 1), (n, 1))
            polys[4][:, :] = 0; polys[4][17] = cref.fr_mont(5)      # a single non-zero coefficient
        dev = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
        got = params.commit_many(dev)
        assert got.shape == (M, 12)
        for m in range(M):
            want = cref.g1_to_affine(cref.best_multiexp(polys[m], pts[:n]))
            assert (affine_of(got[m]) == want).all(), (M, m)
            assert (got[m] == params.commit(dev[m])).all()
        assert (params.commit_many(polys) == got).all()                 # host pointers: mi355_msm_g1_batch_host
        gl = params.commit_many([torch.from_numpy(np.vstack([p, p[:24]]).view(np.int64)).cuda() for p in polys], lagrange=True)
        for m in range(M):
            want = cref.g1_to_affine(cref.best_multiexp(np.vstack([polys[m], polys[m][:24]]), pts[::-1].copy()))
            assert (affine_of(gl[m]) == want).all()
    params.release()


def test_batched_commitments_split_oversized_batches(zk, points):
    """a batch whose coarse histogram would not fit is processed as half batches; results keep their order."""
    import torch
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    n, M = 300, 53
    params = h2.ParamsKZG.from_host(9, points[:512], points[:512])
    rng = np.random.default_rng(9200)
    polys = [rand_fr(rng, n, full=False) for _ in range(M)]
    dev = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
    check(lib.mi355_msm_set_window_bits(16))          # 16 windows x 16 coarse bins per polynomial -> 53 x 256 regions > the LDS budget
    try:
        got = params.commit_many(dev)
    finally:
        check(lib.mi355_msm_set_window_bits(0))
    for m in range(M):
        assert (affine_of(got[m]) == cref.g1_to_affine(cref.best_multiexp(polys[m], points[:n]))).all(), m
    # error behaviour: null polynomial pointer, null output
    arr = (C.c_void_p * 2)(dev[0].data_ptr(), None)
    out = np.zeros((2, 12), dtype=np.uint64)
    assert lib.mi355_msm_g1_batch_dev(params._g, 0, arr, 2, n, zk._capi.ptr(out)) == zk._capi.EBADARG
    assert lib.mi355_msm_g1_batch_dev(params._g, 0, arr, 0, n, zk._capi.ptr(out)) == 0
    assert lib.mi355_msm_g1_batch_dev(params._g, 500, arr, 1, n, zk._capi.ptr(out)) == zk._capi.EBADARG   # slice past the basis
    params.release()


def _affine_to_jac(a):
    a = np.asarray(a, dtype=np.uint64)
    j = np.zeros((a.shape[0], 12), dtype=np.uint64)
    j[:, :8] = a
    one_q = np.array(pyref.to_limbs(pyref.MONT_R % pyref.P_MOD), dtype=np.uint64)
    j[(a != 0).any(axis=1), 8:] = one_q
    return j


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 7, 10])
def test_g1_fft_matches_oracle(zk, points, k):
    """best_fft::<Fr, G1> (mi355_g1_fft_host / _dev) against the oracle's serial restatement; inputs include the identity, repeated
    points (the doubling branch of the butterfly) and a non-normalised Jacobian representative."""
    import torch
    h2 = zk.halo2
    n = 1 << k
    pts = points[100:100 + n].copy()
    if n >= 8:
        pts[3] = 0; pts[5] = pts[4]; pts[6] = pts[4]; pts[6][4:] = cref.g1_to_affine(cref.g1_mul(pts[4], h2.fr(R - 1)))[4:]   # identity, P == Q, P == -Q
    jac = _affine_to_jac(pts)
    if n >= 2:   # scale point 1 to another representative: (X l^2, Y l^3, Z l)
        lam = 0x1234567
        X, Y = (cref.limbs_to_int(cref.f_to_canonical_vec(cref.FQ, jac[1, 4 * i:4 * i + 4].reshape(1, 4))[0]) for i in (0, 1))
        P = pyref.P_MOD
        rep = [X * lam * lam % P, Y * lam ** 3 % P, lam]
        jac[1] = np.concatenate([cref.f_from_canonical_vec(cref.FQ, np.array(pyref.to_limbs(v), dtype=np.uint64).reshape(1, 4))[0] for v in rep])
    w = h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << (28 - k), R))
    want = cref.g1_to_affine(cref.best_fft_g1(jac, w, k))
    a = jac.copy(); h2.best_fft(a, w, k)
    assert all((affine_of(a[i]) == want[i]).all() for i in range(n))
    d = torch.from_numpy(jac.view(np.int64)).cuda(); h2.best_fft(d, w, k)
    assert (d.cpu().numpy().view(np.uint64) == a).all()


def test_downsize_rebuilds_g_lagrange(zk):
    """ParamsKZG::downsize [REF integration/tests/integration.rs:17-22]: g_lagrange of the downsized parameters == oracle g_to_lagrange
    (small k), == the closed-form Lagrange basis of a fresh setup with the same tau (larger k), and commit == commit_lagrange after."""
    import torch
    h2 = zk.halo2
    tau = 0x5343524f4c4c0001
    big = h2.ParamsKZG.setup(7, tau)
    g_host = big.read_g()
    assert (g_host == big._owner[0].cpu().numpy().view(np.uint64).reshape(128, 8)).all()
    big.downsize(5)
    assert (big.k, big.n) == (5, 32)
    w_inv = pow(pow(h2.FR_ROOT_OF_UNITY, 1 << 23, R), R - 2, R)
    want = cref.g_to_lagrange(g_host[:32], 5, h2.fr(w_inv), h2.fr(pow(32, R - 2, R)))
    assert (big.read_g(lagrange=True) == want).all() and (big.read_g() == g_host[:32]).all()
    # the stand-alone transform on caller-owned device memory gives the same points
    assert (h2.g_to_lagrange(big._owner[0], 5).cpu().numpy().view(np.uint64).reshape(32, 8) == want).all()
    big.release()
    p14 = h2.ParamsKZG.setup(14, tau)
    p14.precompute()
    p14.downsize(12)
    p12 = h2.ParamsKZG.setup(12, tau)
    assert (p14.read_g(lagrange=True) == p12.read_g(lagrange=True)).all()
    rng = np.random.default_rng(99)
    evals = rand_fr(rng, 1 << 12)
    dom = h2.EvaluationDomain(3, 12)
    coeffs = evals.copy(); dom.lagrange_to_coeff(coeffs)
    assert (p14.commit_lagrange(evals) == p14.commit(coeffs)).all()
    assert (p14.commit(coeffs) == p12.commit(coeffs)).all()
    with pytest.raises(AssertionError):
        p14.downsize(13)
    p14.release(); p12.release()


@pytest.mark.parametrize("n", [0, 1, 7, 255, 256, 2047, 2048, 2049, 5000, 100003, 1 << 18])
def test_batch_invert_and_prefix_product_match_oracle(zk, n):
    """mi355_fr_batch_invert_dev / mi355_fr_prefix_product_dev against the oracle (ff::BatchInvert, grand-product column): zeros stay
    zero, tile boundaries (2048), ragged tails, in-place operation."""
    import torch
    h2 = zk.halo2
    rng = np.random.default_rng(7000 + n)
    a = rand_fr(rng, n)                         # contains 0, 1, r - 1 when n >= 4
    if n > 2100:
        a[2047] = 0; a[2048] = 0; a[n - 1] = 0
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    h2.batch_invert(d)
    got = d.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert (got == cref.batch_invert(a)).all()
    b = rand_fr(rng, n, full=False)
    if n > 4:
        b[3] = cref.fr_mont(1); b[4] = cref.fr_mont(R - 1)
    src = torch.from_numpy(b.view(np.int64).copy()).cuda()
    z, total = h2.prefix_product(src, want_total=True)
    wz, wt = cref.prefix_product(b)
    assert (z.cpu().numpy().view(np.uint64).reshape(n, 4) == wz).all() and (total == wt).all()
    h2.prefix_product(src, dst=src)             # in place
    assert (src.cpu().numpy().view(np.uint64).reshape(n, 4) == wz).all()
    if n > 3000:                                # a zero factor makes everything after it zero
        b[2500] = 0
        z2, t2 = h2.prefix_product(torch.from_numpy(b.view(np.int64).copy()).cuda(), want_total=True)
        z2 = z2.cpu().numpy().view(np.uint64).reshape(n, 4)
        assert (z2[:2501] == wz[:2501]).all() and (z2[2501:] == 0).all() and (t2 == 0).all()


@pytest.mark.parametrize("n", [0, 1, 7, 256, 2047, 2048, 2049, 100003, 1 << 18])
def test_prefix_sum_matches_oracle(zk, n):
    """mi355_fr_prefix_sum_dev (the running sum phi of the mv-lookup argument) against the oracle: tile boundaries, ragged tails, in place."""
    import torch
    h2 = zk.halo2
    rng = np.random.default_rng(7100 + n)
    b = rand_fr(rng, n)                         # contains 0, 1, r - 1 when n >= 4: sums wrap around the modulus
    src = torch.from_numpy(b.view(np.int64).copy()).cuda()
    z, total = h2.prefix_sum(src, want_total=True)
    wz, wt = cref.prefix_sum(b)
    assert (z.cpu().numpy().view(np.uint64).reshape(n, 4) == wz).all() and (total == wt).all()
    h2.prefix_sum(src, dst=src)
    assert (src.cpu().numpy().view(np.uint64).reshape(n, 4) == wz).all()


def test_log_derivative_lookup_sum_closes(zk):
    """the identity the running sum exists for (mv-lookup / logUp): every looked-up value f[i] is in the table t, m[j] counts how often t[j] is
    hit; then sum_i 1 / (beta + f[i]) - m[i] / (beta + t[i]) = 0 -- computed with batch_invert, vec ops and prefix_sum on the device."""
    import torch
    h2 = zk.halo2
    n = 1 << 14
    rng = np.random.default_rng(32)
    t_int = rng.choice(1 << 40, size=n, replace=False)
    hits = rng.integers(0, n, size=n)
    m_int = np.bincount(hits, minlength=n)
    beta = 0x1234567890ABCDEF1234567890ABCDEF % R
    def col(vals):
        return np.stack([cref.fr_mont(int(v)) for v in vals])
    tb = col((int(x) + beta) % R for x in t_int); fb = col((int(t_int[j]) + beta) % R for j in hits); mm = col(m_int)
    d_t = torch.from_numpy(tb.view(np.int64).copy()).cuda(); d_f = torch.from_numpy(fb.view(np.int64).copy()).cuda(); d_m = torch.from_numpy(mm.view(np.int64).copy()).cuda()
    h2.batch_invert(d_t); h2.batch_invert(d_f)
    h2.fr_vec_op("mul", d_t, d_t, d_m)
    h2.fr_vec_op("sub", d_f, d_f, d_t)
    phi, total = h2.prefix_sum(d_f, want_total=True)
    assert (total == 0).all()
    assert (phi[0].cpu().numpy().view(np.uint64) == 0).all()
    want, _ = cref.prefix_sum(d_f.cpu().numpy().view(np.uint64).reshape(n, 4))
    assert (phi.cpu().numpy().view(np.uint64).reshape(n, 4) == want).all()


def test_permutation_grand_product_closes(zk):
    """the permutation-argument identity the grand product exists for: with numerators a permutation of the denominators,
    z = prefix_product(num * batch_invert(den)) ends at one."""
    import torch
    h2 = zk.halo2
    n = 1 << 16
    rng = np.random.default_rng(31)
    den = rand_fr(rng, n, full=False); den[den.sum(axis=1) == 0] = cref.fr_mont(1)
    num = den[rng.permutation(n)]
    d = torch.from_numpy(den.view(np.int64).copy()).cuda(); m = torch.from_numpy(num.view(np.int64).copy()).cuda()
    h2.batch_invert(d)
    h2.fr_vec_op("mul", d, d, m)
    z, total = h2.prefix_product(d, want_total=True)
    assert (total == cref.fr_mont(1)).all()
    assert (z[0].cpu().numpy().view(np.uint64) == cref.fr_mont(1)).all()


@pytest.mark.parametrize("chunks", [2, 3, 4, 7])
def test_pipelined_msm_chunks_do_not_change_the_result(zk, points, chunks):
    """mi355_msm_set_pipeline: the point range cut into slices whose sort / accumulate / reduce stages overlap on three streams;
    every slice count must give the oracle's point (plain bases and window tables, slices of a registered basis, ragged n)."""
    import torch
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    params = h2.ParamsKZG.from_host(11, points, points[::-1].copy())
    rng = np.random.default_rng(600 + chunks)
    check(lib.mi355_msm_set_pipeline(chunks, 6))
    try:
        for pre in (False, True):
            if pre:
                params.precompute(c=7)
            for off, n in ((0, 2048), (0, 1999), (37, 1500), (0, 64), (5, 63)):
                sc = rand_fr(rng, n)
                want = cref.g1_to_affine(cref.best_multiexp(sc, points[off:off + n]))
                for _ in range(2):      # twice: slot reuse across calls
                    got = affine_of(h2.best_multiexp(sc, params.g_slice(off, n)))
                    assert (got == want).all(), (pre, off, n)
                d = torch.from_numpy(sc.view(np.int64)).cuda()
                assert (affine_of(h2.best_multiexp(d, params.g_slice(off, n))) == want).all()
        zero = np.tile(cref.fr_mont(0), (2048, 1))
        assert (h2.best_multiexp(zero, params.g_slice(0, 2048)) == 0).all()
    finally:
        check(lib.mi355_msm_set_pipeline(0, 0))
    params.release()


def test_device_resident_partial_and_fold(zk, points):
    """mi355_msm_g1_dev_async + mi355_g1_sum_dev (the multi-GPU leg without host round trips): the partial stays in device memory in
    stream order, un-normalised when asked; folding the partials of two point-range shards gives the oracle's MSM."""
    import torch
    h2 = zk.halo2
    lib, check, ptr = zk._capi.lib(), zk._capi.check, zk._capi.ptr
    n = 2048
    params = h2.ParamsKZG.from_host(11, points, points)
    rng = np.random.default_rng(515)
    sc = rand_fr(rng, n)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    want = cref.g1_to_affine(cref.best_multiexp(sc, points))
    parts = torch.zeros(3 * 96, dtype=torch.uint8, device="cuda")
    check(lib.mi355_msm_set_normalise(0))
    try:
        cuts = (0, 700, 2048, 2048)         # third shard empty -> identity partial
        for r in range(3):
            lo, hi = cuts[r], cuts[r + 1]
            check(lib.mi355_msm_g1_dev_async(params._g, lo, ptr(d[lo:hi]) if hi > lo else None, hi - lo, C.c_void_p(parts.data_ptr() + 96 * r)))
    finally:
        check(lib.mi355_msm_set_normalise(1))
    out = np.zeros(12, dtype=np.uint64)
    check(lib.mi355_g1_sum_dev(ptr(parts), 3, ptr(out)))
    assert (affine_of(out) == want).all()
    assert (parts[192:].cpu().numpy() == 0).all()
    # the single-rank form of distributed.sharded_multiexp_device
    got = zk.distributed.sharded_multiexp_device(zk._capi, params._g, d, n)
    assert (affine_of(got) == want).all()
    assert lib.mi355_msm_g1_dev_async(params._g, 0, ptr(d), n, None) == zk._capi.EBADARG
    params.release()


def test_results_are_deterministic_across_runs_and_schedules(zk, points):
    """the reference's (disabled) test_deterministic / test_vk_same intent [REF integration/tests/integration.rs:51-174]: commitments are
    canonical group elements, so repeated runs, other window widths and the batched / pipelined schedules give byte-identical output."""
    import torch
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    params = h2.ParamsKZG.from_host(11, points, points[::-1].copy())
    rng = np.random.default_rng(808)
    sc = rand_fr(rng, 2048)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    ref = params.commit(d).copy()
    outs = [params.commit(d).copy() for _ in range(3)] + [params.commit(sc).copy()]
    for c in (7, 12):
        check(lib.mi355_msm_set_window_bits(c)); outs.append(params.commit(d).copy()); check(lib.mi355_msm_set_window_bits(0))
    params.precompute(c=9); outs.append(params.commit(d).copy())
    outs.append(params.commit_many([d, d])[1].copy())
    check(lib.mi355_msm_set_pipeline(3, 6)); outs.append(params.commit(d).copy()); check(lib.mi355_msm_set_pipeline(0, 0))
    assert all((o == ref).all() for o in outs)
    a = rand_fr(rng, 1 << 11)
    dom = h2.EvaluationDomain(4, 11)
    f1 = a.copy(); dom.coeff_to_lagrange(f1)
    f2 = torch.from_numpy(a.view(np.int64).copy()).cuda(); dom.coeff_to_lagrange(f2); dom.coeff_to_lagrange(f2); dom.lagrange_to_coeff(f2); 
    assert (f2.cpu().numpy().view(np.uint64).reshape(-1, 4) == f1).all()
    params.release()


def test_trace_env_prints_per_call_counters():
    """MI355_TRACE=1: one stderr line per MSM / NTT call with n, ms and the algorithmic GB/s (metrics hook)."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as ge; zk = ge.load_package(); zk.init(0); h2 = zk.halo2\n"
        "from oracle import cref\n"
        "G = cref.g1_generator(); pts = np.tile(G, (64, 1)); sc = np.tile(cref.fr_mont(3), (64, 1))\n"
        "h2.best_multiexp(sc, pts)\n"
        "a = np.tile(cref.fr_mont(5), (256, 1)); h2.best_fft(a, h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << 20, h2.R_MOD)), 8)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI355_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[mi355zk] msm_g1 n=64" in r.stderr and "[mi355zk] ntt_fr n=256" in r.stderr, r.stderr[-2000:]


def test_rccl_allgather_path_on_one_rank(points):
    """distributed.sharded_multiexp_device through a real RCCL communicator (backend "nccl", world size 1 on this one-GPU box): the
    partial is produced on the library stream, gathered by RCCL and folded on the device without host round trips."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    np.save("/tmp/_mi355_pts.npy", points[:512])
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge; from oracle import cref\n"
        "torch.cuda.set_device(0); dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib, check = zk._capi.lib(), zk._capi.check\n"
        "check(lib.mi355_set_stream(torch.cuda.current_stream().cuda_stream))\n"
        "pts = np.load('/tmp/_mi355_pts.npy'); params = h2.ParamsKZG.from_host(9, pts, pts)\n"
        "rng = np.random.default_rng(3); sc = rng.integers(0, 2**64, size=(512, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)\n"
        "d = torch.from_numpy(sc.view(np.int64)).cuda()\n"
        "check(lib.mi355_msm_set_normalise(0))\n"
        "for _ in range(3): got = zk.distributed.sharded_multiexp_device(zk._capi, params._g, d, 512)\n"
        "want = cref.g1_to_affine(cref.best_multiexp(sc, pts))\n"
        "assert (got[:8] == want).all(), (got, want)\n"
        "dist.barrier(); dist.destroy_process_group(); print('RCCL-OK')\n"
    ) % root
    import socket
    with socket.socket() as sk:                      # a free rendezvous port
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_params_file_loader_streams_into_device_memory(zk, points, tmp_path):
    """mi355_srs_load_params_file (Prover::load_params for one degree): exact-length rule, points land in HBM unchanged, G2 tail
    returned, optional on-device validation (what SerdeFormat::RawBytes checks and RawBytesUnchecked skips), downsize on load."""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    k, n = 9, 512
    g, gl = points[:n].copy(), points[n:2 * n].copy()
    g[17] = 0                                                        # an identity point is legal
    g2, s_g2 = bytes(range(128)), bytes(range(128, 256))
    path = str(tmp_path / "params9")
    h2.write_params(path, k, g, gl, g2, s_g2)
    for validate in (False, True):
        p = h2.params_from_file(path, validate=validate)
        assert (p.k, p.n) == (k, n) and p.g2 == g2 and p.s_g2 == s_g2
        assert (p.read_g() == g).all() and (p.read_g(lagrange=True) == gl).all()
        sc = rand_fr(np.random.default_rng(12), n)
        assert (affine_of(p.commit(sc)) == cref.g1_to_affine(cref.best_multiexp(sc, g))).all()
        p.release()
    # a point off the curve / a coordinate that is not reduced: accepted unchecked (as RawBytesUnchecked), rejected with validation
    for what in ("off_curve", "unreduced"):
        bad = gl.copy()
        if what == "off_curve":
            bad[100, 4] ^= np.uint64(1)
        else:
            bad[100, :4] = np.array(pyref.to_limbs(pyref.P_MOD), dtype=np.uint64)   # x = p
        h2.write_params(path, k, g, bad, g2, s_g2)
        p = h2.params_from_file(path); p.release()
        with pytest.raises(capi.Mi355Error):
            h2.params_from_file(path, validate=True)
        assert b"not on the curve" in lib.mi355_last_error()
    # the exact-length rule of load_params, and files that are not there
    h2.write_params(path, k, g, gl, g2, s_g2)
    with open(path, "ab") as f:
        f.write(b"\x00")
    with pytest.raises(capi.Mi355Error):
        h2.params_from_file(path)
    with open(path, "wb") as f:
        f.write((9).to_bytes(4, "little") + bytes(1000))
    with pytest.raises(capi.Mi355Error):
        h2.params_from_file(path)
    with pytest.raises(capi.Mi355Error):
        h2.params_from_file(str(tmp_path / "missing"))
    # load + downsize in one go equals a fresh setup of the smaller degree (synthetic SRS with known tau)
    big = h2.ParamsKZG.setup(11, 77)
    h2.write_params(path, 11, big.read_g(), big.read_g(lagrange=True))
    small = h2.params_from_file(path, validate=True, downsize_to=9)
    ref = h2.ParamsKZG.setup(9, 77)
    assert small.k == 9 and (small.read_g() == ref.read_g()).all() and (small.read_g(lagrange=True) == ref.read_g(lagrange=True)).all()
    big.release(); small.release(); ref.release()


@pytest.mark.parametrize("n", [1, 255, 4097, 100003])
def test_fr_vec_axpy_matches_big_int_arithmetic(zk, n):
    """dst = a + s * b (and dst = s * b), in place too: the linear-combination step of the multi-open argument."""
    import torch
    h2 = zk.halo2
    rng = np.random.default_rng(4400 + n)
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    s_int = int(rng.integers(1, 2**62)) * 0x1000000000000001 % R
    da, db = torch.from_numpy(a.view(np.int64).copy()).cuda(), torch.from_numpy(b.view(np.int64).copy()).cuda()
    out = torch.empty_like(da)
    h2.fr_vec_axpy(out, da, db, h2.fr(s_int))
    ai = [cref.limbs_to_int(x) for x in cref.f_to_canonical_vec(cref.FR, a)]
    bi = [cref.limbs_to_int(x) for x in cref.f_to_canonical_vec(cref.FR, b)]
    want = cref.f_from_canonical_vec(cref.FR, np.array([pyref.to_limbs((x + s_int * y) % R) for x, y in zip(ai, bi)], dtype=np.uint64))
    assert (out.cpu().numpy().view(np.uint64).reshape(n, 4) == want).all()
    h2.fr_vec_axpy(da, da, db, h2.fr(s_int))                      # dst aliases a
    assert torch.equal(da, out)
    h2.fr_vec_axpy(db, None, db, h2.fr(R - 1))                    # dst = -b, in place
    neg = cref.f_from_canonical_vec(cref.FR, np.array([pyref.to_limbs((R - y) % R) for y in bi], dtype=np.uint64))
    assert (db.cpu().numpy().view(np.uint64).reshape(n, 4) == neg).all()


@pytest.mark.parametrize("n", [1, 2, 3, 9, 2047, 2048, 2049, 2050, 4097, 100003])
def test_kate_division_matches_oracle(zk, n):
    """mi355_fr_kate_division_dev against the oracle's restatement of halo2's loop, and the defining identity p(X) - p(z) = (X - z) q(X)."""
    import torch
    h2 = zk.halo2
    rng = np.random.default_rng(5100 + n)
    a = rand_fr(rng, n)
    z_int = int(rng.integers(2, 2**62)) * 0xfffffffb % R
    z = h2.fr(z_int)
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    q = h2.kate_division(d, z)
    want = cref.kate_division(a, z)
    assert q.shape[0] == n - 1 and (q.cpu().numpy().view(np.uint64).reshape(n - 1, 4) == want).all()
    if n >= 3:
        r = h2.fr(0x123456789abcdef % R)
        pr, pz, qr = (cref.limbs_to_int(cref.f_to_canonical_vec(cref.FR, v.reshape(1, 4))[0]) for v in (h2.eval_polynomial(d, r), h2.eval_polynomial(d, z), h2.eval_polynomial(q, r)))
        assert (pr - pz) % R == (0x123456789abcdef - z_int) * qr % R
        # shifted in place: dst == poly + 1 element
        lib, check = zk._capi.lib(), zk._capi.check
        check(lib.mi355_fr_kate_division_dev(C.c_void_p(d.data_ptr() + 32), zk._capi.ptr(d), n, zk._capi.ptr(z)))
        assert (d[1:].cpu().numpy().view(np.uint64).reshape(n - 1, 4) == want).all()


def test_shutdown_releases_everything_and_reinit_works():
    """mi355_shutdown / mi355_init life cycle (own process): after shutdown every compute entry point fails loudly with
    MI355_ENODEVICE, stale handles are gone, and a second init gives a working library again."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as ge; zk = ge.load_package(); capi = zk._capi\n"
        "from oracle import cref\n"
        "lib = capi.lib(); zk.init(0); h2 = zk.halo2\n"
        "G = cref.g1_generator(); pts = np.tile(G, (32, 1)); sc = np.tile(cref.fr_mont(3), (32, 1)); out = np.zeros(12, dtype=np.uint64)\n"
        "p = h2.ParamsKZG.from_host(5, pts, pts); p.precompute(); first = p.commit(sc).copy()\n"
        "a = np.tile(cref.fr_mont(5), (256, 1)); h2.best_fft(a, h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << 20, h2.R_MOD)), 8)\n"
        "assert lib.mi355_shutdown() == 0 and lib.mi355_shutdown() == 0\n"
        "assert lib.mi355_msm_g1_host(p._g, 0, capi.ptr(sc), 32, capi.ptr(out)) == capi.ENODEVICE\n"
        "assert lib.mi355_ntt_fr_host(capi.ptr(a), 8, capi.ptr(sc[0])) == capi.ENODEVICE\n"
        "zk.init(0)\n"
        "assert lib.mi355_msm_g1_host(p._g, 0, capi.ptr(sc), 32, capi.ptr(out)) == capi.EBADARG      # the old handle died with the context\n"
        "q = h2.ParamsKZG.from_host(5, pts, pts); assert (q.commit(sc) == first).all()\n"
        "want = cref.g1_to_affine(cref.g1_mul(G, cref.fr_mont(96))); assert (first[:8] == want).all()\n"
        "print('LIFECYCLE-OK')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LIFECYCLE-OK" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


# ------------------------------------------------------------------------------------------------ G2 (ParamsKZG::setup: s_g2 = tau * G2)
def test_g2_scalar_multiplication_matches_both_oracles(zk, kat):
    h2 = zk.halo2
    gen = h2.g2_generator()
    for tau in (1, 2, 3, 0x5343524F4C4C0001, R - 1, (1 << 253) + 7):
        got = h2.g2_mul(gen, h2.fr(tau))
        assert (got == cref.g2_mul(gen, cref.fr_mont(tau))).all()
        assert list(got) == pyref.g2_to_limbs(pyref.g2_mul(pyref.G2_GEN, tau))
    assert (h2.g2_mul(gen, h2.fr(0)) == 0).all()                                     # 0 * G2 = identity (all-zero G2Affine)
    assert (h2.g2_mul(np.zeros(16, dtype=np.uint64), h2.fr(5)) == 0).all()           # k * identity
    # the production SRS's s_g2 (fixture, [REF release-v0.13.1/evm_verifier.yul:1236-1239]) as the base point
    s_g2 = cref.g2_from_words(kat["yul"]["s_g2_words"])
    assert (h2.g2_mul(s_g2, h2.fr(12345)) == cref.g2_mul(s_g2, cref.fr_mont(12345))).all()
    off = gen.copy(); off[0] ^= np.uint64(1)
    with pytest.raises(zk.Mi355Error):
        h2.g2_mul(off, h2.fr(3))                                                     # not on the twist


def test_setup_writes_a_loadable_params_file_with_real_g2_points(zk, tmp_path):
    """ParamsKZG::setup -> write (RawBytes) -> load_params: g2 = G2 generator, s_g2 = tau * G2 (no longer 128 zero bytes), both bases intact."""
    h2 = zk.halo2
    k, tau = 9, 0xABCDEF0123456789
    p = h2.ParamsKZG.setup(k, tau)
    assert p.g2 == h2.g2_generator().tobytes()
    assert p.s_g2 == cref.g2_mul(cref.g2_generator(), cref.fr_mont(tau)).tobytes()
    path = str(tmp_path / f"params{k}")
    p.write(path)
    assert os.path.getsize(path) == h2.params_file_size(k)
    q = h2.params_from_file(path, validate=True)
    assert q.k == k and q.g2 == p.g2 and q.s_g2 == p.s_g2
    assert (q.read_g() == p.read_g()).all() and (q.read_g(lagrange=True) == p.read_g(lagrange=True)).all()
    g_or, gl_or, _, _ = cref.srs_setup(k, cref.fr_mont(tau), h2.fr(pyref.omega(k)))
    assert (q.read_g() == g_or).all() and (q.read_g(lagrange=True) == gl_or).all()
    p.release(); q.release()


def test_clone_downsized_shares_the_parent_registration(zk):
    """load_params_map's clone + downsize [REF integration/tests/integration.rs:12-22]: prefix view of g (same memory, same tables),
    g_lagrange rebuilt; parent released first, the clone keeps working."""
    h2 = zk.halo2
    lib, check = zk._capi.lib(), zk._capi.check
    big = h2.ParamsKZG.setup(13, 0x77)
    big.precompute(lagrange=False)
    small = big.clone_downsized(11)
    ref = h2.ParamsKZG.setup(11, 0x77)
    pa, pb = C.c_void_p(), C.c_void_p(); ca, cb = C.c_int(), C.c_int()
    check(lib.mi355_srs_pre_dev_ptr(big._g, C.byref(pa), C.byref(ca), None)); check(lib.mi355_srs_pre_dev_ptr(small._g, C.byref(pb), C.byref(cb), None))
    assert pa.value == pb.value and pa.value and ca.value == cb.value                  # ONE window table
    assert (small.read_g() == ref.read_g()).all() and (small.read_g(lagrange=True) == ref.read_g(lagrange=True)).all()
    rng = np.random.default_rng(5)
    sc = rand_fr(rng, 1 << 11)
    want = affine_of(ref.commit(sc))
    assert (affine_of(small.commit(sc)) == want).all() and (affine_of(small.commit_lagrange(sc)) == affine_of(ref.commit_lagrange(sc))).all()
    check(lib.mi355_srs_release(big._g)); check(lib.mi355_srs_release(big._gl))        # parent first
    assert (affine_of(small.commit(sc)) == want).all()
    n_out = C.c_uint64()
    assert lib.mi355_srs_len(big._g, C.byref(n_out)) == zk._capi.EBADARG
    small.release(); ref.release()
