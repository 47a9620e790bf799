"""
The two CPU oracles (C restatement vs pure-Python big ints) must agree with each other and with the
algebraic invariants SURVEY.md §8c lists in lieu of MSM/NTT known-answer vectors.  CPU only.
"""
import random

import numpy as np
import pytest

from oracle import cref, pyref

R, P = pyref.R_MOD, pyref.P_MOD


def fr_vec(vals):
    return cref.f_from_canonical_vec(cref.FR, np.array([pyref.to_limbs(v % R) for v in vals], dtype=np.uint64).reshape(-1, 4))


def fr_ints(arr):
    return [cref.limbs_to_int(x) for x in cref.f_to_canonical_vec(cref.FR, arr)]


def pt_limbs(Pt):
    x, y = pyref.g1_affine_to_limbs(Pt)
    return np.array(x + y, dtype=np.uint64)


def jac_to_py(j):
    a = cref.g1_to_affine(j)
    return pyref.g1_affine_from_limbs(a[:4], a[4:])


def rand_points(rng, n):
    pts = [pyref.g1_mul(pyref.G1_GEN, rng.randrange(1, R)) for _ in range(n)]
    return pts, np.stack([pt_limbs(p) for p in pts])


def test_field_ops_match():
    rng = random.Random(1)
    for w, m in ((cref.FQ, P), (cref.FR, R)):
        for _ in range(200):
            a, b = rng.randrange(m), rng.randrange(m)
            am = cref.f_from_canonical_vec(w, cref.int_to_limbs(a)[None])[0]
            bm = cref.f_from_canonical_vec(w, cref.int_to_limbs(b)[None])[0]
            c = lambda x: cref.limbs_to_int(cref.f_to_canonical_vec(w, x[None])[0])
            assert c(cref.f_mul(w, am, bm)) == a * b % m
            assert c(cref.f_add(w, am, bm)) == (a + b) % m
            assert c(cref.f_sub(w, am, bm)) == (a - b) % m
        for a in (0, 1, m - 1, 2, m - 2, (1 << 253) % m):
            am = cref.f_from_canonical_vec(w, cref.int_to_limbs(a)[None])[0]
            c = lambda x: cref.limbs_to_int(cref.f_to_canonical_vec(w, x[None])[0])
            assert c(cref.f_mul(w, am, am)) == a * a % m
            if a:
                assert c(cref.f_inv(w, am)) == pow(a, -1, m)


def test_g1_ops_match():
    rng = random.Random(2)
    G = pyref.G1_GEN
    assert (cref.g1_generator() == pt_limbs(G)).all()
    for _ in range(10):
        k = rng.randrange(R)
        j = cref.g1_mul(pt_limbs(G), fr_vec([k])[0])
        assert jac_to_py(j) == pyref.g1_mul(G, k)
    A, B = pyref.g1_mul(G, 5), pyref.g1_mul(G, 7)
    ja = cref.g1_mul(pt_limbs(G), fr_vec([5])[0]); jb = cref.g1_mul(pt_limbs(G), fr_vec([7])[0])
    assert jac_to_py(cref.g1_add(ja, jb)) == pyref.g1_add(A, B) == pyref.g1_mul(G, 12)
    assert jac_to_py(cref.g1_add(ja, ja)) == pyref.g1_mul(G, 10)              # add(P,P) -> doubling branch
    assert jac_to_py(cref.g1_double(ja)) == pyref.g1_mul(G, 10)
    assert jac_to_py(cref.g1_add_affine(ja, pt_limbs(A))) == pyref.g1_mul(G, 10)  # madd doubling branch
    assert jac_to_py(cref.g1_add_affine(ja, pt_limbs(pyref.g1_neg(A)))) is None   # P + (-P)
    assert jac_to_py(cref.g1_add_affine(ja, np.zeros(8, dtype=np.uint64))) == A   # + identity
    assert jac_to_py(cref.g1_mul(pt_limbs(G), fr_vec([0])[0])) is None
    assert jac_to_py(cref.g1_mul(pt_limbs(G), fr_vec([R - 1])[0])) == pyref.g1_neg(G)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 100])
def test_msm_oracles_agree(n):
    rng = random.Random(100 + n)
    pts, bases = rand_points(rng, n)
    sc = [rng.randrange(R) for _ in range(n)]
    scal = fr_vec(sc)
    want = pyref.msm(sc, pts)
    assert jac_to_py(cref.msm_naive(scal, bases)) == want
    assert jac_to_py(cref.multiexp_serial(scal, bases)) == want
    for t in (1, 3, 8):
        assert jac_to_py(cref.best_multiexp(scal, bases, threads=t)) == want


def test_msm_edge_cases():
    rng = random.Random(7)
    n = 64
    pts, bases = rand_points(rng, n)
    # all-zero scalars, all-one scalars, unit vector, identity bases, repeated point, P and -P
    assert jac_to_py(cref.best_multiexp(fr_vec([0] * n), bases, 4)) is None
    allsum = None
    for p in pts:
        allsum = pyref.g1_add(allsum, p)
    assert jac_to_py(cref.best_multiexp(fr_vec([1] * n), bases, 4)) == allsum
    e = [0] * n; e[17] = 1
    assert jac_to_py(cref.best_multiexp(fr_vec(e), bases, 4)) == pts[17]
    b2 = bases.copy(); b2[::2] = 0
    sc = [rng.randrange(R) for _ in range(n)]
    want = pyref.msm(sc[1::2], pts[1::2])
    assert jac_to_py(cref.best_multiexp(fr_vec(sc), b2, 4)) == want
    rep = np.stack([bases[3]] * n)
    assert jac_to_py(cref.best_multiexp(fr_vec(sc), rep, 4)) == pyref.g1_mul(pts[3], sum(sc) % R)
    pm = np.stack([pt_limbs(pts[0]), pt_limbs(pyref.g1_neg(pts[0]))] * (n // 2))
    assert jac_to_py(cref.best_multiexp(fr_vec([5] * n), pm, 4)) is None
    # linearity (s+t).P = s.P + t.P
    s2 = [rng.randrange(R) for _ in range(n)]
    lhs = jac_to_py(cref.best_multiexp(fr_vec([(a + b) % R for a, b in zip(sc, s2)]), bases, 4))
    rhs = pyref.g1_add(jac_to_py(cref.best_multiexp(fr_vec(sc), bases, 4)), jac_to_py(cref.best_multiexp(fr_vec(s2), bases, 4)))
    assert lhs == rhs


@pytest.mark.parametrize("k", [0, 1, 2, 3, 6, 10])
def test_ntt_oracles_agree(k):
    rng = random.Random(200 + k)
    n = 1 << k
    a = [rng.randrange(R) for _ in range(n)]
    w = pyref.omega(k)
    want = pyref.ntt(a, w)
    if k <= 6:
        assert want == pyref.dft(a, w)
        assert fr_ints(cref.dft_naive(fr_vec(a), fr_vec([w])[0])) == want
    for t in (1, 8):
        assert fr_ints(cref.best_fft(fr_vec(a), fr_vec([w])[0], k, threads=t)) == want


def test_ntt_invariants():
    k, n = 12, 1 << 12
    rng = random.Random(3)
    w = pyref.omega(k); wm = fr_vec([w])[0]; wim = fr_vec([pow(w, -1, R)])[0]; ninv = fr_vec([pow(n, -1, R)])[0]
    a = [rng.randrange(R) for _ in range(n)]
    am = fr_vec(a)
    f = cref.best_fft(am, wm, k, threads=8)          # threaded path (layers >= 4096 butterflies)
    assert (cref.best_fft(am, wm, k, threads=1) == f).all()
    assert (cref.ifft(f, wim, k, ninv) == am).all()   # round trip, raw Montgomery bytes
    delta = [1] + [0] * (n - 1)
    assert fr_ints(cref.best_fft(fr_vec(delta), wm, k)) == [1] * n
    assert fr_ints(cref.best_fft(fr_vec([1] * n), wm, k)) == [n] + [0] * (n - 1)
    # evaluation semantics: a'[i] = a(omega^i)
    for i in (0, 1, 5, n - 1):
        assert cref.limbs_to_int(cref.f_to_canonical_vec(cref.FR, f[i:i + 1])[0]) == pyref.eval_poly(a, pow(w, i, R))
        assert (cref.eval_polynomial(am, fr_vec([pow(w, i, R)])[0]) == f[i]).all()


def test_coset_extension_matches_horner():
    k, ext_k = 5, 7
    rng = random.Random(4)
    coeffs = [rng.randrange(R) for _ in range(1 << k)]
    z = pyref.FR_ZETA; zi = z * z % R
    ew = pyref.omega(ext_k)
    ext = cref.coeff_to_extended(fr_vec(coeffs), k, ext_k, fr_vec([z])[0], fr_vec([zi])[0], fr_vec([ew])[0])
    got = fr_ints(ext)
    assert got == pyref.coeff_to_extended(coeffs, k, ext_k)
    # [EXT-recalled] caveat: distribute_powers_zeta multiplies by zeta^(i mod 3), i.e. evaluates at zeta*omega^i
    # only up to the identity zeta^3 = 1: p(zeta * x) = sum c_i zeta^(i mod 3) x^i.  Check that reading.
    for i in (0, 1, 9, (1 << ext_k) - 1):
        assert got[i] == pyref.eval_poly(coeffs, z * pow(ew, i, R) % R)
    back = cref.extended_to_coeff(ext, ext_k, fr_vec([z])[0], fr_vec([zi])[0], fr_vec([pow(ew, -1, R)])[0], fr_vec([pow(1 << ext_k, -1, R)])[0])
    assert fr_ints(back) == coeffs + [0] * ((1 << ext_k) - (1 << k))


def test_commit_equals_commit_lagrange_on_synthetic_srs():
    """upstream halo2 `test_commit_lagrange` property [EXT-recalled poly/kzg/commitment.rs]: commit(coeff(a)) == commit_lagrange(a)."""
    k, n = 4, 16
    rng = random.Random(5)
    tau = rng.randrange(2, R)
    w = pyref.omega(k)
    g, gl, gs, gls = cref.srs_setup(k, fr_vec([tau])[0], fr_vec([w])[0])
    assert fr_ints(gs) == [pow(tau, i, R) for i in range(n)]
    assert fr_ints(gls) == pyref.lagrange_scalars(k, tau)
    evals = [rng.randrange(R) for _ in range(n)]           # polynomial in Lagrange basis
    coeffs = pyref.intt(evals, w)
    c1 = jac_to_py(cref.best_multiexp(fr_vec(coeffs), g, 2))
    c2 = jac_to_py(cref.best_multiexp(fr_vec(evals), gl, 2))
    want = pyref.g1_mul(pyref.G1_GEN, pyref.eval_poly(coeffs, tau))   # commit(p) = p(tau).G
    assert c1 == c2 == want


def affine_to_jac(a):
    """[n,8] affine -> [n,12] Jacobian with z = R (identity stays all-zero)."""
    a = np.asarray(a, dtype=np.uint64)
    j = np.zeros((a.shape[0], 12), dtype=np.uint64)
    j[:, :8] = a
    one_q = np.array(pyref.to_limbs(pyref.MONT_R % P), dtype=np.uint64)
    nz = (a != 0).any(axis=1)
    j[nz, 8:] = one_q
    return j


def test_g1_fft_and_g_to_lagrange_oracle():
    """best_fft over G1 points against its definition (row i = MSM of the inputs with scalars omega^(ij)), and g_to_lagrange against the
    closed form L_i(tau) G of the synthetic SRS -- the property ParamsKZG::downsize relies on."""
    k, n = 3, 8
    rng = random.Random(11)
    w = pyref.omega(k)
    tau = rng.randrange(2, R)
    g, gl, _, _ = cref.srs_setup(k, fr_vec([tau])[0], fr_vec([w])[0])
    pts = g.copy(); pts[5] = 0                                     # an identity among the inputs
    got = cref.g1_to_affine(cref.best_fft_g1(affine_to_jac(pts), fr_vec([w])[0], k))
    for i in range(n):
        want = cref.g1_to_affine(cref.msm_naive(fr_vec([pow(w, i * j, R) for j in range(n)]), pts))
        assert (got[i] == want).all(), i
    w_inv, n_inv = pow(w, -1, R), pow(n, -1, R)
    assert (cref.g_to_lagrange(g, k, fr_vec([w_inv])[0], fr_vec([n_inv])[0]) == gl).all()
    # downsize: the first 2^(k-1) powers of tau give the Lagrange basis of the half-size domain
    g2, gl2, _, _ = cref.srs_setup(k - 1, fr_vec([tau])[0], fr_vec([pyref.omega(k - 1)])[0])
    assert (g2 == g[: n // 2]).all()
    w2_inv = pow(pyref.omega(k - 1), -1, R)
    assert (cref.g_to_lagrange(g[: n // 2], k - 1, fr_vec([w2_inv])[0], fr_vec([pow(n // 2, -1, R)])[0]) == gl2).all()


def test_batch_invert_and_grand_product_oracle():
    """ff::BatchInvert and the grand-product column against big-int arithmetic."""
    rng = random.Random(21)
    vals = [rng.randrange(R) for _ in range(300)]
    vals[0] = 0; vals[17] = 0; vals[299] = 0; vals[5] = 1; vals[6] = R - 1
    inv = fr_ints(cref.batch_invert(fr_vec(vals)))
    assert inv == [pow(v, -1, R) if v else 0 for v in vals]
    z, total = cref.prefix_product(fr_vec(vals[1:17]))
    acc, want = 1, []
    for v in vals[1:17]:
        want.append(acc); acc = acc * v % R
    assert fr_ints(z) == want and fr_ints(total.reshape(1, 4)) == [acc]
    z0, t0 = cref.prefix_product(fr_vec([]).reshape(0, 4))
    assert z0.shape[0] == 0 and fr_ints(t0.reshape(1, 4)) == [1]


def test_kate_division_oracle():
    """kate_division against big-int polynomial arithmetic: p(X) = (X - z) q(X) + p(z)."""
    rng = random.Random(33)
    for n in (1, 2, 5, 40):
        coeffs = [rng.randrange(R) for _ in range(n)]
        z = rng.randrange(R)
        q = fr_ints(cref.kate_division(fr_vec(coeffs).reshape(n, 4), fr_vec([z])[0]))
        assert len(q) == n - 1
        back = [0] * n                                    # (X - z) q(X) + p(z)
        for i, c in enumerate(q):
            back[i + 1] = (back[i + 1] + c) % R; back[i] = (back[i] - z * c) % R
        back[0] = (back[0] + pyref.eval_poly(coeffs, z)) % R
        assert back == coeffs


def test_gate_eval_restatement_matches_big_integer_definition():
    """orc_gate_eval (the checker of mi355_fr_gate_eval_dev) against the definition in Python big integers: random term lists with
    positive / negative rotations, a constant term, repeated factors, with and without accumulation."""
    rng = np.random.default_rng(77)
    n = 64
    R_ = pyref.R_MOD
    vals = [[(int(x) * 0x9E3779B97F4A7C15 * int(x) + 12345) % R_ for x in rng.integers(0, 2**63, size=n)] for _ in range(5)]
    polys = [np.stack([cref.fr_mont(v) for v in col]) for col in vals]
    for trial in range(6):
        nt = int(rng.integers(1, 8))
        terms = []
        for j in range(nt):
            ln = 0 if (trial == 0 and j == 0) else int(rng.integers(1, 5))
            c = int(rng.integers(1, 2**62)) * 3 + 1
            terms.append((c, [(int(rng.integers(0, 5)), int(rng.integers(-70, 70))) for _ in range(ln)]))
        coeffs = np.stack([cref.fr_mont(c) for c, _ in terms])
        tl = [len(f) for _, f in terms]; fp = [p for _, f in terms for p, _ in f]; fr_ = [r for _, f in terms for _, r in f]
        base = [int(x) % R_ for x in rng.integers(0, 2**63, size=n)]
        dst0 = np.stack([cref.fr_mont(v) for v in base])
        for acc in (False, True):
            got = cref.gate_eval(polys, coeffs, tl, fp, fr_, n, dst=dst0 if acc else None)
            for i in (0, 1, n // 2, n - 1):
                want = base[i] if acc else 0
                for c, f in terms:
                    t = c
                    for p, r in f:
                        t = t * vals[p][(i + r) % n] % R_
                    want = (want + t) % R_
                assert cref.limbs_to_int(cref.f_to_canonical_vec(cref.FR, got[i:i + 1])[0]) == want


def test_gate_eval_rows_over_threads_equals_the_serial_loop():
    """orc_gate_eval_mt (the at-size checker of the -m gpu tests: rows dealt over threads) == orc_gate_eval == Python big integers"""
    rng = np.random.default_rng(404)
    n = 1 << 13
    polys = [np.stack([cref.fr_mont(int(x)) for x in rng.integers(0, 1 << 62, size=n)]) for _ in range(3)]
    coeffs = np.stack([cref.fr_mont(3), cref.fr_mont(pyref.R_MOD - 1), cref.fr_mont(1), cref.fr_mont(12345)])
    tl, fp, fr_ = [2, 1, 3, 0], [0, 1, 2, 0, 1, 2], [1, -1, n - 1, -3 * n + 5, 0, 1 << 12]
    a = cref.gate_eval(polys, coeffs, tl, fp, fr_, n)
    b = cref.gate_eval(polys, coeffs, tl, fp, fr_, n, threads=5)
    assert (a == b).all()
    c = cref.gate_eval(polys, coeffs, tl, fp, fr_, n, dst=a, threads=3)
    d = cref.gate_eval(polys, coeffs, tl, fp, fr_, n, dst=a)
    assert (c == d).all()
    Rinv = pow(pyref.MONT_R, -1, pyref.R_MOD)
    val = lambda arr, i: pyref.from_limbs([int(x) for x in arr[i % n]]) * Rinv % pyref.R_MOD
    for i in (0, 1, n - 1, 4097):
        want = (3 * val(polys[0], i + 1) * val(polys[1], i - 1) - val(polys[2], i + n - 1) + val(polys[0], i - 3 * n + 5) * val(polys[1], i) * val(polys[2], i + (1 << 12)) + 12345) % pyref.R_MOD
        assert val(b, i) == want
