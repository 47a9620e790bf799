"""
Runs the product's __host__ __device__ limb code (fp.hpp 8x32-bit Montgomery, g1.hpp XYZZ formulas) on the CPU
via tests/hostcheck/host_selftest.cpp and checks it bit-exactly against the oracle.  This is a check OF the
device arithmetic, compiled for x86 -- it is not a product path.  CPU only.
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import cref, pyref

HERE = os.path.dirname(os.path.abspath(__file__))
R, P = pyref.R_MOD, pyref.P_MOD
ONE_Q = np.array(pyref.to_limbs(pyref.MONT_R % P), dtype=np.uint64)


@pytest.fixture(scope="module")
def hs():
    src = os.path.join(HERE, "hostcheck", "host_selftest.cpp")
    so = os.path.join(HERE, "hostcheck", "libhostselftest.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return C.CDLL(so)


def p_(a):
    return a.ctypes.data_as(C.c_void_p)


def op(hs, which, o, a, b=None):
    out = np.zeros(4, dtype=np.uint64)
    a = np.ascontiguousarray(a); b = a if b is None else np.ascontiguousarray(b)
    hs.hs_f_op(which, o, p_(out), p_(a), p_(b))
    return out


def test_field_ops_bit_exact(hs):
    rng = random.Random(11)
    for w, m in ((cref.FQ, P), (cref.FR, R)):
        edge = [0, 1, 2, m - 1, m - 2, (1 << 253) % m, (1 << 32) - 1, 1 << 32, (1 << 224)]
        vals = edge + [rng.randrange(m) for _ in range(300)]
        mont = cref.f_from_canonical_vec(w, np.array([pyref.to_limbs(v) for v in vals], dtype=np.uint64))
        for i in range(len(vals)):
            a, b = mont[i], mont[(i * 7 + 3) % len(vals)]
            assert (op(hs, w, 2, a, b) == cref.f_mul(w, a, b)).all()
            assert (op(hs, w, 0, a, b) == cref.f_add(w, a, b)).all()
            assert (op(hs, w, 7, a, b) == cref.f_mul(w, a, b)).all()      # product-scanning multiplier (fp_asm.hpp, host form)
            assert (op(hs, w, 8, a) == cref.f_mul(w, a, a)).all()         # dedicated squaring
            assert (op(hs, w, 1, a, b) == cref.f_sub(w, a, b)).all()
            assert (op(hs, w, 5, np.array(pyref.to_limbs(vals[i]), dtype=np.uint64)) == a).all()
            assert (op(hs, w, 6, a) == np.array(pyref.to_limbs(vals[i]), dtype=np.uint64)).all()
            assert (op(hs, w, 11, a) == np.array(pyref.to_limbs(vals[i]), dtype=np.uint64)).all()      # reduction-only conversion (k_msm_digits)
            assert pyref.from_limbs(op(hs, w, 11, np.array(pyref.to_limbs(vals[i]), dtype=np.uint64))) == vals[i] * pow(1 << 256, -1, m) % m
        for i in range(0, 40):
            assert (op(hs, w, 3, mont[i]) == cref.f_inv(w, mont[i])).all()
        for i in range(len(vals)):                                        # binary-Euclid inverse used by the one-lane normalisations
            assert (op(hs, w, 9, mont[i]) == cref.f_inv(w, mont[i])).all(), vals[i]
        # division-step ("safegcd") inverse: constant instruction stream, every edge value plus a long random run; 0 -> 0
        more = [rng.randrange(m) for _ in range(3000)] + [(1 << k) % m for k in range(0, 256, 5)] + [m - (1 << k) for k in range(0, 250, 7)]
        mont2 = cref.f_from_canonical_vec(w, np.array([pyref.to_limbs(v) for v in more], dtype=np.uint64))
        for i in range(len(vals)):
            assert (op(hs, w, 10, mont[i]) == cref.f_inv(w, mont[i])).all(), vals[i]
        for i in range(len(more)):
            got = pyref.from_limbs(op(hs, w, 10, mont2[i]))
            assert got == pow(more[i], -1, m) * pyref.MONT_R % m, more[i]


def _pt(Pt):
    x, y = pyref.g1_affine_to_limbs(Pt)
    return np.array(x + y, dtype=np.uint64)


def test_xyzz_formulas_against_oracle(hs):
    rng = random.Random(12)
    n = 24
    pts = [pyref.g1_mul(pyref.G1_GEN, rng.randrange(1, R)) for _ in range(n)]
    bases = np.stack([_pt(p) for p in pts])
    bases[5] = 0  # identity base
    sc = [rng.randrange(R) for _ in range(n)]
    sc[0] = 0; sc[1] = 1; sc[2] = R - 1; sc[3] = 2
    canon = np.array([pyref.to_limbs(s) for s in sc], dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    hs.hs_msm_naive(p_(out), p_(canon), p_(bases), C.c_uint64(n))
    want = cref.g1_to_affine(cref.msm_naive(cref.f_from_canonical_vec(cref.FR, canon), bases))
    assert (out[:8] == want).all() and (out[8:] == ONE_Q).all()


def test_xyzz_special_cases(hs):
    G = pyref.G1_GEN
    A = pyref.g1_mul(G, 12345)

    def xyzz_of(Pt):
        acc = np.zeros(16, dtype=np.uint64); hs.hs_xyzz_madd(p_(acc), p_(_pt(Pt))); return acc

    def aff(acc):
        out = np.zeros(12, dtype=np.uint64); hs.hs_xyzz_to_jac(p_(out), p_(acc))
        return pyref.g1_jacobian_from_limbs(out[:4], out[4:8], out[8:])

    a = xyzz_of(A); assert aff(a) == A
    hs.hs_xyzz_madd(p_(a), p_(_pt(A))); assert aff(a) == pyref.g1_mul(A, 2)           # madd doubling branch
    hs.hs_xyzz_madd(p_(a), p_(_pt(pyref.g1_neg(pyref.g1_mul(A, 2))))); assert aff(a) is None  # madd inverse branch
    hs.hs_xyzz_madd(p_(a), p_(np.zeros(8, dtype=np.uint64))); assert aff(a) is None    # identity + identity
    b = xyzz_of(A); c = xyzz_of(A)
    hs.hs_xyzz_madd(p_(b), p_(_pt(G)))   # b = A+G with a non-trivial ZZ
    hs.hs_xyzz_madd(p_(c), p_(_pt(G)))
    hs.hs_xyzz_add(p_(b), p_(c)); assert aff(b) == pyref.g1_mul(pyref.g1_add(A, G), 2)   # add doubling branch
    d = xyzz_of(pyref.g1_neg(pyref.g1_mul(pyref.g1_add(A, G), 2)))
    hs.hs_xyzz_add(p_(b), p_(d)); assert aff(b) is None                                # add inverse branch
    hs.hs_xyzz_add(p_(b), p_(c)); assert aff(b) == pyref.g1_add(A, G)                   # identity + q
    out = np.zeros(16, dtype=np.uint64)
    for k in (0, 1, 2, 3, 1000, 65535, 2**31 + 5):
        hs.hs_xyzz_mul_small(p_(out), p_(c), C.c_uint32(k)); assert aff(out) == pyref.g1_mul(pyref.g1_add(A, G), k)
    # Jacobian (non-normalised) -> XYZZ
    j = cref.g1_mul(_pt(G), cref.fr_mont(777)); x = np.zeros(16, dtype=np.uint64)
    hs.hs_jac_to_xyzz(p_(x), p_(j)); assert aff(x) == pyref.g1_mul(G, 777)


def test_unsaturated_29bit_arithmetic(hs):
    """fp29.hpp (9 x 29-bit limbs, R' = 2^261): conversions from/to the ABI form, lazy add/sub chains, mul/sqr, zero test."""
    rng = random.Random(29)
    for w, m in ((cref.FQ, P), (cref.FR, R)):
        edge = [0, 1, 2, m - 1, m - 2, (1 << 253) % m, (1 << 29) - 1, 1 << 29, (1 << 232), (1 << 232) - 1]
        vals = edge + [rng.randrange(m) for _ in range(200)]
        mont = cref.f_from_canonical_vec(w, np.array([pyref.to_limbs(v) for v in vals], dtype=np.uint64))

        def o29(op, a, b):
            out = np.zeros(4, dtype=np.uint64); hs.hs_f29_op(w, op, p_(out), p_(np.ascontiguousarray(a)), p_(np.ascontiguousarray(b))); return out

        for i in range(len(vals)):
            a, b = mont[i], mont[(i * 5 + 1) % len(vals)]
            va, vb = vals[i], vals[(i * 5 + 1) % len(vals)]
            c = lambda x: cref.limbs_to_int(cref.f_to_canonical_vec(w, x[None])[0])
            assert (o29(0, a, b) == cref.f_mul(w, a, b)).all()
            assert (o29(1, a, b) == cref.f_mul(w, a, a)).all()
            assert c(o29(2, a, b)) == (va + vb) * vb % m
            assert c(o29(3, a, b)) == (va - vb) % m
            assert c(o29(4, a, b)) == (va - 2 * vb) % m
            assert c(o29(5, a, b)) == (3 * va - vb) % m
            assert c(o29(6, a, b)) == (2 * va - vb) ** 2 % m
            assert o29(7, a, b).tolist() == a.tolist()          # from_sat / to_sat round trip
            assert c(o29(8, a, b)) == ((va - vb) * (vb - 2 * va) - (vb - va) * va) % m    # fused difference of products (signed lazy reduction)
            assert bool(hs.hs_f29_is_zero(w, p_(np.ascontiguousarray(a)), p_(np.ascontiguousarray(a)))) is True
            assert bool(hs.hs_f29_is_zero(w, p_(np.ascontiguousarray(a)), p_(np.ascontiguousarray(b)))) is (va == vb)


def test_xyzz29_bucket_accumulator(hs):
    """g1_29.hpp: long chains of mixed additions with lazy values, negation, identity bases, P+P and P+(-P) inside a bucket."""
    rng = random.Random(31)
    G = pyref.G1_GEN
    pool = [pyref.g1_mul(G, rng.randrange(1, R)) for _ in range(12)]

    def run(seq):  # seq of (point or None, sign)
        pts = np.stack([_pt(p) if p is not None else np.zeros(8, dtype=np.uint64) for p, _ in seq])
        signs = (C.c_uint8 * len(seq))(*[s for _, s in seq])
        out = np.zeros(16, dtype=np.uint64)
        hs.hs_bucket_sum29(p_(out), p_(pts), signs, C.c_uint64(len(seq)))
        jac = np.zeros(12, dtype=np.uint64); hs.hs_xyzz_to_jac(p_(jac), p_(out))
        got = pyref.g1_jacobian_from_limbs(jac[:4], jac[4:8], jac[8:])
        want = None
        for p, s in seq:
            want = pyref.g1_add(want, pyref.g1_neg(p) if s else p)
        assert got == want, seq

    for trial in range(30):
        n = rng.randrange(1, 40)
        run([(rng.choice(pool), rng.randrange(2)) for _ in range(n)])
    A, B = pool[0], pool[1]
    run([(A, 0), (A, 0)])                          # doubling branch
    run([(A, 1), (A, 1), (B, 0)])                  # doubling of a negated point, then a normal add
    run([(A, 0), (A, 1)])                          # annihilation
    run([(A, 0), (A, 1), (B, 1), (None, 0), (B, 1)])   # identity -> re-init -> identity base -> doubling
    run([(A, 0), (B, 0), (pyref.g1_add(A, B), 1)])  # sum hits the negative of the accumulator
    run([(A, 0), (B, 0), (pyref.g1_add(A, B), 0)])  # sum equals the accumulator: doubling with non-trivial ZZ
    run([(None, 0), (None, 1)])
    run([(A, 0)] * 64)                              # 64 P by repeated additions (one doubling, then ordinary adds)


def test_reduce_small_boundaries(hs):
    """quotient estimate of Fp29::reduce_small: exact multiples of p, one below/above, and random values -- over the whole range the
    9-limb form can hold (v < 2^261 = 168 p): the NTT's unit-twiddle butterflies reduce u - v + 64 r < 103 r this way."""
    rng = random.Random(64)
    M = (1 << 29) - 1
    for w, m in ((0, P), (1, R)):
        top = (1 << 261) // m
        vals = [k * m + d for k in range(top + 1) for d in (0, 1, m - 1, m // 2) if k * m + d < (1 << 261)] + [rng.randrange(1 << 261) for _ in range(6000)] + [(1 << 261) - 1]
        for v in vals:
            inp = (C.c_uint32 * 9)(*[(v >> (29 * i)) & M if i < 8 else v >> 232 for i in range(9)])
            out = (C.c_uint32 * 9)()
            hs.hs_f29_reduce_small(w, out, inp)
            got = sum(int(out[i]) << (29 * i) for i in range(9))
            assert got % m == v % m and got < 2 * m and all(out[i] <= M for i in range(8)), (w, v // m)


def test_xyzz29_full_addition_and_doubling(hs):
    """g1_xyzz29_add / g1_xyzz29_dbl: accumulators with lazy coordinates (the output of chains of mixed additions) combined by full
    additions, including equal and opposite partial sums and empty groups, and a double-and-add ladder over a lazy base point."""
    rng = random.Random(77)
    G = pyref.G1_GEN
    pool = [pyref.g1_mul(G, rng.randrange(1, R)) for _ in range(10)]

    def arrays(seq):
        pts = np.stack([_pt(p) if p is not None else np.zeros(8, dtype=np.uint64) for p, _ in seq])
        return pts, (C.c_uint8 * len(seq))(*[s for _, s in seq])

    def total(seq):
        want = None
        for p, s in seq:
            want = pyref.g1_add(want, pyref.g1_neg(p) if s else p)
        return want

    def as_point(out):
        jac = np.zeros(12, dtype=np.uint64); hs.hs_xyzz_to_jac(p_(jac), p_(out))
        return pyref.g1_jacobian_from_limbs(jac[:4], jac[4:8], jac[8:])

    def grouped(seq, group):
        pts, signs = arrays(seq)
        out = np.zeros(16, dtype=np.uint64)
        hs.hs_xyzz29_grouped_sum(p_(out), p_(pts), signs, C.c_uint64(len(seq)), C.c_uint64(group))
        assert as_point(out) == total(seq), (seq, group)

    for trial in range(40):
        n = rng.randrange(1, 60)
        grouped([(rng.choice(pool), rng.randrange(2)) for _ in range(n)], rng.randrange(1, 9))
    A, B, Cp = pool[0], pool[1], pool[2]
    grouped([(A, 0), (B, 0), (A, 0), (B, 0)], 2)                 # equal partial sums: the doubling branch of the full addition
    grouped([(A, 0), (B, 0), (A, 1), (B, 1)], 2)                 # opposite partial sums: annihilation
    grouped([(A, 0), (A, 1), (B, 0), (Cp, 0)], 2)                # an empty (identity) group first
    grouped([(B, 0), (Cp, 0), (A, 0), (A, 1)], 2)                # an identity group second
    grouped([(A, 0)] * 33, 3)                                    # 3A + 3A + ... : doubling then general additions of multiples
    grouped([(None, 0), (None, 0), (A, 0)], 2)
    for trial in range(12):
        seq = [(rng.choice(pool), rng.randrange(2)) for _ in range(rng.randrange(1, 20))]
        k = rng.choice([0, 1, 2, 3, 5, 255, 65536, 2**31 + 12345, 2**32 - 1, rng.randrange(2**32)])
        pts, signs = arrays(seq)
        out = np.zeros(16, dtype=np.uint64)
        hs.hs_xyzz29_ladder(p_(out), p_(pts), signs, C.c_uint64(len(seq)), C.c_uint32(k))
        base = total(seq)
        assert as_point(out) == (pyref.g1_mul(base, k) if (base is not None and k) else None), (seq, k)


def test_glv_decomposition_and_joint_scalar_multiple(hs):
    """glv.hpp: k = k1 + lambda k2 (mod r) with |k_i| < 2^127 for random and extreme scalars, and k * P by the joint double-and-add
    (phi(P) = (beta x, y), signed 2-bit digits) against the big-integer oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_glv", os.path.join(os.path.dirname(HERE), "tools", "gen_glv_constants.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    c = gen.derive()
    lam = c["lam"]
    assert lam == pyref.FR_ZETA and (lam * lam + lam + 1) % R == 0 and pow(c["beta"], 3, P) == 1 and c["beta"] != 1
    rng = random.Random(2024)
    ks = [0, 1, 2, 3, R - 1, R - 2, R // 2, R // 2 + 1, lam, R - lam, (1 << 253), (1 << 127), (1 << 127) - 1, (1 << 128) + 5] + [rng.randrange(R) for _ in range(4000)]
    for k in ks:
        kc = np.array(pyref.to_limbs(k), dtype=np.uint64)
        k1 = (C.c_uint32 * 4)(); k2 = (C.c_uint32 * 4)(); n1 = C.c_int(); n2 = C.c_int()
        hs.hs_glv_decompose(p_(kc), k1, C.byref(n1), k2, C.byref(n2))
        a = sum(int(v) << (32 * i) for i, v in enumerate(k1)) * (-1 if n1.value else 1)
        b = sum(int(v) << (32 * i) for i, v in enumerate(k2)) * (-1 if n2.value else 1)
        assert (a, b) == gen.decompose(k, c), k                                     # the same integers as the generator's formulas
        assert (a + lam * b - k) % R == 0 and abs(a) < (1 << 127) and abs(b) < (1 << 127), k
    G = pyref.G1_GEN
    pts = [G, pyref.g1_mul(G, 5), pyref.g1_mul(G, rng.randrange(1, R)), pyref.g1_mul(G, rng.randrange(1, R))]
    for k in ks[:14] + ks[14:14 + 40]:
        for Pt in pts[: 2 if k in ks[:14] else 4]:
            out = np.zeros(16, dtype=np.uint64)
            hs.hs_g1_mul_glv(p_(out), p_(_pt(Pt)), p_(np.array(pyref.to_limbs(k), dtype=np.uint64)))
            jac = np.zeros(12, dtype=np.uint64); hs.hs_xyzz_to_jac(p_(jac), p_(out))
            got = pyref.g1_jacobian_from_limbs(jac[:4], jac[4:8], jac[8:])
            assert got == (pyref.g1_mul(Pt, k) if k % R else None), (k, Pt)
