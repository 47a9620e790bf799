"""
Runs the product's __host__ __device__ limb code (fp.cuh 8x32-bit Montgomery, g1.cuh XYZZ formulas) on the CPU
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
            assert (op(hs, w, 7, a, b) == cref.f_mul(w, a, b)).all()      # product-scanning multiplier (fp_asm.cuh, host form)
            assert (op(hs, w, 8, a) == cref.f_mul(w, a, a)).all()         # dedicated squaring
            assert (op(hs, w, 1, a, b) == cref.f_sub(w, a, b)).all()
            assert (op(hs, w, 5, np.array(pyref.to_limbs(vals[i]), dtype=np.uint64)) == a).all()
            assert (op(hs, w, 6, a) == np.array(pyref.to_limbs(vals[i]), dtype=np.uint64)).all()
        for i in range(0, 40):
            assert (op(hs, w, 3, mont[i]) == cref.f_inv(w, mont[i])).all()


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
