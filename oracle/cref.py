"""
cref.py -- ctypes binding of oracle/liboracle_bn254.so (the C restatement, see bn254_oracle.c).

TEST INFRASTRUCTURE ONLY.  Arrays are numpy uint64 of shape [n, 4] (field elements, Montgomery
LE limbs), [n, 8] (G1Affine x|y) and [n, 12] / [12] (Jacobian x|y|z), i.e. exactly the byte
layout the C-ABI of the product uses (include/mi355zk.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_bn254.so")
FQ, FR = 0, 1


def usable_cpus() -> int:
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (a container on a 256-thread host is
    typically limited to a few CPUs' worth of time: 256 threads then only contend with each other)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                quota, period = parse(txt)
                if quota != "max":
                    n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
            else:
                quota = int(txt); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bn254_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _fe(x=None):
    a = np.zeros(4, dtype=np.uint64)
    if x is not None:
        a[:] = x
    return a


def f_mul(w, a, b):
    o = _fe(); lib().orc_f_mul(w, _p(o), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))); return o


def f_add(w, a, b):
    o = _fe(); lib().orc_f_add(w, _p(o), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))); return o


def f_sub(w, a, b):
    o = _fe(); lib().orc_f_sub(w, _p(o), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))); return o


def f_inv(w, a):
    o = _fe(); lib().orc_f_inv(w, _p(o), _p(np.ascontiguousarray(a))); return o


def f_pow(w, a, e: int):
    o = _fe(); ee = np.array([(e >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    lib().orc_f_pow(w, _p(o), _p(np.ascontiguousarray(a)), _p(ee)); return o


def f_from_canonical_vec(w, a):
    a = np.ascontiguousarray(a, dtype=np.uint64); o = np.empty_like(a)
    lib().orc_f_from_canonical_vec(w, _p(o), _p(a), C.c_uint64(a.shape[0])); return o


def f_to_canonical_vec(w, a):
    a = np.ascontiguousarray(a, dtype=np.uint64); o = np.empty_like(a)
    lib().orc_f_to_canonical_vec(w, _p(o), _p(a), C.c_uint64(a.shape[0])); return o


def f_mul_vec(w, a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b); o = np.empty_like(a)
    lib().orc_f_mul_vec(w, _p(o), _p(a), _p(b), C.c_uint64(a.shape[0])); return o


def int_to_limbs(x: int):
    return np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def limbs_to_int(l) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def fr_mont(x: int):
    """canonical int -> Montgomery limbs (through the C oracle)."""
    return f_from_canonical_vec(FR, int_to_limbs(x % 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001)[None, :])[0]


def g1_generator():
    o = np.zeros(8, dtype=np.uint64); lib().orc_g1_generator(_p(o)); return o


def g1_is_on_curve(p) -> bool:
    return bool(lib().orc_g1_is_on_curve(_p(np.ascontiguousarray(p))))


def g1_mul(p, s_mont):
    o = np.zeros(12, dtype=np.uint64); lib().orc_g1_mul(_p(o), _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(s_mont))); return o


def g1_mul_generator_vec(scalars, threads: int | None = None):
    """[n,8] affine points scalars[i] * G (threaded test-input generator)."""
    threads = threads or usable_cpus()
    s = np.ascontiguousarray(scalars); o = np.zeros((s.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_mul_generator_vec(_p(o), _p(s), C.c_uint64(s.shape[0]), C.c_int(threads)); return o


def g1_add(p, q):
    o = np.zeros(12, dtype=np.uint64); lib().orc_g1_add(_p(o), _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(q))); return o


def g1_add_affine(p, q):
    o = np.zeros(12, dtype=np.uint64); lib().orc_g1_add_affine(_p(o), _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(q))); return o


def g1_double(p):
    o = np.zeros(12, dtype=np.uint64); lib().orc_g1_double(_p(o), _p(np.ascontiguousarray(p))); return o


def g1_to_affine(p):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    if p.ndim == 1:
        o = np.zeros(8, dtype=np.uint64); lib().orc_g1_to_affine(_p(o), _p(p)); return o
    o = np.zeros((p.shape[0], 8), dtype=np.uint64); lib().orc_g1_to_affine_vec(_p(o), _p(p), C.c_uint64(p.shape[0])); return o


def g1_compress(p) -> bytes:
    o = (C.c_uint8 * 32)(); lib().orc_g1_compress(o, _p(np.ascontiguousarray(p))); return bytes(o)


def g1_decompress(b: bytes):
    o = np.zeros(8, dtype=np.uint64); buf = (C.c_uint8 * 32).from_buffer_copy(b)
    ok = lib().orc_g1_decompress(_p(o), buf); return o if ok else None


def g2_generator():
    o = np.zeros(16, dtype=np.uint64); lib().orc_g2_generator(_p(o)); return o


def g2_is_on_curve(p) -> bool:
    return bool(lib().orc_g2_is_on_curve(_p(np.ascontiguousarray(p, dtype=np.uint64))))


def g2_in_subgroup(p) -> bool:
    return bool(lib().orc_g2_in_subgroup(_p(np.ascontiguousarray(p, dtype=np.uint64))))


def g2_mul(p, s_mont):
    """s * p on the twist: G2Affine (16 limbs: x.c0, x.c1, y.c0, y.c1) in, G2Affine out."""
    o = np.zeros(16, dtype=np.uint64)
    lib().orc_g2_mul(_p(o), _p(np.ascontiguousarray(p, dtype=np.uint64)), _p(np.ascontiguousarray(s_mont))); return o


def g2_from_words(words):
    """the four 32-byte big-endian words of an EVM pairing input (x.c1, x.c0, y.c1, y.c0) -> G2Affine Montgomery limbs"""
    w = [int(x, 16) if isinstance(x, str) else int(x) for x in words]
    q = [w[1], w[0], w[3], w[2]]
    return np.concatenate([f_from_canonical_vec(FQ, np.array([int_to_limbs(v)], dtype=np.uint64))[0] for v in q])


def msm_naive(scalars, bases):
    o = np.zeros(12, dtype=np.uint64); s = np.ascontiguousarray(scalars); b = np.ascontiguousarray(bases)
    lib().orc_msm_naive(_p(o), _p(s), _p(b), C.c_uint64(s.shape[0])); return o


def multiexp_serial(scalars, bases):
    o = np.zeros(12, dtype=np.uint64); s = np.ascontiguousarray(scalars); b = np.ascontiguousarray(bases)
    lib().orc_multiexp_serial(_p(o), _p(s), _p(b), C.c_uint64(s.shape[0])); return o


def best_multiexp(scalars, bases, threads: int | None = None):
    threads = threads or usable_cpus()
    o = np.zeros(12, dtype=np.uint64); s = np.ascontiguousarray(scalars); b = np.ascontiguousarray(bases)
    assert s.shape[0] == b.shape[0]
    lib().orc_best_multiexp(_p(o), _p(s), _p(b), C.c_uint64(s.shape[0]), C.c_int(threads)); return o


def dft_naive(a, omega):
    a = np.ascontiguousarray(a); o = np.empty_like(a)
    lib().orc_dft_naive(_p(o), _p(a), C.c_uint64(a.shape[0]), _p(np.ascontiguousarray(omega))); return o


def best_fft(a, omega, log_n: int, threads: int | None = None):
    """in place on a copy; returns the transformed array."""
    threads = threads or usable_cpus()
    a = np.array(a, dtype=np.uint64, copy=True, order="C"); assert a.shape[0] == 1 << log_n
    lib().orc_best_fft(_p(a), _p(np.ascontiguousarray(omega)), C.c_uint32(log_n), C.c_int(threads)); return a


def best_fft_g1(points_jac, omega, log_n: int):
    """best_fft over G1 (Jacobian [n,12]), on a copy."""
    a = np.array(points_jac, dtype=np.uint64, copy=True, order="C"); assert a.shape == (1 << log_n, 12)
    lib().orc_best_fft_g1(_p(a), _p(np.ascontiguousarray(omega)), C.c_uint32(log_n)); return a


def g_to_lagrange(g_affine, k: int, omega_inv, n_inv):
    g = np.ascontiguousarray(g_affine, dtype=np.uint64); assert g.shape == (1 << k, 8)
    out = np.zeros_like(g)
    lib().orc_g_to_lagrange(_p(out), _p(g), C.c_uint32(k), _p(np.ascontiguousarray(omega_inv)), _p(np.ascontiguousarray(n_inv))); return out


def kate_division(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64); q = np.zeros((max(a.shape[0] - 1, 0), 4), dtype=np.uint64)
    if a.shape[0] > 1:
        lib().orc_kate_division(_p(q), _p(a), C.c_uint64(a.shape[0]), _p(np.ascontiguousarray(b)))
    return q


def batch_invert(a):
    a = np.array(a, dtype=np.uint64, copy=True, order="C"); lib().orc_batch_invert(_p(a), C.c_uint64(a.shape[0])); return a


def prefix_product(v):
    v = np.ascontiguousarray(v, dtype=np.uint64); z = np.zeros_like(v); t = _fe()
    lib().orc_prefix_product(_p(z), _p(v), C.c_uint64(v.shape[0]), _p(t)); return z, t


def prefix_sum(v):
    v = np.ascontiguousarray(v, dtype=np.uint64); z = np.zeros_like(v); t = _fe()
    lib().orc_prefix_sum(_p(z), _p(v), C.c_uint64(v.shape[0]), _p(t)); return z, t


def ifft(a, omega_inv, log_n: int, divisor, threads: int | None = None):
    threads = threads or usable_cpus()
    a = np.array(a, dtype=np.uint64, copy=True, order="C")
    lib().orc_ifft(_p(a), _p(np.ascontiguousarray(omega_inv)), C.c_uint32(log_n), _p(np.ascontiguousarray(divisor)), C.c_int(threads)); return a


def coeff_to_extended(coeffs, k, ext_k, g_coset, g_coset_inv, ext_omega, threads: int | None = None):
    threads = threads or usable_cpus()
    coeffs = np.ascontiguousarray(coeffs); dst = np.zeros((1 << ext_k, 4), dtype=np.uint64)
    lib().orc_coeff_to_extended(_p(dst), _p(coeffs), C.c_uint32(k), C.c_uint32(ext_k), _p(np.ascontiguousarray(g_coset)),
                                _p(np.ascontiguousarray(g_coset_inv)), _p(np.ascontiguousarray(ext_omega)), C.c_int(threads))
    return dst


def extended_to_coeff(a, ext_k, g_coset, g_coset_inv, ext_omega_inv, ext_divisor, threads: int | None = None):
    threads = threads or usable_cpus()
    a = np.array(a, dtype=np.uint64, copy=True, order="C")
    lib().orc_extended_to_coeff(_p(a), C.c_uint32(ext_k), _p(np.ascontiguousarray(g_coset)), _p(np.ascontiguousarray(g_coset_inv)),
                                _p(np.ascontiguousarray(ext_omega_inv)), _p(np.ascontiguousarray(ext_divisor)), C.c_int(threads))
    return a


def eval_polynomial(poly, point):
    o = _fe(); poly = np.ascontiguousarray(poly)
    lib().orc_eval_polynomial(_p(o), _p(poly), C.c_uint64(poly.shape[0]), _p(np.ascontiguousarray(point))); return o


def eval_polynomial_mt(poly, point, threads: int | None = None):
    threads = threads or usable_cpus()
    o = _fe(); poly = np.ascontiguousarray(poly)
    lib().orc_eval_polynomial_mt(_p(o), _p(poly), C.c_uint64(poly.shape[0]), _p(np.ascontiguousarray(point)), C.c_int(threads)); return o


def gate_eval(polys, coeffs, term_len, factor_poly, factor_rot, n: int, dst=None, threads: int = 1):
    """dst[i] (+)= sum_j coeffs[j] * prod_k polys[factor_poly[.]][(i + factor_rot[.]) mod n]  (restated evaluate_h operand shape); dst given: accumulate"""
    polys = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
    arr = (C.c_void_p * max(1, len(polys)))(*[p.ctypes.data for p in polys])
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    tl = np.ascontiguousarray(term_len, dtype=np.uint32); fp = np.ascontiguousarray(factor_poly, dtype=np.uint32); fr_ = np.ascontiguousarray(factor_rot, dtype=np.int32)
    acc = dst is not None
    out = np.array(dst, dtype=np.uint64, copy=True, order="C") if acc else np.zeros((n, 4), dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if threads > 1:   # rows dealt over threads (same loop per row): the at-size checker
        lib().orc_gate_eval_mt(_p(out), arr, _p(coeffs), vp(tl), C.c_uint32(len(tl)), vp(fp), vp(fr_), C.c_uint64(n), C.c_int(1 if acc else 0), C.c_int(threads))
    else:
        lib().orc_gate_eval(_p(out), arr, _p(coeffs), vp(tl), C.c_uint32(len(tl)), vp(fp), vp(fr_), C.c_uint64(n), C.c_int(1 if acc else 0))
    return out


def srs_setup(k: int, tau_mont, omega_mont):
    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64); gl = np.zeros((n, 8), dtype=np.uint64)
    gs = np.zeros((n, 4), dtype=np.uint64); gls = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_srs_setup(_p(g), _p(gl), C.c_uint32(k), _p(np.ascontiguousarray(tau_mont)), _p(np.ascontiguousarray(omega_mont)), _p(gs), _p(gls))
    return g, gl, gs, gls
