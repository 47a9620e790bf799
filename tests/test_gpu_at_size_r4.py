"""
-m gpu: full-vector oracle parity AT THE SIZES THE BENCH LINE QUOTES for the kernels round 3 added or restructured (VERDICT r3 weak #1b, next #1):

  * mi355_fr_gate_eval_dev at 2^24 and 2^26 rows: 6 polynomials, 16 terms, rotations in {+-1, +-2^20, +-(n - 1), +-3n}, every output word
    against the oracle's orc_gate_eval (rows dealt over threads, same loop per row), plain and accumulating; and at 2^28 rows (the extended
    domain of k = 26) on sampled rows against Python big integers -- the row index arithmetic at the largest size a rotation can reach;
  * mi355_fr_prefix_sum_dev, mi355_fr_prefix_product_dev, mi355_fr_kate_division_dev, mi355_fr_batch_invert_dev: all 2^26 words against the
    oracle's serial loops (the scans were restructured in round 3: LDS aliasing, two workgroups per CU);
  * mi355_ntt_fr_batch_host: 5 x 2^24 through the overlapped pipeline (two staging buffers reused, helper thread), sha256 of every result
    against cref.best_fft / cref.ifft;
  * mi355_coset_ntt_fr_batch_dev at 2^22 and mi355_ntt_fr_batch_dev at 2^22 against cref.
The oracle is the checker only (oracle/, TEST INFRASTRUCTURE).
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.test_gpu_metric_size import as_host, host_gib_available
from tests.test_gpu_properties import dev_scalars

pytestmark = pytest.mark.gpu
NPROC = cref.usable_cpus()
R = pyref.R_MOD


@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    return pkg


def gate_terms(n, rng, n_polys=6):
    """16 terms over n_polys polynomials, 36 factors; every rotation class the verdict names appears on a first and on a later factor;
    coefficients: general, 1 and -1 (the multiplication-free paths), one constant term."""
    rots = [1, -1, 1 << 20, -(1 << 20), n - 1, -(n - 1), 3 * n, -3 * n, 0, 2, -2, 0]
    terms = []
    q = 0
    for j in range(16):
        ln = [3, 2, 1, 4, 2, 0, 3, 2, 3, 1, 2, 3, 2, 4, 2, 2][j]
        fac = []
        for _ in range(ln):
            fac.append((int(rng.integers(0, n_polys)), rots[q % len(rots)])); q += 1
        c = cref.fr_mont(1) if j % 5 == 0 else cref.fr_mont(R - 1) if j % 5 == 3 else cref.fr_mont(int(rng.integers(2, 1 << 62)) * 0x9E3779B97F4A7C15 % R)
        terms.append((c, fac))
    return terms


def flat(terms):
    return (np.stack([c for c, _ in terms]), [len(f) for _, f in terms], [p for _, f in terms for p, _ in f], [r for _, f in terms for _, r in f])


@pytest.mark.parametrize("k", [24, 26])
def test_gate_eval_full_vector_at_size(zk, k):
    n = 1 << k
    need = 6 * n * 32 / 2**30 + 3 * n * 32 / 2**30 + 2
    if host_gib_available() < need:
        pytest.skip(f"needs ~{need:.0f} GiB of host memory")
    h2 = zk.halo2
    rng = np.random.default_rng(2400 + k)
    polys_d = [dev_scalars(n, 9100 + 7 * k + i) for i in range(6)]
    polys_h = [as_host(p, n) for p in polys_d]
    terms = gate_terms(n, rng)
    coeffs, tl, fp, fr_ = flat(terms)
    dst = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    want = cref.gate_eval(polys_h, coeffs, tl, fp, fr_, n, threads=NPROC)
    h2.gate_eval(dst, polys_d, terms, n)
    got = as_host(dst, n)
    assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()
    assert (got == want).all()
    # accumulate on top of the previous result, a different (shorter) list: dst is read and written by the same thread
    terms2 = terms[3:9]
    c2, tl2, fp2, fr2 = flat(terms2)
    want2 = cref.gate_eval(polys_h, c2, tl2, fp2, fr2, n, dst=want, threads=NPROC)
    del want
    h2.gate_eval(dst, polys_d, terms2, n, accumulate=True)
    got2 = as_host(dst, n)
    assert (got2 == want2).all()
    # dst aliasing an un-rotated operand is allowed (same thread reads then writes); with a rotation it is refused
    lib, capi = zk._capi.lib(), zk._capi
    t_alias = [(cref.fr_mont(5), [(0, 0), (1, 3)]), (cref.fr_mont(1), [(0, 0)])]
    ca, tla, fpa, fra = flat(t_alias)
    want3 = cref.gate_eval(polys_h, ca, tla, fpa, fra, n, threads=NPROC)
    h2.gate_eval(polys_d[0], polys_d, t_alias, n)
    assert (as_host(polys_d[0], n) == want3).all()
    bad = [(cref.fr_mont(5), [(1, 1)])]
    cb, tlb, fpb, frb = flat(bad)
    arr = (C.c_void_p * 6)(*[q.data_ptr() for q in polys_d])
    rc = lib.mi355_fr_gate_eval_dev(C.c_void_p(polys_d[1].data_ptr()), arr, 6, capi.ptr(np.ascontiguousarray(cb)), (C.c_uint32 * 1)(1), 1, (C.c_uint32 * 1)(1), (C.c_int32 * 1)(1), n, 0)
    assert rc == capi.EBADARG
    del polys_d, dst
    torch.cuda.empty_cache()


def test_gate_eval_2_28_rows_sampled_against_big_integers(zk):
    """the extended domain of k = 26: n = 2^28 rows, rotations scaled by 4 as evaluate_h does on the extended domain (+-4, +-4 * 2^20, +-3n);
    256 sampled rows (first, last, around the wrap points) recomputed with Python integers from the operands' own words."""
    h2 = zk.halo2
    n = 1 << 28
    free, _ = torch.cuda.mem_get_info()
    if free < 30 * 2**30:
        pytest.skip("needs ~26 GiB of HBM")
    polys_d = [dev_scalars(n, 2800 + i) for i in range(2)]
    dst = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    terms = [(cref.fr_mont(7), [(0, 4), (1, -4)]), (cref.fr_mont(1), [(1, 4 << 20), (0, -(4 << 20)), (0, 0)]), (cref.fr_mont(R - 1), [(0, 3 * n - 4), (1, -(3 * n) + 8)]),
             (cref.fr_mont(11), [(1, n - 4)]), (cref.fr_mont(13), [])]
    h2.gate_eval(dst, polys_d, terms, n)
    rng = np.random.default_rng(28)
    rows = np.concatenate([np.arange(0, 16), np.arange(n - 16, n), (1 << 20) * 4 + np.arange(-8, 8), n - (4 << 20) + np.arange(-8, 8), rng.integers(0, n, size=192)]).astype(np.int64)
    Rinv = pow(pyref.MONT_R, -1, R)

    def word(t, idx):   # canonical value of element idx of a device vector
        return pyref.from_limbs([int(x) & (2**64 - 1) for x in t[idx].cpu().tolist()]) * Rinv % R
    for i in rows.tolist():
        want = 0
        for c, fac in terms:
            v = pyref.from_limbs([int(x) for x in c]) * Rinv % R
            for p, r in fac:
                v = v * word(polys_d[p], (i + r) % n) % R
            want = (want + v) % R
        assert word(dst, i) == want, i
    del polys_d, dst
    torch.cuda.empty_cache()


def test_scans_full_vector_2_26(zk):
    """prefix sum (mv-lookup phi), grand product, kate_division and batch inversion: every one of the 2^26 output words against the oracle's
    serial loops; the totals as well"""
    if host_gib_available() < 10:
        pytest.skip("needs ~8 GiB of host memory")
    h2 = zk.halo2
    k = 26
    n = 1 << k
    v = dev_scalars(n, 2660)
    host = as_host(v, n)
    # running sum
    d, tot = h2.prefix_sum(v, want_total=True)
    want, wtot = cref.prefix_sum(host)
    got = as_host(d, n)
    assert (got == want).all() and (tot == wtot).all()
    del want, got
    # grand product
    d2, tot2 = h2.prefix_product(v, dst=d, want_total=True)
    want, wtot = cref.prefix_product(host)
    got = as_host(d2, n)
    assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()
    assert (got == want).all() and (tot2 == wtot).all()
    del want, got
    # kate_division by (X - z): n - 1 coefficients
    z = h2.fr(0x1234567890ABCDEF1234567890ABCDEF % R)
    q = h2.kate_division(v, z)
    want = cref.kate_division(host, z)
    got = as_host(q, n - 1)
    assert (got == want).all()
    del want, got, q
    # batch inversion in place (zeros stay zero): a few zeros planted
    v[12345] = 0; v[n - 1] = 0; v[0] = 0
    host = as_host(v, n)
    h2.batch_invert(v)
    want = cref.batch_invert(host)
    assert (as_host(v, n) == want).all()
    del v, d, want
    torch.cuda.empty_cache()


def test_ntt_batch_host_5_x_2_24_matches_oracle(zk):
    """the host-pointer column loop: five polynomials of 2^24 through upload | transform | download with two staging buffers and the helper
    thread (each staging buffer is reused at least twice), forward then inverse, sha256 against cref"""
    if host_gib_available() < 10:
        pytest.skip("needs ~8 GiB of host memory")
    h2 = zk.halo2
    k = 24
    n = 1 << k
    dom = h2.EvaluationDomain(2, k)
    polys = [as_host(dev_scalars(n, 2400 + i), n).copy() for i in range(5)]
    batch = [p.copy() for p in polys]
    h2.best_fft_many(batch, dom.omega, k)
    for a, p in zip(batch, polys):
        want = cref.best_fft(p, dom.omega, k, threads=NPROC)
        assert hashlib.sha256(a.tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()
    h2.best_fft_many(batch, dom.omega_inv, k, divisor=dom.ifft_divisor)
    for a, p in zip(batch, polys):
        assert (a == p).all()
    torch.cuda.empty_cache()


def test_coset_and_plain_batches_on_resident_buffers_2_22(zk):
    """mi355_coset_ntt_fr_batch_dev (coeff_to_extended_part of the scroll fork: coefficients scaled by factor^i, then the transform) and
    mi355_ntt_fr_batch_dev, four polynomials of 2^22, against the oracle: the scaling by plain field arithmetic, the transform by best_fft"""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    k = 22
    n = 1 << k
    dom = h2.EvaluationDomain(5, k)
    factor_int = h2.FR_ZETA * pow(h2.fr_to_int(dom.extended_omega), 3, R) % R   # zeta * extended_omega^3: part 3
    factor = h2.fr(factor_int)
    srcs = [dev_scalars(n, 2200 + i) for i in range(4)]
    hosts = [as_host(s, n).copy() for s in srcs]
    dsts = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    capi.check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * 4)(*[d.data_ptr() for d in dsts]), (C.c_void_p * 4)(*[s.data_ptr() for s in srcs]), 4, k, capi.ptr(factor), capi.ptr(dom.omega)))
    # factor^i for all i by repeated squaring blocks (vectorised through the oracle's f_mul_vec)
    pw = np.zeros((n, 4), dtype=np.uint64); pw[0] = cref.fr_mont(1)
    m = 1
    while m < n:
        step = np.tile(cref.fr_mont(pow(factor_int, m, R)), (m, 1))
        pw[m:2 * m] = cref.f_mul_vec(cref.FR, pw[:m], step)
        m *= 2
    for d, hsrc, s in zip(dsts, hosts, srcs):
        want = cref.best_fft(cref.f_mul_vec(cref.FR, hsrc, pw), dom.omega, k, threads=NPROC)
        assert (as_host(d, n) == want).all()
        assert (as_host(s, n) == hsrc).all()          # the sources are not modified
    # in place (dst == src) is the other documented form
    capi.check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * 1)(srcs[0].data_ptr()), (C.c_void_p * 1)(srcs[0].data_ptr()), 1, k, capi.ptr(factor), capi.ptr(dom.omega)))
    assert (as_host(srcs[0], n) == as_host(dsts[0], n)).all()
    # plain batch, forward and inverse
    bufs = [dev_scalars(n, 2200 + i) for i in range(1, 4)]
    h2.best_fft_many(bufs, dom.omega, k)
    for b, hsrc in zip(bufs, hosts[1:]):
        assert (as_host(b, n) == cref.best_fft(hsrc, dom.omega, k, threads=NPROC)).all()
    h2.best_fft_many(bufs, dom.omega_inv, k, divisor=dom.ifft_divisor)
    for b, hsrc in zip(bufs, hosts[1:]):
        assert (as_host(b, n) == hsrc).all()
    torch.cuda.empty_cache()


def test_eval_polynomial_batch_equals_single_calls_and_oracle(zk):
    """mi355_eval_polynomial_batch_dev (step 9 of create_proof with one synchronisation): 300 evaluations (two scratch chunks) over 7 polynomials of
    2^18 + a ragged tail check at n = 1000, against single calls and the oracle's Horner"""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    for n in (1 << 18, 1000, 4096):
        polys = [dev_scalars(n, 1800 + i) for i in range(7)]
        hosts = [as_host(p, n) for p in polys]
        rng = np.random.default_rng(18)
        B = 300 if n > 4096 else 9 if n == 1000 else 37   # odd batches (9, 37): the pointer table in front of the point / result arrays is padded to keep them 16-byte aligned
        which = [int(rng.integers(0, 7)) for _ in range(B)]
        pts = np.ascontiguousarray(np.stack([h2.fr(int(rng.integers(1, 1 << 62)) * 0x9E3779B97F4A7C15 % R) for _ in range(B)]))
        out = np.zeros((B, 4), dtype=np.uint64)
        arr = (C.c_void_p * B)(*[polys[w].data_ptr() for w in which])
        capi.check(lib.mi355_eval_polynomial_batch_dev(arr, B, n, capi.ptr(pts), capi.ptr(out)))
        assert (h2.eval_polynomial_many([polys[w] for w in which], pts) == out).all()      # the halo2.py wrapper of the same call
        for i in range(0, B, 37):
            assert (out[i] == h2.eval_polynomial(polys[which[i]], pts[i])).all()
            assert (out[i] == cref.eval_polynomial(hosts[which[i]], pts[i])).all()
    capi.check(lib.mi355_eval_polynomial_batch_dev(None, 0, 0, None, None))


def test_interleave_and_mem_info(zk):
    """mi355_fr_interleave_dev: dst[i * Q + q] = parts[q][i] for Q = 1, 4, 8 (numpy as the checker), overlap and argument rules;
    mi355_mem_info: the library's own accounting moves with mi355_buf_alloc / _free / _trim"""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    n = (1 << 16) + 3                       # not a power of two: the kernel has no such requirement
    for Q in (1, 4, 8):
        parts = [dev_scalars(n, 700 + q) for q in range(Q)]
        dst = torch.empty((Q * n, 4), dtype=torch.int64, device="cuda")
        arr = (C.c_void_p * Q)(*[p.data_ptr() for p in parts])
        capi.check(lib.mi355_fr_interleave_dev(C.c_void_p(dst.data_ptr()), arr, Q, n))
        want = np.stack([as_host(p, n) for p in parts], axis=1).reshape(Q * n, 4)
        assert (as_host(dst, Q * n) == want).all()
        assert (as_host(h2.interleave(parts), Q * n) == want).all()
    arr9 = (C.c_void_p * 9)(*([parts[0].data_ptr()] * 9))
    assert lib.mi355_fr_interleave_dev(C.c_void_p(dst.data_ptr()), arr9, 9, n) == capi.EBADARG              # more than 8 parts
    arr1 = (C.c_void_p * 1)(dst.data_ptr())
    assert lib.mi355_fr_interleave_dev(C.c_void_p(dst.data_ptr()), arr1, 1, n) == capi.EBADARG              # dst overlaps a part
    free0, tot, live0, pool0, ws0 = (C.c_uint64() for _ in range(5))
    capi.check(lib.mi355_buf_trim())
    capi.check(lib.mi355_mem_info(0, C.byref(free0), C.byref(tot), C.byref(live0), C.byref(pool0), C.byref(ws0)))
    assert tot.value > (200 << 30) and free0.value <= tot.value and pool0.value == 0
    b = h2.DeviceBuffer(1 << 30)
    free1, live1, pool1 = C.c_uint64(), C.c_uint64(), C.c_uint64()
    capi.check(lib.mi355_mem_info(0, C.byref(free1), None, C.byref(live1), C.byref(pool1), None))
    assert live1.value == live0.value + (1 << 30) and free1.value <= free0.value - (1 << 30) + (64 << 20)
    b.free()
    capi.check(lib.mi355_mem_info(0, None, None, C.byref(live1), C.byref(pool1), None))
    assert live1.value == live0.value and pool1.value == 1 << 30                                              # freed = pooled, still held
    capi.check(lib.mi355_buf_trim())
    capi.check(lib.mi355_mem_info(0, C.byref(free1), None, None, C.byref(pool1), None))
    assert pool1.value == 0 and free1.value >= free0.value - (64 << 20)
    assert lib.mi355_mem_info(99, None, None, None, None, None) == capi.EBADARG
    mi = h2.mem_info()
    assert mi["total"] == tot.value and mi["pooled"] == 0


def test_host_alloc_is_page_locked_memory_the_entry_points_accept(zk):
    """mi355_host_alloc / _free: the block is ordinary addressable host memory (numpy can wrap it), uploads and host-pointer calls take it like any
    pointer, results equal those from pageable memory"""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    k = 16
    n = 1 << k
    p = C.c_void_p()
    capi.check(lib.mi355_host_alloc(32 * n, C.byref(p)))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, 4))
    vals = as_host(dev_scalars(n, 1616), n)
    arr[:] = vals
    dom = h2.EvaluationDomain(2, k)
    capi.check(lib.mi355_ntt_fr_host(p, k, capi.ptr(dom.omega)))                       # in place, through the pinned block
    assert (arr == cref.best_fft(vals, dom.omega, k, threads=4)).all()
    buf = h2.DeviceBuffer(32 * n)
    capi.check(lib.mi355_buf_upload(C.c_void_p(buf.data_ptr()), p, 32 * n))
    assert (buf.fr() == arr).all()
    buf.free()
    capi.check(lib.mi355_host_free(p))
    capi.check(lib.mi355_host_free(None))
    assert lib.mi355_host_alloc(0, C.byref(p)) == capi.EBADARG


def _factor_powers(factor_int, n):
    pw = np.zeros((n, 4), dtype=np.uint64); pw[0] = cref.fr_mont(1)
    m = 1
    while m < n:
        pw[m:2 * m] = cref.f_mul_vec(cref.FR, pw[:m], np.tile(cref.fr_mont(pow(factor_int, m, R)), (m, 1)))
        m *= 2
    return pw


@pytest.mark.parametrize("k", [8, 9, 10, 13, 16, 18, 19, 20, 21, 22, 24])
def test_coset_shift_folded_into_the_first_pass_matches_oracle(zk, k):
    """round 6: up to 2^24 the coset shift a[i] *= f^i rides on the first pass of the transform (a 2^log_m-entry table on load, f^column on the inter-level
    twiddles) instead of running as k_distribute_powers; k = 8 (one pass) still takes the separate kernel, as does everything under MI355_NTT_COSET_FOLD_MAX_LOG=0 (a proof
    with that setting is one of tests/test_plonk_protocol.py's byte-equality cases).  Every word against the
    oracle (the scaling by plain field multiplications, the transform by best_fft), single and batched entry points, two coset factors (two table sets on one plan), in place
    and out of place, sources untouched."""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    n = 1 << k
    dom = h2.EvaluationDomain(9, k)                                                   # Q = 8 parts
    srcs = [dev_scalars(n, 6100 + 7 * k + i) for i in range(3)]
    hosts = [as_host(s, n).copy() for s in srcs]
    for part in (5, 0):
        factor_int = h2.FR_ZETA * pow(h2.fr_to_int(dom.extended_omega), part, R) % R
        factor = h2.fr(factor_int)
        pw = _factor_powers(factor_int, n)
        wants = [cref.best_fft(cref.f_mul_vec(cref.FR, hsrc, pw), dom.omega, k, threads=NPROC) for hsrc in hosts]
        out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        capi.check(lib.mi355_coset_ntt_fr_dev(capi.ptr(out), capi.ptr(srcs[0]), k, capi.ptr(factor), capi.ptr(dom.omega)))
        assert (as_host(out, n) == wants[0]).all(), f"single coset transform, part {part}"
        dsts = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(3)]
        capi.check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * 3)(*[d.data_ptr() for d in dsts]), (C.c_void_p * 3)(*[s.data_ptr() for s in srcs]), 3, k, capi.ptr(factor), capi.ptr(dom.omega)))
        for i in range(3):
            assert (as_host(dsts[i], n) == wants[i]).all(), f"batched coset transform {i}, part {part}"
            assert (as_host(srcs[i], n) == hosts[i]).all(), "a source was modified"
        inplace = srcs[1].clone()
        capi.check(lib.mi355_coset_ntt_fr_dev(capi.ptr(inplace), capi.ptr(inplace), k, capi.ptr(factor), capi.ptr(dom.omega)))
        assert (as_host(inplace, n) == wants[1]).all(), "in place"
    # the plain transform of the same plan is untouched by the coset tables it now owns
    plain = srcs[2].clone()
    dom.coeff_to_lagrange(plain)
    assert (as_host(plain, n) == cref.best_fft(hosts[2], dom.omega, k, threads=NPROC)).all()
