"""
-m gpu: full-vector parity AT THE SIZES THE METRIC NAMES -- no sampling, no round-trip-only checks.

  * NTT at k = 26 (BASELINE.json `metric`: "NTT Fr-butterflies/sec at k=26"): every one of the 2^26 output words, forward and inverse,
    equal to the oracle's best_fft / EvaluationDomain::ifft (the restatement of halo2's radix-2 layers, all usable host threads; ~10 s each);
  * extended_to_coeff at 2^28 (the extended domain of k = 26 with four quotient chunks [REF integration/configs/layer4.config:3-10]):
    a random evaluation vector in general position, the 2^26 coefficients the caller keeps (and a 2^20-word sample of the rest) equal to
    the oracle's orc_extended_to_coeff;
  * MSM at 2^24 INDEPENDENT random points (not an SRS the device generated from a known tau): HIP vs the oracle's best_multiexp
    (restatement of halo2's multiexp_serial per thread + fold), window tables off and on, device and host scalars.
The oracle is the checker only (oracle/, TEST INFRASTRUCTURE).
"""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import affine_of
from tests.test_gpu_headline import last_run
from tests.test_gpu_properties import dev_scalars

pytestmark = pytest.mark.gpu
NPROC = cref.usable_cpus()


def host_gib_available() -> float:
    """what this process may still allocate: MemAvailable capped by the cgroup limit (a box that runs out of memory is a lost box)"""
    avail = 0.0
    with open("/proc/meminfo") as f:
        for line in f:
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / (1 << 20)
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(p).read().strip()
            if v.isdigit():
                used = 0
                for q in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                    try:
                        used = int(open(q).read().strip()); break
                    except OSError:
                        pass
                avail = min(avail, (int(v) - used) / (1 << 30))
        except OSError:
            pass
    return avail


@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    return pkg


def as_host(t, n):
    return t.cpu().numpy().view(np.uint64).reshape(n, 4)


def test_ntt_2_26_full_vector_matches_oracle(zk):
    """the size the metric is quoted on: forward and inverse, all 2^26 words against cref.best_fft / cref.ifft"""
    if host_gib_available() < 14:
        pytest.skip("needs ~12 GiB of host memory for the oracle's copies at 2^26")
    h2 = zk.halo2
    k = 26
    n = 1 << k
    dom = h2.EvaluationDomain(2, k)
    a = dev_scalars(n, 2626)
    host = as_host(a, n)
    want_f = cref.best_fft(host, dom.omega, k, threads=NPROC)          # cref copies its input
    dom.coeff_to_lagrange(a)
    got_f = as_host(a, n)
    assert hashlib.sha256(got_f.tobytes()).digest() == hashlib.sha256(want_f.tobytes()).digest()
    assert (got_f == want_f).all()
    del got_f
    want_i = cref.ifft(want_f, dom.omega_inv, k, dom.ifft_divisor, threads=NPROC)
    del want_f
    dom.lagrange_to_coeff(a)
    got_i = as_host(a, n)
    assert (got_i == want_i).all() and (got_i == host).all()
    del a
    torch.cuda.empty_cache()


def test_extended_to_coeff_2_28_matches_oracle_on_the_kept_coefficients(zk):
    """EvaluationDomain::extended_to_coeff on a 2^28-point vector in general position: the 2^26 coefficients create_proof keeps (h(X) is cut
    into four 2^26 pieces, the first is compared in full here, the rest by a 2^20-word sample) equal to the oracle's."""
    if host_gib_available() < 30:
        pytest.skip("needs ~26 GiB of host memory (8 GiB vector, the oracle's copy and twiddles)")
    h2 = zk.halo2
    k, j = 26, 5          # j = 5: extended_k = k + ceil(log2(j - 1)) = 28
    dom = h2.EvaluationDomain(j, k)
    assert dom.extended_k == 28
    en = 1 << 28
    a = dev_scalars(en, 2828)
    host = as_host(a, en)
    want = cref.extended_to_coeff(host, 28, dom.g_coset, dom.g_coset_inv, dom.extended_omega_inv, dom.extended_ifft_divisor, threads=NPROC)
    del host
    dom.extended_to_coeff(a)
    kept = as_host(a[: 1 << k].contiguous(), 1 << k)
    assert (kept == want[: 1 << k]).all()
    idx = np.random.default_rng(28).integers(1 << k, en, size=1 << 20)
    got_s = as_host(a[torch.from_numpy(idx).cuda()].contiguous(), 1 << 20)
    assert (got_s == want[idx]).all()
    del a, want
    torch.cuda.empty_cache()


def test_msm_2_24_independent_points_matches_oracle_best_multiexp(zk):
    """2^24 pairs on independent points k_i G (generated on the device with the fixed-base kernel; a 4096-point sample checked on the curve
    and four points against the oracle's own scalar multiple), HIP vs best_multiexp of the oracle, tables off and on."""
    if host_gib_available() < 8:
        pytest.skip("needs ~6 GiB of host memory")
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    k = 24
    n = 1 << k
    ks = dev_scalars(n, 2400)
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    capi.check(lib.mi355_g1_fixed_base_mul_dev(capi.ptr(pts), capi.ptr(ks), n))
    capi.check(lib.mi355_synchronize())
    points = pts.cpu().numpy().view(np.uint64).reshape(n, 8)
    ks_h = as_host(ks, n)
    G = cref.g1_generator()
    for i in (0, 1, n // 3, n - 1):                       # the inputs are what they claim to be
        assert (points[i] == cref.g1_to_affine(cref.g1_mul(G, ks_h[i]))).all()
    # y^2 = x^3 + 3 on a 4096-point sample, in the field through the oracle
    x, y = np.ascontiguousarray(points[:, :4]), np.ascontiguousarray(points[:, 4:])
    x3 = cref.f_mul_vec(cref.FQ, cref.f_mul_vec(cref.FQ, x, x), x)
    y2 = cref.f_mul_vec(cref.FQ, y, y)
    three = np.array(pyref.to_limbs(3 * pyref.MONT_R % pyref.P_MOD), dtype=np.uint64)
    rhs = np.stack([cref.f_add(cref.FQ, x3[i], three) for i in range(0, n, n // 4096)])
    assert (y2[:: n // 4096] == rhs).all()
    del x, y, x3, y2
    sc = dev_scalars(n, 2401)
    sc_h = as_host(sc, n)
    want = cref.g1_to_affine(cref.best_multiexp(sc_h, points, threads=NPROC))
    params = h2.ParamsKZG.from_host(k, points, points)
    got = affine_of(params.commit(sc))
    assert (got == want).all() and not last_run(zk)["shared"]
    params.precompute(lagrange=False)
    got_t = affine_of(params.commit(sc))
    run = last_run(zk)
    assert (got_t == want).all() and run["shared"] and run["c"] >= 20, run
    assert (affine_of(params.commit(sc_h)) == want).all()       # host-pointer path (point-range slices)
    assert last_run(zk)["host_slices"] >= 2
    params.release()
    del pts, ks, sc
    torch.cuda.empty_cache()
