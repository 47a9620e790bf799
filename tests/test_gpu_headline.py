"""
-m gpu: the BENCHMARKED configuration under pytest.

  * the headline schedule -- window tables (mi355_srs_precompute), one shared bucket set, c >= 20, 16384-entry level-2 tiles,
    shift-packed table rows -- at k = 22 / 24 / 26, uniform and witness-like scalars, both bases (g through commit, g_lagrange through
    commit_lagrange), each result checked in the field: commit(p) = p(tau) G [SURVEY 8c: the strongest large-N oracle], with
    mi355_msm_last_run / _last_plan asserting that the table schedule really ran;
  * BASELINE.json configs[1] verbatim: 2^20 random scalars x 2^20 independent random points, HIP vs the oracle's best_multiexp
    (restatement of halo2's multiexp_serial / best_multiexp), bit-exact after to_affine;
  * BASELINE.json configs[2] verbatim: 2^24 coefficients forward + inverse, every output word equal to the oracle's best_fft / ifft
    (raw Montgomery bytes);
  * the host-pointer entry point (what the Rust shim calls) at a size where it is cut into point-range slices.
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
from tests.gpu_common import affine_of
from tests.test_gpu_properties import dev_scalars, field_commit

pytestmark = pytest.mark.gpu
R = pyref.R_MOD
TAU = 0x5343524F4C4C0001
NPROC = cref.usable_cpus()   # the cgroup quota, not the host's 256 hardware threads


@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    return pkg


def last_run(zk):
    lib = zk._capi.lib()
    dev, ex, sh, sl = C.c_int(), C.c_char_p(), C.c_int(), C.c_int()
    zk._capi.check(lib.mi355_msm_last_run(C.byref(dev), C.byref(ex), C.byref(sh), C.byref(sl)))
    c, w, e = C.c_int(), C.c_int(), C.c_uint64()
    zk._capi.check(lib.mi355_msm_last_plan(C.byref(c), C.byref(w), C.byref(e)))
    return {"devices": dev.value, "exchange": ex.value.decode(), "shared": bool(sh.value), "host_slices": sl.value, "c": c.value, "windows": w.value, "entries": e.value}


@pytest.mark.parametrize("k,kind,both", [(22, "uniform", True), (24, "witness", False), (24, "uniform", False), (26, "uniform", True), (26, "witness", False)])
def test_headline_schedule_commit_equals_field_evaluation(zk, k, kind, both):
    """the schedule bench.py times: tables on, shared buckets, c >= 20."""
    h2 = zk.halo2
    n = 1 << k
    params = h2.ParamsKZG.setup(k, TAU + k)
    params.precompute(lagrange=both)
    sc = dev_scalars(n, 4200 + k, kind)
    got = affine_of(params.commit(sc))
    run = last_run(zk)
    assert run["shared"] and run["c"] >= 20 and run["windows"] == (255 + run["c"] - 1) // run["c"] and run["entries"] == n * run["windows"], run
    want = field_commit(sc, TAU + k)
    assert (got == want).all(), f"commit != p(tau) G at k={k} ({kind})"
    if both:
        # the Lagrange basis through the same schedule: commit_lagrange(evals) == commit(ifft(evals)) == p(tau) G
        coeffs = sc.clone()
        dom = h2.EvaluationDomain(2, k)
        dom.coeff_to_lagrange(sc)                      # sc now holds the evaluations of the polynomial with coefficients `coeffs`
        got_l = affine_of(params.commit_lagrange(sc))
        assert last_run(zk)["shared"]
        assert (got_l == want).all(), f"commit_lagrange != p(tau) G at k={k}"
        del coeffs
    params.release()
    del sc
    torch.cuda.empty_cache()


def test_msm_2_20_random_points_matches_oracle_best_multiexp(zk):
    """BASELINE config #2: 2^20 random scalars / points, bit-exact vs best_multiexp (oracle restatement, all host threads)."""
    h2 = zk.halo2
    n = 1 << 20
    rng = np.random.default_rng(20)
    ks = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); ks[:, 3] &= np.uint64((1 << 60) - 1)
    points = cref.g1_mul_generator_vec(ks, threads=NPROC)          # independent points k_i G
    sc = dev_scalars(n, 2020).cpu().numpy().view(np.uint64).reshape(n, 4)
    want = cref.g1_to_affine(cref.best_multiexp(sc, points, threads=NPROC))
    # ad-hoc bases (best_multiexp on a slice that is not a registered SRS)
    got = affine_of(h2.best_multiexp(sc, points))
    assert (got == want).all()
    # the same points registered as a basis, without and with window tables, host and device scalars
    params = h2.ParamsKZG.from_host(20, points, points)
    assert (affine_of(params.commit(sc)) == want).all()
    params.precompute(lagrange=False)
    assert (affine_of(params.commit(sc)) == want).all() and last_run(zk)["shared"]
    assert (affine_of(params.commit(torch.from_numpy(sc.view(np.int64)).cuda())) == want).all()
    params.release()


def test_ntt_2_24_full_vector_matches_oracle(zk):
    """BASELINE config #3: 2^24 coefficients, forward and inverse, every word equal to best_fft / EvaluationDomain::ifft of the oracle."""
    h2 = zk.halo2
    k = 24
    dom = h2.EvaluationDomain(2, k)
    a = dev_scalars(1 << k, 2424)
    host = a.cpu().numpy().view(np.uint64).reshape(1 << k, 4)
    want_f = cref.best_fft(host.copy(), dom.omega, k, threads=NPROC)
    dom.coeff_to_lagrange(a)
    got_f = a.cpu().numpy().view(np.uint64).reshape(1 << k, 4)
    assert hashlib.sha256(got_f.tobytes()).digest() == hashlib.sha256(want_f.tobytes()).digest() and (got_f == want_f).all()
    want_i = cref.ifft(want_f.copy(), dom.omega_inv, k, dom.ifft_divisor, threads=NPROC)
    dom.lagrange_to_coeff(a)
    got_i = a.cpu().numpy().view(np.uint64).reshape(1 << k, 4)
    assert (got_i == want_i).all() and (got_i == host).all()


@pytest.mark.parametrize("k", [20, 23])
def test_host_pointer_path_sliced(zk, k):
    """mi355_msm_g1_host, the entry point the Rust shim calls: k = 23 is cut into point-range slices whose bucket sets are folded
    before ONE reduction tail; the result must equal the device-resident call and the field evaluation."""
    h2 = zk.halo2
    n = 1 << k
    params = h2.ParamsKZG.setup(k, TAU + 77)
    params.precompute(lagrange=False)
    sc = dev_scalars(n, 5000 + k)
    want = field_commit(sc, TAU + 77)
    sc_host = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
    got = affine_of(params.commit(sc_host))
    run = last_run(zk)
    assert (got == want).all()
    assert (run["host_slices"] == 1) if k < 22 else (2 <= run["host_slices"] <= 8), run
    assert (affine_of(params.commit(sc)) == want).all()
    # witness-like scalars (giant buckets straddle slices) and a ragged length through a prefix slice
    wl = dev_scalars(n, 6000 + k, "witness")
    assert (affine_of(params.commit(wl.cpu().numpy().view(np.uint64).reshape(n, 4))) == field_commit(wl, TAU + 77)).all()
    m = n - 12345
    part = sc_host[:m].copy()
    got_p = affine_of(h2.best_multiexp(part, params.g_slice(0, m)))
    assert (got_p == field_commit(sc[:m].contiguous(), TAU + 77)).all()
    params.release()
