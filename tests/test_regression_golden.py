"""
Frozen end-to-end regression vectors (tests/golden/regression.json, written by tests/golden/make_regression.py from the pure-Python
oracle): the C oracle (CPU) and the HIP path (-m gpu) must reproduce the committed bytes.  These are regression vectors, not
reference KATs -- the reference has none at the MSM / NTT boundary.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import cref, pyref

R = pyref.R_MOD
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regression.json")))


def _fr(vals):
    return cref.f_from_canonical_vec(cref.FR, np.array([pyref.to_limbs(v % R) for v in vals], dtype=np.uint64).reshape(-1, 4))


def _canon_bytes(arr):
    return np.ascontiguousarray(cref.f_to_canonical_vec(cref.FR, arr)).tobytes()


def _inputs():
    g = np.stack([cref.g1_decompress(bytes.fromhex(h)) for h in GOLD["srs"]["g_compressed"]])
    gl = np.stack([cref.g1_decompress(bytes.fromhex(h)) for h in GOLD["srs"]["g_lagrange_compressed"]])
    sc = _fr([int(h, 16) for h in GOLD["msm"]["scalars_canonical_hex"]])
    return g, gl, sc


def _ntt_input():
    # the generator's splitmix64 stream, restated
    mask = (1 << 64) - 1
    x = int(GOLD["ntt"]["seed"], 16) & mask
    out = []
    for _ in range(1 << GOLD["ntt"]["k"]):
        w = []
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & mask
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
            w.append(z ^ (z >> 31))
        out.append((w[0] | (w[1] << 64) | (w[2] << 128) | (w[3] << 192)) % R)
    assert hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in out)).hexdigest() == GOLD["ntt"]["input_sha256"]
    return out


def test_c_oracle_reproduces_the_frozen_vectors():
    g, gl, sc = _inputs()
    k = GOLD["srs"]["k"]
    w = _fr([pyref.omega(k)])[0]
    assert cref.g1_compress(cref.g1_to_affine(cref.best_multiexp(sc, gl, 2))).hex() == GOLD["msm"]["commit_lagrange_compressed"]
    coeffs = cref.ifft(sc, _fr([pow(pyref.omega(k), -1, R)])[0], k, _fr([pow(1 << k, -1, R)])[0], 1)
    assert hashlib.sha256(_canon_bytes(coeffs)).hexdigest() == GOLD["msm"]["coeffs_sha256"]
    assert cref.g1_compress(cref.g1_to_affine(cref.best_multiexp(coeffs, g, 2))).hex() == GOLD["msm"]["commit_lagrange_compressed"]
    k2 = GOLD["ntt"]["k"]
    a = _fr(_ntt_input())
    f = cref.best_fft(a, _fr([pyref.omega(k2)])[0], k2, 2)
    assert hashlib.sha256(_canon_bytes(f)).hexdigest() == GOLD["ntt"]["output_sha256_canonical_le"]
    assert [format(cref.limbs_to_int(x), "064x") for x in cref.f_to_canonical_vec(cref.FR, f[:4])] == GOLD["ntt"]["output_first4_hex"]
    assert w is not None


@pytest.mark.gpu
def test_hip_path_reproduces_the_frozen_vectors():
    import __graft_entry__ as ge
    zk = ge.load_package(); zk.init(0); h2 = zk.halo2
    g, gl, sc = _inputs()
    k = GOLD["srs"]["k"]
    params = h2.ParamsKZG.from_host(k, g, gl)
    for tables in (False, True):
        if tables:
            params.precompute()
        c = params.commit_lagrange(sc)
        assert h2.g1_to_bytes(c).hex() == GOLD["msm"]["commit_lagrange_compressed"]
        dom = h2.EvaluationDomain(3, k)
        coeffs = sc.copy(); dom.lagrange_to_coeff(coeffs)
        assert hashlib.sha256(_canon_bytes(coeffs)).hexdigest() == GOLD["msm"]["coeffs_sha256"]
        assert h2.g1_to_bytes(params.commit(coeffs)).hex() == GOLD["msm"]["commit_lagrange_compressed"]
    params.release()
    k2 = GOLD["ntt"]["k"]
    a = _fr(_ntt_input())
    f = a.copy(); h2.best_fft(f, h2.fr(pyref.omega(k2)), k2)
    assert hashlib.sha256(_canon_bytes(f)).hexdigest() == GOLD["ntt"]["output_sha256_canonical_le"]
    d4 = h2.EvaluationDomain(4, 4)
    assert d4.extended_k == GOLD["coset"]["extended_k"]
    ext = d4.coeff_to_extended(a[:16].copy())
    assert hashlib.sha256(_canon_bytes(np.asarray(ext))).hexdigest() == GOLD["coset"]["output_sha256_canonical_le"]
