"""MSM and NTT known answers DERIVED FROM THE REFERENCE'S RELEASED PROOFS (tests/golden/released_kats.json, written by tests/golden/make_golden.py).

The reference holds no output vector of `best_multiexp` or `best_fft`.  Its released proofs hold something as good for a handful of inputs: the verifier's last step is ONE
multi-scalar multiplication over the proof's own points (verifying-key and proof commitments, the generator, the two SHPLONK points; 19 / 24 / 19 terms), with scalars the
transcript dictates, and the pairing equation  e(result, G2) e(W', -[s]G2) == 1  holds only for the right group element -- so (scalars, points) -> result is an MSM instance whose
answer the reference's data certifies.  Likewise the verifier needs the instance polynomial's value at the challenge x, and the proof verifies only with the right one: an inverse
transform of the instance column over the proof's own domain (2^25 / 2^26 rows), evaluated at x, must reproduce it.

  CPU    the C restatement (oracle/bn254_oracle.c: best_multiexp, multiexp_serial, the naive sum; ifft + eval_polynomial at 2^25) reproduces both; the stored results satisfy the
         pairing equation under oracle/pairing.py (so the fixture cannot have drifted from what the released verifier accepts)
  GPU    the product, through the C-ABI: mi355_msm_g1_adhoc_host on the same terms, a registered basis (mi355_msm_g1_host) with and without window tables; the instance column
         through mi355_intt_fr_dev at the proofs' full domain sizes and mi355_eval_polynomial_dev at x
"""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import cref, pairing, pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
KATS = {k: v for k, v in json.load(open(os.path.join(GOLD, "released_kats.json"))).items() if not k.startswith("_")}
YUL = json.load(open(os.path.join(GOLD, "kat.json")))["yul"]
NEG_S_G2 = pyref.g2_from_evm_words([int(w, 16) for w in YUL["s_g2_words"]])
NAMES = sorted(KATS)


def msm_inputs(name):
    m = KATS[name]["msm"]
    scalars = np.stack([cref.fr_mont(int(s, 16)) for s in m["scalars"]])
    points = np.stack([np.array(pyref.g1_affine_to_limbs((int(x, 16), int(y, 16))), dtype=np.uint64).reshape(8) for x, y in m["points"]])
    want = np.array(pyref.g1_affine_to_limbs((int(m["result"][0], 16), int(m["result"][1], 16))), dtype=np.uint64).reshape(8)
    return scalars, points, want


def instance_column(name):
    e = KATS[name]["instance_eval"]
    k = KATS[name]["k"]
    col = np.zeros((1 << k, 4), dtype=np.uint64)
    col[:len(e["instances"])] = np.stack([cref.fr_mont(int(v, 16)) for v in e["instances"]])
    return k, col, cref.fr_mont(int(e["x"], 16)), cref.fr_mont(int(e["value"], 16))


def test_fixture_shape():
    assert NAMES == ["batch_proof", "bundle_proof", "chunk_proof"]
    assert [len(KATS[n]["msm"]["scalars"]) for n in NAMES] == [24, 19, 19] and [KATS[n]["k"] for n in NAMES] == [26, 26, 25]
    assert [len(KATS[n]["instance_eval"]["instances"]) for n in NAMES] == [23, 25, 44]


@pytest.mark.parametrize("name", NAMES)
def test_stored_results_satisfy_the_released_verifiers_pairing(name):
    m = KATS[name]["msm"]
    res = (int(m["result"][0], 16), int(m["result"][1], 16)); wp = (int(m["w_prime"][0], 16), int(m["w_prime"][1], 16))
    assert pyref.g1_is_on_curve(res) and pyref.g1_is_on_curve(wp)
    assert pairing.pairing_product_is_one([(res, pyref.G2_GEN), (wp, NEG_S_G2)])
    assert not pairing.pairing_product_is_one([(pyref.g1_add(res, pyref.G1_GEN), pyref.G2_GEN), (wp, NEG_S_G2)])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_multiexp_reproduces_the_certified_results(name):
    """oracle/bn254_oracle.c on an MSM whose answer comes from the reference's data: the Pippenger restatement (threads 1, 3, 16), the serial one, the naive sum, and the big-integer
    oracle -- all equal the stored result"""
    scalars, points, want = msm_inputs(name)
    for got in (cref.best_multiexp(scalars, points, 1), cref.best_multiexp(scalars, points, 3), cref.best_multiexp(scalars, points, 16), cref.multiexp_serial(scalars, points),
                cref.msm_naive(scalars, points)):
        assert (cref.g1_to_affine(got).reshape(8) == want).all()
    m = KATS[name]["msm"]
    acc = None
    for s, (x, y) in zip(m["scalars"], m["points"]):
        acc = pyref.g1_add(acc, pyref.g1_mul((int(x, 16), int(y, 16)), int(s, 16)))
    assert acc == (int(m["result"][0], 16), int(m["result"][1], 16))
    # one scalar off by one: another group element
    bad = scalars.copy(); bad[3] = cref.fr_mont((int(m["scalars"][3], 16) + 1) % pyref.R_MOD)
    assert not (cref.g1_to_affine(cref.best_multiexp(bad, points, 4)).reshape(8) == want).all()


def test_oracle_inverse_transform_reproduces_the_instance_value_of_the_chunk_proof():
    """the instance column of the released chunk proof (44 public inputs in 2^25 rows) through the C restatement of EvaluationDomain::ifft, then eval_polynomial at the proof's
    challenge x: the value the verifier used -- and without which the proof does not verify.  Pins best_fft's output order, the inverse root and the divisor at the proof's size."""
    k, col, x, want = instance_column("chunk_proof")
    w_inv = cref.fr_mont(pow(pyref.omega(k), -1, pyref.R_MOD)); n_inv = cref.fr_mont(pow(1 << k, -1, pyref.R_MOD))
    coeffs = cref.ifft(col, w_inv, k, n_inv)
    assert (cref.eval_polynomial_mt(coeffs, x).reshape(4) == want).all()
    assert not (cref.eval_polynomial_mt(coeffs, cref.fr_mont(5)).reshape(4) == want).all()


# ------------------------------------------------------------------------------------------------ the product
@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    yield pkg


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_multiexp_reproduces_the_certified_results(zk, name):
    from tests.gpu_common import affine_of
    h2 = zk.halo2
    scalars, points, want = msm_inputs(name)
    assert (affine_of(h2.best_multiexp(scalars, points)) == want).all()
    import torch
    n = scalars.shape[0]
    gen = np.array(pyref.g1_affine_to_limbs(pyref.G1_GEN), dtype=np.uint64).reshape(1, 8)
    basis = np.concatenate([points, np.repeat(gen, 32 - n, axis=0)])              # a registered 2^5 basis: the terms, then generators that meet zero scalars
    padded = np.concatenate([scalars, np.zeros((32 - n, 4), np.uint64)])
    params = h2.ParamsKZG.from_host(5, basis, basis)
    assert (affine_of(params.commit(padded)) == want).all()                       # the commit path (mi355_msm_g1_host on a registered basis), table-free
    assert (affine_of(h2.best_multiexp(scalars, params.g_slice(0, n))) == want).all()   # a prefix of the basis
    assert (affine_of(params.commit(torch.from_numpy(padded.view(np.int64)).cuda())) == want).all()   # scalars resident (mi355_msm_g1_dev)
    params.precompute()
    assert (affine_of(params.commit(padded)) == want).all()                       # and through window tables
    params.release()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_inverse_transform_reproduces_the_instance_value(zk, name):
    """at the proofs' own domain sizes (2^25 for the chunk proof, 2^26 for the batch and bundle proofs): instance column -> mi355_intt_fr_dev -> mi355_eval_polynomial_dev at x"""
    h2 = zk.halo2
    k, col, x, want = instance_column(name)
    dom = h2.EvaluationDomain(2, k)
    buf = h2.DeviceBuffer.from_host(col)
    dom.lagrange_to_coeff(buf)
    assert (np.asarray(h2.eval_polynomial(buf, x)).reshape(4) == want).all()
    buf.free()
