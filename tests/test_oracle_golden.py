"""
Pins BOTH CPU oracles (oracle/bn254_oracle.c via oracle/cref.py, and oracle/pyref.py) against the
golden vectors decoded from the reference's own fixtures (tests/golden/kat.json, produced by
tests/golden/make_golden.py; SURVEY.md Appendix A, KAT A1-A9).  CPU only.
"""
import numpy as np
import pytest

from oracle import cref, pyref

R, P = pyref.R_MOD, pyref.P_MOD


def L(x):  # list -> np limbs
    return np.array(x, dtype=np.uint64)


@pytest.mark.parametrize("which,k,root_pow", [("chunk_protocol", 25, 8), ("batch_proof", 26, 4)])
def test_A1_A2_domain_generators(kat, which, k, root_pow):
    pr = kat[which] if which == "chunk_protocol" else kat[which]["protocol"]
    dom = pr["domain"]
    assert dom["k"] == k and dom["n"] == 1 << k
    gen = pyref.from_mont(pyref.from_limbs(dom["gen"]), R)
    # python oracle: omega_k derived from the multiplicative generator 7
    assert gen == pyref.omega(k) == pow(pyref.FR_ROOT_OF_UNITY, root_pow, R)
    assert pow(gen, 1 << k, R) == 1 and pow(gen, 1 << (k - 1), R) != 1
    assert pyref.from_mont(pyref.from_limbs(dom["gen_inv"]), R) == pow(gen, -1, R)
    assert pyref.from_mont(pyref.from_limbs(dom["n_inv"]), R) == pow(1 << k, -1, R)
    # C oracle: same facts in Montgomery limbs, bit-exact
    root_m = cref.fr_mont(pyref.FR_ROOT_OF_UNITY)
    assert (cref.f_pow(cref.FR, root_m, root_pow) == L(dom["gen"])).all()
    one_m = cref.fr_mont(1)
    assert (cref.f_mul(cref.FR, L(dom["gen"]), L(dom["gen_inv"])) == one_m).all()
    assert (cref.f_inv(cref.FR, L(dom["gen"])) == L(dom["gen_inv"])).all()
    assert (cref.f_inv(cref.FR, cref.fr_mont(1 << k)) == L(dom["n_inv"])).all()
    assert (one_m == L(pyref.to_limbs(pyref.MONT_R % R))).all()


def _points(pr):
    return [(p["x"], p["y"]) for p in pr["preprocessed"]]


@pytest.mark.parametrize("which", ["chunk_protocol", "batch_proof", "chunk_proof"])
def test_A3_preprocessed_points_on_curve(kat, which):
    pr = kat[which] if which == "chunk_protocol" else kat[which]["protocol"]
    for x, y in _points(pr):
        assert pyref.g1_is_on_curve(pyref.g1_affine_from_limbs(x, y))
        assert cref.g1_is_on_curve(L(x + y))
    # and a corrupted point is rejected by both
    x, y = _points(pr)[0]
    bad = list(y); bad[0] ^= 1
    assert not pyref.g1_is_on_curve(pyref.g1_affine_from_limbs(x, bad))
    assert not cref.g1_is_on_curve(L(x + bad))


@pytest.mark.parametrize("vk,proto,k,npts", [("vk_chunk", "chunk_protocol", 25, 7), ("vk_batch_agg", "batch_proof", 26, 9)])
def test_A4_compressed_g1_codec(kat, vk, proto, k, npts):
    raw = bytes.fromhex(kat[vk])
    pr = kat[proto] if proto == "chunk_protocol" else kat[proto]["protocol"]
    assert int.from_bytes(raw[0:4], "big") == k and len(raw) == 8 + 32 * npts
    pts = _points(pr)
    for i in range(npts):
        word = raw[8 + 32 * i: 40 + 32 * i]
        want = pyref.g1_affine_from_limbs(*pts[i])
        assert pyref.g1_decompress(word) == want
        assert pyref.g1_compress(want) == word
        got = cref.g1_decompress(word)
        assert got is not None and (got == L(pts[i][0] + pts[i][1])).all()
        assert cref.g1_compress(L(pts[i][0] + pts[i][1])) == word
    # the vk embedded in the proof json equals the released file where both exist
    if vk == "vk_batch_agg":
        assert bytes.fromhex(kat["batch_proof"]["vk"]) == raw
    else:
        assert bytes.fromhex(kat["chunk_proof"]["vk"]) == raw


@pytest.mark.parametrize("which,n_commit,n_eval", [("chunk_proof", 9, 17), ("batch_proof", 12, 27)])
def test_A5_A6_proof_layout(kat, which, n_commit, n_eval):
    proof = bytes.fromhex(kat[which]["proof"])
    words = [proof[i:i + 32] for i in range(0, len(proof), 32)]
    assert len(words) == n_commit + n_eval + 2
    assert kat[which]["protocol"]["n_evaluations"] == n_eval
    nw = kat[which]["protocol"]["num_witness"]
    assert sum(nw) + kat[which]["protocol"]["quotient_num_chunk"] == n_commit
    for w in words[:n_commit] + words[-2:]:
        pt = pyref.g1_decompress(w)
        assert pt is not None and pyref.g1_is_on_curve(pt)
        c = cref.g1_decompress(w)
        assert c is not None and cref.g1_is_on_curve(c) and cref.g1_compress(c) == w
    for w in words[n_commit:n_commit + n_eval]:
        assert int.from_bytes(w, "little") < R


def test_A7_instances(kat):
    inst = bytes.fromhex(kat["chunk_proof"]["instances"])
    words = [int.from_bytes(inst[i:i + 32], "big") for i in range(0, len(inst), 32)]
    assert len(words) == 44 and all(w < (1 << 88) for w in words[:12]) and all(w < 256 for w in words[12:])


def test_A9_moduli(kat):
    assert int(kat["yul"]["f_p"], 16) == P and int(kat["yul"]["f_q"], 16) == R


def test_constants_rederived():
    # the constants hard-coded in the C oracle, re-derived from the moduli alone
    for w, m in ((cref.FQ, P), (cref.FR, R)):
        one = cref.f_from_canonical_vec(w, cref.int_to_limbs(1)[None])[0]
        assert cref.limbs_to_int(one) == (1 << 256) % m
        x = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % m
        xm = cref.f_from_canonical_vec(w, cref.int_to_limbs(x)[None])[0]
        assert cref.limbs_to_int(xm) == x * (1 << 256) % m
        assert cref.limbs_to_int(cref.f_to_canonical_vec(w, xm[None])[0]) == x
    assert pow(pyref.FR_ZETA, 3, R) == 1 and pyref.FR_ZETA != 1
    assert pyref.FR_ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pyref.omega(24) == 0x1951441010B2B95A6E47A6075066A50A036F5BA978C050F2821DF86636C0FACB
    assert pyref.omega(20) == 0x2A14464F1FF42DE3856402B62520E670745E39FADA049D5B2F0E1E3182673378


def test_g2_fixture_pins_the_twist_arithmetic(kat):
    """KAT A8 [REF release-v0.13.1/evm_verifier.yul:1230-1239]: the pairing inputs g2 and s_g2 of the production SRS.
    * the G2 generator constant of both oracles (and of the host mirror) equals the fixture words, in the (x.c1, x.c0, y.c1, y.c0) order
      of the EVM precompile;
    * both fixture points satisfy y^2 = x^3 + 3 / (9 + u) over Fq2 (pins Fq2 multiplication and the twist constant);
    * both are annihilated by r (pins the G2 addition / doubling formulas: a wrong formula leaves the order-r subgroup);
    * the C oracle (Jacobian) and the Python oracle (affine) agree on a scalar multiple."""
    import numpy as np
    import __graft_entry__ as ge
    g2w, s2w = kat["yul"]["g2_words"], kat["yul"]["s_g2_words"]
    gen_c = cref.g2_generator()
    assert (gen_c == cref.g2_from_words(g2w)).all()
    assert pyref.g2_from_evm_words(g2w) == pyref.G2_GEN
    assert list(gen_c) == pyref.g2_to_limbs(pyref.G2_GEN)
    h2 = ge.load_package().halo2
    assert (h2.g2_generator() == gen_c).all()                       # the product's constant (host mirror)
    s_c = cref.g2_from_words(s2w)
    s_py = pyref.g2_from_evm_words(s2w)
    for pt_c, pt_py in ((gen_c, pyref.G2_GEN), (s_c, s_py)):
        assert cref.g2_is_on_curve(pt_c) and pyref.g2_is_on_curve(pt_py)
        assert cref.g2_in_subgroup(pt_c)
    assert pyref.g2_mul(s_py, R) is None and pyref.g2_mul(pyref.G2_GEN, R) is None
    off = gen_c.copy(); off[0] ^= np.uint64(1)
    assert not cref.g2_is_on_curve(off)
    for tau in (1, 2, 0x5343524F4C4C0001, R - 1):
        assert list(cref.g2_mul(gen_c, cref.fr_mont(tau))) == pyref.g2_to_limbs(pyref.g2_mul(pyref.G2_GEN, tau))
    assert (cref.g2_mul(s_c, cref.fr_mont(0)) == 0).all()
