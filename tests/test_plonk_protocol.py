"""create_proof driven by the reference's own constraint systems.

The reference ships the PlonkProtocol of its chunk proof (layer 2, k = 25: [REF release-v0.13.1/chunk.protocol]) and of its batch proof (layer 4, k = 26:
the base64 `protocol` of [REF integration/tests/test_data/full_proof_batch_agg_1.json]); tests/golden/protocol_layer{2,4}.json are those files
(tests/golden/make_golden.py).  Here:
  CPU   the halo2-base rule (scroll-prover_amd/protocols.py) reproduces both fixtures node for node; DELTA and the proof word counts are pinned by them; the
        C++ host side (protocol reader + recogniser, circuit builder, Blake2b transcript) agrees with the Python restatement; the CPU prover's proofs have the
        reference's sizes (896 B / 1 312 B) and verify; tampered proofs and witnesses do not
  GPU   the HIP prover (include/mi355zk_plonk.hpp through the C-ABI) emits proof bytes IDENTICAL to the CPU restatement of halo2's create_proof on the same
        circuit instance and randomness, for every layer, with resident and recomputed proving-key cosets and on several device slots; at full size
        (k = 25 / 26, the fixtures' own files) the verifier -- which walks the JSON expression tree itself -- accepts what the device produced
"""
import json
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import plonk, pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
zk = ge.load_package()
protocols = zk.protocols
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
TAU0 = 0x5343524F4C4C0001


def fixture(layer):
    return json.load(open(os.path.join(GOLD, f"protocol_layer{layer}.json")))["protocol"]


def exe():
    ge.build()
    return ge.build_cpp("test_plonk_replay")


# ------------------------------------------------------------------------------------------------ CPU: the protocols
@pytest.mark.parametrize("layer", [2, 4])
def test_halo2_base_rule_reproduces_the_reference_protocols(layer):
    g, f = protocols.layer_protocol(layer), fixture(layer)
    for key in ("domain", "num_instance", "num_witness", "num_challenge", "evaluations", "queries", "quotient"):
        assert g[key] == f[key], key
    assert g["num_preprocessed"] == len(f["preprocessed"])
    assert f["domain"] == (KAT["chunk_protocol"] if layer == 2 else KAT["batch_proof"]["protocol"])["domain"]


def test_layer_configs_are_the_reference_files():
    cfg = json.load(open(os.path.join(GOLD, "layer_configs.json")))
    for layer, c in protocols.LAYER_CONFIGS.items():
        for key, v in c.items():
            assert cfg[str(layer)][key] == v, (layer, key)


def test_delta_is_pinned_by_the_fixtures():
    """the constants next to `Identity` in the permutation argument are 1, DELTA, DELTA^2, ...: halo2curves' Fr::DELTA = 7^(2^28), of order (r - 1) / 2^28"""
    d = protocols.FR_DELTA
    assert pow(d, (pyref.R_MOD - 1) >> 28, pyref.R_MOD) == 1 and d != 1
    for layer, ncols in ((2, 3), (4, 5)):
        pr = plonk.Protocol(fixture(layer))
        got = [c[2] for ch in pr.perm for c in ch["columns"]]
        assert got == [pow(d, j, pyref.R_MOD) for j in range(ncols)]


def test_proof_word_counts_match_the_released_proofs():
    """commitments | evaluations | 2 SHPLONK points: 28 words (896 B) for the chunk proof, 41 (1 312 B) for the batch proof; the bundle's EVM
    proof = 12 accumulator words + uncompressed points + evaluations = 51 words [REF release-v0.13.1/proof.data]"""
    for layer, nbytes in ((2, len(bytes.fromhex(KAT["chunk_proof"]["proof"]))), (4, len(bytes.fromhex(KAT["batch_proof"]["proof"])))):
        pr = plonk.Protocol(fixture(layer))
        assert 32 * (sum(pr.num_witness) + pr.Q + len(pr.evaluations) + 2) == nbytes
    p6 = plonk.Protocol(protocols.layer_protocol(6))
    assert 12 + 2 * (sum(p6.num_witness) + p6.Q + 2) + len(p6.evaluations) == len(bytes.fromhex(KAT["bundle_proof_data"])) // 32


def test_released_proofs_parse_under_their_protocols():
    """the released proofs, read with THEIR protocols by the verifier's reader: every commitment word decompresses on the curve, every evaluation is canonical, nothing is left over
    (test_reference_released_proofs_verify below goes all the way)"""
    for layer, blob in ((2, KAT["chunk_proof"]["proof"]), (4, KAT["batch_proof"]["proof"])):
        pr = plonk.Protocol(fixture(layer)); proof = bytes.fromhex(blob)
        T = plonk.Transcript(proof)
        for _ in range(sum(pr.num_witness) + pr.Q):
            T.read_point()
        for _ in pr.evaluations:
            T.read_scalar()
        T.read_point(); T.read_point()
        assert T.pos == len(proof)


def released(name):
    blob = bytes.fromhex(KAT[name]["instances"])
    return [int.from_bytes(blob[i:i + 32], "big") for i in range(0, len(blob), 32)], bytes.fromhex(KAT[name]["proof"])


NEG_S_G2 = pyref.g2_from_evm_words([int(w, 16) for w in KAT["yul"]["s_g2_words"]])      # the second G2 point of the released EVM verifier's pairing call: -[s]G2 of the reference's SRS


def test_poseidon_constants_reproduce_a_published_vector():
    """the Grain LFSR, the rejection sampling, the Cauchy matrix and the round schedule of oracle/poseidon.py at T = 3, R_F = 8, R_P = 57: circomlib's poseidon([1, 2])"""
    from oracle import poseidon
    assert poseidon.hash_circomlib([1, 2]) == 0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A
    rc, mds = poseidon.parameters(5, 8, 60)
    assert len(rc) == 68 * 5 and len(set(rc)) == len(rc) and all(len(row) == 5 for row in mds)


@pytest.mark.parametrize("name", ["chunk_proof", "batch_proof"])
def test_released_accumulators_satisfy_the_pairing(name):
    """the first 12 instances of a released proof are the KZG accumulator the proof carries forward (two G1 points, 3 limbs of 88 bits per coordinate); with the G2 words of the
    released EVM verifier, e(lhs, G2) e(rhs, -[s]G2) == 1: pins oracle/pairing.py and the reading of those words on the reference's data"""
    from oracle import pairing
    inst, _ = released(name)
    c = [inst[3 * i] + (inst[3 * i + 1] << 88) + (inst[3 * i + 2] << 176) for i in range(4)]
    lhs, rhs = (c[0], c[1]), (c[2], c[3])
    assert all(v < (1 << 88) for v in inst[:12]) and pyref.g1_is_on_curve(lhs) and pyref.g1_is_on_curve(rhs)
    assert pyref.g2_is_on_curve(NEG_S_G2)
    assert pairing.pairing_product_is_one([(lhs, pyref.G2_GEN), (rhs, NEG_S_G2)])
    assert not pairing.pairing_product_is_one([(lhs, pyref.G2_GEN), (pyref.g1_neg(rhs), NEG_S_G2)])
    assert not pairing.pairing_product_is_one([(pyref.g1_add(lhs, pyref.G1_GEN), pyref.G2_GEN), (rhs, NEG_S_G2)])


@pytest.mark.parametrize("name,layer", [("chunk_proof", 2), ("batch_proof", 4)])
def test_reference_released_proofs_verify(name, layer):
    """THE PIN OF THE RESTATEMENT: the proofs the reference released [REF integration/tests/test_data/full_proof_1.json chunk_proofs[0]; full_proof_batch_agg_1.json], made by the
    real prover on the production SRS, are ACCEPTED by oracle/plonk.py's verifier -- Poseidon transcript (oracle/poseidon.py), the protocol file's own preprocessed commitments and
    initial transcript scalar, the numerator tree evaluated at x, instance polynomial from the public values, SHPLONK over the rotation sets with the i-th set at v^i and its j-th
    polynomial at y^j, and the pairing against the released verifier's -[s]G2.  One wrong constant, word order, rotation set or power breaks the pairing equation; so do a
    flipped proof word and a changed instance."""
    pr = plonk.Protocol(fixture(layer))
    inst, proof = released(name)
    res = plonk.verify(pr, None, inst, proof, transcript="poseidon", neg_s_g2=NEG_S_G2)
    assert res["ok"] and res["pairing"], res
    nc, ne = sum(pr.num_witness) + pr.Q, len(pr.evaluations)
    for word in (1, nc - 1, nc, nc + ne - 1, nc + ne, nc + ne + 1):
        bad = bytearray(proof); bad[32 * word + 2] ^= 1
        try:
            ok = plonk.verify(pr, None, inst, bytes(bad), transcript="poseidon", neg_s_g2=NEG_S_G2)["ok"]
        except AssertionError:
            ok = False
        assert not ok, f"the released proof with word {word} altered was accepted"
    for i in (0, 12, len(inst) - 1):
        bad_inst = list(inst); bad_inst[i] ^= 1
        assert not plonk.verify(pr, None, bad_inst, proof, transcript="poseidon", neg_s_g2=NEG_S_G2)["ok"]
    assert not plonk.verify(pr, None, inst, proof, transcript="blake2b", neg_s_g2=NEG_S_G2)["ok"]          # and it is the Poseidon transcript that makes it so


def test_more_stored_proofs_verify():
    """six more of the 318 chunk proofs the reference stores as inputs of its batch tests (the first of six different files; tests/golden/make_golden.py --verify-all ran ALL 318
    through the same verifier when the fixtures were made: tests/golden/released_proofs_verified.json), and the second stored batch proof"""
    pr = plonk.Protocol(fixture(2))
    assert len(KAT["more_chunk_proofs"]) == 6
    for m in KAT["more_chunk_proofs"]:
        ib, proof = bytes.fromhex(m["instances"]), bytes.fromhex(m["proof"])
        assert proof != bytes.fromhex(KAT["chunk_proof"]["proof"])
        assert plonk.verify(pr, None, [int.from_bytes(ib[i:i + 32], "big") for i in range(0, len(ib), 32)], proof, transcript="poseidon", neg_s_g2=NEG_S_G2)["ok"], m["source"]
    ib, proof = bytes.fromhex(KAT["batch_proof_2"]["instances"]), bytes.fromhex(KAT["batch_proof_2"]["proof"])
    assert plonk.verify(plonk.Protocol(fixture(4)), None, [int.from_bytes(ib[i:i + 32], "big") for i in range(0, len(ib), 32)], proof, transcript="poseidon", neg_s_g2=NEG_S_G2)["ok"]
    rec = json.load(open(os.path.join(GOLD, "released_proofs_verified.json")))
    assert rec["stored_chunk_proofs"] == rec["accepted_by_oracle_plonk_verify"] == 318


def test_vkey_files_hold_the_protocols_preprocessed_commitments_in_order():
    """[REF release-v0.13.1/vk_chunk.vkey] = u32 k | u32 fixed columns | the SAME seven commitments, in the same order, as `preprocessed` of [REF release-v0.13.1/chunk.protocol]:
    the .vkey layout our keygen writes (fixed columns, then the permutation's sigma columns) is the reference's"""
    vk = bytes.fromhex(KAT["vk_chunk"])
    pts = [pyref.g1_decompress(vk[8 + 32 * i:8 + 32 * i + 32]) for i in range(7)]
    assert pts == [(plonk.fq_limbs_mont_to_int(p["x"]), plonk.fq_limbs_mont_to_int(p["y"])) for p in fixture(2)["preprocessed"]]
    assert int.from_bytes(vk[:4], "big") == 25 and len(vk) == 8 + 7 * 32
    # and the batch proof's key [REF integration/tests/test_data/vk_batch_agg.vkey] (== the `vk` stored with full_proof_batch_agg_1.json) against the layer-4 protocol: nine commitments, same order
    vk4 = bytes.fromhex(KAT["vk_batch_agg"])
    assert vk4.hex() == KAT["batch_proof"]["vk"] and int.from_bytes(vk4[:4], "big") == 26 and len(vk4) == 8 + 9 * 32
    assert [pyref.g1_decompress(vk4[8 + 32 * i:8 + 32 * i + 32]) for i in range(9)] == [(plonk.fq_limbs_mont_to_int(p["x"]), plonk.fq_limbs_mont_to_int(p["y"])) for p in fixture(4)["preprocessed"]]


def bundle_inputs():
    pd, pi = bytes.fromhex(KAT["bundle_proof_data"]), bytes.fromhex(KAT["bundle_pi_data"])
    words = lambda b: [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]
    vk = bytes.fromhex(KAT["vk_bundle"])
    return words(pd[:384]) + words(pi), pd[384:], [pyref.g1_decompress(vk[8 + 32 * i:8 + 32 * i + 32]) for i in range(7)]


def test_released_bundle_evm_proof_verifies():
    """Layer 6 has NO protocol fixture: its constraint system here is GENERATED by the halo2-base rule from [REF integration/configs/layer6.config] (scroll-prover_amd/protocols.py).
    The released bundle proof [REF release-v0.13.1/proof.data, pi.data] verifies under it: calldata = 12 accumulator limbs | 13 public-input words | the proof proper (points
    uncompressed), Keccak transcript (oracle/keccak.py, EvmTranscript) started from the released verifier's own first word [REF release-v0.13.1/evm_verifier.yul:66], preprocessed
    commitments = [REF release-v0.13.1/vk_bundle.vkey] in file order, pairing against the verifier's -[s]G2.  Pins the generator on a layer it was not fitted to, the Keccak
    transcript, and the .vkey layout."""
    inst, proof, pre = bundle_inputs()
    assert len(inst) == protocols.LAYER_NUM_INSTANCE[6] == 25 and len(proof) == 11 * 64 + 17 * 32
    pr = plonk.Protocol(protocols.layer_protocol(6))
    kw = dict(transcript="evm", neg_s_g2=NEG_S_G2, preprocessed=pre, initial_state=int(KAT["yul"]["transcript_initial_state"]))
    assert plonk.verify(pr, None, inst, proof, **kw)["ok"]
    bad = bytearray(proof); bad[64 * 9 + 5] ^= 1                                  # an evaluation word
    assert not plonk.verify(pr, None, inst, bytes(bad), **kw)["ok"]
    assert not plonk.verify(pr, None, inst[:12] + [inst[12] ^ 1] + inst[13:], proof, **kw)["ok"]
    assert not plonk.verify(pr, None, inst, proof, **dict(kw, preprocessed=pre[1:] + pre[:1]))["ok"]
    # its accumulator obeys the pairing too, and the first public-input word is the digest the release publishes
    from oracle import pairing
    c = [inst[3 * i] + (inst[3 * i + 1] << 88) + (inst[3 * i + 2] << 176) for i in range(4)]
    assert pairing.pairing_product_is_one([((c[0], c[1]), pyref.G2_GEN), ((c[2], c[3]), NEG_S_G2)])


def test_pairing_is_bilinear_and_non_degenerate():
    """oracle/pairing.py on its own: e(aP, bQ) == e(P, Q)^(ab) == e(abP, Q) == e(P, abQ), e(P, Q) != 1 of order r, e(P, Q) e(-P, Q) == 1, and the product form with several pairs"""
    from oracle import pairing
    G1, G2 = pyref.G1_GEN, pyref.G2_GEN
    a, b = 0x1234567890ABCDEF1234567, 0xFEDCBA0987654321
    e = pairing.pairing(G2, G1)
    assert not (e == pairing.F12.one()) and (e ** pyref.R_MOD) == pairing.F12.one()
    lhs = pairing.pairing(pyref.g2_mul(G2, b), pyref.g1_mul(G1, a))
    assert lhs == e ** (a * b % pyref.R_MOD) == pairing.pairing(G2, pyref.g1_mul(G1, a * b % pyref.R_MOD)) == pairing.pairing(pyref.g2_mul(G2, a * b % pyref.R_MOD), G1)
    assert pairing.pairing_product_is_one([(G1, G2), (pyref.g1_neg(G1), G2)])
    assert pairing.pairing_product_is_one([(pyref.g1_mul(G1, a), pyref.g2_mul(G2, b)), (pyref.g1_mul(G1, 3), pyref.g2_mul(G2, 5)), (pyref.g1_neg(pyref.g1_mul(G1, (a * b + 15) % pyref.R_MOD)), G2)])
    assert pairing.miller_loop(None, G1) == pairing.F12.one() and pairing.miller_loop(G2, None) == pairing.F12.one()
    x = pairing.F12([3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8])
    assert x * x.inv() == pairing.F12.one()


def test_keccak256_vectors():
    from oracle import keccak
    assert keccak.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert keccak.keccak256(bytes(136)).hex() != keccak.keccak256(bytes(135)).hex()


def test_released_bundle_evm_proof_parses_under_the_generated_layer6_protocol():
    """layer 6 has no protocol fixture; the halo2-base rule gives it layer 2's system at k = 26.  The released bundle proof [REF release-v0.13.1/proof.data] (EVM layout:
    32-byte big-endian words, points uncompressed) agrees word for word with that shape: 12 accumulator limbs below 2^88, then exactly num_witness + Q = 9 point pairs ON THE
    CURVE, 17 canonical scalars, and the two SHPLONK points -- a shape check of the generated protocol against a proof the reference really emitted"""
    p6 = plonk.Protocol(protocols.layer_protocol(6))
    b = bytes.fromhex(KAT["bundle_proof_data"])
    w = [int.from_bytes(b[32 * i:32 * i + 32], "big") for i in range(len(b) // 32)]
    on_curve = lambda x, y: x < pyref.P_MOD and y < pyref.P_MOD and (y * y - x * x * x - 3) % pyref.P_MOD == 0
    nc, ne = sum(p6.num_witness) + p6.Q, len(p6.evaluations)
    assert len(w) == 12 + 2 * nc + ne + 4 and all(v < (1 << 88) for v in w[:12])
    assert all(on_curve(w[12 + 2 * i], w[13 + 2 * i]) for i in range(nc))
    assert all(v < pyref.R_MOD for v in w[12 + 2 * nc:12 + 2 * nc + ne])
    assert not any(on_curve(w[i], w[i + 1]) for i in range(12 + 2 * nc, 12 + 2 * nc + ne - 1))        # no point hides among the evaluations: the split is where the protocol says
    assert on_curve(w[-4], w[-3]) and on_curve(w[-2], w[-1])


def test_recognised_structure_of_the_fixtures():
    p2, p4 = plonk.Protocol(fixture(2)), plonk.Protocol(fixture(4))
    assert (p2.last, p2.blind, p2.Q, len(p2.gates), len(p2.perm), len(p2.lookups)) == (-7, 6, 4, 1, 1, 1)
    assert (p4.last, p4.blind, p4.Q, len(p4.gates), len(p4.perm), len(p4.lookups)) == (-7, 6, 4, 2, 2, 1)
    assert [len(c["columns"]) for c in p4.perm] == [3, 2]                                       # chunks of degree - 2 = 3
    rs = lambda p: [(sorted(s["rots"]), len(s["polys"])) for s in plonk.rotation_sets(p.queries)]
    assert rs(p2) == [([0, 1, 2, 3], 1), ([0, 1], 2), ([0], 10)]
    assert rs(p4) == [([0, 1, 2, 3], 2), ([0], 13), ([-7, 0, 1], 1), ([0, 1], 2)]                # the z(w^-7 X) link opens z_0 at a third point


def test_expression_evaluator():
    P = protocols
    leaf = dict(poly=lambda p, r: 10 * p + r + 1, challenge=lambda j: 100 + j, identity=lambda: 7, lagrange=lambda i: 1000 + i)
    e = P.DP([P.Poly(1), P.Prod(P.Poly(2, 1), P.Ch(0)), P.Neg(P.Const(5))], P.Ch(3))
    assert plonk.evaluate(e, **leaf) == ((11 * 103 + 22 * 100) * 103 - 5) % pyref.R_MOD
    assert plonk.evaluate(P.Sum(P.IDENTITY, P.Lag(-7)), **leaf) == 7 + 993


# ------------------------------------------------------------------------------------------------ CPU: host side of the prover vs the restatement
def test_cpp_transcript_equals_hashlib_blake2b():
    out = subprocess.run([exe(), "--transcript-selftest"], capture_output=True, text=True, timeout=60)
    got = json.loads(out.stdout)
    T = plonk.Transcript(); T.common_scalar(5)
    assert "%064x" % T.squeeze() == got["c1"]
    T.write_point((1, 2)); T.write_scalar(0xDEADBEEF)
    assert "%064x" % T.squeeze() == got["c2"] and T.out.hex() == got["proof"]
    assert "%064x" % plonk.vk_transcript_repr(bytes([7] * 40)) == got["vk_repr"]


def test_cpp_poseidon_transcript_equals_the_restatement_that_verifies_the_released_proofs():
    """the product's Poseidon transcript (include/mi355zk_transcript.hpp: Grain constants generated in C++, the sponge, points as (x mod r, y mod r)) against oracle/poseidon.py,
    which the reference's released proofs pin: first / last round constant, two matrix entries, the same stream as the Blake2b test, and a squeeze after every buffer length 0 .. 9
    on one running sponge (short chunk + 1, exact multiple of RATE + empty permutation, nothing buffered)"""
    from oracle import poseidon
    got = json.loads(subprocess.run([exe(), "--transcript-selftest"], capture_output=True, text=True, timeout=60).stdout)["poseidon"]
    rc, mds = poseidon.parameters(5, 8, 60)
    assert ["%064x" % v for v in (rc[0], rc[-1], mds[0][0], mds[4][4])] == [got["rc0"], got["rc_last"], got["mds00"], got["mds44"]]
    T = plonk.PoseidonTranscript(); T.common_scalar(5)
    assert "%064x" % T.squeeze() == got["c1"]
    T.write_point((1, 2)); T.write_scalar(0xDEADBEEF)
    assert "%064x" % T.squeeze() == got["c2"] and got["proof_equal"] is True
    L = plonk.PoseidonTranscript()
    for ln in range(10):
        for i in range(ln):
            L.common_scalar(1000 * ln + i)
        assert "%064x" % L.squeeze() == got["by_length"][ln], ln
    # layer 6's Keccak transcript in the EVM layout, same stream; Keccak-256 itself on 200 bytes (two blocks)
    from oracle import keccak
    ev = json.loads(subprocess.run([exe(), "--transcript-selftest"], capture_output=True, text=True, timeout=60).stdout)["evm"]
    E = plonk.EvmTranscript(); E.common_scalar(5)
    assert "%064x" % E.squeeze() == ev["c1"] and "%064x" % E.squeeze() == ev["c1b"]
    E.write_point((1, 2)); E.write_scalar(0xDEADBEEF)
    assert "%064x" % E.squeeze() == ev["c2"] and E.out.hex() == ev["proof"] and len(E.out) == 96
    assert keccak.keccak256(bytes((i * 7 + 1) & 255 for i in range(200))).hex() == ev["keccak_200"]
    # a coordinate above r is absorbed reduced: (x mod r, y mod r)
    big = plonk.PoseidonTranscript(); big.common_point((pyref.R_MOD + 5, 7)); ref = plonk.PoseidonTranscript(); ref.common_scalar(5); ref.common_scalar(7)
    assert big.squeeze() == ref.squeeze()


def build_instance(tmp_path, layer, k, **shape):
    d = str(tmp_path / f"l{layer}"); os.makedirs(d, exist_ok=True)
    proto = protocols.write(layer, os.path.join(d, "p.json"), k, **shape)
    out = subprocess.run([exe(), "--protocol", proto, "--out", d, "--builder-only", "--threads", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return plonk.ProofInputs.load(d)


SMALL = [(2, 6, {}), (4, 7, {}), (6, 6, {}), (5, 7, {}), (1, 7, {}), (0, 7, dict(advice=40, fixed=8, lookups=3, perm_columns=12, degree=5)), (0, 6, dict(advice=36, fixed=7, lookups=9, perm_columns=20, degree=9))]


@pytest.mark.parametrize("layer,k,shape", SMALL)
def test_cpu_prove_and_verify(tmp_path, layer, k, shape):
    """the C++ builder's circuit instance satisfies the protocol (the CPU prover asserts: grand products return to 1, running sums to 0, the quotient fits Q pieces),
    the proof has the layout's size and verifies; flipping any one word of it does not"""
    inp, man = build_instance(tmp_path, layer, k, **shape)
    pr = inp.pr
    vk = plonk.keygen_vk(pr, inp.pre, inp.tau)
    proof = plonk.prove(inp, vk)
    assert len(proof) == 32 * (sum(pr.num_witness) + pr.Q + len(pr.evaluations) + 2)
    if layer in (2, 6):
        assert len(proof) == 896 and len(vk) == 232
    if layer == 4:
        assert len(proof) == 1312 and len(vk) == 296
    assert plonk.verify(pr, vk, inp.instances, proof, inp.tau)["ok"]
    nc = sum(pr.num_witness) + pr.Q
    for word in (0, nc - 1, nc, nc + len(pr.evaluations) - 1, nc + len(pr.evaluations), nc + len(pr.evaluations) + 1):
        bad = bytearray(proof); bad[32 * word + 1] ^= 4
        try:
            ok = plonk.verify(pr, vk, inp.instances, bytes(bad), inp.tau)["ok"]
        except AssertionError:   # not a curve point / not canonical
            ok = False
        assert not ok, f"a proof with word {word} altered was accepted"
    wrong_inst = list(inp.instances); wrong_inst[0] = (wrong_inst[0] + 1) % pyref.R_MOD
    assert not plonk.verify(pr, vk, wrong_inst, proof, inp.tau)["ok"]
    if layer == 2:            # the trapdoor check and the pairing check are the same statement: with -[tau]G2 of this synthetic SRS the pairing accepts the proof, and not a tampered one
        sg2 = pyref.g2_mul(pyref.G2_GEN, inp.tau)
        neg = (sg2[0], tuple((-c) % pyref.P_MOD for c in sg2[1]))
        assert plonk.verify(pr, vk, inp.instances, proof, None, neg_s_g2=neg)["pairing"]
        bad = bytearray(proof); bad[32 * nc + 7] ^= 1
        assert not plonk.verify(pr, vk, inp.instances, bytes(bad), None, neg_s_g2=neg)["ok"]
    if layer == 6:            # layer 6 in the EVM layout: points uncompressed, big-endian words, Keccak challenges
        pe = plonk.prove(inp, vk, transcript="evm")
        assert len(pe) == 11 * 64 + 17 * 32 and plonk.verify(pr, vk, inp.instances, pe, inp.tau, transcript="evm")["ok"]
        bad = bytearray(pe); bad[64 * 9 + 40] ^= 2
        assert not plonk.verify(pr, vk, inp.instances, bytes(bad), inp.tau, transcript="evm")["ok"]
    if layer in (2, 4):       # the same circuit under the transcript the reference proves these layers with
        pp = plonk.prove(inp, vk, transcript="poseidon")
        assert len(pp) == len(proof) and pp != proof and pp[:32 * pr.num_witness[0]] == proof[:32 * pr.num_witness[0]]
        assert plonk.verify(pr, vk, inp.instances, pp, inp.tau, transcript="poseidon")["ok"] and not plonk.verify(pr, vk, inp.instances, pp, inp.tau)["ok"]


REF_INITIAL_STATE = json.load(open(os.path.join(GOLD, "protocol_layer2.json")))["protocol"]["transcript_initial_state"]   # [REF release-v0.13.1/chunk.protocol]


def with_initial_state(tmp_path, layer, k):
    """a generated protocol that CARRIES a transcript_initial_state, as the reference's protocol files do"""
    d = str(tmp_path / "stated"); os.makedirs(d, exist_ok=True)
    proto = protocols.layer_protocol(layer, k)
    assert not proto.get("transcript_initial_state")
    proto["transcript_initial_state"] = REF_INITIAL_STATE
    path = os.path.join(d, "p.json")
    json.dump(proto, open(path, "w"), separators=(",", ":"))
    return d, path


def test_a_protocol_s_own_initial_state_keys_the_transcript(tmp_path):
    """a verifier built from a PlonkProtocol absorbs the file's `transcript_initial_state` (snark-verifier never hashes key bytes), so prover and verifier
    start from it whenever the file carries one; the .vkey hash is the convention for generated protocols only.  Same circuit, same randomness: the proofs agree up to the first
    challenge (the advice commitments) and differ afterwards; each verifies under its own protocol only -- also by the file-only route the released proofs take (vk_bytes = None)."""
    assert REF_INITIAL_STATE and len(REF_INITIAL_STATE) == 4
    d, path = with_initial_state(tmp_path, 2, 6)
    out = subprocess.run([exe(), "--protocol", path, "--out", d, "--builder-only", "--threads", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    inp, _ = plonk.ProofInputs.load(d)
    stated = inp.pr
    plain = plonk.Protocol({k_: v for k_, v in stated.d.items() if k_ != "transcript_initial_state"})
    vk = plonk.keygen_vk(stated, inp.pre, inp.tau)
    assert plonk.vk_transcript_scalar(stated, vk) == plonk.limbs_mont_to_int(REF_INITIAL_STATE) != plonk.vk_transcript_repr(vk) == plonk.vk_transcript_scalar(plain, vk)
    p_stated = plonk.prove(inp, vk, transcript="poseidon")
    inp.pr = plain
    p_plain = plonk.prove(inp, vk, transcript="poseidon")
    na = 32 * stated.num_witness[0]
    assert p_stated[:na] == p_plain[:na] and p_stated[na:] != p_plain[na:]
    assert plonk.verify(stated, vk, inp.instances, p_stated, inp.tau, transcript="poseidon")["ok"] and plonk.verify(plain, vk, inp.instances, p_plain, inp.tau, transcript="poseidon")["ok"]
    assert not plonk.verify(plain, vk, inp.instances, p_stated, inp.tau, transcript="poseidon")["ok"] and not plonk.verify(stated, vk, inp.instances, p_plain, inp.tau, transcript="poseidon")["ok"]
    pre = [pyref.g1_decompress(vk[8 + 32 * i:8 + 32 * i + 32]) for i in range(stated.num_pre)]
    assert plonk.verify(stated, None, inp.instances, p_stated, inp.tau, transcript="poseidon", preprocessed=pre)["ok"]       # nothing but the protocol file and the commitments


def test_a_witness_that_breaks_a_gate_yields_a_rejected_proof(tmp_path):
    """the quotient of a violated constraint system is no polynomial: the 4n extended evaluations still interpolate to SOMETHING, the proof has its 1 312 bytes,
    and the verifier's identity h(x) (x^n - 1) == numerator(x) fails at the random x"""
    inp, _ = build_instance(tmp_path, 4, 7)
    vk = plonk.keygen_vk(inp.pr, inp.pre, inp.tau)
    inp.advice[0][3] = (inp.advice[0][3] + 1) % pyref.R_MOD       # the output cell of the first vertical gate
    proof = plonk.prove(inp, vk)
    assert len(proof) == 1312 and not plonk.verify(inp.pr, vk, inp.instances, proof, inp.tau)["ok"]


def test_replay_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the replay runs under -m gpu")
    rec = zk.replay.run(2, 6, out_dir=str(tmp_path))
    assert not rec["ok"] and rec["returncode"] == 2 and "mi355_init" in rec["error"]


# ------------------------------------------------------------------------------------------------ GPU: the HIP prover vs the CPU restatement, byte for byte
def check_against_restatement(rec):
    assert rec.get("ok"), rec.get("error")
    inp, man = plonk.ProofInputs.load(rec["out_dir"])
    vk = plonk.keygen_vk(inp.pr, inp.pre, inp.tau)
    assert rec["vk"] == vk, "verifying key (commit_lagrange of the fixed / sigma columns) differs"
    assert rec["transcript"] == ("evm" if inp.pr.d.get("layer") == 6 else "poseidon")           # the reference's choice per layer
    want = plonk.prove(inp, vk, transcript=rec["transcript"])
    got = rec["proof"]
    first = next((i // 32 for i in range(0, min(len(got), len(want)), 32) if got[i:i + 32] != want[i:i + 32]), None)
    assert got == want, f"proof bytes differ from the CPU restatement from word {first} on ({len(got)} vs {len(want)} bytes)"
    assert plonk.verify(inp.pr, rec["vk"], inp.instances, got, inp.tau, transcript=rec["transcript"])["ok"]
    pr = inp.pr
    nw = sum(pr.num_witness)
    assert rec["msm"] == nw + pr.Q + 2 and rec["intt"] == nw and rec["evals"] == len(pr.evaluations) and rec["proof_bytes"] == len(want)   # nw - 1 witness polynomials + the instance column
    return rec


GPU_CASES = [
    (2, 7, [], {}, {}), (4, 8, [], {}, {}), (6, 7, [], {}, {}), (5, 8, [], {}, {}), (1, 8, [], {}, {}), (3, 8, [], {}, {}),
    (4, 10, ["--pk-cosets", "on-the-fly"], {}, {}), (2, 9, ["--no-tables", "--proofs", "3"], {}, {}), (3, 9, ["--pinned-witness", "--upload-threads", "3", "--early-intt", "1", "--tables", "lagrange"], {}, {}),
    (4, 9, ["--devices", "2"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "6"}, {}),
    (2, 8, ["--devices", "3", "--pk-cosets", "on-the-fly"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "6"}, {}),
    (3, 9, ["--devices", "8"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "6"}, {}),      # the target node's device count: 8 MSM shards, 4 quotient parts on 4 of the slots
    (0, 8, ["--devices", "8"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "5"}, dict(advice=40, fixed=8, lookups=3, perm_columns=12, degree=9)),   # Q = 8: one quotient part per slot
    (0, 8, [], {}, dict(advice=40, fixed=8, lookups=3, perm_columns=12, degree=5)),
    (0, 13, ["--sparse-uploads", "--assign-density", "0.3", "--upload-threads", "2"], {}, dict(advice=40, fixed=8, lookups=3, perm_columns=12, degree=5)),   # mostly-zero columns cross PCIe as (index, value) pairs
    (3, 12, ["--sparse-uploads"], {"MI355_PLAN_PREFIX_MIN": "0"}, {}),
    (4, 11, ["--no-packed-multiplicities"], {}, {}),                                                                                                           # by default the multiplicity columns cross PCIe as 4-byte counts + their blinding rows; here as 32-byte words                                                                                         # and the plan without common-prefix groups
    (0, 7, [], {}, dict(advice=70, fixed=9, lookups=10, perm_columns=30, degree=9)),
    (3, 10, [], {"MI355_NTT_COSET_FOLD_MAX_LOG": "0"}, {}),                                                                                                    # the coset shift as its own k_distribute_powers pass (round 5's schedule; by default it rides on the first pass of the transform)
]


@pytest.mark.gpu
@pytest.mark.parametrize("layer,k,args,env,shape", GPU_CASES)
def test_gpu_proof_bytes_equal_the_cpu_restatement(tmp_path, layer, k, args, env, shape):
    rec = check_against_restatement(zk.replay.run(layer, k, out_dir=str(tmp_path), args=["--dump-inputs"] + args, env=env, **shape))
    if layer in (2, 6):   # layer 6 in the EVM layout: 11 uncompressed points + 17 words = 1 248 bytes, the 39 words that follow the accumulator in [REF release-v0.13.1/proof.data]
        assert (rec["msm"], rec["intt"], rec["evals"], rec["proof_bytes"]) == (11, 5, 17, 896 if layer == 2 else 1248) and rec["coset_ntt"] >= 20
    if layer == 4:
        assert (rec["msm"], rec["intt"], rec["evals"], rec["proof_bytes"]) == (14, 8, 27, 1312) and rec["coset_ntt"] >= 32


@pytest.mark.gpu
def test_gpu_proof_starts_from_the_protocol_s_initial_state(tmp_path):
    """the device prover under a protocol file that carries `transcript_initial_state` (as [REF release-v0.13.1/chunk.protocol] does): bytes equal the CPU restatement, which
    starts from the file's scalar (test_a_protocol_s_own_initial_state_keys_the_transcript), and differ from the proof of the same circuit under the .vkey-hash convention"""
    d, path = with_initial_state(tmp_path, 2, 7)
    stated = check_against_restatement(zk.replay.run(2, None, out_dir=d, args=["--dump-inputs", "--proofs", "1"], protocol_file=path))
    plain = check_against_restatement(zk.replay.run(2, 7, out_dir=str(tmp_path / "plain"), args=["--dump-inputs", "--proofs", "1"]))
    na = 32 * 1
    assert stated["vk"] == plain["vk"] and stated["proof"][:na] == plain["proof"][:na] and stated["proof"][na:] != plain["proof"][na:]


@pytest.mark.gpu
def test_gpu_proof_of_a_broken_witness_is_rejected(tmp_path):
    """one advice cell off by one: the device still emits 1 312 well-formed bytes, and the verifier refuses them (the quotient is no polynomial)"""
    rec = zk.replay.run(4, 8, out_dir=str(tmp_path), args=["--corrupt-witness", "--proofs", "1"])
    assert rec.get("ok"), rec.get("error")
    pr = plonk.Protocol(json.load(open(rec["protocol_path"])))
    inst = plonk.mont_to_ints(np.frombuffer(rec["instances"], dtype=np.uint64).reshape(-1, 4))
    assert len(rec["proof"]) == 1312 and not plonk.verify(pr, rec["vk"], inst, rec["proof"], TAU0 + 4, transcript=rec["transcript"])["ok"]


def verify_record(rec, layer):
    assert rec.get("ok"), rec.get("error")
    pr = plonk.Protocol(json.load(open(rec["protocol_path"])))
    inst = plonk.mont_to_ints(np.frombuffer(rec["instances"], dtype=np.uint64).reshape(-1, 4))
    tau = TAU0 + (rec["layer"] if rec["layer"] >= 0 else 0)
    res = plonk.verify(pr, rec["vk"], inst, rec["proof"], tau, transcript=rec["transcript"])
    assert res["ok"], res
    return pr


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [2, 4])
def test_gpu_full_size_proof_of_the_reference_protocol_verifies(tmp_path, layer):
    """the reference's OWN protocol file at its own size (k = 25 / 26): proven on the device, verified from the bytes by the tree-walking verifier"""
    rec = zk.replay.run(layer, out_dir=str(tmp_path), protocol_file=os.path.join(GOLD, f"protocol_layer{layer}.json"), timeout=1500)
    pr = verify_record(rec, layer)
    want = {2: (25, 11, 5, 20, 17, 896), 4: (26, 14, 8, 32, 27, 1312)}[layer]
    assert (pr.k, rec["msm"], rec["intt"], rec["evals"], rec["proof_bytes"]) == (want[0], want[1], want[2], want[4], want[5])
    assert rec["coset_ntt"] == want[3] or rec["pk_cosets"] == "on-the-fly"
    assert rec["rotation_sets"] == (3 if layer == 2 else 4)


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [3, 0])
def test_gpu_full_size_many_column_layers_verify(tmp_path, layer):
    """layer 3 (k = 21, 93 advice columns, 32 grand products) and the layer-0 stand-in (k = 20, 800 advice columns, degree 9) at full size"""
    rec = zk.replay.run(layer, out_dir=str(tmp_path), timeout=1500)
    verify_record(rec, layer)


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [1, 5, 6])
def test_gpu_full_size_remaining_layers_verify(tmp_path, layer):
    """the layers round 5 verified at full size only inside bench.py: layer 1 (k = 24, halo2-base rule on layer1.config), layer 5 (k = 21) and
    layer 6 -- k = 26, the Keccak transcript and the EVM proof layout of the bundle proof [REF integration/src/prove.rs:95-103]: 11 uncompressed points + 17 words = 1 248 bytes,
    the 39 words that follow the accumulator in [REF release-v0.13.1/proof.data] -- proven on the device, verified from the bytes"""
    rec = zk.replay.run(layer, out_dir=str(tmp_path), timeout=1500)
    pr = verify_record(rec, layer)
    assert pr.k == {1: 24, 5: 21, 6: 26}[layer] and rec["transcript"] == ("evm" if layer == 6 else "poseidon")
    if layer == 6:
        assert (rec["msm"], rec["intt"], rec["evals"], rec["proof_bytes"], rec["rotation_sets"]) == (11, 5, 17, 1248, 3)


@pytest.mark.gpu
def test_gpu_prover_invisible_choices(tmp_path):
    """What the verifier cannot pin: the prover's own randomness.  The SAME witness proven with another blinding stream gives other proof bytes that verify
    just as well, and with the blinding rows and the z / phi blinding values left ZERO the proof still verifies -- so `verify() accepts` says nothing about how blinding is
    drawn or placed (only that the rows it occupies are the protocol's unusable rows); each variant still equals the CPU restatement byte for byte on the same inputs."""
    base = check_against_restatement(zk.replay.run(4, 9, out_dir=str(tmp_path / "a"), args=["--dump-inputs", "--proofs", "1"]))
    other = check_against_restatement(zk.replay.run(4, 9, out_dir=str(tmp_path / "b"), args=["--dump-inputs", "--proofs", "1", "--blind-seed", "77"]))
    zero = check_against_restatement(zk.replay.run(4, 9, out_dir=str(tmp_path / "c"), args=["--dump-inputs", "--proofs", "1", "--zero-blinding"]))
    assert base["vk"] == other["vk"] == zero["vk"] and base["instances"] == other["instances"] == zero["instances"]      # same circuit, same public inputs
    assert len({base["proof"], other["proof"], zero["proof"]}) == 3                                                      # three different proofs of it
    ia, _ = plonk.ProofInputs.load(str(tmp_path / "a")); ib, _ = plonk.ProofInputs.load(str(tmp_path / "b")); ic, _ = plonk.ProofInputs.load(str(tmp_path / "c"))
    u = ia.pr.usable
    for x, y in zip(ia.advice, ib.advice):
        assert x[: u - 8] == y[: u - 8]                                                                                  # the witness proper is the same (cells next to the wrap-around may follow the blinding rows)
    assert all(v == 0 for col in ic.advice for v in col[u + 1:]) and all(v == 0 for zb in ic.z_blind for v in zb)


@pytest.mark.gpu
def test_gpu_one_prover_process_holds_three_layers(tmp_path):
    """a chunk prover's shape [REF integration/src/prove.rs:30-43]: ONE process with the SRS of three degrees, the proving keys and witnesses of layers 0, 1, 2 resident under
    plan_residency, the three proofs back to back, two rounds; every proof verified from its bytes"""
    rec = zk.replay.run_process([0, 1, 2], ks={0: 9, 1: 11, 2: 12}, out_dir=str(tmp_path), shapes={0: dict(advice=40, fixed=8, lookups=3, perm_columns=12, degree=9)})
    assert rec.get("ok"), rec.get("error")
    assert rec["plan"]["fits"] and len(rec["layers"]) == 3 and rec["rounds"] == 2
    for lay in rec["layers"]:
        pr = plonk.Protocol(json.load(open(lay["protocol_path"])))
        inst = plonk.mont_to_ints(np.frombuffer(lay["instances"], dtype=np.uint64).reshape(-1, 4))
        assert plonk.verify(pr, lay["vk"], inst, lay["proof"], int(lay["tau"], 16), transcript=lay["transcript"])["ok"], lay["layer"]
        assert lay["cosets_resident"] and lay["proof_bytes"] == len(lay["proof"])
