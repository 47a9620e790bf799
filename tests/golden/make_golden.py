#!/usr/bin/env python3
"""
make_golden.py -- regenerates tests/golden/kat.json from the reference's own fixtures.

Run in the build container (needs /root/reference); the output is committed because
/root/reference does not exist on the GPU box.  Nothing here computes anything: it only
copies/decodes bytes so that the tests can check OUR oracles and kernels against values
that come from the reference (KATs A1-A9).

Sources:
  [REF release-v0.13.1/chunk.protocol]                     domain (k=25) + 7 preprocessed G1 points
  [REF release-v0.13.1/vk_chunk.vkey, vk_batch.vkey, vk_bundle.vkey]
  [REF integration/tests/test_data/vk_batch_agg.vkey]
  [REF integration/tests/test_data/full_proof_1.json]        chunk proof (896 B), instances, vk, protocol
  [REF integration/tests/test_data/full_proof_batch_agg_1.json]  batch proof (1312 B), vk, protocol (k=26)
  [REF release-v0.13.1/evm_verifier.yul:17-18,1230-1239]    moduli, g2, s_g2 words
  [REF release-v0.13.1/proof.data, pi.data]                 bundle EVM proof words

Also writes tests/golden/protocol_layer2.json / protocol_layer4.json: the COMPLETE snark-verifier PlonkProtocol of the chunk proof
(layer 2, k = 25: [REF release-v0.13.1/chunk.protocol] == [REF integration/tests/test_data/chunk_chunk_0.protocol] == the base64
`protocol` of full_proof_1.json) and of the batch proof (layer 4, k = 26: the base64 `protocol` of
[REF integration/tests/test_data/full_proof_batch_agg_1.json]) -- quotient.numerator (the constraint system as an expression tree),
queries, evaluations, num_witness, num_challenge, domain -- re-serialised compactly, nothing added or changed, plus the six
layer configs [REF integration/configs/layer{1..6}.config] as tests/golden/layer_configs.json.
"""
import base64, json, os, re, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rd(p, mode="rb"):
    with open(os.path.join(REF, p), mode) as f:
        return f.read()


def proto_digest(pr):
    return {"domain": pr["domain"], "preprocessed": pr["preprocessed"], "num_witness": pr["num_witness"],
            "num_instance": pr["num_instance"], "quotient_num_chunk": pr["quotient"]["num_chunk"],
            "n_evaluations": len(pr["evaluations"])}


out = {"_generated_by": "tests/golden/make_golden.py", "_source": "scroll-tech/scroll-prover fixtures (see docstring)"}
out["chunk_protocol"] = proto_digest(json.loads(rd("release-v0.13.1/chunk.protocol", "r")))
for name, path in [("vk_chunk", "release-v0.13.1/vk_chunk.vkey"), ("vk_batch", "release-v0.13.1/vk_batch.vkey"),
                   ("vk_bundle", "release-v0.13.1/vk_bundle.vkey"), ("vk_batch_agg", "integration/tests/test_data/vk_batch_agg.vkey")]:
    out[name] = rd(path).hex()
cp = json.loads(rd("integration/tests/test_data/full_proof_1.json", "r"))["chunk_proofs"][0]
out["chunk_proof"] = {"protocol": proto_digest(json.loads(base64.b64decode(cp["protocol"]))),
                      "proof": base64.b64decode(cp["proof"]).hex(), "instances": base64.b64decode(cp["instances"]).hex(),
                      "vk": base64.b64decode(cp["vk"]).hex()}
bp = json.loads(rd("integration/tests/test_data/full_proof_batch_agg_1.json", "r"))
out["batch_proof"] = {"protocol": proto_digest(json.loads(base64.b64decode(bp["protocol"]))),
                      "proof": base64.b64decode(bp["proof"]).hex(), "instances": base64.b64decode(bp["instances"]).hex(),
                      "vk": base64.b64decode(bp["vk"]).hex()}
yul = rd("release-v0.13.1/evm_verifier.yul", "r").splitlines()
out["yul"] = {"f_p": re.search(r"0x[0-9a-f]+", yul[16]).group(0), "f_q": re.search(r"0x[0-9a-f]+", yul[17]).group(0),
              "g2_words": [re.search(r", (0x[0-9a-f]+)\)", yul[i]).group(1) for i in range(1229, 1233)],
              "s_g2_words": [re.search(r", (0x[0-9a-f]+)\)", yul[i]).group(1) for i in range(1235, 1239)],
              # the transcript's first word (the verifying key's scalar of layer 6) [REF release-v0.13.1/evm_verifier.yul:66]; NOTE the second G2 point above is -[s]G2: the
              # verifier's pairing call is e(lhs, G2) e(rhs, -[s]G2) == 1 (tests/test_plonk_protocol.py::test_released_accumulators_satisfy_the_pairing)
              "transcript_initial_state": re.search(r"mstore\(0x0, (\d+)\)", yul[65]).group(1)}
out["bundle_proof_data"] = rd("release-v0.13.1/proof.data").hex()
out["bundle_pi_data"] = rd("release-v0.13.1/pi.data").hex()
l2 = json.loads(rd("release-v0.13.1/chunk.protocol", "r"))
assert l2 == json.loads(rd("integration/tests/test_data/chunk_chunk_0.protocol", "r")) == json.loads(base64.b64decode(cp["protocol"]))
l4 = json.loads(base64.b64decode(bp["protocol"]))
assert l4 == json.loads(base64.b64decode(json.loads(rd("integration/tests/test_data/full_proof_batch_agg_2.json", "r"))["protocol"]))
# every chunk proof the reference stores as an input of its batch tests carries this same constraint system, and reads as 9 points | 17 scalars | 2 points under it
import glob
P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def parses(proof: bytes, n_points: int, n_scalars: int) -> bool:
    def point(b):
        v = int.from_bytes(b, "little"); x = v & ((1 << 254) - 1)
        y2 = (x * x * x + 3) % P_MOD; y = pow(y2, (P_MOD + 1) // 4, P_MOD)
        return x < P_MOD and y * y % P_MOD == y2
    words = [proof[32 * i:32 * i + 32] for i in range(len(proof) // 32)]
    return (len(words) == n_points + n_scalars + 2 and all(point(w) for w in words[:n_points]) and all(int.from_bytes(w, "little") < R_MOD for w in words[n_points:n_points + n_scalars])
            and all(point(w) for w in words[-2:]))


same_system = 0
more = []          # a few more stored chunk proofs (one per source file, the first of each), for the verifier test: same protocol, other witnesses
verify_all = "--verify-all" in sys.argv
if verify_all:     # every stored chunk proof through the CPU restatement of the verifier (about 0.7 s each)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import plonk as _plonk, pyref as _pyref
    _neg = _pyref.g2_from_evm_words([int(w, 16) for w in out["yul"]["s_g2_words"]])
    _l2p = _plonk.Protocol(l2)
verified = 0
for path in ["integration/tests/test_data/full_proof_batch_prove_1.json", "integration/tests/test_data/batch-task-no-encode.json", "integration/tests/test_data/batch-task-with-blob.json",
             "integration/tests/test_data/batch-task-with-blob-raw.json"] + sorted(os.path.relpath(p, REF) for p in glob.glob(os.path.join(REF, "integration/tests/test_data/batch_tasks/*.json"))):
    for c in json.loads(rd(path, "r")).get("chunk_proofs", []):
        pr_ = json.loads(base64.b64decode(c["protocol"]))
        assert pr_["quotient"] == l2["quotient"] and pr_["queries"] == l2["queries"] and pr_["evaluations"] == l2["evaluations"], path
        assert parses(base64.b64decode(c["proof"]), 9, 17), path
        assert pr_["preprocessed"] == l2["preprocessed"] and pr_["transcript_initial_state"] == l2["transcript_initial_state"], path     # one verifying key throughout
        if same_system == 0 or (len(more) < 6 and all(m["source"] != path for m in more) and base64.b64decode(c["proof"]).hex() != out["chunk_proof"]["proof"]):
            more.append({"source": path, "proof": base64.b64decode(c["proof"]).hex(), "instances": base64.b64decode(c["instances"]).hex()})
        if verify_all:
            ib = base64.b64decode(c["instances"])
            ok = _plonk.verify(_l2p, None, [int.from_bytes(ib[i:i + 32], "big") for i in range(0, len(ib), 32)], base64.b64decode(c["proof"]), transcript="poseidon", neg_s_g2=_neg)["ok"]
            assert ok, (path, same_system)
            verified += 1
        same_system += 1
print("chunk proofs in the reference's test data with this constraint system, each parsing as 9 points | 17 scalars | 2 points:", same_system)
out["more_chunk_proofs"] = more
bp2 = json.loads(rd("integration/tests/test_data/full_proof_batch_agg_2.json", "r"))
out["batch_proof_2"] = {"proof": base64.b64decode(bp2["proof"]).hex(), "instances": base64.b64decode(bp2["instances"]).hex()}
if verify_all:
    print("stored chunk proofs ACCEPTED by oracle/plonk.py's verifier (Poseidon transcript, pairing against the released verifier's -[s]G2):", verified, "of", same_system)
    with open(os.path.join(HERE, "released_proofs_verified.json"), "w") as f:
        json.dump({"_generated_by": "tests/golden/make_golden.py --verify-all", "stored_chunk_proofs": same_system, "accepted_by_oracle_plonk_verify": verified}, f, indent=1)
for name, pr, src in (("protocol_layer2.json", l2, "release-v0.13.1/chunk.protocol == integration/tests/test_data/chunk_chunk_0.protocol == base64 `protocol` of integration/tests/test_data/full_proof_1.json"),
                      ("protocol_layer4.json", l4, "base64 `protocol` of integration/tests/test_data/full_proof_batch_agg_1.json (== full_proof_batch_agg_2.json)")):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump({"_generated_by": "tests/golden/make_golden.py", "_source": "scroll-tech/scroll-prover: " + src,
                   "_what": "GOLDEN VECTOR (data fixture of the reference, not source code): the snark-verifier PlonkProtocol the reference ships with this proof, values untouched",
                   "_stored_proofs_with_this_constraint_system": same_system if name == "protocol_layer2.json" else 2,
                   "protocol": pr}, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")
# ---- known-answer vectors DERIVED from the released proofs (needs this repository's oracle/ and scroll-prover_amd/protocols.py: the script is run from a checkout).  The verifier's final
# step is one multi-scalar multiplication over the proof's own points, and the pairing equation certifies its RESULT -- so (scalars, points) -> result is an MSM instance whose answer the
# reference's data vouches for; likewise the value of the instance polynomial at the challenge x (the verifier needs exactly it) is what an inverse transform of the instance column,
# evaluated at x, must give.  tests/test_released_kats.py feeds both to the C oracle's best_multiexp / best_fft and, under -m gpu, to the device.
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import plonk as _plonk, pyref as _pyref, pairing as _pairing
import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("protocols", os.path.join(os.path.dirname(os.path.dirname(HERE)), "scroll-prover_amd", "protocols.py")); _protocols = _ilu.module_from_spec(_spec); _spec.loader.exec_module(_protocols)
_neg = _pyref.g2_from_evm_words([int(w, 16) for w in out["yul"]["s_g2_words"]])
_words = lambda b: [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]
_hx = lambda v: "%064x" % v
kats = {"_generated_by": "tests/golden/make_golden.py", "_what": "MSM and instance-polynomial known answers derived from the reference's released proofs: every `result` satisfies e(result, G2) e(w_prime, -[s]G2) == 1"}
_pd, _pi, _vkb = bytes.fromhex(out["bundle_proof_data"]), bytes.fromhex(out["bundle_pi_data"]), bytes.fromhex(out["vk_bundle"])
for name, pr_json, inst, proof, kw in (
        ("chunk_proof", l2, _words(bytes.fromhex(out["chunk_proof"]["instances"])), bytes.fromhex(out["chunk_proof"]["proof"]), dict(transcript="poseidon")),
        ("batch_proof", l4, _words(bytes.fromhex(out["batch_proof"]["instances"])), bytes.fromhex(out["batch_proof"]["proof"]), dict(transcript="poseidon")),
        ("bundle_proof", _protocols.layer_protocol(6), _words(_pd[:384]) + _words(_pi), _pd[384:],
         dict(transcript="evm", preprocessed=[_pyref.g1_decompress(_vkb[8 + 32 * i:8 + 32 * i + 32]) for i in range(7)], initial_state=int(out["yul"]["transcript_initial_state"])))):
    _pr = _plonk.Protocol(pr_json)
    res = _plonk.verify(_pr, None, inst, proof, neg_s_g2=_neg, **kw)
    assert res["ok"], name
    m = res["msm"]
    assert _pairing.pairing_product_is_one([(m["result"], _pyref.G2_GEN), (m["w_prime"], _neg)])
    x = res["challenges"]["x"]
    kats[name] = {"k": _pr.k,
                  "msm": {"scalars": [_hx(v) for v in m["scalars"]], "points": [[_hx(p_[0]), _hx(p_[1])] for p_ in m["points"]], "result": [_hx(m["result"][0]), _hx(m["result"][1])],
                          "w_prime": [_hx(m["w_prime"][0]), _hx(m["w_prime"][1])]},
                  "instance_eval": {"x": _hx(x), "instances": [_hx(v) for v in inst], "value": _hx(sum(v * _plonk.lagrange_at(_pr, i, x) for i, v in enumerate(inst)) % _pyref.R_MOD)}}
with open(os.path.join(HERE, "released_kats.json"), "w") as f:
    json.dump(kats, f, indent=1)
print("wrote released_kats.json:", {k_: len(v["msm"]["scalars"]) for k_, v in kats.items() if not k_.startswith("_")}, "MSM terms")
with open(os.path.join(HERE, "layer_configs.json"), "w") as f:
    json.dump({str(i): json.loads(rd(f"integration/configs/layer{i}.config", "r")) for i in range(1, 7)}, f, indent=1)
with open(os.path.join(HERE, "kat.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", os.path.join(HERE, "kat.json"), os.path.getsize(os.path.join(HERE, "kat.json")), "bytes")
