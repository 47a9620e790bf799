#!/usr/bin/env python3
"""verify_released.py -- run the CPU restatement of the verifier (oracle/plonk.py: TEST INFRASTRUCTURE) on the reference's released proofs and print what it finds.

  python tools/verify_released.py                      the proofs committed under tests/golden/ (see below), about ten seconds
  python tools/verify_released.py --all /root/reference every chunk proof stored under integration/tests/test_data (318), about three minutes

Committed: the chunk proof of full_proof_1.json, six more stored chunk proofs, both batch proofs (Poseidon transcript, the fixtures' own protocols) and the released bundle proof
(Keccak transcript, EVM layout, the layer-6 protocol GENERATED from layer6.config, vk_bundle.vkey).  Each is checked with a real pairing against the -[s]G2 of the released EVM
verifier, then once more with one byte flipped (must fail).  Same checks as tests/test_plonk_protocol.py, as a command.
"""
import base64, glob, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util as ilu
from oracle import plonk, pyref

spec = ilu.spec_from_file_location("protocols", os.path.join(ROOT, "scroll-prover_amd", "protocols.py")); protocols = ilu.module_from_spec(spec); spec.loader.exec_module(protocols)
GOLD = os.path.join(ROOT, "tests", "golden")
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
NEG = pyref.g2_from_evm_words([int(w, 16) for w in KAT["yul"]["s_g2_words"]])
words = lambda b: [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]


def check(label, pr, inst, proof, flip_at, **kw):
    t0 = time.time()
    ok = plonk.verify(pr, None, inst, proof, neg_s_g2=NEG, **kw)["ok"]
    bad = bytearray(proof); bad[flip_at] ^= 1
    try:
        ok_bad = plonk.verify(pr, None, inst, bytes(bad), neg_s_g2=NEG, **kw)["ok"]
    except AssertionError:
        ok_bad = False
    print(f"{label:58s} {len(proof):5d} B  {'ACCEPTED' if ok else 'rejected'}   one byte flipped: {'accepted (!)' if ok_bad else 'rejected'}   {time.time() - t0:.1f} s", flush=True)
    return ok and not ok_bad


def main():
    l2 = plonk.Protocol(json.load(open(os.path.join(GOLD, "protocol_layer2.json"))))
    l4 = plonk.Protocol(json.load(open(os.path.join(GOLD, "protocol_layer4.json"))))
    good = True
    if "--all" in sys.argv:
        ref = sys.argv[sys.argv.index("--all") + 1]
        td = os.path.join(ref, "integration", "tests", "test_data")
        files = [os.path.join(td, f) for f in ("full_proof_batch_prove_1.json", "batch-task-no-encode.json", "batch-task-with-blob.json", "batch-task-with-blob-raw.json")] + sorted(glob.glob(os.path.join(td, "batch_tasks", "*.json")))
        n = 0
        for f in files:
            for i, c in enumerate(json.load(open(f)).get("chunk_proofs", [])):
                good &= check(f"{os.path.basename(f)} chunk_proofs[{i}]", l2, words(base64.b64decode(c["instances"])), base64.b64decode(c["proof"]), 32 * 12 + 3, transcript="poseidon"); n += 1
        print(n, "stored chunk proofs;", "all accepted, all tampered copies rejected" if good else "FAILURES above")
        return 0 if good else 1
    c = KAT["chunk_proof"]
    good &= check("chunk proof (full_proof_1.json), layer 2, k = 25", l2, words(bytes.fromhex(c["instances"])), bytes.fromhex(c["proof"]), 100, transcript="poseidon")
    for m in KAT["more_chunk_proofs"]:
        good &= check("chunk proof (" + os.path.basename(m["source"]) + ")", l2, words(bytes.fromhex(m["instances"])), bytes.fromhex(m["proof"]), 500, transcript="poseidon")
    for name in ("batch_proof", "batch_proof_2"):
        good &= check(f"batch proof ({name}), layer 4, k = 26", l4, words(bytes.fromhex(KAT[name]["instances"])), bytes.fromhex(KAT[name]["proof"]), 700, transcript="poseidon")
    pd, pi, vk = bytes.fromhex(KAT["bundle_proof_data"]), bytes.fromhex(KAT["bundle_pi_data"]), bytes.fromhex(KAT["vk_bundle"])
    good &= check("bundle proof (proof.data + pi.data), layer 6, k = 26, EVM", plonk.Protocol(protocols.layer_protocol(6)), words(pd[:384]) + words(pi), pd[384:], 64 * 9 + 7, transcript="evm",
                  preprocessed=[pyref.g1_decompress(vk[8 + 32 * i:8 + 32 * i + 32]) for i in range(7)], initial_state=int(KAT["yul"]["transcript_initial_state"]))
    print("all accepted, all tampered copies rejected" if good else "FAILURES above")
    return 0 if good else 1


if __name__ == "__main__":
    sys.exit(main())
