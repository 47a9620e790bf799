"""first GPU contact of the protocol-driven prover: small k, every layer; GPU proof bytes vs the CPU restatement, then the verifier"""
import os, sys, time, json, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
zk = ge.load_package()
from oracle import plonk
cases = [(2, 7, [], None), (4, 8, [], None), (6, 7, [], None), (5, 8, [], None), (1, 8, [], None), (3, 8, [], None),
         (4, 10, ["--pk-cosets", "on-the-fly"], None), (2, 9, ["--devices", "2"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "6"})]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1:]]
for layer, k, args, env in cases:
    t0 = time.time()
    rec = zk.replay.run(layer, k, args=["--dump-inputs", "--proofs", "1"] + args, env=env)
    if not rec.get("ok"):
        print("LAYER", layer, "k", k, "REPLAY FAILED", rec.get("error")); continue
    try:
        inp, man = plonk.ProofInputs.load(rec["out_dir"])
        vk = plonk.keygen_vk(inp.pr, inp.pre, inp.tau)
        want = plonk.prove(inp, vk)
        got = rec["proof"]
        same = got == want
        first = next((i // 32 for i in range(0, min(len(got), len(want)), 32) if got[i:i + 32] != want[i:i + 32]), None)
        ver = plonk.verify(inp.pr, rec["vk"], inp.instances, got, inp.tau)
        print("LAYER", layer, "k", k, args, "vk_same", vk == rec["vk"], "proof_bytes", len(got), len(want), "IDENTICAL" if same else f"DIFFER at word {first}", "verify", ver["ok"], ver.get("error"),
              "msm", rec["msm"], "intt", rec["intt"], "coset", rec["coset_ntt"], "evals", rec["evals"], "plan", rec["plan"], "ms", rec["resident_ms"], "wall", round(time.time() - t0, 1), flush=True)
    except Exception:
        print("LAYER", layer, "k", k, "CHECK FAILED"); traceback.print_exc()
