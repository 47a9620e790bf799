"""verified A/Bs of round 5 (every replay's proof goes through oracle/plonk.py's verifier): common-prefix groups, upload variants on a 60 %-zero layer-0 witness, the layer-0 sensitivity band"""
import sys, json, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
zk = ge.load_package()
from oracle import plonk
KEEP = ("ok", "verified", "error", "k", "resident_ms", "first_proof_ms", "step_ms", "plan", "pk_cosets", "window_table_bases", "sparse_uploads", "pinned_witness", "msm", "intt", "coset_ntt", "evals", "proof_bytes", "gate_launches", "gate_eval_process_totals", "hbm", "circuit", "protocol")
def run(layer, **kw):
    rec = zk.replay.run(layer, **kw)
    if rec.get("ok"):
        pr = plonk.Protocol(json.load(open(rec["protocol_path"])))
        inst = plonk.mont_to_ints(np.frombuffer(rec["instances"], dtype=np.uint64).reshape(-1, 4))
        rec["verified"] = bool(plonk.verify(pr, rec["vk"], inst, rec["proof"], 0x5343524F4C4C0001 + max(rec["layer"], 0))["ok"])
    return {k: rec.get(k) for k in KEEP}
out = {}
for layer in (0, 3):
    for pm in ("16", "0"):
        out[f"prefix_L{layer}_min{pm}"] = run(layer, env={"MI355_PLAN_PREFIX_MIN": pm})
for tag, args in (("sparse60_plain", ["--assign-density", "0.4"]), ("sparse60_sparse", ["--assign-density", "0.4", "--sparse-uploads", "--upload-threads", "2"])):
    out["uploads_L0_" + tag] = run(0, args=args)
for adv, lk, deg, fx, pc in ((400, 30, 5, 60, 75), (1600, 120, 9, 240, 300)):
    out[f"band_L0_{adv}_{lk}_{deg}"] = run(0, advice=adv, lookups=lk, degree=deg, fixed=fx, perm_columns=pc, timeout=1500)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_job3_ab.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v.get("ok"), "verified", v.get("verified"), v.get("resident_ms"), (v.get("step_ms") or {}).get("2_3_advice_lookup_commits"), (v.get("step_ms") or {}).get("7_quotient"), v.get("plan"), v.get("pk_cosets"), (v.get("hbm") or {}).get("peak_used_gib"), v.get("error"))
