"""r06 job 7: create_proof without the device-wide synchronisations of steps 1 / 4 (uploads order themselves): byte equality + timing at k = 26 / 25 / 20"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_plonk_protocol.py", "-m", "gpu", "-x", "-q"], cwd=ROOT, capture_output=True, text=True)
out["tests"] = r.stdout[-400:]; print(r.stdout[-600:], r.stderr[-300:], flush=True)
import __graft_entry__ as ge
zk = ge.load_package()
for layer in (4, 6, 2, 1, 0, 3):
    rec = zk.replay.run(layer, args=["--proofs", "3"], timeout=1200)
    keep = {k: rec.get(k) for k in ("ok", "k", "resident_ms", "step_ms", "error")}
    out[f"layer{layer}"] = keep
    print(layer, json.dumps(keep), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_job7.json"), "w"), indent=1, default=str)
