"""r06 job 1: buffer tests (range checks, slab segregation), then --phase-profile replays of the many-column layers (tail share of ONE proof, keygen excluded)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
zk = ge.load_package()
out = {}
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_buffers.py", "-m", "gpu", "-x", "-q"], cwd=ROOT, capture_output=True, text=True)
out["buffers_tests"] = r.stdout[-600:]
print(out["buffers_tests"], flush=True)
for layer in (0, 3, 5, 2):
    rec = zk.replay.run(layer, args=["--phase-profile"], timeout=1200)
    keep = {k: rec.get(k) for k in ("ok", "k", "msm", "intt", "coset_ntt", "resident_ms", "first_proof_ms", "step_ms", "phase_profile", "error", "hbm")}
    out[f"layer{layer}"] = keep
    print(layer, json.dumps(keep), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_phase_profile.json"), "w"), indent=1)
