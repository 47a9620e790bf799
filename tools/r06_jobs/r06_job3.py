"""r06 job 3: the whole -m gpu suite on the round-6 tree, then the gate-kernel multiplier A/B (LD_PRELOAD of the -DZK_GATE_CHAIN=false build) and the one-process chunk prover"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
t0 = time.time()
r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-x", "-q", "--durations=10"], cwd=ROOT, capture_output=True, text=True)
open(os.path.join(ROOT, "gpurun_out", "r06_gpu_suite.log"), "w").write(r.stdout[-6000:] + "\n---- stderr ----\n" + r.stderr[-2000:])
out["suite"] = {"rc": r.returncode, "tail": r.stdout[-400:], "wall_s": time.time() - t0}
print(r.stdout[-1200:], flush=True)
import __graft_entry__ as ge
zk = ge.load_package()
variant = os.path.join(ROOT, "scroll-prover_amd", "libmi355zk_gatenochain.so")
for layer in (0, 4):
    for tag, env in (("chain", {}), ("plain", {"LD_PRELOAD": variant})):
        rec = zk.replay.run(layer, args=["--proofs", "3"], env=env, timeout=1200)
        keep = {k: rec.get(k) for k in ("ok", "k", "resident_ms", "step_ms", "error")}
        out[f"gate_layer{layer}_{tag}"] = keep
        print(layer, tag, json.dumps(keep), flush=True)
rec = zk.replay.run_process([0, 1, 2], timeout=2400)
for lay in rec.get("layers", []):
    for key in ("proof", "vk", "instances"):
        lay.pop(key, None)
out["chunk_prover_process"] = rec
print(json.dumps({k: v for k, v in rec.items() if k != "layers"}), flush=True)
for lay in rec.get("layers", []): print(json.dumps(lay), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_job3.json"), "w"), indent=1, default=str)
