"""r06 job 4: the fold at k = 25 / 26 behind the free-HBM guard: full-size proofs verify, A/B against the cap of 24, prover processes {0,1,2} and {3,4}"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_plonk_protocol.py", "tests/test_gpu_at_size_r4.py", "-m", "gpu", "-x", "-q", "-k", "full_size or coset"], cwd=ROOT, capture_output=True, text=True)
out["tests"] = r.stdout[-500:]; print(r.stdout[-800:], r.stderr[-300:], flush=True)
import __graft_entry__ as ge
zk = ge.load_package()
for layer in (4, 2, 6, 1):
    for tag, env in (("cap26", {}), ("cap24", {"MI355_NTT_COSET_FOLD_MAX_LOG": "24"})):
        rec = zk.replay.run(layer, args=["--proofs", "3"], env=env, timeout=1200)
        keep = {k: rec.get(k) for k in ("ok", "k", "resident_ms", "step_ms", "hbm", "error")}
        out[f"layer{layer}_{tag}"] = keep
        print(layer, tag, json.dumps(keep), flush=True)
for name, layers in (("chunk", [0, 1, 2]), ("batch", [3, 4])):
    rec = zk.replay.run_process(layers, timeout=2400)
    for lay in rec.get("layers", []):
        for key in ("proof", "vk", "instances"): lay.pop(key, None)
    out["process_" + name] = rec
    print(name, json.dumps({k: v for k, v in rec.items() if k != "layers"}), flush=True)
    for lay in rec.get("layers", []): print(json.dumps(lay), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_job4.json"), "w"), indent=1, default=str)
