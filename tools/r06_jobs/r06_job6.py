"""r06 job 6: residency by value (chunk prover: layer 0 lean, layers 1 + 2 resident) + the lean key's cosets recomputed in one batched call"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_plonk_protocol.py", "-m", "gpu", "-x", "-q", "-k", "cpu_restatement or prover_process"], cwd=ROOT, capture_output=True, text=True)
out["tests"] = r.stdout[-400:]; print(r.stdout[-600:], r.stderr[-300:], flush=True)
import __graft_entry__ as ge
zk = ge.load_package()
for name, layers in (("chunk", [0, 1, 2]), ("batch", [3, 4]), ("chunk_again", [0, 1, 2])):
    rec = zk.replay.run_process(layers, timeout=2400)
    for lay in rec.get("layers", []):
        for key in ("proof", "vk", "instances"): lay.pop(key, None)
    out["process_" + name] = rec
    print(name, json.dumps({k: v for k, v in rec.items() if k != "layers"})[:900], flush=True)
    for lay in rec.get("layers", []): print(json.dumps(lay)[:500], flush=True)
for layer, args in ((0, ["--pk-cosets", "on-the-fly"]), (0, [])):
    rec = zk.replay.run(layer, args=args + ["--proofs", "3"], timeout=1200)
    keep = {k: rec.get(k) for k in ("ok", "k", "resident_ms", "step_ms", "pk_cosets", "hbm", "error")}
    out[f"layer{layer}_{'lean' if args else 'resident'}"] = keep
    print(layer, args, json.dumps(keep), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_job6.json"), "w"), indent=1, default=str)
