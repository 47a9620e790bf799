#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest "tests/test_plonk_protocol.py::test_gpu_proof_bytes_equal_the_cpu_restatement" -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_plonk.py 40 23 2>&1 | grep -v "^ok" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x"]
exec(open("tools/_scratch/r05_job4.py").read().split("out = {}")[0])
out = {}
for layer in (4, 0, 3):
    for tag, args in (("plain", []), ("packed_m", ["--packed-multiplicities"])):
        out[f"L{layer}_{tag}"] = run(layer, args=args, protocol_file=os.path.join(ROOT, "tests", "golden", f"protocol_layer{layer}.json") if layer == 4 else None)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_packed_m_ab.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v.get("ok"), "verified", v.get("verified"), v.get("resident_ms"), (v.get("step_ms") or {}).get("2_3_advice_lookup_commits"), v.get("sparse_uploads"), v.get("error"))
PY
