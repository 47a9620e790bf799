#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
r = bench.prover_process((0, 1, 2))
print(json.dumps(r)[:2500])
json.dump(r, open("gpurun_out/r05_prover_process_0_1_2_trim.json", "w"), indent=1)
PY
