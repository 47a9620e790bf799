#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for layers in ((0, 1, 2), (3, 4)):
    r = bench.prover_process(layers)
    print(json.dumps(r)[:2300])
    json.dump(r, open(f"gpurun_out/r05_prover_process_{'_'.join(map(str, layers))}_slabs.json", "w"), indent=1)
PY
