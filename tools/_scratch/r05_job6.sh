#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest "tests/test_plonk_protocol.py::test_gpu_one_prover_process_holds_three_layers" -x -q -m gpu 2>&1 | tail -5
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for layers in ((0, 1, 2), (3, 4)):
    r = bench.prover_process(layers)
    print(json.dumps(r)[:3000])
    json.dump(r, open(f"gpurun_out/r05_prover_process_{'_'.join(map(str, layers))}.json", "w"), indent=1)
PY
