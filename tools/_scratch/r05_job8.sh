#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_buffers.py tests/test_gpu_faults.py "tests/test_plonk_protocol.py::test_gpu_proof_bytes_equal_the_cpu_restatement" tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -6
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for layers in ((0, 1, 2), (3, 4)):
    r = bench.prover_process(layers)
    print(json.dumps(r)[:2300])
    json.dump(r, open(f"gpurun_out/r05_prover_process_{'_'.join(map(str, layers))}_slabs.json", "w"), indent=1)
for lay in (0, 4):
    r = bench.replay_create_proof(lay)
    print(lay, r.get("ok"), r.get("resident_ms"), r.get("first_proof_ms"), r.get("step_ms"), (r.get("hbm") or {}), r.get("window_table_bases"), r.get("pk_cosets"), r.get("error"))
PY
