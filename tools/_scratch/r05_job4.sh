#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cpp_mirror.py tests/test_plonk_protocol.py -x -q -m gpu --durations=5 2>&1 | tail -12 > gpurun_out/r05_job4_tests.log
tail -4 gpurun_out/r05_job4_tests.log
python tools/_scratch/r05_job4.py 2>&1 | tail -12
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
tail -3 gpurun_out/r05_collect.log
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_err.log
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("metric", "value", "ms_per_step", "verified_against_field_check")}, l["config"], l["roofline"]["frac"], l["roofline"]["traffic"], l["roofline"]["traffic_source"])
PY
