#!/bin/bash
# second GPU job of round 5: new tests, then the profile passes, then the default bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_multi.py tests/test_bench_launch.py "tests/test_plonk_protocol.py::test_gpu_proof_bytes_equal_the_cpu_restatement" -x -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/r05_job2_tests.log
tail -6 gpurun_out/r05_job2_tests.log
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
tail -12 gpurun_out/r05_collect.log
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_err.log
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("metric", "value", "ms_per_step", "verified_against_field_check")}, l["config"], l["roofline"]["frac"], l["cpu_baseline"])
print({k: v.get("cpu_baseline") for k, v in l.get("sizes", {}).items()})
PY
