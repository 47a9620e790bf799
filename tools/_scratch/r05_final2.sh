#!/bin/bash
# closing evidence of round 5 on the final tree (with the slabs): the whole -m gpu suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -14 > gpurun_out/r05_gpu_suite.log; grep -n "passed\|failed\|Error" gpurun_out/r05_gpu_suite.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok"
timeout 1200 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_err.log
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("metric", "value", "ms_per_step", "verified_against_field_check")}, l["config"]["proxies_ms"], l["config"]["layer_ms"], l["config"]["all_checks"], l["roofline"]["frac"], l["roofline"]["traffic"])
PY
