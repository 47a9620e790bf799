#!/bin/bash
# round-4 closing session: the full -m gpu suite, smoke and the default bench line on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40) > gpurun_out/r04_gpu_suite.log 2>&1; tail -6 gpurun_out/r04_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py) > gpurun_out/r04_bench_line_final.json 2> gpurun_out/r04_bench_line_final.err; tail -4 gpurun_out/r04_bench_line_final.err
python tools/probe_host_copies.py 2>/dev/null | tail -10 > gpurun_out/r04_host_copy_rates.log; cat gpurun_out/r04_host_copy_rates.log
