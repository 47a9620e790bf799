#!/bin/bash
# round-2 GPU session Q: accumulate segment length from the actual entry count (witness-like columns) -- suite, fuzz, degenerate distributions, bench
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -rf > $O/pytest_all.log 2>&1 ) 2> $O/pytest.time
echo "rc=$?" >> $O/pytest_all.log
timeout 400 python tools/fuzz_gpu.py 150 51 > $O/fuzz.log 2>&1
timeout 300 python tools/bench_giant_buckets.py > $O/giant.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_all.log; cat $O/pytest.time; tail -1 $O/fuzz.log; grep -v amdgpu $O/giant.log | tail -12; python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['msm_phase_ms']); print(d['witness_like']); print(d['sizes']['k20']['msm_ms_per_commit'], d['sizes']['k24']['msm_ms_per_commit'])"
