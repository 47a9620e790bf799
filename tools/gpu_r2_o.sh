#!/bin/bash
# round-2 GPU session O: 4-lane fix-up + shorter reduce chains for small bucket sets -- parity, fuzz, small-size latencies (A/B on the chunk knob), bench
O=gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_regression_golden.py -m gpu -q -rf -x > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 400 python tools/fuzz_gpu.py 150 41 > $O/fuzz.log 2>&1
timeout 300 python tools/bench_small_sizes.py > $O/small_chunk2.log 2>&1
MI355_REDUCE_MIN_CHUNK=8 timeout 300 python tools/bench_small_sizes.py > $O/small_chunk8.log 2>&1
MI355_REDUCE_MIN_CHUNK=1 timeout 300 python tools/bench_small_sizes.py > $O/small_chunk1.log 2>&1
MI355_REDUCE_MIN_CHUNK=4 timeout 300 python tools/bench_small_sizes.py > $O/small_chunk4.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.log; tail -1 $O/fuzz.log; for f in chunk2 chunk8 chunk1 chunk4; do echo $f; grep -v amdgpu $O/small_$f.log; done; head -c 250 $O/bench.json
