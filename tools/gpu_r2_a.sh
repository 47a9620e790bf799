#!/bin/bash
# round-2 GPU session A: new tests first (continue on failure), then the whole -m gpu suite, then the bench with the host-API leg
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_multi.py tests/test_multi_gpu_gloo.py tests/test_cpp_mirror.py -m gpu -q -rA --durations=15 > gpurun_out/r2a/pytest_new.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r2a/pytest_new.log
timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_headline.py --deselect tests/test_gpu_multi.py > gpurun_out/r2a/pytest_all.log 2>&1
echo "all tests rc=$?" >> gpurun_out/r2a/pytest_all.log
timeout 300 python bench.py --steps 5 --warmup 2 --host-api > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
MI355_HOST_CHUNKS=4 timeout 200 python bench.py --steps 3 --warmup 1 --host-api --no-ntt --no-cpu-baseline > gpurun_out/r2a/bench_hc4.json 2>> gpurun_out/r2a/bench.err
MI355_HOST_CHUNKS=16 timeout 200 python bench.py --steps 3 --warmup 1 --host-api --no-ntt --no-cpu-baseline > gpurun_out/r2a/bench_hc16.json 2>> gpurun_out/r2a/bench.err
tail -5 gpurun_out/r2a/pytest_new.log; tail -3 gpurun_out/r2a/pytest_all.log; head -c 600 gpurun_out/r2a/bench.json
