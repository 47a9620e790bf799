#!/bin/bash
# round-2 GPU session BB: persistent scatter kernels with next-tile prefetch -- parity, fuzz, A/B
O=gpurun_out/r2bb; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_regression_golden.py -m gpu -q -x > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 400 python tools/fuzz_gpu.py 200 121 > $O/fuzz.log 2>&1
for m in 1 0 1 0; do
  echo "== MI355_SORT_PERSIST=$m"
  MI355_SORT_PERSIST=$m timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu | grep "witness\|uniform"
  MI355_SORT_PERSIST=$m timeout 300 python tools/bench_small_sizes.py 20 22 24 2>&1 | grep -v amdgpu | cut -c1-24,88-
done > $O/ab.log 2>&1
tail -3 $O/pytest.log; tail -1 $O/fuzz.log; cat $O/ab.log
