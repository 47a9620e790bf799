#!/bin/bash
# round-2 GPU session CC: level-1 scatter fused with the digit extraction (no digit plane) -- parity, fuzz (default + multi-device / sliced host), A/B
O=gpurun_out/r2cc; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_regression_golden.py tests/test_gpu_multi.py tests/test_gpu_headline.py -m gpu -q -x > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 400 python tools/fuzz_gpu.py 200 131 > $O/fuzz.log 2>&1
MI355_ALLOW_DUP_DEVICES=1 FUZZ_DEVICES=0,0,0 MI355_SHARD_MIN_LOG=4 MI355_HOST_SLICE_MIN_LOG=6 timeout 400 python tools/fuzz_gpu.py 100 132 > $O/fuzz_multi.log 2>&1
for m in 1 0 1 0; do
  echo "== MI355_SORT_FUSED=$m"
  MI355_SORT_FUSED=$m timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu
  MI355_SORT_FUSED=$m timeout 300 python tools/bench_small_sizes.py 20 22 24 2>&1 | grep -v amdgpu | cut -c1-24,88-
done > $O/ab.log 2>&1
tail -3 $O/pytest.log; tail -1 $O/fuzz.log; tail -1 $O/fuzz_multi.log; cat $O/ab.log
