#!/bin/bash
# round-2 GPU session AA: segmented fix-up (k_msm_segfix) -- correctness (parity, properties, fuzz in three modes), A/B against the per-bucket kernels
O=gpurun_out/r2aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_regression_golden.py tests/test_gpu_multi.py -m gpu -q -x > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 400 python tools/fuzz_gpu.py 200 101 > $O/fuzz.log 2>&1
MI355_ALLOW_DUP_DEVICES=1 FUZZ_DEVICES=0,0,0 MI355_SHARD_MIN_LOG=4 MI355_HOST_SLICE_MIN_LOG=6 timeout 400 python tools/fuzz_gpu.py 100 102 > $O/fuzz_multi.log 2>&1
for m in 1 0; do
  echo "== MI355_FIXUP_MODE=$m"
  MI355_FIXUP_MODE=$m timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu
  MI355_FIXUP_MODE=$m timeout 200 python tools/bench_giant_buckets.py 2>&1 | grep -v amdgpu | cut -c1-150
  MI355_FIXUP_MODE=$m timeout 300 python tools/bench_small_sizes.py 10 14 18 20 22 2>&1 | grep -v amdgpu | cut -c1-24,112-
done > $O/ab.log 2>&1
echo "== MI355_FIXUP_MODE=1 MI355_SEG_MIN=16, full spread (segment = entries / threads would need the 40% rule off: compare reduce only)" >> $O/ab.log
tail -3 $O/pytest.log; tail -1 $O/fuzz.log; tail -1 $O/fuzz_multi.log; cat $O/ab.log
