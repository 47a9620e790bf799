#!/bin/bash
# round-2 GPU session Y: slice-major block order of the giant-bucket fix-up; thresholds again
O=gpurun_out/r2y; mkdir -p $O
export TMPDIR=/tmp
for hm in 2048 8192; do
  echo "== MI355_FIXUP_HUGE_MIN=$hm"
  MI355_FIXUP_HUGE_MIN=$hm timeout 200 python tools/bench_giant_buckets.py 2>&1 | grep -v amdgpu | cut -c1-150
  MI355_FIXUP_HUGE_MIN=$hm timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu
done > $O/calib.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "giant or edge or witness or msm" 2>&1 | tail -2 >> $O/calib.log
timeout 300 python tools/fuzz_gpu.py 100 81 2>&1 | tail -1 >> $O/calib.log
cat $O/calib.log
