#!/bin/bash
# round-2 GPU session X: giant-bucket fix-up -- threshold between the workgroup-per-bucket and the several-workgroups path
O=gpurun_out/r2x; mkdir -p $O
export TMPDIR=/tmp
for hm in 2048 8192 32768 131072 1000000000; do
  echo "== MI355_FIXUP_HUGE_MIN=$hm"
  MI355_FIXUP_HUGE_MIN=$hm timeout 200 python tools/bench_giant_buckets.py 2>&1 | grep -v amdgpu | grep -v "^uniform\|r - 1\|99%" | cut -c1-150
  MI355_FIXUP_HUGE_MIN=$hm timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu | grep -v uniform
done > $O/calib.log 2>&1
cat $O/calib.log
