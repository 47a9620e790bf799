#!/bin/bash
# round-2 GPU session T: second calibration pass (segment minimum x serial fix-up span), 2^26 columns and the 2^24 degenerate set
O=gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
for cfg in "256 8" "256 3" "192 8" "384 8" "512 8" "256 16"; do
  set -- $cfg
  echo "== MI355_SEG_MIN=$1 MI355_FIXUP_SERIAL_MAX=$2" >> $O/calib.log
  MI355_SEG_MIN=$1 MI355_FIXUP_SERIAL_MAX=$2 timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu >> $O/calib.log
  MI355_SEG_MIN=$1 MI355_FIXUP_SERIAL_MAX=$2 timeout 200 python tools/bench_giant_buckets.py 2>&1 | grep -v amdgpu | cut -c1-60 >> $O/calib.log
done
cat $O/calib.log
