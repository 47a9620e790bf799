#!/bin/bash
# round-2 GPU session S: calibration of the accumulate segment minimum and of the serial fix-up span on witness-like / degenerate columns (2^26)
O=gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
for cfg in "16 32" "64 32" "128 32" "256 32" "4096 32" "16 8" "16 3" "64 8" "128 8" "64 3"; do
  set -- $cfg
  echo "== MI355_SEG_MIN=$1 MI355_FIXUP_SERIAL_MAX=$2" >> $O/calib.log
  MI355_SEG_MIN=$1 MI355_FIXUP_SERIAL_MAX=$2 timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu >> $O/calib.log
done
cat $O/calib.log
