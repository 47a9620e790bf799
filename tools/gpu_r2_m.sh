#!/bin/bash
# round-2 GPU session M: whole -m gpu suite (wall time recorded) + randomised cross-checks after the inverse / precompute changes
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -rf --durations=15 > $O/pytest_all.log 2>&1 ) 2> $O/pytest.time
echo "rc=$?" >> $O/pytest_all.log
timeout 400 python tools/fuzz_gpu.py 120 31 > $O/fuzz_default.log 2>&1
MI355_ALLOW_DUP_DEVICES=1 FUZZ_DEVICES=0,0,0 MI355_SHARD_MIN_LOG=4 MI355_HOST_SLICE_MIN_LOG=6 timeout 400 python tools/fuzz_gpu.py 100 32 > $O/fuzz_multi_slices.log 2>&1
tail -25 $O/pytest_all.log; cat $O/pytest.time; tail -2 $O/fuzz_default.log; tail -2 $O/fuzz_multi_slices.log
