#!/bin/bash
# round-2 GPU session P: calibration of the four-lane fix-up threshold (k = 18 .. 23) + bench
O=gpurun_out/r2p; mkdir -p $O
export TMPDIR=/tmp
for L in 0 16 17 18 19; do MI355_FIXUP_LANES_MAX_LOG=$L timeout 300 python tools/bench_small_sizes.py 16 18 19 20 21 22 23 2>&1 | grep -v amdgpu > $O/small_L$L.log; done
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
for L in 0 16 17 18 19; do echo "lanes_max_log=$L"; cut -c1-24,112- $O/small_L$L.log; done; head -c 250 $O/bench.json
