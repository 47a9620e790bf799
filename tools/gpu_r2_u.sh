#!/bin/bash
# round-2 GPU session U: segment = entries / (40 % of the launched threads): columns at 2^26, degenerate set at 2^24, small sizes, fuzz
O=gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu > $O/cols26.log
MI355_SEG_MIN=4096 timeout 200 python tools/bench_witness_like.py 26 2>&1 | grep -v amdgpu > $O/cols26_worstcase_seg.log
timeout 200 python tools/bench_giant_buckets.py 2>&1 | grep -v amdgpu > $O/giant24.log
MI355_SEG_MIN=4096 timeout 200 python tools/bench_giant_buckets.py 2>&1 | grep -v amdgpu > $O/giant24_worstcase_seg.log
timeout 300 python tools/bench_small_sizes.py 14 18 20 22 2>&1 | grep -v amdgpu > $O/small.log
timeout 400 python tools/fuzz_gpu.py 150 61 > $O/fuzz.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x > $O/pytest.log 2>&1
cat $O/cols26.log; echo worst; cat $O/cols26_worstcase_seg.log; cut -c1-140 $O/giant24.log; echo worst; cut -c1-140 $O/giant24_worstcase_seg.log; cat $O/small.log; tail -1 $O/fuzz.log; tail -2 $O/pytest.log
