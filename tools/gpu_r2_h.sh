#!/bin/bash
# round-2 GPU session H: G1 DFT on the 29-bit field -- parity, timing at size; reduction tail on the chained multiplier -- whole MSM parity + bench
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_headline.py -m gpu -q -x > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_g1fft.py 16 20 22 24 > $O/g1fft.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix > $O/bench.json 2> $O/bench.err
timeout 300 python tools/bench_small_sizes.py > $O/small.log 2>&1
tail -3 $O/pytest.log; cat $O/g1fft.log; head -c 300 $O/bench.json
