#!/bin/bash
# round-2 GPU session I: after the early-clobber fix of the chained multiplier -- whole suite, fuzz (three modes), G1 DFT timing, bench (default and variant 0)
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf > $O/pytest_all.log 2>&1
echo "rc=$?" >> $O/pytest_all.log
timeout 500 python tools/fuzz_gpu.py 150 21 > $O/fuzz_default.log 2>&1
MI355_ALLOW_DUP_DEVICES=1 FUZZ_DEVICES=0,0,0 MI355_SHARD_MIN_LOG=4 MI355_HOST_SLICE_MIN_LOG=6 timeout 500 python tools/fuzz_gpu.py 150 22 > $O/fuzz_multi_slices.log 2>&1
MI355_ACC_VARIANT=0 timeout 400 python tools/fuzz_gpu.py 60 23 > $O/fuzz_v0.log 2>&1
timeout 600 python tools/bench_g1fft.py 16 20 22 24 > $O/g1fft.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
MI355_ACC_VARIANT=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-ntt > $O/bench_v0.json 2>> $O/bench.err
tail -3 $O/pytest_all.log; tail -1 $O/fuzz_default.log; tail -1 $O/fuzz_multi_slices.log; tail -1 $O/fuzz_v0.log; cat $O/g1fft.log; head -c 300 $O/bench.json
