#!/bin/bash
# round-2 GPU session C: whole -m gpu suite, default bench (host-API, table-free, proof-mix legs), single-process 2-slot bench (test mode)
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf --durations=12 > $O/pytest_all.log 2>&1
echo "rc=$?" >> $O/pytest_all.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
MI355_ALLOW_DUP_DEVICES=1 timeout 300 python bench.py --gpus 2 --single-process --logn 24 --steps 3 --warmup 1 --no-proof-mix --no-cpu-baseline --no-ntt > $O/bench_single2.json 2>> $O/bench.err
MI355_HOST_CHUNKS=4 timeout 300 python bench.py --steps 3 --warmup 1 --no-proof-mix --no-cpu-baseline --no-ntt --no-table-free > $O/bench_hc4.json 2>> $O/bench.err
tail -4 $O/pytest_all.log; head -c 400 $O/bench.json
