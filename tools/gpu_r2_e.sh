#!/bin/bash
# round-2 GPU session E: whole suite on the final source, fuzz (default; several device slots + sliced host path), c = 23 / 24 tables, N = 2 bench control flow on one GPU
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf > $O/pytest_all.log 2>&1
echo "rc=$?" >> $O/pytest_all.log
timeout 500 python tools/fuzz_gpu.py 120 11 > $O/fuzz_default.log 2>&1
MI355_ALLOW_DUP_DEVICES=1 FUZZ_DEVICES=0,0,0 MI355_SHARD_MIN_LOG=4 MI355_HOST_SLICE_MIN_LOG=6 timeout 500 python tools/fuzz_gpu.py 120 12 > $O/fuzz_multi_slices.log 2>&1
MI355_HOST_SLICE_MIN_LOG=5 timeout 400 python tools/fuzz_gpu.py 80 13 > $O/fuzz_slices.log 2>&1
for C in 23 24; do
  timeout 300 python bench.py --steps 5 --warmup 2 --table-bits $C --no-cpu-baseline --no-ntt --no-proof-mix --no-table-free > $O/bench_c$C.json 2>> $O/bench.err
done
MI355_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --logn 22 --no-ntt > $O/bench_share2.json 2>> $O/bench.err
tail -3 $O/pytest_all.log; tail -1 $O/fuzz_default.log; tail -1 $O/fuzz_multi_slices.log; tail -1 $O/fuzz_slices.log; head -c 300 $O/bench_c24.json
