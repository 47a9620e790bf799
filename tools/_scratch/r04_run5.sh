#!/bin/bash
# round-4 GPU session 5: witness-upload threads A/B on the k = 20 layer, the default bench line, kernel stats of the headline legs, the N > 1 launch path on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
for U in 1 2 4; do timeout 400 ./tests/cpp/test_create_proof_replay --layer 0 --upload-threads $U --no-check > gpurun_out/r04_upload_threads_L0_u$U.json 2>&1; python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r04_upload_threads_L0_u$U.json') if l.startswith('{')][0]); print('upload_threads', d['upload_threads'], 'resident_ms', d['resident_ms'], d['step_ms'])"; done
timeout 300 ./tests/cpp/test_create_proof_replay --layer 4 --upload-threads 1 --no-check > gpurun_out/r04_upload_threads_L4_u1.json 2>&1; tail -c 400 gpurun_out/r04_upload_threads_L4_u1.json
(time python bench.py) > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; tail -3 gpurun_out/r04_bench_line.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kstats -o kstats -- python $R/bench.py --no-proof-mix --no-batch-legs --no-host-api --no-table-free --no-witness-like --no-sizes --no-cpu-baseline > $R/gpurun_out/r04_kstats_bench_line.json 2> $R/gpurun_out/r04_kstats.err); ls gpurun_out/kstats
MI355_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --logn 22 --steps 3 --no-proof-mix --no-cpu-baseline > gpurun_out/r04_share_gpu_2ranks.json 2> gpurun_out/r04_share_gpu_2ranks.err; tail -c 600 gpurun_out/r04_share_gpu_2ranks.json; tail -2 gpurun_out/r04_share_gpu_2ranks.err
