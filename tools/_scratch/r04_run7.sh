#!/bin/bash
# round-4 GPU session 7: where the wall time of a layer-0 / layer-3 proof goes (kernel time against wall time), sanitizer re-run with flushed output, fuzz on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_buffers.py tests/test_gpu_at_size_r4.py -m gpu -q -k "gate_eval and not 26" --timeout 600 2>&1 | tail -2
for L in 0 3; do (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_L$L -o kt -- $R/tests/cpp/test_create_proof_replay --layer $L --no-check --proofs 1 > $R/gpurun_out/r04_kt_L$L.json 2> $R/gpurun_out/r04_kt_L$L.err); tail -c 300 gpurun_out/r04_kt_L$L.json; done
bash tools/r04_asan.sh 2>&1 | tail -8
(time python tools/fuzz_gpu.py 1500 404) > gpurun_out/r04_fuzz.log 2>&1; tail -4 gpurun_out/r04_fuzz.log
