#!/bin/bash
# round-4 GPU session 12: kernel timelines of one layer-4 and one layer-1 proof (where is the device idle?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
for L in 4 1; do (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/kt_L$L -o kt -- $R/tests/cpp/test_create_proof_replay --layer $L --no-check --proofs 2 > $R/gpurun_out/r04_kt_L$L.json 2> $R/gpurun_out/r04_kt_L$L.err); tail -c 200 gpurun_out/r04_kt_L$L.json; done
