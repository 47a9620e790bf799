#!/bin/bash
# round-4 GPU session 15: kernel timeline of one layer-0 proof on the final tree (busy fraction per phase)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_L0_final -o kt -- $R/tests/cpp/test_create_proof_replay --layer 0 --no-check --proofs 2 > $R/gpurun_out/r04_kt_L0_final.json 2> $R/gpurun_out/r04_kt_L0_final.err); tail -c 200 gpurun_out/r04_kt_L0_final.json
