#!/bin/bash
# round-4 GPU session 13: upload order (committed columns first), one uploading thread, commit batches by bytes: replay tests, then all layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m pytest tests/test_cpp_mirror.py tests/test_gpu_buffers.py -m gpu -q --timeout 900 2>&1 | tail -2
rm -f gpurun_out/r04_layers_after_overlap.log
for L in 0 1 2 3 4 5 6; do timeout 500 ./tests/cpp/test_create_proof_replay --layer $L > gpurun_out/r04d_replay_L$L.json 2>&1; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04d_replay_L$L.json') if l.startswith('{')][0]); print('layer $L proof ms', d['resident_ms'], 'first', d['first_proof_ms'], 'ok', d['ok'], d['semantic_check'], d['trapdoor_check'], d['step_ms'], 'peak', d['hbm']['peak_used_gib'])" | tee -a gpurun_out/r04_layers_after_overlap.log; done
