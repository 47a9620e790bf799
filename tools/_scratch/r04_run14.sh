#!/bin/bash
# round-4 GPU session 14: batched NTT passes (blockIdx.y = polynomial) behind the batch entry points: parity, small-size timing on / off, layers 0 / 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_buffers.py tests/test_gpu_at_size_r4.py tests/test_cpp_mirror.py tests/test_gpu_multi.py -m gpu -q --timeout 900 -k "not gate_eval_full and not scans and not 2_28" 2>&1 | tail -2
for B in 22 0 22; do echo "MI355_NTT_BATCH_MAX_LOG=$B"; MI355_NTT_BATCH_MAX_LOG=$B python tools/bench_ntt_small.py 2>/dev/null | grep "^k="; done | tee gpurun_out/r04_ntt_batched_passes_ab.log
for L in 0 3; do for B in 22 0; do MI355_NTT_BATCH_MAX_LOG=$B timeout 400 ./tests/cpp/test_create_proof_replay --layer $L $( [ $B = 0 ] && echo --no-check ) > gpurun_out/r04_nb_L$L.json 2>&1; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_nb_L$L.json') if l.startswith('{')][0]); print('layer $L MI355_NTT_BATCH_MAX_LOG=$B proof ms', d['resident_ms'], 'ok', d['ok'], d['semantic_check'], {k: d['step_ms'][k] for k in ('2_3_advice_lookup_commits','6_to_coeff','7_quotient')})" | tee -a gpurun_out/r04_ntt_batched_passes_ab.log; done; done
