#!/bin/bash
# round-4 GPU session 11: page-locked witness columns (mi355_host_alloc): test, then layers 0 / 3 / 1 with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_at_size_r4.py tests/test_cpp_mirror.py -m gpu -q --timeout 900 -k "host_alloc or pinned or (replay and layer)" 2>&1 | tail -2
ulimit -l
rm -f gpurun_out/r04_pinned_witness_ab.log
for L in 0 3 1; do for P in "" "--pinned-witness"; do timeout 500 ./tests/cpp/test_create_proof_replay --layer $L $P $( [ -z "$P" ] && echo --no-check ) > gpurun_out/r04_pin_L${L}.json 2>&1; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_pin_L${L}.json') if l.startswith('{')][0]); print('layer $L pinned_witness', d['pinned_witness'], 'proof ms', d['resident_ms'], 'first', d['first_proof_ms'], 'ok', d['ok'], d['semantic_check'], {k: d['step_ms'][k] for k in ('1_instance', '2_3_advice_lookup_commits','4_products','6_to_coeff','7_quotient')})" | tee -a gpurun_out/r04_pinned_witness_ab.log || tail -3 gpurun_out/r04_pin_L${L}.json; done; done
