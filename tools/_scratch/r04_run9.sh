#!/bin/bash
# round-4 GPU session 9: buffer allocation / free without the device lock: buffer, fault and replay tests; layers 0 / 3 / 1 / 4 with and without the early inverse transforms
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_buffers.py tests/test_gpu_faults.py tests/test_cpp_mirror.py tests/test_gpu_multi.py -m gpu -q --timeout 900 2>&1 | tail -3
rm -f gpurun_out/r04_early_intt_ab.log
for L in 0 3 1 4; do for E in 0 1; do timeout 400 ./tests/cpp/test_create_proof_replay --layer $L --early-intt $E $( [ $E = 0 ] && echo --no-check ) > gpurun_out/r04_early_L${L}_e$E.json 2>&1; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_early_L${L}_e$E.json') if l.startswith('{')][0]); print('layer $L early_intt $E proof ms', d['resident_ms'], 'ok', d['ok'], d['semantic_check'], {k: d['step_ms'][k] for k in ('1_instance', '2_3_advice_lookup_commits','4_products','6_to_coeff','7_quotient')}, 'peak', d['hbm']['peak_used_gib'])" | tee -a gpurun_out/r04_early_intt_ab.log; done; done
