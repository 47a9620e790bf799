#!/bin/bash
# round-4 GPU session 6: the full -m gpu suite on the final tree, smoke, sanitizer run with the final library, layer-4 residency A/B, sorter key split A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40) > gpurun_out/r04_gpu_suite.log 2>&1; tail -6 gpurun_out/r04_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p /tmp/asanlib && cp scroll-prover_amd/libmi355zk_asan.so /tmp/asanlib/libmi355zk.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
{
  echo "sanitizer build: $(ls -la scroll-prover_amd/libmi355zk_asan.so)"
  echo "== test_shim_replay"; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_shim_replay 2>&1 | tail -4; echo "rc=$?"
  echo "== replay layer 3, k = 11, two device slots, 3 upload threads"; MI355_ALLOW_DUP_DEVICES=1 MI355_SHARD_MIN_LOG=6 LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_create_proof_replay --layer 3 --k 11 --devices 2 --upload-threads 3 2>&1 | tail -c 700; echo "rc=$?"
  echo "== replay layer 0 (reduced), host api, cosets on the fly"; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_create_proof_replay --layer 0 --k 10 --advice 40 --fixed 5 --lookups 4 --perm 12 --host-api --pk-cosets on-the-fly --upload-threads 4 2>&1 | tail -c 700; echo "rc=$?"
  echo "== halo2 mirror"; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_halo2_mirror 2>&1 | tail -3; echo "rc=$?"
  echo "== loaded library:"; LD_LIBRARY_PATH=/tmp/asanlib ldd ./tests/cpp/test_shim_replay | grep mi355
  echo "== reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' /dev/stdin < /dev/null)"
} > gpurun_out/r04_asan_gpu.log 2>&1
echo "sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' gpurun_out/r04_asan_gpu.log)"; grep "rc=\|all checks" gpurun_out/r04_asan_gpu.log | tr '\n' ' '; echo
unset ASAN_OPTIONS UBSAN_OPTIONS
for cfg in "--tables off" "--pk-cosets on-the-fly --tables on" ""; do timeout 400 ./tests/cpp/test_create_proof_replay --layer 4 --no-check $cfg > gpurun_out/r04_L4_cfg.json 2>&1; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_L4_cfg.json') if l.startswith('{')][0]); print('layer 4 [$cfg] tables', d['window_table_bases'], 'pk', d['pk_cosets'], 'proof ms', d['resident_ms'], 'coset_ntt', d['coset_ntt'], 'peak GiB', d['hbm']['peak_used_gib'], d['step_ms'])" | tee -a gpurun_out/r04_L4_residency_ab.log; done
for FB in 11 12 10 11; do MI355_SORT_FB=$FB python bench.py --steps 8 --no-proof-mix --no-batch-legs --no-host-api --no-table-free --no-witness-like --no-sizes --no-cpu-baseline --no-ntt 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('MI355_SORT_FB=$FB ms_per_step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['msm_phase_ms'].items()})" | tee -a gpurun_out/r04_sort_fb_ab.log; done
