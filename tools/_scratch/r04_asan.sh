#!/bin/bash
# sanitizer session: the ASAN + UBSAN host build of the library under the compiled callers, complete logs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/asan
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p /tmp/asanlib && cp scroll-prover_amd/libmi355zk_asan.so /tmp/asanlib/libmi355zk.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
run() { name=$1; shift; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 "$@" > gpurun_out/asan/$name.log 2>&1; echo "$name rc=$? reports=$(grep -c 'ERROR: AddressSanitizer\|runtime error:\|AddressSanitizer CHECK' gpurun_out/asan/$name.log) $(grep -c 'all checks passed' gpurun_out/asan/$name.log)"; }
run shim_replay ./tests/cpp/test_shim_replay
MI355_ALLOW_DUP_DEVICES=1 MI355_SHARD_MIN_LOG=6 run replay_L3_two_slots ./tests/cpp/test_create_proof_replay --layer 3 --k 11 --devices 2 --upload-threads 3
run replay_L3_one_slot ./tests/cpp/test_create_proof_replay --layer 3 --k 11 --upload-threads 3
run replay_L0_host_api ./tests/cpp/test_create_proof_replay --layer 0 --k 10 --advice 40 --fixed 5 --lookups 4 --perm 12 --host-api --pk-cosets on-the-fly --upload-threads 4
run halo2_mirror ./tests/cpp/test_halo2_mirror
grep -h -A25 "ERROR: AddressSanitizer\|AddressSanitizer CHECK\|runtime error:" gpurun_out/asan/*.log | head -120
