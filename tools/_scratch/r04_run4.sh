#!/bin/bash
# round-4 GPU session 4: lock-free uploads, batched coset shift / evaluations; sanitizer run through the compiled callers; replays
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_at_size_r4.py tests/test_cpp_mirror.py tests/test_gpu_buffers.py tests/test_gpu_parity.py -k "eval or coset or create_proof_replay or buffer or upload or distribute or quotient" -m gpu -q --timeout 900 > gpurun_out/r04_run4_tests.log 2>&1; tail -3 gpurun_out/r04_run4_tests.log
python tools/bench_ntt_small.py > gpurun_out/r04_ntt_small_after.log 2>&1; cat gpurun_out/r04_ntt_small_after.log
for L in 0 3 4 1; do timeout 500 ./tests/cpp/test_create_proof_replay --layer $L > gpurun_out/r04c_replay_L$L.json 2>&1; tail -c 200 gpurun_out/r04c_replay_L$L.json; done
# sanitizers: the host side of the library (pool, handle tables, worker threads, batch paths) under ASAN + UBSAN, driven by the compiled callers
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p /tmp/asanlib && cp scroll-prover_amd/libmi355zk_asan.so /tmp/asanlib/libmi355zk.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
{
  echo "== test_shim_replay"; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_shim_replay 2>&1 | tail -15; echo "rc=$?"
  echo "== replay layer 3, k = 11, two device slots"; MI355_ALLOW_DUP_DEVICES=1 MI355_SHARD_MIN_LOG=6 LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_create_proof_replay --layer 3 --k 11 --devices 2 2>&1 | tail -c 1500; echo "rc=$?"
  echo "== replay layer 0 (reduced), host api"; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_create_proof_replay --layer 0 --k 10 --advice 40 --fixed 5 --lookups 4 --perm 12 --host-api --pk-cosets on-the-fly 2>&1 | tail -c 1500; echo "rc=$?"
  echo "== halo2 mirror"; LD_LIBRARY_PATH=/tmp/asanlib LD_PRELOAD=$ASAN_RT timeout 300 ./tests/cpp/test_halo2_mirror 2>&1 | tail -5; echo "rc=$?"
  echo "== loaded library:"; LD_LIBRARY_PATH=/tmp/asanlib ldd ./tests/cpp/test_shim_replay | grep mi355
} > gpurun_out/r04_asan_gpu.log 2>&1
grep -c "ERROR: AddressSanitizer\|runtime error:" gpurun_out/r04_asan_gpu.log; tail -12 gpurun_out/r04_asan_gpu.log
