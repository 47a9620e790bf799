#!/bin/bash
# round-4 GPU session 3: new-API tests, two-level NTT A/B at 2^19..2^22, sanitizer run of the buffer/batch GPU tests, roctx ranges, replays of the other layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_at_size_r4.py tests/test_cpp_mirror.py -k "eval_polynomial_batch or create_proof_replay" -m gpu -q --timeout 900 > gpurun_out/r04_run3_tests.log 2>&1; tail -2 gpurun_out/r04_run3_tests.log
MI355_NTT_TWO_LEVEL_MAX_LOG=20 python -m pytest tests/test_gpu_parity.py tests/test_regression_golden.py -m gpu -q --timeout 600 > gpurun_out/r04_run3_two_level_parity.log 2>&1; tail -2 gpurun_out/r04_run3_two_level_parity.log
python tools/bench_ntt_small.py > gpurun_out/r04_ntt_small_default.log 2>&1
MI355_NTT_TWO_LEVEL_MAX_LOG=20 python tools/bench_ntt_small.py > gpurun_out/r04_ntt_small_two_level.log 2>&1
python tools/bench_ntt_small.py >> gpurun_out/r04_ntt_small_default.log 2>&1
cat gpurun_out/r04_ntt_small_default.log gpurun_out/r04_ntt_small_two_level.log
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
(ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$ASAN_RT MI355ZK_LIB=$R/scroll-prover_amd/libmi355zk_asan.so timeout 600 python -m pytest tests/test_gpu_buffers.py tests/test_gpu_multi.py -m gpu -q -p no:cacheprovider --timeout 500) > gpurun_out/r04_asan_gpu.log 2>&1; tail -3 gpurun_out/r04_asan_gpu.log; grep -c "ERROR: AddressSanitizer\|runtime error" gpurun_out/r04_asan_gpu.log
(cd /tmp && export TMPDIR=/tmp && MI355_TRACE=2 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d $R/gpurun_out/roctx -o roctx -- $R/tests/cpp/test_create_proof_replay --layer 4 --k 16 --proofs 1 > $R/gpurun_out/r04_roctx_run.log 2>&1); ls gpurun_out/roctx | head
for L in 4 6 2 1 5 0; do timeout 500 ./tests/cpp/test_create_proof_replay --layer $L > gpurun_out/r04b_replay_L$L.json 2>&1; tail -c 300 gpurun_out/r04b_replay_L$L.json; done
