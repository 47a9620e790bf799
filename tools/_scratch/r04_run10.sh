#!/bin/bash
# round-4 GPU session 10: raw scratch between the NTT passes (MI355_NTT_RAW_SCRATCH=1): parity, then alternating timing runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
MI355_NTT_RAW_SCRATCH=1 python -m pytest tests/test_gpu_parity.py tests/test_regression_golden.py tests/test_gpu_properties.py tests/test_gpu_metric_size.py -m gpu -q --timeout 900 -k "not msm" 2>&1 | tail -3
for i in 1 2 3; do for RAW in 0 1; do echo -n "MI355_NTT_RAW_SCRATCH=$RAW  "; MI355_NTT_RAW_SCRATCH=$RAW python tools/bench_ntt_knobs.py 2>/dev/null | tail -1; done; done | tee gpurun_out/r04_ntt_raw_scratch_ab.log
