#!/bin/bash
# round-2 GPU session N: GLV in the G1 DFT -- oracle parity (k <= 10), downsize against the closed-form Lagrange basis at k = 16 .. 24, C++ mirror, small-size table
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_cpp_mirror.py tests/test_gpu_multi.py -m gpu -q -rf -k "g1_fft or downsize or lagrange or mirror or shim or params" > $O/pytest_g1fft.log 2>&1
echo "rc=$?" >> $O/pytest_g1fft.log
timeout 600 python tools/bench_g1fft.py 16 20 22 24 > $O/g1fft.log 2>&1
timeout 300 python tools/bench_small_sizes.py > $O/small_sizes.log 2>&1
tail -3 $O/pytest_g1fft.log; grep -v amdgpu $O/g1fft.log; grep -v amdgpu $O/small_sizes.log
