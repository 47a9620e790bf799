#!/bin/bash
# round-2 GPU session F: column-block chained multiplier, A/B on one box (alternating runs): accumulate variant 4, NTT / everything in the chain library
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout 400 tools/microbench > $O/microbench.log 2>&1
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix"
for i in 1 2; do
  timeout 300 $B > $O/base_$i.json 2>> $O/bench.err
  MI355_ACC_VARIANT=4 timeout 300 $B > $O/acc4_$i.json 2>> $O/bench.err
  MI355ZK_LIB=$PWD/scroll-prover_amd/libmi355zk_chain.so MI355_ACC_VARIANT=4 timeout 300 $B > $O/chainlib_$i.json 2>> $O/bench.err
done
MI355ZK_LIB=$PWD/scroll-prover_amd/libmi355zk_chain.so MI355_ACC_VARIANT=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x > $O/pytest_chainlib.log 2>&1
echo "rc=$?" >> $O/pytest_chainlib.log
grep -E "Fq29::mul |mul_c|sqr_c|madd chain [234] waves" $O/microbench.log; tail -2 $O/pytest_chainlib.log
