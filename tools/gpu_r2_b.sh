#!/bin/bash
# round-2 GPU session B: chained-mad A/B (microbench, accumulate variant, NTT variant library), NTT changes validated, default bench, kernel trace
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 tools/microbench > $O/microbench.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_headline.py -m gpu -q -x -k "fft or ntt or domain or coset or extended or quotient or headline" > $O/pytest_ntt.log 2>&1
echo "rc=$?" >> $O/pytest_ntt.log
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
MI355_ACC_VARIANT=4 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free > $O/bench_acc4.json 2>> $O/bench.err
MI355ZK_LIB=$PWD/scroll-prover_amd/libmi355zk_chain.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free > $O/bench_chainlib.json 2>> $O/bench.err
MI355ZK_LIB=$PWD/scroll-prover_amd/libmi355zk_chain.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fft or ntt or multiexp" > $O/pytest_chainlib.log 2>&1
echo "rc=$?" >> $O/pytest_chainlib.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(ls $O/prof/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats.md > /dev/null
find $O/prof -name "*.db" -size +20M -delete
tail -3 $O/pytest_ntt.log; tail -2 $O/pytest_chainlib.log; grep -i "mul_c\|chained\|Fq29::mul (" $O/microbench.log
