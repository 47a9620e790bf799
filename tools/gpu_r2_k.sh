#!/bin/bash
# round-2 GPU session K: division-step inverse + shared-inversion precompute -- parity file, small-size latencies (default build vs the tail on the
# un-chained multiplier), default bench
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_regression_golden.py -m gpu -q -rf -x > $O/pytest_parity.log 2>&1 ) 2> $O/pytest.time
echo "rc=$?" >> $O/pytest_parity.log
timeout 300 python tools/bench_small_sizes.py > $O/small_default.log 2>&1
MI355ZK_LIB=$R/scroll-prover_amd/libmi355zk_nochain_tail.so timeout 300 python tools/bench_small_sizes.py > $O/small_nochain_tail.log 2>&1
( time timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -3 $O/pytest_parity.log; cat $O/pytest.time; cat $O/small_default.log; echo; cat $O/small_nochain_tail.log; head -c 300 $O/bench.json
