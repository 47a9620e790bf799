#!/bin/bash
# round-2 GPU session Z: state after the second sitting -- whole suite, smoke, default bench, kernel statistics of the bench run
O=gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q -rf > $O/pytest_all.log 2>&1 ) 2> $O/pytest.time
echo "rc=$?" >> $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-proof-mix > $R/$O/bench_prof.json 2> $R/$O/prof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats.md > /dev/null
rm -rf $O/prof
grep -n "passed\|failed" $O/pytest_all.log | tail -2; cat $O/pytest.time | head -2; tail -1 $O/smoke.log; cat $O/bench.time | head -2; head -c 300 $O/bench.json; echo; head -30 $O/kernel_stats.md
