#!/bin/bash
# round-2 GPU session G: final source -- whole suite, default bench twice, variant 0 for reference, kernel trace
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -rf > $O/pytest_all.log 2>&1
echo "rc=$?" >> $O/pytest_all.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
MI355_ACC_VARIANT=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-ntt > $O/bench_v0.json 2>> $O/bench.err
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-ntt > $O/bench_v4.json 2>> $O/bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-proof-mix > $R/$O/bench_prof.json 2> $R/$O/prof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats.md > /dev/null
rm -rf $O/prof
tail -3 $O/pytest_all.log; head -c 300 $O/bench.json
