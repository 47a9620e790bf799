#!/bin/bash
# round-2 final GPU session: whole suite, smoke, long fuzz (default / three device slots + sliced host path / segmented fix-up forced), default bench,
# kernel statistics of the default bench run and of a headline-only run (only the timed uniform 2^26 MSMs are launched)
O=gpurun_out/r2final; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q -rf > $O/pytest_all.log 2>&1 ) 2> $O/pytest.time
echo "rc=$?" >> $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python tools/fuzz_gpu.py 1200 201 > $O/fuzz_default.log 2>&1
MI355_ALLOW_DUP_DEVICES=1 FUZZ_DEVICES=0,0,0 MI355_SHARD_MIN_LOG=4 MI355_HOST_SLICE_MIN_LOG=6 timeout 600 python tools/fuzz_gpu.py 500 202 > $O/fuzz_multi_slices.log 2>&1
MI355_FIXUP_MODE=1 timeout 600 python tools/fuzz_gpu.py 500 203 > $O/fuzz_segfix.log 2>&1
( time timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-proof-mix > $R/$O/bench_prof.json 2> $R/$O/prof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats.md > /dev/null
rm -rf $O/prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-proof-mix --no-host-api --no-table-free --no-sizes --no-ntt --no-witness-like > $R/$O/bench_headline_only.json 2> $R/$O/prof2.err
cd $R
DB=$(find $O/prof2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_headline_only.md > /dev/null
rm -rf $O/prof2
grep -n "passed\|failed" $O/pytest_all.log | tail -2; tail -1 $O/smoke.log; tail -1 $O/fuzz_default.log; tail -1 $O/fuzz_multi_slices.log; tail -1 $O/fuzz_segfix.log; head -2 $O/bench.time; head -c 200 $O/bench.json; echo; head -14 $O/kernel_stats_headline_only.md | cut -c1-120
