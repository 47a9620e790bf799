#!/bin/bash
# round-2 GPU session V: kernel timeline of the "16 distinct values" column at 2^24 (192 buckets of 10^6 entries)
O=gpurun_out/r2v; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof -o d16 -- python $R/tools/bench_giant_buckets.py distinct > $R/$O/run.log 2> $R/$O/err.log
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB > $O/timeline_distinct16_k24.md 2>&1
rm -rf $O/prof
grep -v amdgpu $O/run.log; cat $O/timeline_distinct16_k24.md
