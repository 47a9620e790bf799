#!/bin/bash
# round-2 GPU session R: kernel timeline of the witness-like 2^26 MSM
O=gpurun_out/r2r; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof -o wl -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-proof-mix --no-host-api --no-table-free --no-sizes --no-ntt > $R/$O/bench.json 2> $R/$O/err.log
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB > $O/timeline_wl_k26.md 2>&1
rm -rf $O/prof
cat $O/timeline_wl_k26.md
