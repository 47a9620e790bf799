#!/bin/bash
# round-2 GPU session L: kernel timelines of uniform MSMs at 2^10 .. 2^20 (where does the latency-bound tail go)
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 10 14 18 20; do
  cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$O/p$k -o t -- python $R/tools/trace_small_msm.py $k > /dev/null 2> $R/$O/err$k.log
  cd $R
  DB=$(find $O/p$k -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_timeline.py $DB > $O/timeline_k$k.md 2>&1
  rm -rf $O/p$k
done
cat $O/timeline_k10.md $O/timeline_k18.md
