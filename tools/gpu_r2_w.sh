#!/bin/bash
# round-2 GPU session W: why is the giant-bucket fix-up slow on the 16-distinct-values column -- timelines with other segment factors
O=gpurun_out/r2w; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sf in 4 8 32; do
  cd /tmp && MI355_SEG_FACTOR=$sf timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof$sf -o d16 -- python $R/tools/bench_giant_buckets.py distinct > $R/$O/run$sf.log 2> $R/$O/err$sf.log
  cd $R
  DB=$(find $O/prof$sf -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_timeline.py $DB k_msm_accumulate k_msm_final29 > $O/timeline_sf$sf.md 2>&1
  rm -rf $O/prof$sf
  echo "== MI355_SEG_FACTOR=$sf"; grep -v amdgpu $O/run$sf.log; cat $O/timeline_sf$sf.md
done
