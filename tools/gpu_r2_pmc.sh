#!/bin/bash
# round-2 second sitting: HBM-traffic counters (separate --pmc passes, kernel trace only) of the final kernels on the headline-only workload
O=gpurun_out/r2pmc; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-sizes --no-ntt --no-witness-like"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/$O/pmc_$C -o p -- $BENCH > $R/$O/pmc_$C.json 2> $R/$O/pmc_$C.err
  DB=$(find $R/$O/pmc_$C -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_query.py $DB k_msm > $R/$O/pmc_${C}_msm.txt 2>&1
  [ -n "$DB" ] && python $R/tools/pmc_query.py $DB k_sort > $R/$O/pmc_${C}_sort.txt 2>&1
  rm -rf $R/$O/pmc_$C
done
cd $R
cat $O/pmc_FETCH_SIZE_msm.txt $O/pmc_WRITE_SIZE_msm.txt $O/pmc_FETCH_SIZE_sort.txt $O/pmc_WRITE_SIZE_sort.txt
