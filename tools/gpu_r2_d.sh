#!/bin/bash
# round-2 GPU session D: batched-affine candidate microbench, downsize at size, HBM-traffic and SQ counters of the final kernels
O=gpurun_out/r2d; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 tools/microbench > $O/microbench.log 2>&1
timeout 600 python tools/bench_g1fft.py 20 22 24 > $O/g1fft.log 2>&1
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/$O/pmc_$C -o p -- $BENCH --no-ntt > $R/$O/pmc_$C.json 2> $R/$O/pmc_$C.err
  DB=$(find $R/$O/pmc_$C -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_query.py $DB k_msm_accumulate > $R/$O/pmc_${C}_accumulate.txt 2>&1
  [ -n "$DB" ] && python $R/tools/pmc_query.py $DB k_sort > $R/$O/pmc_${C}_sort.txt 2>&1
  rm -rf $R/$O/pmc_$C
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $R/$O/pmc_sq -o p -- $BENCH > $R/$O/pmc_sq.json 2> $R/$O/pmc_sq.err
DB=$(find $R/$O/pmc_sq -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/pmc_query.py $DB k_ > $R/$O/pmc_sq_all.txt 2>&1
rm -rf $R/$O/pmc_sq
cd $R
cat $O/g1fft.log; grep -i "affine\|plain madd" $O/microbench.log; head -3 $O/pmc_FETCH_SIZE_accumulate.txt
