#!/bin/bash
# collect_profiles.sh <tag> -- the rocprofv3 passes behind profiles/<tag>_*: kernel statistics of the headline run, HBM traffic counters
# (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel trace only -- gpurun refuses --pmc together with sys/hip/hsa tracing) and SQ counters
# of the MSM and NTT kernels at 2^26.  Run on the GPU box from the repository root; raw per-dispatch lines go to gpurun_out/<tag>_*.txt,
# from which the tables under profiles/ are written by hand (counter corrections: profiles/r01_pmc_msm_k26.md, MI355X_MICROARCH.md "HBM").
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-sizes --no-witness-like --no-batch-legs"
# 1. per-kernel statistics, headline legs only (timed uniform MSMs + the NTT leg)
rm -rf /tmp/p_stats; rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o bench -- python $ROOT/bench.py --steps 5 --warmup 1 $LEGS > $OUT/${TAG}_stats_bench_line.json 2> /dev/null
python $ROOT/tools/rocpd_summary.py $(find /tmp/p_stats -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats.txt
# 2./3. HBM traffic: one un-warmed commitment + one forward and inverse transform per pass
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
  DB=$(find /tmp/p_$C -name "*.db" | head -1)
  for K in k_msm_accumulate k_msm_digits k_sort_l1 k_sort_l2_hist k_sort_l2_scatter k_ntt29_strided k_ntt29_final k_eval_poly; do python $ROOT/tools/pmc_query.py $DB $K | head -8; done > $OUT/${TAG}_pmc_$C.txt
done
# 4. SQ counters
rm -rf /tmp/p_sq; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d /tmp/p_sq -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
DB=$(find /tmp/p_sq -name "*.db" | head -1)
for K in k_msm_accumulate k_ntt29_strided k_ntt29_final k_sort_l1 k_sort_l2_scatter k_msm_digits; do python $ROOT/tools/pmc_query.py $DB $K | head -6; done > $OUT/${TAG}_pmc_sq.txt
tail -5 $OUT/${TAG}_kernel_stats.txt; head -3 $OUT/${TAG}_pmc_FETCH_SIZE.txt
