#!/bin/bash
# round-2 second sitting: SQ counters (one --pmc pass, kernel trace only) of the final kernels: uniform 2^26 MSM + 2^26 NTT + the k = 24 leg
O=gpurun_out/r2sq; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $R/$O/pmc_sq -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-witness-like --no-sizes > $R/$O/pmc_sq.json 2> $R/$O/pmc_sq.err
DB=$(find $R/$O/pmc_sq -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/pmc_query.py $DB k_ > $R/$O/pmc_sq_all.txt 2>&1
rm -rf $R/$O/pmc_sq
cd $R
grep -E "k_msm_digits|k_sort|k_msm_accumulate|k_msm_segfix|k_msm_bucket_reduce|k_ntt29|k_eval" $O/pmc_sq_all.txt | head -30
