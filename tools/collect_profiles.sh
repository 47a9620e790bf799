#!/bin/bash
# collect_profiles.sh <tag> -- the rocprofv3 passes behind profiles/<tag>_*.  Run on the GPU box from the repository root.
#   A  headline legs of bench.py (timed uniform MSMs at 2^26 + the NTT leg): --kernel-trace --stats; FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one
#      pass, and gpurun refuses --pmc together with sys / hip / hsa tracing); SQ counters
#   B  one create_proof of the reference's layer-4 protocol (k = 26; tests/cpp/test_plonk_replay): the NTT passes THROUGH THE BATCHED ENTRY POINTS and k_fr_gate_eval,
#      same three counter passes + kernel statistics; the program's own record carries the algorithmic side (gate_eval_process_totals)
#   C  kernel statistics of a layer-0 and a layer-3 proof (kernel time vs wall; the process includes keygen)
#   D  the same proofs' own phase totals without a profiler (keygen excluded)
# Raw per-dispatch lines go to gpurun_out/<tag>_*.txt; tools/pmc_report.py <tag> turns them into profiles/<tag>_pmc_k26.md, profiles/<tag>_gate_eval.md and
# profiles/pmc_latest.json (with the source hash bench.py checks before it re-emits the recorded traffic).
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-host-api --no-table-free --no-proof-mix --no-sizes --no-witness-like --no-batch-legs"
Q="python $ROOT/tools/pmc_query.py"
# ---- A
rm -rf /tmp/p_stats; rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o bench -- python $ROOT/bench.py --steps 5 --warmup 1 $LEGS > $OUT/${TAG}_stats_bench_line.json 2> /dev/null
python $ROOT/tools/rocpd_summary.py $(find /tmp/p_stats -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
  DB=$(find /tmp/p_$C -name "*.db" | head -1)
  for K in k_msm_accumulate k_msm_digits k_sort_l1 k_sort_l2_hist k_sort_l2_scatter k_ntt29_strided k_ntt29_final k_eval_poly; do $Q $DB $K | head -8; done > $OUT/${TAG}_pmc_$C.txt
done
SQC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
rm -rf /tmp/p_sq; rocprofv3 --kernel-trace --pmc $SQC -d /tmp/p_sq -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 $LEGS > /dev/null 2>&1
DB=$(find /tmp/p_sq -name "*.db" | head -1)
for K in k_msm_accumulate k_ntt29_strided k_ntt29_final k_sort_l1 k_sort_l2_scatter k_msm_digits; do $Q $DB $K | head -6; done > $OUT/${TAG}_pmc_sq.txt
# ---- B
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import __graft_entry__ as ge
ge.load_package().protocols.write(0, "/tmp/${TAG}_layer0.json"); ge.load_package().protocols.write(3, "/tmp/${TAG}_layer3.json")
PY
EXE=$ROOT/tests/cpp/test_plonk_replay
L4="$EXE --protocol $ROOT/tests/golden/protocol_layer4.json --proofs 1"
mkdir -p /tmp/${TAG}_l4
rm -rf /tmp/q_stats; rocprofv3 --kernel-trace --stats -d /tmp/q_stats -o l4 -- $L4 --out /tmp/${TAG}_l4 > $OUT/${TAG}_L4_stats_record.json 2> /dev/null
python $ROOT/tools/rocpd_summary.py $(find /tmp/q_stats -name "*.db" | head -1) > $OUT/${TAG}_L4_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/q_$C; rocprofv3 --kernel-trace --pmc $C -d /tmp/q_$C -o l4 -- $L4 --out /tmp/${TAG}_l4 > $OUT/${TAG}_L4_${C}_record.json 2> /dev/null
  DB=$(find /tmp/q_$C -name "*.db" | head -1)
  for K in k_fr_gate_eval k_ntt29_strided k_ntt29_final k_fr_interleave k_fr_batch_inv k_fr_scan k_kate k_fr_vec k_eval_poly; do $Q $DB $K; done > $OUT/${TAG}_L4_pmc_$C.txt
done
rm -rf /tmp/q_sq; rocprofv3 --kernel-trace --pmc $SQC -d /tmp/q_sq -o l4 -- $L4 --out /tmp/${TAG}_l4 > $OUT/${TAG}_L4_sq_record.json 2> /dev/null
DB=$(find /tmp/q_sq -name "*.db" | head -1)
for K in k_fr_gate_eval k_ntt29_strided k_ntt29_final; do $Q $DB $K; done > $OUT/${TAG}_L4_pmc_sq.txt
# ---- C
for L in 0 3; do
  mkdir -p /tmp/${TAG}_l$L; rm -rf /tmp/c_stats
  rocprofv3 --kernel-trace --stats -d /tmp/c_stats -o l -- $EXE --protocol /tmp/${TAG}_layer$L.json --out /tmp/${TAG}_l$L --proofs 1 > $OUT/${TAG}_L${L}_stats_record.json 2> /dev/null
  python $ROOT/tools/rocpd_summary.py $(find /tmp/c_stats -name "*.db" | head -1) > $OUT/${TAG}_L${L}_kernel_stats.txt
done
# ---- D  one proof's own phase totals (the library's events, keygen excluded, no profiler attached): the replay's --phase-profile
for L in 0 3; do $EXE --protocol /tmp/${TAG}_layer$L.json --out /tmp/${TAG}_l$L --phase-profile > $OUT/${TAG}_L${L}_phase_record.json 2> /dev/null; done
$EXE --protocol $ROOT/tests/golden/protocol_layer4.json --out /tmp/${TAG}_l4 --phase-profile > $OUT/${TAG}_L4_phase_record.json 2> /dev/null
tail -5 $OUT/${TAG}_kernel_stats.txt; head -3 $OUT/${TAG}_pmc_FETCH_SIZE.txt; head -3 $OUT/${TAG}_L4_pmc_FETCH_SIZE.txt; wc -l $OUT/${TAG}_*
