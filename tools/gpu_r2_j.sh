#!/bin/bash
# round-2 GPU session J (second sitting of the round): batch_normalize parity, default bench with the k = 20 / 24 legs, kernel timeline of a 2^20 MSM
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rf -k "batch_normalize or g1_sum or best_multiexp_matches" > $O/pytest_new.log 2>&1
echo "rc=$?" >> $O/pytest_new.log
( time timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof20 -o b20 -- python $R/bench.py --logn 20 --steps 20 --warmup 3 --no-cpu-baseline --no-proof-mix --no-host-api --no-table-free > $R/$O/bench20.json 2> $R/$O/prof20.err
cd $R
DB=$(find $O/prof20 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_k20.md > /dev/null && python tools/rocpd_timeline.py $DB > $O/timeline_k20.md 2>&1 && python tools/rocpd_timeline.py $DB k_ntt29_strided k_ntt29_final > $O/timeline_ntt20.md 2>&1
rm -rf $O/prof20
tail -3 $O/pytest_new.log; cat $O/bench.time; head -c 400 $O/bench.json; echo; cat $O/timeline_k20.md
