#!/bin/bash
# round-2 GPU session DD: full inter-level twiddle table for big NTT levels -- parity (2^24 full vector vs oracle, round trips to 2^28), A/B
O=gpurun_out/r2dd; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -q -x -k "ntt or fft or domain or coset or extended or quotient" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
for m in 26 0 26 0; do
  echo "== MI355_NTT_DIRECT2_MAX_LOG=$m"
  MI355_NTT_DIRECT2_MAX_LOG=$m timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-proof-mix --no-host-api --no-table-free --no-witness-like 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ntt 2^26', round(d['ntt']['ms_per_transform'],3), 'ms; 2^24', round(d['sizes']['k24']['ntt_ms_per_transform'],3), '; 2^20', round(d['sizes']['k20']['ntt_ms_per_transform'],4), '; msm', round(d['ms_per_step'],2))"
done > $O/ab.log 2>&1
tail -3 $O/pytest.log; cat $O/ab.log
