#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_buffers.py "tests/test_gpu_at_size_r4.py::test_eval_polynomial_batch_equals_single_calls_and_oracle" "tests/test_plonk_protocol.py::test_gpu_proof_bytes_equal_the_cpu_restatement" -x -q -m gpu --durations=5 2>&1 | tail -15 > gpurun_out/r05_job3_tests.log
tail -5 gpurun_out/r05_job3_tests.log
python tools/bench_narrow_uploads.py 24 > gpurun_out/r05_narrow_uploads.json 2>gpurun_out/r05_narrow_err.log; cat gpurun_out/r05_narrow_uploads.json | cut -c1-1500
python - <<'PY'
import sys, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as ge
zk = ge.load_package()
def slim(r): return {k: r.get(k) for k in ("ok", "error", "k", "resident_ms", "first_proof_ms", "step_ms", "plan", "pk_cosets", "window_table_bases", "sparse_uploads", "pinned_witness", "msm", "coset_ntt", "gate_launches", "gate_eval_process_totals", "hbm", "circuit")}
out = {}
# A/B: common-prefix groups of the plan compiler (layer 0 and layer 3, full size)
for layer in (0, 3):
    for pm in ("16", "0"):
        out[f"prefix_L{layer}_min{pm}"] = slim(zk.replay.run(layer, env={"MI355_PLAN_PREFIX_MIN": pm}))
# sparse uploads on a 60 %-zero witness (assigned gates on 40 % of the rows), pageable and page-locked
for tag, args in (("dense_plain", []), ("sparse60_plain", ["--assign-density", "0.4"]), ("sparse60_sparse", ["--assign-density", "0.4", "--sparse-uploads", "--upload-threads", "2"]),
                  ("sparse60_pinned", ["--assign-density", "0.4", "--pinned-witness"])):
    out["uploads_L0_" + tag] = slim(zk.replay.run(0, args=args))
# layer-0 sensitivity band
for adv, lk, deg, fx, pc in ((400, 30, 5, 60, 75), (1600, 120, 9, 240, 300)):
    out[f"band_L0_{adv}_{lk}_{deg}"] = slim(zk.replay.run(0, advice=adv, lookups=lk, degree=deg, fixed=fx, perm_columns=pc, timeout=1500))
json.dump(out, open("gpurun_out/r05_job3_ab.json", "w"), indent=1)
for k, v in out.items():
    print(k, v.get("ok"), v.get("resident_ms"), (v.get("step_ms") or {}).get("2_3_advice_lookup_commits"), (v.get("step_ms") or {}).get("7_quotient"), v.get("plan"), v.get("pk_cosets"), (v.get("hbm") or {}).get("peak_used_gib"), v.get("error"))
PY
