"""r06 job 2: the folded coset shift -- parity, then A/B (MI355_NTT_COSET_FOLD_MAX_LOG=0 restores k_distribute_powers) on the coset batch microbenchmark and on layers 0 / 3 / 5"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
def run(cmd, env=None, timeout=1500):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)
    return r.returncode, r.stdout, r.stderr
rc, so, se = run([sys.executable, "-m", "pytest", "tests/test_gpu_at_size_r4.py", "tests/test_gpu_buffers.py", "-m", "gpu", "-x", "-q", "-k", "coset or narrow or slabs or batch"])
out["parity"] = so[-800:]; print(so[-1500:], se[-500:], flush=True)
rc, so, se = run([sys.executable, "-m", "pytest", "tests/test_plonk_protocol.py", "-m", "gpu", "-x", "-q", "-k", "cpu_restatement or invisible or initial_state"])
out["plonk_bytes"] = so[-400:]; print(so[-600:], se[-300:], flush=True)
MB = r'''
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib, capi = zk._capi.lib(), zk._capi
from tests.test_gpu_properties import dev_scalars
res = {}
for k, M in ((20, 64), (21, 32), (22, 16), (18, 128)):
    n = 1 << k; dom = h2.EvaluationDomain(9, k)
    srcs = [dev_scalars(n, 10 + i) for i in range(M)]; dsts = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(M)]
    fac = h2.fr(h2.FR_ZETA * pow(h2.fr_to_int(dom.extended_omega), 3, h2.R_MOD) %% h2.R_MOD)
    call = lambda: capi.check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * M)(*[d.data_ptr() for d in dsts]), (C.c_void_p * M)(*[s.data_ptr() for s in srcs]), M, k, capi.ptr(fac), capi.ptr(dom.omega)))
    call(); call(); capi.check(lib.mi355_synchronize())
    t = time.perf_counter()
    for _ in range(5): call()
    capi.check(lib.mi355_synchronize()); dt = (time.perf_counter() - t) / 5
    res["k%%d" %% k] = {"batch": M, "us_per_transform": dt / M * 1e6}
import json; print("MB" + json.dumps(res))
''' % ROOT
for tag, env in (("fold", {}), ("separate", {"MI355_NTT_COSET_FOLD_MAX_LOG": "0"})):
    rc, so, se = run([sys.executable, "-c", MB], env)
    line = next((l for l in so.splitlines() if l.startswith("MB")), None)
    out["microbench_" + tag] = json.loads(line[2:]) if line else (so + se)[-600:]
    print(tag, out["microbench_" + tag], flush=True)
import __graft_entry__ as ge
zk = ge.load_package()
for layer in (0, 3, 5):
    for tag, env in (("fold", {}), ("separate", {"MI355_NTT_COSET_FOLD_MAX_LOG": "0"})):
        rec = zk.replay.run(layer, args=["--phase-profile"], env=env, timeout=1200)
        keep = {k: rec.get(k) for k in ("ok", "k", "resident_ms", "first_proof_ms", "step_ms", "phase_profile", "error")}
        keep["gate"] = rec.get("gate_eval_process_totals")
        out[f"layer{layer}_{tag}"] = keep
        print(layer, tag, json.dumps(keep), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_coset_fold_ab.json"), "w"), indent=1)
