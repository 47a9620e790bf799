"""r06 job 5: two-pass plans at 2^19 / 2^20 (MI355_NTT_TWO_LEVEL_MAX_LOG=20) with the coset fold in place: microbenchmark + layer 0 / 3 proofs"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
def run(cmd, env=None, timeout=1500):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)
    return r.returncode, r.stdout, r.stderr
MB = r'''
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib, capi = zk._capi.lib(), zk._capi
from tests.test_gpu_properties import dev_scalars
res = {}
for k, M in ((20, 64), (19, 64), (21, 32)):
    n = 1 << k; dom = h2.EvaluationDomain(9, k)
    srcs = [dev_scalars(n, 10 + i) for i in range(M)]; dsts = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(M)]
    fac = h2.fr(h2.FR_ZETA * pow(h2.fr_to_int(dom.extended_omega), 3, h2.R_MOD) %% h2.R_MOD)
    call = lambda: capi.check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * M)(*[d.data_ptr() for d in dsts]), (C.c_void_p * M)(*[s.data_ptr() for s in srcs]), M, k, capi.ptr(fac), capi.ptr(dom.omega)))
    call(); call(); capi.check(lib.mi355_synchronize())
    t = time.perf_counter()
    for _ in range(5): call()
    capi.check(lib.mi355_synchronize()); dt = (time.perf_counter() - t) / 5
    plain = lambda: h2.best_fft_many(dsts, dom.omega, k)
    plain(); capi.check(lib.mi355_synchronize()); t = time.perf_counter()
    for _ in range(5): plain()
    capi.check(lib.mi355_synchronize()); dp = (time.perf_counter() - t) / 5
    res["k%%d" %% k] = {"batch": M, "coset_us": dt / M * 1e6, "plain_us": dp / M * 1e6}
import json; print("MB" + json.dumps(res))
''' % ROOT
for tag, env in (("three_pass", {}), ("two_pass", {"MI355_NTT_TWO_LEVEL_MAX_LOG": "20"}), ("three_pass_again", {})):
    rc, so, se = run([sys.executable, "-c", MB], env)
    line = next((l for l in so.splitlines() if l.startswith("MB")), None)
    out["microbench_" + tag] = json.loads(line[2:]) if line else (so + se)[-600:]
    print(tag, out["microbench_" + tag], flush=True)
import __graft_entry__ as ge
zk = ge.load_package()
for layer in (0,):
    for tag, env in (("three_pass", {}), ("two_pass", {"MI355_NTT_TWO_LEVEL_MAX_LOG": "20"})):
        rec = zk.replay.run(layer, args=["--proofs", "3"], env=env, timeout=1200)
        keep = {k: rec.get(k) for k in ("ok", "k", "resident_ms", "step_ms", "error")}
        out[f"layer{layer}_{tag}"] = keep
        print(layer, tag, json.dumps(keep), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_two_pass_ab.json"), "w"), indent=1)
