"""r06 job 10: the first level of a 2^26 transform on the [k][column] table (2.4 GB, built only into spare HBM) against the lo x hi product (MI355_NTT_DIRECT2_MAX_LOG=25)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
def run(cmd, env=None, timeout=1500):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)
    return r.returncode, r.stdout, r.stderr
rc, so, se = run([sys.executable, "-m", "pytest", "tests/test_gpu_metric_size.py", "tests/test_gpu_faults.py", "-m", "gpu", "-x", "-q"])
out["parity"] = so[-300:]; print(so[-400:], se[-300:], flush=True)
MB = r'''
import sys, os, time, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib, capi = zk._capi.lib(), zk._capi
from tests.test_gpu_properties import dev_scalars
res = {}
for k in (26, 25):
    n = 1 << k; dom = h2.EvaluationDomain(2, k); poly = dev_scalars(n, 7)
    rec = {}
    for name, fn in (("fwd", lambda: dom.coeff_to_lagrange(poly)), ("inv", lambda: dom.lagrange_to_coeff(poly))):
        fn(); fn(); capi.check(lib.mi355_synchronize()); t = time.perf_counter()
        for _ in range(10): fn()
        capi.check(lib.mi355_synchronize()); rec[name + "_ms"] = (time.perf_counter() - t) / 10 * 1e3
    res["k%%d" %% k] = rec
import json; print("MB" + json.dumps(res))
''' % ROOT
for tag, env in (("table", {}), ("lo_x_hi", {"MI355_NTT_DIRECT2_MAX_LOG": "25"}), ("table_again", {}), ("lo_x_hi_again", {"MI355_NTT_DIRECT2_MAX_LOG": "25"})):
    rc, so, se = run([sys.executable, "-c", MB], env)
    line = next((l for l in so.splitlines() if l.startswith("MB")), None)
    out[tag] = json.loads(line[2:]) if line else (so + se)[-600:]
    print(tag, out[tag], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_direct2_k26_ab.json"), "w"), indent=1)
