"""r06 job 9: the [k][column] twiddle layout for SMALL levels too (MI355_NTT_DIRECT2_MIN_LOG=0; default 21): parity under the knob, then transform times 2^16 ... 2^26"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
def run(cmd, env=None, timeout=1500):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=e, timeout=timeout)
    return r.returncode, r.stdout, r.stderr
rc, so, se = run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_at_size_r4.py", "tests/test_gpu_buffers.py", "tests/test_gpu_metric_size.py", "-m", "gpu", "-x", "-q", "-k", "fft or ntt or coset or extended or batch"], {"MI355_NTT_DIRECT2_MIN_LOG": "0"})
out["parity_under_knob"] = so[-400:]; print(so[-600:], se[-300:], flush=True)
MB = r'''
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib, capi = zk._capi.lib(), zk._capi
from tests.test_gpu_properties import dev_scalars
res = {}
for k in (16, 18, 20, 22, 24, 26):
    n = 1 << k; dom = h2.EvaluationDomain(2, k); M = max(1, min(64, (1 << 26) >> k))
    polys = [dev_scalars(n, 10 + i) for i in range(M)]
    f = lambda: h2.best_fft_many(polys, dom.omega, k)
    i_ = lambda: h2.best_fft_many(polys, dom.omega_inv, k, divisor=dom.ifft_divisor)
    rec = {}
    for name, fn in (("fwd", f), ("inv", i_)):
        fn(); fn(); capi.check(lib.mi355_synchronize()); t = time.perf_counter()
        for _ in range(5): fn()
        capi.check(lib.mi355_synchronize()); rec[name + "_us"] = (time.perf_counter() - t) / 5 / M * 1e6
    res["k%%d" %% k] = dict(batch=M, **rec)
import json; print("MB" + json.dumps(res))
''' % ROOT
for tag, env in (("default_min21", {}), ("min0", {"MI355_NTT_DIRECT2_MIN_LOG": "0"}), ("default_again", {}), ("min0_again", {"MI355_NTT_DIRECT2_MIN_LOG": "0"})):
    rc, so, se = run([sys.executable, "-c", MB], env)
    line = next((l for l in so.splitlines() if l.startswith("MB")), None)
    out[tag] = json.loads(line[2:]) if line else (so + se)[-600:]
    print(tag, out[tag], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_direct2_small_levels_ab.json"), "w"), indent=1)
