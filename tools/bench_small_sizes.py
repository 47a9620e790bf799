"""latency of small MSMs / NTTs (device-resident and host-pointer entry points): guidance for the offload thresholds of the binding"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
import ctypes as C
lib = zk._capi.lib()
def prof(name):
    ms, cnt = C.c_double(), C.c_uint64(); zk._capi.check(lib.mi355_profile_get(name.encode(), C.byref(ms), C.byref(cnt))); return ms.value
for k in [int(x) for x in (sys.argv[1:] or (10, 12, 14, 16, 18, 20, 22))]:
    n = 1 << k
    p = h2.ParamsKZG.setup(k, 0x5343524f4c4c0001); p.precompute()
    sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); sc[:, 3] &= (1 << 59) - 1
    host = sc.cpu().numpy().view(np.uint64)
    a = sc.clone(); host2 = host.copy()   # transforms run in place on a preallocated buffer (a fresh numpy copy per call would time page faults, not PCIe)
    w = h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << (28 - k), h2.R_MOD))
    res = {}
    for name, fn in (("msm_dev", lambda: p.commit(sc)), ("msm_host", lambda: p.commit(host)), ("ntt_dev", lambda: h2.best_fft(a, w, k)), ("ntt_host", lambda: h2.best_fft(host2, w, k))):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t) / 10 * 1e3
    zk._capi.check(lib.mi355_profile_reset()); zk._capi.check(lib.mi355_profile_enable(1))
    for _ in range(5): p.commit(sc)
    zk._capi.check(lib.mi355_profile_enable(0))
    ph = " ".join(f"{q[4:]}={prof(q) / 5:.3f}" for q in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce"))
    print(f"k={k}: " + "  ".join(f"{a_}={b_:.3f} ms" for a_, b_ in res.items()) + "  | phases ms: " + ph, flush=True)
    p.release()
