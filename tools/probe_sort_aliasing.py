"""does the sorter's time depend on the region stride?  uniform commitments over n = 2^26 and a few nearby lengths (the level-1 scatter writes 1024 streams whose
starts are one region apart: 3 * 2^18 entries * 8 B = 6 MiB exactly at 2^26, c = 22); phase times per length, three repetitions each"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib = zk._capi.lib(); check = zk._capi.check; ptr = zk._capi.ptr
k = 26
p = h2.ParamsKZG.setup(k, 0x5343524F4C4C0001); p.precompute(lagrange=False)
sc = torch.randint(0, 2**62, (1 << k, 4), dtype=torch.int64, device="cuda"); sc[:, 3] &= (1 << 59) - 1
def prof(name):
    ms, cnt = C.c_double(), C.c_uint64(); check(lib.mi355_profile_get(name.encode(), C.byref(ms), C.byref(cnt))); return ms.value / max(1, cnt.value)
out = np.zeros(12, dtype=np.uint64)
for n in (1 << 26, (1 << 26) - 4096, (1 << 26) - 123456, 66000000, 65000000, 1 << 26):
    for rep in range(2):
        check(lib.mi355_msm_g1_dev(p._g, 0, ptr(sc), n, ptr(out)))
    check(lib.mi355_profile_reset()); check(lib.mi355_profile_enable(1))
    for rep in range(3):
        check(lib.mi355_msm_g1_dev(p._g, 0, ptr(sc), n, ptr(out)))
    check(lib.mi355_profile_enable(0))
    tot = prof("msm_total")
    print(f"n={n}: total {tot:.2f} ms ({tot / n * 1e6:.4f} ns/pair)  digits={prof('msm_digits'):.2f} sort={prof('msm_sort'):.2f} ({prof('msm_sort') / n * 1e6:.4f} ns/pair) accumulate={prof('msm_accumulate'):.2f} reduce={prof('msm_reduce'):.2f}", flush=True)
