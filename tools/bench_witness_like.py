"""phase times of ONE witness-like commitment at 2^k (60 % zero, 20 % small, 10 % 64-bit from 4096 distinct values, 10 % uniform; bench.py's generator)
and of two degenerate columns, window tables on: the workload for calibrating the accumulate segment / fix-up knobs (MI355_SEG_MIN,
MI355_FIXUP_SERIAL_MAX, MI355_FIXUP_LANES_MAX_LOG are read at init, so every setting is its own process)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
import bench
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib = zk._capi.lib(); check = zk._capi.check; ptr = zk._capi.ptr
k = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << k
p = h2.ParamsKZG.setup(k, 0x5343524F4C4C0001); p.precompute(lagrange=False)
dev = torch.device("cuda", 0)
def prof(name):
    ms, cnt = C.c_double(), C.c_uint64(); check(lib.mi355_profile_get(name.encode(), C.byref(ms), C.byref(cnt))); return ms.value
cols = {"witness-like": bench.witness_like_scalars(n, 0x5343524F4C4C0004, dev, h2)}
small = torch.from_numpy(np.stack([h2.fr(v) for v in range(256)]).view(np.int64)).to(dev)
cols["bytes"] = torch.cat([torch.index_select(small, 0, torch.randint(0, 256, (min(1 << 22, n - lo),), device=dev)) for lo in range(0, n, 1 << 22)]).contiguous()   # torch's gather refuses 2^26 indices in one launch
cols["all ones"] = small[1].repeat(n, 1).contiguous()
cols["uniform"] = bench.rand_scalars(n, 7, dev)
for name, sc in cols.items():
    p.commit(sc); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): p.commit(sc)
    dt = (time.perf_counter() - t0) / 3 * 1e3
    check(lib.mi355_profile_reset()); check(lib.mi355_profile_enable(1)); p.commit(sc); check(lib.mi355_profile_enable(0))
    print(f"{name}: {dt:.2f} ms  " + " ".join(f"{q[4:]}={prof(q):.2f}" for q in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce")), flush=True)
