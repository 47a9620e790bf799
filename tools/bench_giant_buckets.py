"""degenerate scalar distributions (giant buckets): all ones (selector columns), all equal, ones and zeros -- phase times at 2^24"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
from oracle import cref
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
lib, check = zk._capi.lib(), zk._capi.check
k = 24; n = 1 << k
p = h2.ParamsKZG.setup(k, 0x5343524f4c4c0001); p.precompute()
one = torch.from_numpy(cref.fr_mont(1).view(np.int64)).cuda()
rnd = torch.from_numpy(cref.fr_mont(0x1234567890abcdef1234567890abcdef1234567890abcdef).view(np.int64)).cuda()
uni = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); uni[:, 3] &= (1 << 59) - 1
cases = {"uniform": uni, "all ones": one.repeat(n, 1).contiguous(), "all equal (random value)": rnd.repeat(n, 1).contiguous()}
half = one.repeat(n, 1).contiguous(); half[::2] = 0; cases["ones and zeros"] = half
tab = torch.from_numpy(np.stack([cref.fr_mont((0x9E3779B97F4A7C15F39CC0605CEDC8341082276BF3A27251F86C6A11D0C18E95 * (7 * i_ + 3)) % (1 << 248)) for i_ in range(16)]).view(np.int64)).cuda()   # fixed values: runs are comparable
cases["16 distinct random values"] = tab[torch.randint(0, 16, (n,), device="cuda")].contiguous()
sparse = uni.clone(); sparse[torch.rand(n, device="cuda") < 0.99] = 0; cases["99% zeros"] = sparse
cases["all r - 1"] = torch.from_numpy(cref.fr_mont(cref.R_MOD - 1 if hasattr(cref, "R_MOD") else 21888242871839275222246405745257275088548364400416034343698204186575808495616).view(np.int64)).cuda().repeat(n, 1).contiguous()
small = torch.from_numpy(np.stack([cref.fr_mont(v) for v in range(256)]).view(np.int64)).cuda()
cases["bytes (uniform 0..255)"] = small[torch.randint(0, 256, (n,), device="cuda")].contiguous()
check(lib.mi355_profile_enable(1))
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, sc in cases.items():
    if only and only not in name: continue
    p.commit(sc); torch.cuda.synchronize(); check(lib.mi355_profile_reset()); t = time.perf_counter()
    for _ in range(3): p.commit(sc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    import ctypes as C
    ph = {}
    for nm in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce"):
        ms, cnt = C.c_double(), C.c_uint64(); check(lib.mi355_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))); ph[nm] = round(ms.value / max(1, cnt.value), 3)
    print(f"{name}: {dt*1e3:.2f} ms {ph}", flush=True)
