"""NTT and coset-NTT times at the sizes of the many-column layers (2^19 .. 2^22), single transforms and a batch of 32 resident polynomials through
mi355_coset_ntt_fr_batch_dev (what step 7 of create_proof issues per coset part).  A/B: MI355_NTT_TWO_LEVEL_MAX_LOG=20 python tools/bench_ntt_small.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
lib, capi = zk._capi.lib(), zk._capi
print("MI355_NTT_TWO_LEVEL_MAX_LOG =", os.environ.get("MI355_NTT_TWO_LEVEL_MAX_LOG", "(default 18)"))
for k in (19, 20, 21, 22):
    n = 1 << k
    dom = h2.EvaluationDomain(5, k)
    polys = [torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda") for _ in range(32)]
    for p in polys: p[:, 3] &= (1 << 59) - 1
    dsts = [torch.empty_like(p) for p in polys]
    a = polys[0].clone()
    dom.coeff_to_lagrange(a); dom.lagrange_to_coeff(a); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): dom.coeff_to_lagrange(a); dom.lagrange_to_coeff(a)
    capi.check(lib.mi355_synchronize()); torch.cuda.synchronize(); single = (time.perf_counter() - t) / 100
    factor = h2.fr(h2.FR_ZETA)
    da = (C.c_void_p * 32)(*[d.data_ptr() for d in dsts]); sa = (C.c_void_p * 32)(*[p.data_ptr() for p in polys])
    capi.check(lib.mi355_coset_ntt_fr_batch_dev(da, sa, 32, k, capi.ptr(factor), capi.ptr(dom.omega)))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): capi.check(lib.mi355_coset_ntt_fr_batch_dev(da, sa, 32, k, capi.ptr(factor), capi.ptr(dom.omega)))
    torch.cuda.synchronize(); batch = (time.perf_counter() - t) / (5 * 32)
    print(f"k={k}: ntt {single*1e3:.4f} ms   coset ntt in a batch of 32: {batch*1e3:.4f} ms each", flush=True)
