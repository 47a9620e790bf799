"""A full params_map resident at once: what `Prover::load_params_map(dir, &[20, 21, 24, 25, 26])` followed by clone + downsize per
degree [REF integration/tests/integration.rs:12-22], [REF bin/src/trace_prover.rs:35-36] leaves in HBM with window tables on every basis --
g of the largest degree registered once (prefix views for the smaller degrees share its memory and tables), one g_lagrange per degree rebuilt on
the device (inverse DFT over G1 points), window tables per Lagrange basis.  Prints the time of each step, HBM in use after it, and checks one
commitment per degree and basis in the field.  ~1 min on one MI355X."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import cref
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
TAU = 0x5343524F4C4C0001
def used_gib():
    free, total = torch.cuda.mem_get_info(); return (total - free) / 2**30
def check(params, k, lagrange):
    n = 1 << k
    g = torch.Generator(device="cuda"); g.manual_seed(k * 2 + lagrange)
    sc = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda", generator=g); sc[:, 3] &= (1 << 59) - 1
    t = time.perf_counter(); out = params.commit_lagrange(sc) if lagrange else params.commit(sc); ms = (time.perf_counter() - t) * 1e3
    coeffs = sc
    if lagrange:
        coeffs = sc.clone(); h2.EvaluationDomain(2, k).lagrange_to_coeff(coeffs)
    want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial_mt(coeffs.cpu().numpy().view(np.uint64), cref.fr_mont(TAU))))
    return bool((np.asarray(out)[:8] == want).all()), ms
K = int(sys.argv[1]) if len(sys.argv) > 1 else 26
degrees = [d for d in (25, 24, 21, 20) if d < K]
t0 = time.perf_counter(); top = h2.ParamsKZG.setup(K, TAU); torch.cuda.synchronize()
print(f"setup k={K} (synthetic SRS on the device): {time.perf_counter() - t0:.2f} s, HBM in use {used_gib():.1f} GiB", flush=True)
t0 = time.perf_counter(); top.precompute(); print(f"window tables of g and g_lagrange at k={K}: {time.perf_counter() - t0:.2f} s, HBM in use {used_gib():.1f} GiB", flush=True)
pm = {K: top}
for d in degrees:
    t0 = time.perf_counter(); p = top.clone_downsized(d); t1 = time.perf_counter()
    p.precompute(); t2 = time.perf_counter()
    pm[d] = p
    print(f"degree {d}: clone + downsize (prefix view of g, g_lagrange rebuilt on the device) {t1 - t0:.2f} s, window tables {t2 - t1:.2f} s, HBM in use {used_gib():.1f} GiB", flush=True)
for d, p in sorted(pm.items()):
    ok_c, ms_c = check(p, d, False); ok_l, ms_l = check(p, d, True)
    print(f"degree {d}: commit {ms_c:.2f} ms ok={ok_c}  commit_lagrange {ms_l:.2f} ms ok={ok_l}", flush=True)
print(f"params_map of degrees {sorted(pm)} resident: {used_gib():.1f} GiB of HBM in use", flush=True)
