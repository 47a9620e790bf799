import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in (20, 24, 26):
    n = 1 << k
    a = torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda"); a[:, 3] &= (1 << 59) - 1
    dom = h2.EvaluationDomain(2, k)
    dom.coeff_to_lagrange(a); dom.lagrange_to_coeff(a); torch.cuda.synchronize()
    reps = 20 if k <= 20 else 6
    t = time.perf_counter()
    for _ in range(reps): dom.coeff_to_lagrange(a); dom.lagrange_to_coeff(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / (2 * reps)
    print(f"k={k}: {dt*1e3:.3f} ms", end="  ")
print()
