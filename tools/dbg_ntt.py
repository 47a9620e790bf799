import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.test_gpu_properties import dev_scalars
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in [int(x) for x in sys.argv[1:]]:
    n = 1 << k
    dom = h2.EvaluationDomain(2, k)
    a = dev_scalars(n, 300 + k); orig = a.clone()
    dom.coeff_to_lagrange(a); f1 = a.clone()
    a2 = orig.clone(); dom.coeff_to_lagrange(a2)
    print(k, "fwd deterministic:", torch.equal(f1, a2))
    dom.lagrange_to_coeff(a)
    bad = (a != orig).any(dim=1).nonzero().flatten()
    print(k, "roundtrip bad count", bad.numel(), bad[:10].tolist(), bad[-5:].tolist() if bad.numel() else "")
    if k <= 24:
        want = cref.best_fft(orig.cpu().numpy().view(np.uint64), dom.omega, k)
        fb = (torch.from_numpy(want.view(np.int64)).cuda() != f1).any(dim=1).nonzero().flatten()
        print(k, "fwd vs oracle bad", fb.numel(), fb[:10].tolist())
