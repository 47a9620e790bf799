import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import rand_fr
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
for k in [int(x) for x in sys.argv[1:]]:
    n = 1 << k
    rng = np.random.default_rng(k)
    a = rand_fr(rng, n)
    w = h2.fr(pyref.omega(k))
    want = cref.best_fft(a, w, k)
    host = a.copy(); h2.best_fft(host, w, k)
    d = torch.from_numpy(a.view(np.int64)).cuda(); torch.cuda.synchronize()
    h2.best_fft(d, w, k); torch.cuda.synchronize()
    dev = d.cpu().numpy().view(np.uint64)
    hb = (host != want).any(axis=1); db = (dev != want).any(axis=1)
    print(k, "host bad", int(hb.sum()), "dev bad", int(db.sum()), "first bad dev", np.nonzero(db)[0][:8].tolist())
