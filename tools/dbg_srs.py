import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import cref, pyref
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
k, n = 6, 64
tau = 0x1F2E3D4C5B6A79880123456789ABCDEF
params = h2.ParamsKZG.setup(k, tau)
g = params._owner[0].cpu().numpy().view(np.uint64).reshape(n, 8)
gl = params._owner[1].cpu().numpy().view(np.uint64).reshape(n, 8)
og, ogl, gs, gls = cref.srs_setup(k, h2.fr(tau), h2.fr(pyref.omega(k)))
print("g match rows", (g == og).all(axis=1).sum(), "gl match rows", (gl == ogl).all(axis=1).sum())
print("on curve g", sum(cref.g1_is_on_curve(g[i]) for i in range(n)), "gl", sum(cref.g1_is_on_curve(gl[i]) for i in range(n)))
# fixed base mul directly
sc = torch.from_numpy(gs.view(np.int64)).cuda()
out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
zk._capi.check(zk._capi.lib().mi355_g1_fixed_base_mul_dev(zk._capi.ptr(out), zk._capi.ptr(sc), n))
torch.cuda.synchronize()
o = out.cpu().numpy().view(np.uint64).reshape(n, 8)
print("fixed-base with oracle scalars match rows", (o == og).all(axis=1).sum())
for i in range(3): print(i, g[i][:2], og[i][:2])
