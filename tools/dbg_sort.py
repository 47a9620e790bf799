import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import rand_fr, rand_points, affine_of
zk = ge.load_package(); zk.init(0); h2 = zk.halo2; lib = zk._capi.lib(); check = zk._capi.check
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cbits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pts = rand_points(rng, n); sc = rand_fr(rng, n)
check(lib.mi355_msm_set_window_bits(cbits))
got = affine_of(h2.best_multiexp(sc, pts))
want = cref.g1_to_affine(cref.best_multiexp(sc, pts))
print("match", (got == want).all())
c_, w_, e_ = C.c_int(), C.c_int(), C.c_uint64(); check(lib.mi355_msm_last_plan(C.byref(c_), C.byref(w_), C.byref(e_)))
c, W = c_.value, w_.value; nb = 1 << (c - 1); nbk = W * nb
def rd(role, count, dt):
    a = np.zeros(count, dtype=dt); check(lib.mi355_debug_ws_read(role.encode(), 0, zk._capi.ptr(a), a.nbytes)); return a
enc = rd("msm.digits", n * W, np.uint32).reshape(W, n)
offsets = rd("msm.offsets", nbk + 1, np.uint32)
tot = int(offsets[-1])
sorted_ = rd("msm.sorted", max(tot,1), np.uint32)
# reference from enc
mag = (enc & 0x7fffffff).astype(np.int64)
print("c", c, "W", W, "entries nonzero", int((mag > 0).sum()), "offsets total", tot)
ref_hist = np.zeros(nbk + 1, dtype=np.int64)
for w in range(W):
    m = mag[w][mag[w] > 0] - 1
    np.add.at(ref_hist, w * nb + m, 1)
ref_off = np.concatenate([[0], np.cumsum(ref_hist[:-1])])
print("offsets ok", (ref_off == offsets).all())
bad = 0
for b in np.nonzero(ref_hist)[0][:2000]:
    w = b // nb; bucket = b % nb
    ents = sorted_[offsets[b]:offsets[b + 1]]
    idx = ents & 0x7fffffff
    want_idx = np.nonzero(mag[w] == bucket + 1)[0]
    if sorted(idx.tolist()) != want_idx.tolist(): bad += 1
    else:
        sg = (ents >> 31) == (enc[w][idx] >> 31)
        if not sg.all(): bad += 1
print("bad buckets", bad)
# digits check vs python
canon = cref.f_to_canonical_vec(cref.FR, sc)
ok = True
for i in range(min(n, 50)):
    k = cref.limbs_to_int(canon[i]); tot_ = 0
    for w in range(W):
        e = int(enc[w][i]); d = (e & 0x7fffffff) * (-1 if e >> 31 else 1); tot_ += d << (c * w)
    ok &= (tot_ == k)
print("digits recompose ok", ok)
