"""Soak run on an MI355X (TEST TOOL): a few hundred rounds of the calls a prover process makes -- commitments from device and host scalars,
transforms, resident buffers allocated / uploaded / freed, batches, scans, a params clone + downsize + release -- with the results of every
round compared with the first round's, and the free device memory compared before / after (the library's pool and workspace must not grow
once warm).      python tools/soak_gpu.py [rounds] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
zk = ge.load_package(); zk.init(0); h2 = zk.halo2
lib, check = zk._capi.lib(), zk._capi.check
n = 1 << k
rng = np.random.default_rng(5)


def rand_fr(m):
    a = rng.integers(0, 2**64, size=(m, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a


params = h2.ParamsKZG.setup(k, 0x5343524f4c4c0001); params.precompute()
dom = h2.EvaluationDomain(2, k)
cols = [rand_fr(n) for _ in range(4)]
dev_col = torch.from_numpy(cols[0].view(np.int64)).cuda()


def one_round():
    out = []
    out.append(params.commit(dev_col).copy())                       # device scalars
    out.append(params.commit_lagrange(cols[1]).copy())              # host scalars
    out.append(params.commit_many([cols[2], cols[3]], lagrange=True).copy())
    a = cols[1].copy(); dom.lagrange_to_coeff(a); out.append(a[::4097].copy())
    bufs = [h2.DeviceBuffer.from_host(c) for c in cols]             # pool: allocate, use, free
    h2.best_fft_many(bufs, dom.omega_inv, k, divisor=dom.ifft_divisor)
    out.append(bufs[2].fr()[::4097].copy())
    hp = [c.copy() for c in cols[:3]]
    h2.best_fft_many(hp, dom.omega, k)                             # host batch: overlapped copies, helper thread
    out.append(hp[2][::4097].copy())
    ext = h2.DeviceBuffer(32 * dom.extended_len())
    dom.coeff_to_extended(bufs[0], out=ext); dom.extended_to_coeff(ext)
    out.append(ext.download(32 * 64).copy())
    out.append(h2.eval_polynomial(bufs[1], h2.fr(12345)).copy())
    inv = h2.DeviceBuffer.from_host(cols[3]); h2.batch_invert(inv); z = h2.DeviceBuffer(inv.nbytes); h2.prefix_product(inv, dst=z); out.append(z.fr()[::4097].copy()); inv.free(); z.free()
    ext.free()
    for b in bufs:
        b.free()
    if k >= 12:
        small = params.clone_downsized(k - 2); out.append(small.commit_lagrange(cols[0][: n >> 2]).copy()); small.release()
    return out


first = one_round(); one_round()
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
t0 = time.time(); bad = 0
for r in range(rounds):
    got = one_round()
    for a, b in zip(got, first):
        if not np.array_equal(np.asarray(a), np.asarray(b)):
            bad += 1
    if (r + 1) % 25 == 0:
        torch.cuda.synchronize(); f, _ = torch.cuda.mem_get_info()
        print(f"{r + 1} rounds, {bad} mismatches, free memory {f / 2**30:.2f} GiB (start {free0 / 2**30:.2f}), {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
grown = (free0 - free1) / 2**20
print(f"SOAK DONE {rounds} rounds k={k}: {bad} mismatches, device memory grown by {grown:.1f} MiB")
params.release()
sys.exit(1 if bad or grown > 64 else 0)
