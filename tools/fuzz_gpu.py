"""Randomised cross-check of the MSM / NTT / scan entry points against the oracle on an MI355X (TEST TOOL, not part of the suites):
random sizes, window widths, window tables, batches, pipeline chunks, scalar distributions and degenerate point sets.
    python tools/fuzz_gpu.py [iterations] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
from oracle import cref, pyref

R = pyref.R_MOD
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
# FUZZ_DEVICES=0,0 (with MI355_ALLOW_DUP_DEVICES=1): several device slots behind the process -> sharded bases, worker threads, exchange;
# MI355_HOST_SLICE_MIN_LOG=6: even small host-pointer MSMs are cut into point-range slices (bucket-set fold + one reduction tail)
_devs = os.environ.get("FUZZ_DEVICES")
zk = ge.load_package(); zk.init([int(x) for x in _devs.split(",")] if _devs else 0); h2 = zk.halo2
lib, check, ptr = zk._capi.lib(), zk._capi.check, zk._capi.ptr

G = cref.g1_generator()
NP = 4096
base_sc = rng.integers(0, 2**64, size=(NP, 4), dtype=np.uint64); base_sc[:, 3] &= np.uint64((1 << 60) - 1)
t0 = time.time()
pts_all = cref.g1_to_affine(np.stack([cref.g1_mul(G, base_sc[i]) for i in range(NP)]))
print(f"{NP} points in {time.time() - t0:.1f} s", flush=True)


def scalars(n, kind):
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    if kind == "sparse":
        a[rng.random(n) < 0.8] = 0
    elif kind == "small":
        a = np.stack([cref.fr_mont(int(v)) for v in rng.integers(0, 300, size=n)]) if n else a
    elif kind == "equal":
        a[:] = a[0] if n else a
    elif kind == "extreme":
        ext = [cref.fr_mont(R - 1), cref.fr_mont(1), cref.fr_mont(0), cref.fr_mont((1 << 253) + 5), cref.fr_mont(R - 2)]
        for i in range(n):
            if rng.random() < 0.5:
                a[i] = ext[int(rng.integers(0, len(ext)))]
    return a


fails = 0
for it in range(iters):
    n = int(rng.integers(1, 3000)) if rng.random() < 0.8 else int(rng.integers(3000, NP))
    off = int(rng.integers(0, NP - n + 1))
    pts = pts_all[off:off + n].copy()
    mode = rng.random()
    if mode < 0.2 and n > 4:            # duplicates, negatives, identities
        pts[1] = pts[0]; pts[2] = pts[0]; pts[2][4:] = cref.g1_to_affine(cref.g1_mul(pts[0], cref.fr_mont(R - 1)))[4:]; pts[3] = 0
    kbits = max(1, int(np.ceil(np.log2(n))))
    full = np.zeros((1 << kbits, 8), dtype=np.uint64); full[:n] = pts
    params = h2.ParamsKZG.from_host(kbits, full, full)
    pre = rng.random() < 0.5
    if pre:
        params.precompute(c=int(rng.integers(2, 15)) if rng.random() < 0.7 else 0)
    c_force = int(rng.integers(2, 17)) if (not pre and rng.random() < 0.5) else 0
    chunks = int(rng.integers(2, 6)) if rng.random() < 0.3 else 0
    M = int(rng.integers(1, 10)) if rng.random() < 0.4 else 1
    kind = ["uniform", "sparse", "small", "equal", "extreme"][int(rng.integers(0, 5))]
    polys = [scalars(n, kind) for _ in range(M)]
    check(lib.mi355_msm_set_window_bits(c_force))
    if chunks:
        check(lib.mi355_msm_set_pipeline(chunks, 4))
    try:
        if M == 1:
            dev = rng.random() < 0.5
            x = torch.from_numpy(polys[0].view(np.int64)).cuda() if dev else polys[0]
            got = [h2.best_multiexp(x, params.g_slice(0, n))]
        else:
            got = list(params.commit_many([torch.from_numpy(p.view(np.int64)).cuda() for p in polys]) if rng.random() < 0.5 else params.commit_many(polys))
    finally:
        check(lib.mi355_msm_set_window_bits(0)); check(lib.mi355_msm_set_pipeline(0, 0))
    for m in range(M):
        want = cref.g1_to_affine(cref.best_multiexp(polys[m], pts))
        g = np.asarray(got[m])
        ok = (g[:8] == want).all() if (g[8:] != 0).any() else (want == 0).all()
        if not ok:
            fails += 1
            print("MISMATCH", dict(it=it, n=n, off=off, pre=pre, c_force=c_force, chunks=chunks, M=M, kind=kind, m=m), flush=True)
    params.release()
    # scans / NTT on a random length
    ln = int(rng.integers(1, 6000))
    a = scalars(ln, ["uniform", "sparse", "extreme"][int(rng.integers(0, 3))])
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    h2.batch_invert(d)
    if not (d.cpu().numpy().view(np.uint64).reshape(ln, 4) == cref.batch_invert(a)).all():
        fails += 1; print("MISMATCH batch_invert", ln, flush=True)
    z, tot = h2.prefix_product(torch.from_numpy(a.view(np.int64).copy()).cuda(), want_total=True)
    wz, wt = cref.prefix_product(a)
    if not ((z.cpu().numpy().view(np.uint64).reshape(ln, 4) == wz).all() and (tot == wt).all()):
        fails += 1; print("MISMATCH prefix_product", ln, flush=True)
    zz = cref.fr_mont(int(rng.integers(0, 2**62)))
    q = h2.kate_division(torch.from_numpy(a.view(np.int64).copy()).cuda(), zz)
    if not (q.cpu().numpy().view(np.uint64).reshape(ln - 1, 4) == cref.kate_division(a, zz)).all():
        fails += 1; print("MISMATCH kate_division", ln, flush=True)
    k = int(rng.integers(0, 13))
    b = scalars(1 << k, "uniform")
    w = h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << (28 - k), R))
    f = b.copy(); h2.best_fft(f, w, k)
    if not (f == cref.best_fft(b, w, k)).all():
        fails += 1; print("MISMATCH ntt", k, flush=True)
    # coset transforms (round 6: the shift a[i] *= f^i rides on the first pass for multi-pass plans, the separate kernel below 2^9): ANY field element as the factor -- zero and one
    # included --, single and batched entry points, in place and out of place, against the scaling by plain multiplications + best_fft
    kc = int(rng.integers(0, 15))
    nc = 1 << kc
    f_int = [0, 1, R - 1, int(rng.integers(2, 2**62)), int.from_bytes(rng.bytes(31), "little") % R][int(rng.integers(0, 5))]
    fac = cref.fr_mont(f_int)
    wc = h2.fr(pow(h2.FR_ROOT_OF_UNITY, 1 << (28 - kc), R))
    Mc = int(rng.integers(1, 5))
    srcs_h = [scalars(nc, ["uniform", "sparse", "extreme"][int(rng.integers(0, 3))]) for _ in range(Mc)]
    pw = np.zeros((nc, 4), dtype=np.uint64); pw[0] = cref.fr_mont(1)
    mm = 1
    while mm < nc:
        pw[mm:2 * mm] = cref.f_mul_vec(cref.FR, pw[:mm], np.tile(cref.fr_mont(pow(f_int, mm, R)), (mm, 1))); mm *= 2
    wants = [cref.best_fft(cref.f_mul_vec(cref.FR, hsrc, pw), wc, kc) for hsrc in srcs_h]
    srcs_d = [torch.from_numpy(x.view(np.int64).copy()).cuda() for x in srcs_h]
    inplace = rng.random() < 0.3
    dsts_d = srcs_d if inplace else [torch.empty((nc, 4), dtype=torch.int64, device="cuda") for _ in range(Mc)]
    if Mc == 1 and rng.random() < 0.5:
        check(lib.mi355_coset_ntt_fr_dev(ptr(dsts_d[0]), ptr(srcs_d[0]), kc, ptr(fac), ptr(wc)))
    else:
        check(lib.mi355_coset_ntt_fr_batch_dev((C.c_void_p * Mc)(*[x.data_ptr() for x in dsts_d]), (C.c_void_p * Mc)(*[x.data_ptr() for x in srcs_d]), Mc, kc, ptr(fac), ptr(wc)))
    for i_ in range(Mc):
        if not (dsts_d[i_].cpu().numpy().view(np.uint64).reshape(nc, 4) == wants[i_]).all():
            fails += 1; print("MISMATCH coset_ntt", dict(k=kc, f=hex(f_int), M=Mc, i=i_, inplace=inplace), flush=True)
        if not inplace and not (srcs_d[i_].cpu().numpy().view(np.uint64).reshape(nc, 4) == srcs_h[i_]).all():
            fails += 1; print("MISMATCH coset_ntt modified its source", dict(k=kc, M=Mc, i=i_), flush=True)
    if it % 25 == 24:
        print(f"{it + 1} iterations, {fails} mismatches, {time.time() - t0:.0f} s", flush=True)
print("FUZZ DONE", iters, "iterations", fails, "mismatches")
sys.exit(1 if fails else 0)
