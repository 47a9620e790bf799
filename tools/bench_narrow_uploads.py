#!/usr/bin/env python3
"""bench_narrow_uploads.py [log_n] -- what a witness column costs to put into HBM: the plain 32-byte upload (pageable and page-locked source) against the packed forms
(1 / 2 / 4 / 8-byte canonical cells, mi355_buf_upload_packed) and the sparse form (non-zero cells as (index, value) pairs: mi355_host_compact_nonzero +
mi355_buf_upload_sparse) on SURVEY 8d's witness-like mix (60 % zero, 20 % bytes, 10 % 64-bit, 10 % uniform) and on a selector-like column (90 % zero).
Prints one JSON line; every variant's device contents are compared with the plain upload's."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import cref
zk = ge.load_package(); zk.init(0)
h2, lib, check, ptr = zk.halo2, zk._capi.lib(), zk._capi.check, zk._capi.ptr
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << k
rng = np.random.default_rng(1)
def mont(vals64):
    return cref.f_from_canonical_vec(cref.FR, np.stack([vals64, np.zeros_like(vals64), np.zeros_like(vals64), np.zeros_like(vals64)], axis=1))
def timed(f, reps=5):
    f(); check(lib.mi355_synchronize())
    t = time.perf_counter()
    for _ in range(reps):
        f()
    check(lib.mi355_synchronize())
    return (time.perf_counter() - t) / reps
out = {"log_n": k, "column_bytes": 32 * n}
u = rng.random(n)
col = np.zeros((n, 4), dtype=np.uint64)
uni = u >= 0.9; col[uni] = rng.integers(0, 1 << 62, size=(int(uni.sum()), 4), dtype=np.uint64)
w64 = (u >= 0.8) & ~uni; col[w64] = mont(rng.integers(0, 1 << 63, size=int(w64.sum()), dtype=np.uint64))
byt = (u >= 0.6) & (u < 0.8); col[byt] = mont(rng.integers(1, 256, size=int(byt.sum()), dtype=np.uint64))
dst = h2.DeviceBuffer(32 * n); ref = h2.DeviceBuffer(32 * n); ref.upload(col)
t_plain = timed(lambda: dst.upload(col))
out["plain_pageable"] = {"ms": t_plain * 1e3, "column_GBps": 32 * n / t_plain / 1e9}
pin = C.c_void_p(); check(lib.mi355_host_alloc(32 * n, C.byref(pin)))
C.memmove(pin, col.ctypes.data, 32 * n)
t_pin = timed(lambda: check(lib.mi355_buf_upload(C.c_void_p(dst.data_ptr()), pin, 32 * n)))
out["plain_page_locked"] = {"ms": t_pin * 1e3, "column_GBps": 32 * n / t_pin / 1e9}
idx = np.empty(n, dtype=np.uint32); vals = np.empty((n, 4), dtype=np.uint64); cnt = C.c_uint64()
for threads in (1, 4, 16):
    t_c = timed(lambda: check(lib.mi355_host_compact_nonzero(ptr(col), n, ptr(idx), ptr(vals), C.byref(cnt), threads)))
    out[f"compact_{threads}_threads_ms"] = t_c * 1e3
def sparse():
    check(lib.mi355_host_compact_nonzero(ptr(col), n, ptr(idx), ptr(vals), C.byref(cnt), 16))
    check(lib.mi355_buf_upload_sparse(C.c_void_p(dst.data_ptr()), n, ptr(idx), ptr(vals), cnt.value))
t_s = timed(sparse)
ok = bool((dst.fr() == ref.fr()).all())
out["sparse_witness_like"] = {"ms": t_s * 1e3, "column_GBps": 32 * n / t_s / 1e9, "nonzero_fraction": cnt.value / n, "link_bytes": cnt.value * 36, "equals_plain": ok, "vs_plain_pageable": t_plain / t_s}
sel = np.zeros((n, 4), dtype=np.uint64); on = rng.random(n) < 0.1; sel[on] = mont(np.ones(int(on.sum()), dtype=np.uint64))
ref.upload(sel)
def sparse_sel():
    check(lib.mi355_host_compact_nonzero(ptr(sel), n, ptr(idx), ptr(vals), C.byref(cnt), 16))
    check(lib.mi355_buf_upload_sparse(C.c_void_p(dst.data_ptr()), n, ptr(idx), ptr(vals), cnt.value))
t_ss = timed(sparse_sel)
out["sparse_selector_like"] = {"ms": t_ss * 1e3, "column_GBps": 32 * n / t_ss / 1e9, "equals_plain": bool((dst.fr() == ref.fr()).all())}
for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
    v = rng.integers(0, 1 << (8 * np.dtype(dt).itemsize - 1), size=n, dtype=np.uint64).astype(dt)
    ref.upload(mont(v.astype(np.uint64)))
    t_p = timed(lambda: check(lib.mi355_buf_upload_packed(C.c_void_p(dst.data_ptr()), ptr(v), n, v.dtype.itemsize)))
    out[f"packed_{v.dtype.itemsize}_byte"] = {"ms": t_p * 1e3, "column_GBps": 32 * n / t_p / 1e9, "equals_plain": bool((dst.fr() == ref.fr()).all()), "vs_plain_pageable": t_plain / t_p}
print(json.dumps(out))
