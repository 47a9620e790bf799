"""
-m gpu: out-of-memory and fault drills.  The Rust shim's policy is "non-zero return code -> log and run
the CPU path" and the caller's only safety net is catch_unwind [REF bin/src/prove_utils.rs:27-41,45-77]: the library must turn a full HBM into
MI355_EOOM, never into a hang or an abort, and must keep serving calls afterwards.

  (a) mi355_buf_alloc until MI355_EOOM, free, then an MSM and an NTT succeed (and are correct);
  (b) HBM held by POOLED (freed) blocks does not starve the library's own allocations: workspace growth, twiddle tables and window tables
      release the pool and retry;
  (c) mi355_srs_precompute with HBM nearly full returns MI355_EOOM and the handle still commits, table-free, with the right result;
  (d) a 2^28 extended_to_coeff whose 8 GiB scratch cannot be allocated returns an error (no hang); after freeing memory it succeeds;
  (e) a cross-slot mi355_buf_copy followed IMMEDIATELY by a write to the source: the copy must have read the old contents;
  (f) kill -9 of a process in the middle of a stream of MSMs leaves the device usable by the next process.
"""
import ctypes as C
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from oracle import cref, pyref
from tests.gpu_common import affine_of, rand_fr

pytestmark = pytest.mark.gpu
R = pyref.R_MOD
TAU = 0x5343524F4C4C0666
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zk():
    pkg = ge.load_package()
    pkg.init(0)
    return pkg


def fill_hbm(lib, capi, sizes=(16 << 30, 1 << 30, 64 << 20), cap=64):
    """mi355_buf_alloc blocks of decreasing size until each size fails with MI355_EOOM; returns the live block pointers"""
    blocks = []
    for sz in sizes:
        for _ in range(cap):
            p = C.c_void_p()
            rc = lib.mi355_buf_alloc(sz, 0, C.byref(p))
            if rc != 0:
                assert rc == capi.EOOM, (rc, lib.mi355_last_error())
                break
            blocks.append(p.value)
        else:
            raise AssertionError(f"{cap} blocks of {sz} bytes never exhausted the device")
    return blocks


def free_all(lib, capi, blocks):
    for p in blocks:
        capi.check(lib.mi355_buf_free(C.c_void_p(p)))


def commit_is_right(params, vals, tau):
    want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(vals, cref.fr_mont(tau))))
    return bool((affine_of(params.commit(vals)) == want).all())


def test_buf_alloc_until_oom_then_msm_and_ntt_still_work(zk):
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    torch.cuda.empty_cache()
    blocks = fill_hbm(lib, capi)
    assert len(blocks) >= 8
    assert b"hipMalloc" in lib.mi355_last_error()
    # with HBM full, a call that needs fresh workspace reports the same code (no hang, no abort) ...
    k = 16
    n = 1 << k
    rng = np.random.default_rng(5)
    vals = rand_fr(rng, n)
    free_all(lib, capi, blocks)              # ... and once the blocks are back (pooled, released on demand) everything works
    # the pool serves the LIBRARY's allocations on demand; another allocator in the process (torch's, which ParamsKZG.setup uses for its two
    # staging tensors) only sees what HIP reports free, so the blocks are handed back first
    capi.check(lib.mi355_buf_trim())
    params = h2.ParamsKZG.setup(k, TAU)
    assert commit_is_right(params, vals, TAU)
    dom = h2.EvaluationDomain(2, k)
    a = vals.copy()
    dom.coeff_to_lagrange(a)
    assert (a == cref.best_fft(vals, dom.omega, k, threads=4)).all()
    params.release()
    capi.check(lib.mi355_buf_trim())


def test_pooled_blocks_do_not_starve_workspace_and_tables(zk):
    """fill HBM through mi355_buf_alloc, FREE everything (the blocks stay in the library's pool: HIP still sees a full device), then ask for
    things that need hipMalloc inside the library: an NTT plan + scratch at a size not used before, and window tables"""
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    torch.cuda.empty_cache()
    k = 23
    n = 1 << k
    keep = h2.DeviceBuffer(32 * n)
    vals = rand_fr(np.random.default_rng(6), n)
    keep.upload(vals)
    params = h2.ParamsKZG.setup(18, TAU + 1)
    blocks = fill_hbm(lib, capi)
    free_all(lib, capi, blocks)
    free_before, _ = torch.cuda.mem_get_info()
    assert free_before < (2 << 30), "the pool should still hold the device's memory"
    dom = h2.EvaluationDomain(2, k)
    w_odd = h2.fr(pow(h2.fr_to_int(dom.omega), 3, R))                     # a generator of the same order nobody has a plan for
    capi.check(lib.mi355_ntt_fr_dev(C.c_void_p(keep.data_ptr()), k, capi.ptr(w_odd)))   # new plan: twiddle tables + 256 MiB of scratch
    got = keep.fr()
    assert (got == cref.best_fft(vals, w_odd, k, threads=cref.usable_cpus())).all()
    params.precompute()                                                   # window tables: ~1 GiB of hipMalloc
    v18 = rand_fr(np.random.default_rng(7), 1 << 18)
    assert commit_is_right(params, v18, TAU + 1)
    keep.free(); params.release()
    capi.check(lib.mi355_buf_trim())


def test_precompute_oom_leaves_a_handle_that_commits_table_free(zk):
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    torch.cuda.empty_cache()
    k = 20
    params = h2.ParamsKZG.setup(k, TAU + 2)
    vals = rand_fr(np.random.default_rng(8), 1 << k)
    assert commit_is_right(params, vals, TAU + 2)        # sizes the MSM workspace while memory is plentiful (table-free schedule)
    blocks = fill_hbm(lib, capi)   # down to 64 MiB blocks: less than that is left
    rc = lib.mi355_srs_precompute(params._g, 0, 0)  # ~1 GiB of tables for 2^20 points: cannot fit
    assert rc == capi.EOOM, (rc, lib.mi355_last_error())
    assert commit_is_right(params, vals, TAU + 2)        # same handle, no tables: still serves, still right
    run_c, run_w, run_e = C.c_int(), C.c_int(), C.c_uint64()
    capi.check(lib.mi355_msm_last_plan(C.byref(run_c), C.byref(run_w), C.byref(run_e)))
    free_all(lib, capi, blocks)
    capi.check(lib.mi355_srs_precompute(params._g, 0, 0))   # with memory back the tables build, and the result does not change
    assert commit_is_right(params, vals, TAU + 2)
    params.release()
    capi.check(lib.mi355_buf_trim())


def test_extended_to_coeff_2_28_without_room_for_scratch_is_an_error_not_a_hang(zk):
    h2 = zk.halo2
    lib, capi = zk._capi.lib(), zk._capi
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs 8 GiB for the vector and room to show the failure")
    dom = h2.EvaluationDomain(5, 26)
    ext = 28
    p = C.c_void_p()
    # this process may already own an 8 GiB scratch arena from an earlier test: the drill needs it gone, so re-initialise the library
    capi.check(lib.mi355_shutdown()); zk.init(0)
    capi.check(lib.mi355_buf_alloc(32 << ext, 0, C.byref(p)))
    capi.check(lib.mi355_buf_zero(p, 32 << ext))
    blocks = fill_hbm(lib, capi, sizes=(16 << 30, 4 << 30, 1 << 30))   # less than 1 GiB left: the 8 GiB scratch cannot be had
    t0 = time.perf_counter()
    rc = lib.mi355_extended_to_coeff_dev(p, ext, capi.ptr(dom.g_coset), capi.ptr(dom.g_coset_inv), capi.ptr(dom.extended_omega_inv), capi.ptr(dom.extended_ifft_divisor))
    assert rc == capi.EOOM, (rc, lib.mi355_last_error())
    assert time.perf_counter() - t0 < 30
    free_all(lib, capi, blocks)
    capi.check(lib.mi355_extended_to_coeff_dev(p, ext, capi.ptr(dom.g_coset), capi.ptr(dom.g_coset_inv), capi.ptr(dom.extended_omega_inv), capi.ptr(dom.extended_ifft_divisor)))
    out = np.zeros((4, 4), dtype=np.uint64)
    capi.check(lib.mi355_buf_download(capi.ptr(out), p, 128))
    assert (out == 0).all()                    # the transform of the zero vector
    capi.check(lib.mi355_buf_free(p))
    capi.check(lib.mi355_buf_trim())


def test_cross_slot_copy_then_immediate_overwrite_of_the_source():
    """two device slots (the same physical GPU twice: two contexts, two streams): copy a 1 GiB block from slot 1 to slot 0 and overwrite the
    source on slot 1 in the next call.  The copy runs on slot 0's stream; slot 1's stream must wait for it."""
    from tests.test_gpu_buffers import _reinit
    pkg = ge.load_package()
    pkg.init(0)
    _reinit(pkg, [0, 0], {"MI355_ALLOW_DUP_DEVICES": "1"})
    try:
        h2 = pkg.halo2
        lib, capi = pkg._capi.lib(), pkg._capi
        n = 1 << 25
        vals = np.ascontiguousarray(np.random.default_rng(9).integers(0, 2**62, size=(n, 4), dtype=np.uint64))
        for trial in range(3):
            src = h2.DeviceBuffer.from_host(vals, slot=1)
            dst = h2.DeviceBuffer(32 * n, slot=0)
            capi.check(lib.mi355_synchronize())
            capi.check(lib.mi355_buf_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), 32 * n))
            if trial == 0:
                capi.check(lib.mi355_buf_zero(C.c_void_p(src.data_ptr()), 32 * n))        # a kernel-side overwrite on slot 1's stream
            elif trial == 1:
                old_ptr = src.data_ptr()
                src.free()                                                                 # back to the pool, handed out again at once ...
                again = h2.DeviceBuffer.from_host(np.zeros_like(vals), slot=1)             # ... to a "fresh" upload (copy stream)
                assert again.data_ptr() == old_ptr
                src = again
            else:
                capi.check(lib.mi355_fr_vec_op_dev(0, C.c_void_p(src.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(src.data_ptr()), n))
            got = dst.fr()
            assert (got == vals).all(), trial
            src.free(); dst.free()
    finally:
        _reinit(pkg, 0, {})


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
zk = ge.load_package(); zk.init(0)
h2 = zk.halo2
params = h2.ParamsKZG.setup(22, 12345)
vals = np.ascontiguousarray(np.random.default_rng(1).integers(0, 2**60, size=(1 << 22, 4), dtype=np.uint64))
buf = h2.DeviceBuffer.from_host(vals)
print("READY", flush=True)
while True:
    params.commit(buf)
"""


def test_kill_9_mid_msm_leaves_the_device_usable():
    env = dict(os.environ)
    child = subprocess.Popen([sys.executable, "-c", CHILD % ROOT], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
    try:
        line = ""
        t0 = time.time()
        while "READY" not in line and time.time() - t0 < 240:
            line = child.stdout.readline()
            if not line and child.poll() is not None:
                break
        assert "READY" in line, "the child never reached its MSM loop"
        time.sleep(1.0)                      # several hundred MSMs deep into its loop
        os.kill(child.pid, signal.SIGKILL)   # the exact PID we started
        child.wait(timeout=60)
    finally:
        if child.poll() is None:
            child.kill()
    # the next user of the device: this process (fresh library state) computes and checks a commitment and a transform
    pkg = ge.load_package()
    pkg.init(0)
    h2 = pkg.halo2
    k = 18
    vals = rand_fr(np.random.default_rng(10), 1 << k)
    params = h2.ParamsKZG.setup(k, TAU + 3)
    assert commit_is_right(params, vals, TAU + 3)
    dom = h2.EvaluationDomain(2, k)
    a = vals.copy(); dom.coeff_to_lagrange(a)
    assert (a == cref.best_fft(vals, dom.omega, k, threads=4)).all()
    params.release()
