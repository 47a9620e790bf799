"""CPU-only: libmi355zk.so loads, exports every symbol include/mi355zk.h declares, the ctypes table matches the header,
and -- there being no GPU here -- compute entry points fail loudly with MI355_ENODEVICE instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zk():
    ge.build()
    return ge.load_package()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mi355zk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_\w+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(zk):
    syms = header_symbols()
    assert len(syms) >= 30
    lib = zk._capi.lib()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mi355zk.h but not exported"
    assert sorted(zk._capi.SIGNATURES) == syms, "ctypes signature table and header disagree"


def test_header_cites_the_reference_interfaces():
    txt = open(os.path.join(ROOT, "include", "mi355zk.h")).read()
    for needle in ("best_multiexp", "best_fft", "commit_lagrange", "coeff_to_extended", "[REF integration/src/prove.rs:37,67,96]", "[REF docker/chain-prover/gpu/Dockerfile:7-8]"):
        assert needle in txt


def test_no_gpu_means_loud_failure_not_fallback(zk):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    lib = zk._capi.lib()
    assert lib.mi355_init(0) == zk._capi.ENODEVICE
    assert b"no HIP device" in lib.mi355_last_error() or b"device" in lib.mi355_last_error()
    out = np.zeros(12, dtype=np.uint64); sc = np.zeros((4, 4), dtype=np.uint64); bs = np.zeros((4, 8), dtype=np.uint64)
    assert lib.mi355_msm_g1_adhoc_host(zk._capi.ptr(bs), zk._capi.ptr(sc), 4, zk._capi.ptr(out)) == zk._capi.ENODEVICE
    assert lib.mi355_ntt_fr_host(zk._capi.ptr(sc), 2, zk._capi.ptr(sc[0])) == zk._capi.ENODEVICE
    with pytest.raises(zk.Mi355Error):
        zk.halo2.best_multiexp(sc, bs)
    with pytest.raises(zk.Mi355Error):
        zk.init(0)
    # every compute entry point added later in the round fails the same way (no silent CPU path anywhere)
    import ctypes as C
    capi, ptr = zk._capi, zk._capi.ptr
    h, k = C.c_uint64(), C.c_uint32()
    arr = (C.c_void_p * 1)(sc.ctypes.data)
    jac = np.zeros((4, 12), dtype=np.uint64)
    calls = [
        lib.mi355_msm_g1_batch_host(1, 0, arr, 1, 4, ptr(out)),
        lib.mi355_msm_g1_batch_dev(1, 0, arr, 1, 4, ptr(out)),
        lib.mi355_msm_g1_dev_async(1, 0, ptr(sc), 4, ptr(out)),
        lib.mi355_g1_sum_dev(ptr(jac), 4, ptr(out)),
        lib.mi355_g1_batch_normalize_host(ptr(jac), ptr(bs), 4),
        lib.mi355_g1_batch_normalize_dev(ptr(jac), ptr(bs), 4),
        lib.mi355_g1_fft_host(ptr(jac), 2, ptr(sc[0])),
        lib.mi355_g_to_lagrange_dev(ptr(bs), ptr(bs), 2, ptr(sc[0]), ptr(sc[1])),
        lib.mi355_srs_downsize(1, 1, ptr(sc[0]), ptr(sc[1]), C.byref(h)),
        lib.mi355_srs_read_host(1, 0, 1, ptr(bs)),
        lib.mi355_srs_load_params_file(b"/nonexistent", 0, C.byref(k), C.byref(h), C.byref(h), None, None),
        lib.mi355_fr_batch_invert_dev(ptr(sc), 4),
        lib.mi355_fr_prefix_product_dev(ptr(sc), ptr(sc), 4, None),
        lib.mi355_fr_prefix_sum_dev(ptr(sc), ptr(sc), 4, None),
        lib.mi355_fr_kate_division_dev(ptr(sc), ptr(sc), 4, ptr(sc[0])),
        lib.mi355_fr_vec_axpy_dev(ptr(sc), ptr(sc), ptr(sc), ptr(sc[0]), 4),
        lib.mi355_eval_polynomial_host(ptr(sc), 4, ptr(sc[0]), ptr(sc[1])),
        lib.mi355_srs_register_host(ptr(bs), 4, C.byref(h)),
    ]
    assert all(rc == capi.ENODEVICE for rc in calls), calls
    # several devices behind one process: the same loud failure; the bookkeeping entry points work without a device
    ids = (C.c_int * 2)(0, 1)
    assert lib.mi355_init_multi(ids, 2) == capi.ENODEVICE
    nd = C.c_int(-1)
    assert lib.mi355_device_count(C.byref(nd)) == capi.OK and nd.value == 0
    assert lib.mi355_srs_register_prefix(12345, 4, C.byref(h)) == capi.EBADARG
    dv, ex, shd, sl = C.c_int(), C.c_char_p(), C.c_int(), C.c_int()
    assert lib.mi355_msm_last_run(C.byref(dv), C.byref(ex), C.byref(shd), C.byref(sl)) == capi.OK and ex.value in (b"none", b"rccl_allgather", b"device_copy")


def test_missing_library_raises(zk, monkeypatch):
    monkeypatch.setattr(zk._capi, "_lib", None)
    monkeypatch.setattr(zk._capi, "LIB_PATH", "/nonexistent/libmi355zk.so")
    with pytest.raises(zk.Mi355Error):
        zk._capi.lib()
