"""ctypes binding of libmi355zk.so -- signatures exactly as declared in include/mi355zk.h."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MI355ZK_LIB: another build of the same library (A/B experiments with compile-time switches, tools/build_variant.py); never a fallback
LIB_PATH = os.environ.get("MI355ZK_LIB") or os.path.join(HERE, "libmi355zk.so")

OK, EBADARG, ENODEVICE, EOOM, EHIP, ERCCL = 0, 1, 2, 3, 4, 5
_vp, _u64, _u32, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int

# name -> (restype, argtypes): one row per symbol of include/mi355zk.h (tests/test_capi_symbols.py cross-checks the header)
SIGNATURES = {
    "mi355_init": (_int, [_int]),
    "mi355_init_multi": (_int, [C.POINTER(_int), _int]),
    "mi355_device_count": (_int, [C.POINTER(_int)]),
    "mi355_shutdown": (_int, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_version": (C.c_char_p, []),
    "mi355_set_stream": (_int, [_vp]),
    "mi355_reset_stream": (_int, []),
    "mi355_synchronize": (_int, []),
    "mi355_buf_alloc": (_int, [_u64, _int, C.POINTER(_vp)]),
    "mi355_buf_free": (_int, [_vp]),
    "mi355_buf_trim": (_int, []),
    "mi355_buf_slot": (_int, [_vp, C.POINTER(_int)]),
    "mi355_buf_upload": (_int, [_vp, _vp, _u64]),
    "mi355_buf_download": (_int, [_vp, _vp, _u64]),
    "mi355_buf_upload_packed": (_int, [_vp, _vp, _u64, C.c_uint32]),
    "mi355_buf_upload_sparse": (_int, [_vp, _u64, _vp, _vp, _u64]),
    "mi355_host_compact_nonzero": (_int, [_vp, _u64, _vp, _vp, C.POINTER(_u64), _int]),
    "mi355_buf_copy": (_int, [_vp, _vp, _u64]),
    "mi355_buf_zero": (_int, [_vp, _u64]),
    "mi355_host_alloc": (_int, [_u64, C.POINTER(_vp)]),
    "mi355_host_free": (_int, [_vp]),
    "mi355_mem_info": (_int, [_int, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "mi355_srs_register_host": (_int, [_vp, _u64, C.POINTER(_u64)]),
    "mi355_srs_register_dev": (_int, [_vp, _u64, _int, C.POINTER(_u64)]),
    "mi355_srs_register_prefix": (_int, [_u64, _u64, C.POINTER(_u64)]),
    "mi355_srs_release": (_int, [_u64]),
    "mi355_srs_precompute": (_int, [_u64, _u64, _int]),
    "mi355_srs_pre_dev_ptr": (_int, [_u64, C.POINTER(_vp), C.POINTER(_int), C.POINTER(_int)]),
    "mi355_srs_len": (_int, [_u64, C.POINTER(_u64)]),
    "mi355_srs_dev_ptr": (_int, [_u64, C.POINTER(_vp)]),
    "mi355_msm_g1_host": (_int, [_u64, _u64, _vp, _u64, _vp]),
    "mi355_msm_g1_dev": (_int, [_u64, _u64, _vp, _u64, _vp]),
    "mi355_msm_g1_dev_async": (_int, [_u64, _u64, _vp, _u64, _vp]),
    "mi355_g1_sum_dev": (_int, [_vp, _u64, _vp]),
    "mi355_msm_g1_batch_dev": (_int, [_u64, _u64, C.POINTER(_vp), _u32, _u64, _vp]),
    "mi355_msm_g1_batch_host": (_int, [_u64, _u64, C.POINTER(_vp), _u32, _u64, _vp]),
    "mi355_msm_set_pipeline": (_int, [_u32, _u32]),
    "mi355_msm_g1_adhoc_host": (_int, [_vp, _vp, _u64, _vp]),
    "mi355_g1_sum_host": (_int, [_vp, _u64, _vp]),
    "mi355_g1_batch_normalize_dev": (_int, [_vp, _vp, _u64]),
    "mi355_g1_batch_normalize_host": (_int, [_vp, _vp, _u64]),
    "mi355_msm_set_normalise": (_int, [_int]),
    "mi355_msm_set_window_bits": (_int, [_int]),
    "mi355_ntt_fr_host": (_int, [_vp, _u32, _vp]),
    "mi355_ntt_fr_dev": (_int, [_vp, _u32, _vp]),
    "mi355_intt_fr_host": (_int, [_vp, _u32, _vp, _vp]),
    "mi355_intt_fr_dev": (_int, [_vp, _u32, _vp, _vp]),
    "mi355_coeff_to_extended_host": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "mi355_coeff_to_extended_dev": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "mi355_extended_to_coeff_host": (_int, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "mi355_extended_to_coeff_dev": (_int, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "mi355_distribute_powers_fr_dev": (_int, [_vp, _u64, _vp]),
    "mi355_coset_ntt_fr_dev": (_int, [_vp, _vp, _u32, _vp, _vp]),
    "mi355_ntt_fr_batch_host": (_int, [C.POINTER(_vp), _u32, _u32, _vp, _vp]),
    "mi355_ntt_fr_batch_dev": (_int, [C.POINTER(_vp), _u32, _u32, _vp, _vp]),
    "mi355_coset_ntt_fr_batch_dev": (_int, [C.POINTER(_vp), C.POINTER(_vp), _u32, _u32, _vp, _vp]),
    "mi355_fr_gate_eval_dev": (_int, [_vp, C.POINTER(_vp), _u32, _vp, C.POINTER(_u32), _u32, C.POINTER(_u32), C.POINTER(C.c_int32), _u64, _int]),
    "mi355_fr_interleave_dev": (_int, [_vp, C.POINTER(_vp), _u32, _u64]),
    "mi355_fr_vec_axpy_dev": (_int, [_vp, _vp, _vp, _vp, _u64]),
    "mi355_fr_vec_op_dev": (_int, [_int, _vp, _vp, _vp, _u64]),
    "mi355_fr_vec_mul_periodic_dev": (_int, [_vp, _u64, _vp, _u32]),
    "mi355_g1_fft_dev": (_int, [_vp, _u32, _vp]),
    "mi355_g1_fft_host": (_int, [_vp, _u32, _vp]),
    "mi355_g_to_lagrange_dev": (_int, [_vp, _vp, _u32, _vp, _vp]),
    "mi355_srs_load_params_file": (_int, [C.c_char_p, _u32, C.POINTER(_u32), C.POINTER(_u64), C.POINTER(_u64), _vp, _vp]),
    "mi355_srs_downsize": (_int, [_u64, _u32, _vp, _vp, C.POINTER(_u64)]),
    "mi355_srs_read_host": (_int, [_u64, _u64, _u64, _vp]),
    "mi355_fr_kate_division_dev": (_int, [_vp, _vp, _u64, _vp]),
    "mi355_fr_batch_invert_dev": (_int, [_vp, _u64]),
    "mi355_fr_prefix_product_dev": (_int, [_vp, _vp, _u64, _vp]),
    "mi355_fr_prefix_sum_dev": (_int, [_vp, _vp, _u64, _vp]),
    "mi355_eval_polynomial_dev": (_int, [_vp, _u64, _vp, _vp]),
    "mi355_eval_polynomial_batch_dev": (_int, [C.POINTER(_vp), _u32, _u64, _vp, _vp]),
    "mi355_eval_polynomial_host": (_int, [_vp, _u64, _vp, _vp]),
    "mi355_srs_setup_dev": (_int, [_vp, _vp, _u32, _vp, _vp]),
    "mi355_g1_fixed_base_mul_dev": (_int, [_vp, _vp, _u64]),
    "mi355_g2_mul_host": (_int, [_vp, _vp, _vp]),
    "mi355_profile_enable": (_int, [_int]),
    "mi355_profile_get": (_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "mi355_profile_reset": (_int, []),
    "mi355_debug_ws_read": (_int, [C.c_char_p, _u64, _vp, _u64]),
    "mi355_msm_last_plan": (_int, [C.POINTER(_int), C.POINTER(_int), C.POINTER(_u64)]),
    "mi355_msm_last_run": (_int, [C.POINTER(_int), C.POINTER(C.c_char_p), C.POINTER(_int), C.POINTER(_int)]),
}


class Mi355Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mi355zk error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Loads libmi355zk.so; raises loudly when it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(ENODEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch (device memory / streams / torch.distributed plumbing) bundles its own libamdhip64.so.7; importing it FIRST
        # makes the dynamic loader hand that same HIP runtime to libmi355zk.so (matched by SONAME), so the process has one
        # runtime and torch tensors / streams are usable from the library.  Without torch the ROCm install's runtime is used.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise Mi355Error(rc, lib().mi355_last_error().decode())


_initialised = False


def init(device_id=0) -> None:
    """device_id: an int (one process per GPU) or a list of ints (mi355_init_multi: several devices behind one process)."""
    global _initialised
    if isinstance(device_id, (list, tuple)):
        ids = (C.c_int * len(device_id))(*device_id)
        check(lib().mi355_init_multi(ids, len(device_id)))
        device_id = device_id[0]
    else:
        check(lib().mi355_init(device_id))
    _initialised = True
    # device-resident operands come from torch: run the library on torch's current stream so that kernels are ordered
    # with the tensor producers/consumers (the host-pointer ABI used by the Rust shim is synchronous and unaffected)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(device_id)
            check(lib().mi355_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    except ImportError:  # pragma: no cover
        pass


def shutdown() -> None:
    global _initialised
    check(lib().mi355_shutdown())
    _initialised = False


def ptr(x) -> C.c_void_p:
    """numpy array (host) | torch tensor (device or host) | int address -> void*"""
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    assert x.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return x.ctypes.data_as(C.c_void_p)  # data_as keeps a reference to the array, so temporaries stay alive for the call
