"""
halo2.py -- host-side mirror of the halo2_proofs operator surface that scroll-prover reaches through
`gen_halo2_chunk_proof` / `gen_batch_proof` / `gen_bundle_proof` [REF integration/src/prove.rs:37,67,96], implemented
on libmi355zk.so.  Names, argument meaning and error behaviour follow halo2_proofs@e5ddf67 [EXT-recalled]:

  best_multiexp(coeffs, bases)          src/arithmetic.rs   panics (here: AssertionError) on length mismatch
  best_fft(a, omega, log_n)             src/arithmetic.rs   in place, natural -> natural
  EvaluationDomain(j, k)                src/poly/domain.rs  lagrange_to_coeff, coeff_to_lagrange, coeff_to_extended,
                                                            extended_to_coeff, constants omega/omega_inv/extended_omega/g_coset
  ParamsKZG                             src/poly/kzg/commitment.rs  setup, commit, commit_lagrange, downsize, n, k

Host arrays are numpy uint64: [n, 4] field elements (Montgomery LE limbs), [n, 8] G1Affine, [12] G1 (Jacobian).
Device-resident operands are torch.uint8/uint64 CUDA tensors; they are passed by address (torch is plumbing only).
The constants (omega, n^-1, ...) are computed with Python integers -- host set-up, exactly as EvaluationDomain::new does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FR_S = 28
FR_ROOT_OF_UNITY = pow(7, (R_MOD - 1) >> FR_S, R_MOD)
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23  # halo2curves bn256::Fr::ZETA [EXT-recalled]
_M64 = (1 << 64) - 1


def fr(x: int) -> np.ndarray:
    """canonical integer -> 4 x u64 Montgomery limbs (the in-memory form of halo2curves Fr)."""
    v = (x % R_MOD) * (1 << 256) % R_MOD
    return np.array([(v >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def fr_to_int(limbs) -> int:
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return v * pow(1 << 256, -1, R_MOD) % R_MOD


def _is_device(x) -> bool:
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


# ------------------------------------------------------------------------------------------ arithmetic.rs
def best_multiexp(coeffs, bases) -> np.ndarray:
    """sum_i coeffs[i] * bases[i] -> G1 (12 u64: normalised Jacobian).  bases: host [n,8] array or a ParamsKZG basis handle slice."""
    n = int(coeffs.shape[0])
    out = np.zeros(12, dtype=np.uint64)
    if isinstance(bases, SrsSlice):
        assert n == bases.n, "best_multiexp: coeffs.len() != bases.len()"
        fn = lib().mi355_msm_g1_dev if _is_device(coeffs) else lib().mi355_msm_g1_host
        check(fn(bases.handle, bases.offset, ptr(coeffs), n, ptr(out)))
        return out
    assert n == int(bases.shape[0]), "best_multiexp: coeffs.len() != bases.len()"
    check(lib().mi355_msm_g1_adhoc_host(ptr(np.ascontiguousarray(bases)), ptr(np.ascontiguousarray(coeffs)), n, ptr(out)))
    return out


def best_fft(a, omega: np.ndarray, log_n: int) -> None:
    """in place; a is a host [2^log_n, 4] uint64 array or a device tensor of 2^log_n * 32 bytes (G = Fr), or -- the G = G1
    instantiation used by g_to_lagrange -- a host [2^log_n, 12] array / device tensor of 2^log_n * 96 bytes of Jacobian points."""
    if _is_device(a):
        nbytes = a.numel() * a.element_size()
        if nbytes == 96 << log_n:
            check(lib().mi355_g1_fft_dev(ptr(a), log_n, ptr(omega)))
            return
        assert nbytes == 32 << log_n
        check(lib().mi355_ntt_fr_dev(ptr(a), log_n, ptr(omega)))
    elif a.ndim == 2 and a.shape[1] == 12:
        assert a.shape == (1 << log_n, 12) and a.dtype == np.uint64
        check(lib().mi355_g1_fft_host(ptr(a), log_n, ptr(omega)))
    else:
        assert a.shape == (1 << log_n, 4) and a.dtype == np.uint64
        check(lib().mi355_ntt_fr_host(ptr(a), log_n, ptr(omega)))


def best_fft_many(polys, omega: np.ndarray, log_n: int, divisor=None) -> None:
    """`for a in polys: best_fft(a, omega, log_n)` (divisor: ... followed by a *= divisor, i.e. EvaluationDomain::ifft) as ONE call:
    mi355_ntt_fr_batch_host / _dev deal the independent transforms over the bound devices (host arrays round-robin; device buffers on the
    device that owns them).  In place; all host arrays or all device buffers."""
    if len(polys) == 0:
        return
    dev = _is_device(polys[0])
    assert all(_is_device(q) == dev for q in polys), "best_fft_many: mix of host and device polynomials"
    if dev:
        arr = (C.c_void_p * len(polys))(*[q.data_ptr() for q in polys])
        check(lib().mi355_ntt_fr_batch_dev(arr, len(polys), log_n, ptr(omega), ptr(divisor)))
    else:
        for q in polys:
            assert q.shape == (1 << log_n, 4) and q.dtype == np.uint64 and q.flags["C_CONTIGUOUS"]
        arr = (C.c_void_p * len(polys))(*[q.ctypes.data for q in polys])
        check(lib().mi355_ntt_fr_batch_host(arr, len(polys), log_n, ptr(omega), ptr(divisor)))


def compact_nonzero(col: np.ndarray, threads: int = 8):
    """(indices, values) of the non-zero cells of a [n, 4] u64 column, in index order (mi355_host_compact_nonzero: host threads, no device)"""
    col = np.ascontiguousarray(col, dtype=np.uint64)
    n = col.shape[0]
    idx = np.empty(n, dtype=np.uint32); vals = np.empty((n, 4), dtype=np.uint64); cnt = C.c_uint64()
    check(lib().mi355_host_compact_nonzero(ptr(col), n, ptr(idx), ptr(vals), C.byref(cnt), threads))
    return idx[: cnt.value].copy(), vals[: cnt.value].copy()


class DeviceBuffer:
    """One mi355_buf_alloc block: the Python twin of the Rust shim's DevicePoly (rust_shim/mi355zk.rs).  Quacks like a device tensor for
    the wrappers of this module (data_ptr / numel / element_size); `slot` picks the device of an mi355_init_multi process."""

    def __init__(self, nbytes: int, slot: int = 0):
        p = C.c_void_p()
        check(lib().mi355_buf_alloc(nbytes, slot, C.byref(p)))
        self._ptr, self.nbytes, self.slot = p.value, nbytes, slot

    @classmethod
    def from_host(cls, arr: np.ndarray, slot: int = 0) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes, slot)
        b.upload(arr)
        return b

    @classmethod
    def from_packed(cls, values: np.ndarray, slot: int = 0) -> "DeviceBuffer":
        """a column whose cells are known to fit 1 / 2 / 4 / 8 bytes (values: uint8 / 16 / 32 / 64, CANONICAL integers): that many bytes per cell cross PCIe, the device
        expands to Montgomery words (mi355_buf_upload_packed)"""
        values = np.ascontiguousarray(values)
        assert values.dtype in (np.uint8, np.uint16, np.uint32, np.uint64) and values.ndim == 1
        b = cls(32 * values.shape[0], slot)
        check(lib().mi355_buf_upload_packed(C.c_void_p(b._ptr), ptr(values), values.shape[0], values.dtype.itemsize))
        return b

    @classmethod
    def from_sparse(cls, col: np.ndarray, slot: int = 0, threads: int = 8) -> "DeviceBuffer":
        """a mostly-zero column ([n, 4] u64 Montgomery words): only its non-zero cells cross PCIe (mi355_host_compact_nonzero + mi355_buf_upload_sparse)"""
        idx, vals = compact_nonzero(col, threads)
        b = cls(col.nbytes, slot)
        check(lib().mi355_buf_upload_sparse(C.c_void_p(b._ptr), col.shape[0], ptr(idx), ptr(vals), idx.shape[0]))
        return b

    def data_ptr(self) -> int:
        return self._ptr

    def numel(self) -> int:
        return self.nbytes

    def element_size(self) -> int:
        return 1

    is_cuda = True

    def upload(self, arr: np.ndarray, offset: int = 0) -> None:
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(lib().mi355_buf_upload(C.c_void_p(self._ptr + offset), ptr(arr), arr.nbytes))

    def download(self, nbytes: int | None = None, offset: int = 0) -> np.ndarray:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes // 8, dtype=np.uint64)
        check(lib().mi355_buf_download(ptr(out), C.c_void_p(self._ptr + offset), nbytes))
        return out

    def fr(self) -> np.ndarray:
        return self.download().reshape(-1, 4)

    def free(self) -> None:
        if self._ptr:
            check(lib().mi355_buf_free(C.c_void_p(self._ptr)))
            self._ptr = 0


def g_to_lagrange(g_dev, k: int):
    """g_to_lagrange(g_projective, k) [EXT-recalled halo2_proofs src/arithmetic.rs]: g_lagrange = n^-1 * best_fft(g, omega^-1, k), as a new
    device tensor of 2^k affine points; g_dev: device tensor (or raw device address) holding at least 2^k affine points."""
    import torch
    n = 1 << k
    w_inv = pow(pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD), R_MOD - 2, R_MOD)
    out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
    src = C.c_void_p(g_dev) if isinstance(g_dev, int) else ptr(g_dev)
    check(lib().mi355_g_to_lagrange_dev(src, ptr(out), k, ptr(fr(w_inv)), ptr(fr(pow(n, R_MOD - 2, R_MOD)))))
    return out


def eval_polynomial(poly, point: np.ndarray) -> np.ndarray:
    """halo2_proofs::arithmetic::eval_polynomial: sum_i poly[i] * point^i -> Fr (4 x u64 Montgomery limbs)."""
    out = np.zeros(4, dtype=np.uint64)
    if _is_device(poly):
        n = poly.numel() * poly.element_size() // 32
        check(lib().mi355_eval_polynomial_dev(ptr(poly), n, ptr(point), ptr(out)))
    else:
        poly = np.ascontiguousarray(poly, dtype=np.uint64)
        check(lib().mi355_eval_polynomial_host(ptr(poly), poly.shape[0], ptr(point), ptr(out)))
    return out


def eval_polynomial_many(polys, points) -> np.ndarray:
    """[eval_polynomial(p, x) for p, x in zip(polys, points)] on device-resident coefficient vectors of equal length with ONE synchronisation
    (mi355_eval_polynomial_batch_dev): step 9 of create_proof evaluates every queried (polynomial, rotation) pair.  -> [len(polys), 4] u64."""
    m = len(polys)
    out = np.zeros((m, 4), dtype=np.uint64)
    if m == 0:
        return out
    n = polys[0].numel() * polys[0].element_size() // 32
    assert all(q.numel() * q.element_size() // 32 == n for q in polys), "eval_polynomial_many: polynomials of equal length"
    pts = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.uint64) for x in points]))
    arr = (C.c_void_p * m)(*[q.data_ptr() for q in polys])
    check(lib().mi355_eval_polynomial_batch_dev(arr, m, n, ptr(pts), ptr(out)))
    return out


def interleave(parts, out=None):
    """the extended-domain vector from its Q coset parts (scroll fork: part q = the evaluations at zeta * extended_omega^(q + Q i)):
    out[i * Q + q] = parts[q][i] (mi355_fr_interleave_dev); what extended_to_coeff inverts."""
    import torch
    q = len(parts)
    n = parts[0].numel() * parts[0].element_size() // 32
    if out is None:
        out = torch.empty((q * n, 4), dtype=torch.int64, device=parts[0].device)
    arr = (C.c_void_p * q)(*[p_.data_ptr() for p_ in parts])
    check(lib().mi355_fr_interleave_dev(ptr(out), arr, q, n))
    return out


def mem_info(slot: int = 0) -> dict:
    """HBM accounting of one bound device (mi355_mem_info): what HIP reports free / in total and what the library holds, in bytes."""
    v = [C.c_uint64() for _ in range(5)]
    check(lib().mi355_mem_info(slot, *[C.byref(x) for x in v]))
    return dict(zip(("free", "total", "live_buffers", "pooled", "workspace"), (x.value for x in v)))


def fr_vec_op(op: str, dst, a, b):
    """element-wise add / sub / mul of device-resident Fr vectors (pointwise steps of the quotient construction)."""
    n = a.numel() * a.element_size() // 32
    check(lib().mi355_fr_vec_op_dev({"add": 0, "sub": 1, "mul": 2}[op], ptr(dst), ptr(a), ptr(b), n))
    return dst


def fr_vec_axpy(dst, a, b, scalar: np.ndarray):
    """dst = a + scalar * b on device-resident Fr vectors (a = None: dst = scalar * b)."""
    n = b.numel() * b.element_size() // 32
    check(lib().mi355_fr_vec_axpy_dev(ptr(dst), ptr(a) if a is not None else None, ptr(b), ptr(scalar), n))
    return dst


def gate_eval(dst, polys, terms, n: int, accumulate: bool = False):
    """dst[i] (+)= sum_j c_j * prod_k polys[p][(i + rot) mod n] in ONE launch (mi355_fr_gate_eval_dev): the operand shape of halo2's
    evaluate_h (rotated columns, sums of products).  terms: list of (c_j: [4] u64 Montgomery, [(poly index, rotation in elements), ...])."""
    coeffs = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c, _ in terms])) if terms else np.zeros((0, 4), dtype=np.uint64)
    term_len = (C.c_uint32 * max(1, len(terms)))(*[len(f) for _, f in terms])
    flat = [pr for _, f in terms for pr in f]
    fp = (C.c_uint32 * max(1, len(flat)))(*[p for p, _ in flat])
    fr_ = (C.c_int32 * max(1, len(flat)))(*[r for _, r in flat])
    arr = (C.c_void_p * max(1, len(polys)))(*[q.data_ptr() for q in polys])
    check(lib().mi355_fr_gate_eval_dev(ptr(dst), arr, len(polys), ptr(coeffs), term_len, len(terms), fp, fr_, n, 1 if accumulate else 0))
    return dst


def kate_division(poly, z: np.ndarray, dst=None):
    """halo2_proofs::arithmetic::kate_division: (poly(X) - poly(z)) / (X - z) on a device-resident coefficient vector -> n - 1 coefficients."""
    import torch
    n = poly.numel() * poly.element_size() // 32
    if dst is None:
        dst = torch.empty((max(n - 1, 0), 4), dtype=torch.int64, device=poly.device)
    check(lib().mi355_fr_kate_division_dev(ptr(dst) if n > 1 else None, ptr(poly), n, ptr(z)))
    return dst


def batch_invert(a):
    """ff::BatchInvert on a device-resident Fr vector, in place: a[i] = a[i]^-1, zeros stay zero."""
    check(lib().mi355_fr_batch_invert_dev(ptr(a), a.numel() * a.element_size() // 32))
    return a


def prefix_product(src, dst=None, want_total: bool = False):
    """the grand-product column of the permutation / lookup arguments: dst[0] = 1, dst[i] = prod_{j<i} src[j] (device tensors; dst
    defaults to a new tensor).  want_total: also return prod_{j<n} src[j] (the value that must be one for a valid argument)."""
    import torch
    if dst is None:
        dst = torch.empty_like(src)
    total = np.zeros(4, dtype=np.uint64) if want_total else None
    check(lib().mi355_fr_prefix_product_dev(ptr(dst), ptr(src), src.numel() * src.element_size() // 32, ptr(total) if want_total else None))
    return (dst, total) if want_total else dst


def prefix_sum(src, dst=None, want_total: bool = False):
    """the running sum phi of the log-derivative (mv-lookup) argument: dst[0] = 0, dst[i] = sum_{j<i} src[j] (device buffers; dst defaults to a
    new tensor).  want_total: also return sum_{j<n} src[j] (zero for a valid argument)."""
    import torch
    if dst is None:
        dst = torch.empty_like(src)
    total = np.zeros(4, dtype=np.uint64) if want_total else None
    check(lib().mi355_fr_prefix_sum_dev(ptr(dst), ptr(src), src.numel() * src.element_size() // 32, ptr(total) if want_total else None))
    return (dst, total) if want_total else dst


# halo2curves bn256 G2 generator (x.c0, x.c1, y.c0, y.c1) [EXT-recalled src/bn256/curve.rs]; the same four words are the first pairing
# input of the released verifier [REF release-v0.13.1/evm_verifier.yul:1230-1233] (tests/test_oracle_golden.py)
G2_GENERATOR = (0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED, 0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
                0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA, 0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B)


def fq(x: int) -> np.ndarray:
    v = (x % P_MOD) * (1 << 256) % P_MOD
    return np.array([(v >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def g2_generator() -> np.ndarray:
    """G2Affine (16 u64: x.c0 | x.c1 | y.c0 | y.c1, Montgomery) of the generator: the `g2` field of every ParamsKZG."""
    return np.concatenate([fq(c) for c in G2_GENERATOR])


def g2_mul(point: np.ndarray, scalar: np.ndarray) -> np.ndarray:
    """scalar * point on the twist (mi355_g2_mul_host): ParamsKZG::setup's s_g2 = tau * G2."""
    out = np.zeros(16, dtype=np.uint64)
    check(lib().mi355_g2_mul_host(ptr(np.ascontiguousarray(point, dtype=np.uint64)), ptr(scalar), ptr(out)))
    return out


def g1_sum(points: np.ndarray) -> np.ndarray:
    """fold of per-GPU partial results: results.iter().fold(identity, |a, b| a + b)."""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    check(lib().mi355_g1_sum_host(ptr(points), points.shape[0], ptr(out)))
    return out


def batch_normalize(points, out=None):
    """group::Curve::batch_normalize: n Jacobian G1 (any representative, [n,12] u64) -> n affine points ([n,8] u64, identity = (0, 0)).
    Host arrays go through mi355_g1_batch_normalize_host; device tensors (96 n bytes in, 64 n bytes out, `out` required) stay in HBM."""
    if _is_device(points):
        n = points.numel() * points.element_size() // 96
        assert out is not None and out.numel() * out.element_size() == 64 * n, "batch_normalize: device output tensor of 64 n bytes required"
        check(lib().mi355_g1_batch_normalize_dev(ptr(points), ptr(out), n))
        return out
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    res = np.zeros((points.shape[0], 8), dtype=np.uint64)
    check(lib().mi355_g1_batch_normalize_host(ptr(points), ptr(res), points.shape[0]))
    return res


# ------------------------------------------------------------------------------------------ poly/domain.rs
class EvaluationDomain:
    """EvaluationDomain::new(j, k): n = 2^k, quotient_poly_degree = j - 1, extended_k minimal with 2^extended_k >= n * (j - 1)."""

    def __init__(self, j: int, k: int):
        self.k, self.n = k, 1 << k
        self.quotient_poly_degree = j - 1
        ext = k
        while (1 << ext) < self.n * self.quotient_poly_degree:
            ext += 1
        assert ext <= FR_S, "extended_k exceeds the two-adicity of Fr"
        self.extended_k = ext
        ew = pow(FR_ROOT_OF_UNITY, 1 << (FR_S - ext), R_MOD)
        w = pow(ew, 1 << (ext - k), R_MOD)
        self._omega, self._extended_omega = w, ew
        self.omega, self.omega_inv = fr(w), fr(pow(w, -1, R_MOD))
        self.extended_omega, self.extended_omega_inv = fr(ew), fr(pow(ew, -1, R_MOD))
        self.g_coset, self.g_coset_inv = fr(FR_ZETA), fr(FR_ZETA * FR_ZETA % R_MOD)
        # t_evaluations: (zeta * extended_omega^i)^n - 1 for i < 2^(extended_k - k) (it is periodic), stored inverted as halo2 does
        q = 1 << (ext - k)
        self.t_evaluations_inv = np.stack([fr(pow((pow(FR_ZETA, self.n, R_MOD) * pow(ew, i * self.n, R_MOD) - 1) % R_MOD, -1, R_MOD)) for i in range(q)])
        self.ifft_divisor = fr(pow(self.n, -1, R_MOD))
        self.extended_ifft_divisor = fr(pow(1 << ext, -1, R_MOD))

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def coeff_to_lagrange(self, a):
        best_fft(a, self.omega, self.k)
        return a

    def lagrange_to_coeff(self, a):
        """ifft: best_fft(a, omega_inv, k) then a[i] *= n^-1."""
        if _is_device(a):
            check(lib().mi355_intt_fr_dev(ptr(a), self.k, ptr(self.omega_inv), ptr(self.ifft_divisor)))
        else:
            check(lib().mi355_intt_fr_host(ptr(a), self.k, ptr(self.omega_inv), ptr(self.ifft_divisor)))
        return a

    def coeff_to_extended(self, a, out=None):
        """zero-pad to the extended domain, move into the coset g_coset * H, transform.  Returns a new array (host) or fills `out` (device)."""
        if _is_device(a):
            assert out is not None and _is_device(out)
            check(lib().mi355_coeff_to_extended_dev(ptr(out), ptr(a), self.k, self.extended_k, ptr(self.g_coset), ptr(self.g_coset_inv), ptr(self.extended_omega)))
            return out
        dst = np.zeros((self.extended_len(), 4), dtype=np.uint64)
        check(lib().mi355_coeff_to_extended_host(ptr(dst), ptr(np.ascontiguousarray(a)), self.k, self.extended_k, ptr(self.g_coset), ptr(self.g_coset_inv), ptr(self.extended_omega)))
        return dst

    def coeff_to_extended_part(self, a, part: int, out):
        """scroll-fork style: evaluations of the 2^k coefficients on the coset (g_coset * extended_omega^part) * H  (device tensors)."""
        assert _is_device(a) and _is_device(out)
        factor = fr(FR_ZETA * pow(self._extended_omega, part, R_MOD) % R_MOD)
        check(lib().mi355_coset_ntt_fr_dev(ptr(out), ptr(a), self.k, ptr(factor), ptr(self.omega)))
        return out

    def divide_by_vanishing_poly(self, a):
        """EvaluationDomain::divide_by_vanishing_poly on extended-coset evaluations (device tensor, in place): a[i] *= t_evaluations[i % len]^-1."""
        assert _is_device(a)
        n = a.numel() * a.element_size() // 32
        check(lib().mi355_fr_vec_mul_periodic_dev(ptr(a), n, ptr(self.t_evaluations_inv), self.t_evaluations_inv.shape[0]))
        return a

    def extended_to_coeff(self, a):
        """inverse of coeff_to_extended, truncated to n * quotient_poly_degree coefficients (host) / in place (device, caller truncates)."""
        if _is_device(a):
            check(lib().mi355_extended_to_coeff_dev(ptr(a), self.extended_k, ptr(self.g_coset), ptr(self.g_coset_inv), ptr(self.extended_omega_inv), ptr(self.extended_ifft_divisor)))
            return a
        a = np.array(a, dtype=np.uint64, copy=True, order="C")
        check(lib().mi355_extended_to_coeff_host(ptr(a), self.extended_k, ptr(self.g_coset), ptr(self.g_coset_inv), ptr(self.extended_omega_inv), ptr(self.extended_ifft_divisor)))
        return a[: self.n * self.quotient_poly_degree]


# ------------------------------------------------------------------------------------------ poly/kzg/commitment.rs
class SrsSlice:
    """&params.g[..n] / &params.g_lagrange[..n]: a registered, HBM-resident basis plus (offset, n)."""

    def __init__(self, handle: int, offset: int, n: int):
        self.handle, self.offset, self.n = handle, offset, n


class ParamsKZG:
    """ParamsKZG<Bn256> { k, n, g, g_lagrange } with both bases resident in HBM for the life of the object
    (the reference keeps them in the process-wide params_map [REF bin/src/trace_prover.rs:35-43])."""

    def __init__(self, k: int, g_handle: int, gl_handle: int, owner=None):
        self.k, self.n = k, 1 << k
        self._g, self._gl, self._owner = g_handle, gl_handle, owner
        self.g2, self.s_g2 = bytes(128), bytes(128)   # G2Affine bytes as in the params file; setup() / params_from_file() fill them

    @classmethod
    def from_host(cls, k: int, g: np.ndarray, g_lagrange: np.ndarray) -> "ParamsKZG":
        """what `Prover::load_params_map` hands down, registered once."""
        hs = []
        for arr in (g, g_lagrange):
            arr = np.ascontiguousarray(arr, dtype=np.uint64)
            assert arr.shape == (1 << k, 8)
            h = C.c_uint64()
            check(lib().mi355_srs_register_host(ptr(arr), 1 << k, C.byref(h)))
            hs.append(h.value)
        return cls(k, hs[0], hs[1])

    @classmethod
    def setup(cls, k: int, tau: int) -> "ParamsKZG":
        """ParamsKZG::setup(k, rng) with the toxic scalar given explicitly: g[i] = tau^i G, g_lagrange[i] = L_i(tau) G, built on the device."""
        import torch
        n = 1 << k
        g = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
        gl = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
        w = pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD)
        check(lib().mi355_srs_setup_dev(ptr(g), ptr(gl), k, ptr(fr(tau)), ptr(fr(w))))
        hs = []
        for t in (g, gl):
            h = C.c_uint64()
            check(lib().mi355_srs_register_dev(ptr(t), n, 0, C.byref(h)))
            hs.append(h.value)
        p = cls(k, hs[0], hs[1], owner=(g, gl))
        p.g2 = g2_generator().tobytes()                               # g2 = G2 generator, s_g2 = tau * g2 [EXT-recalled ParamsKZG::setup]
        p.s_g2 = g2_mul(g2_generator(), fr(tau)).tobytes()
        return p

    def clone_downsized(self, k: int) -> "ParamsKZG":
        """`let mut p = params.clone(); p.downsize(k)` of load_params_map [REF integration/tests/integration.rs:12-22]: the clone's g is a
        PREFIX VIEW of this object's registration (mi355_srs_register_prefix: same device memory, same window tables), its g_lagrange is
        rebuilt on the device.  Either object may be released first."""
        assert k <= self.k
        h = C.c_uint64()
        check(lib().mi355_srs_register_prefix(self._g, 1 << k, C.byref(h)))
        w_inv = pow(pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD), R_MOD - 2, R_MOD)
        hl = C.c_uint64()
        check(lib().mi355_srs_downsize(h.value, k, ptr(fr(w_inv)), ptr(fr(pow(1 << k, R_MOD - 2, R_MOD))), C.byref(hl)))
        p = ParamsKZG(k, h.value, hl.value, owner=self._owner)
        p.g2, p.s_g2 = getattr(self, "g2", bytes(128)), getattr(self, "s_g2", bytes(128))
        return p

    def commit(self, poly_coeff) -> np.ndarray:
        """commit(&self, poly: &Polynomial<_, Coeff>, _: Blind) = best_multiexp(poly, &self.g[..poly.len()])"""
        n = poly_coeff.shape[0] if not _is_device(poly_coeff) else poly_coeff.numel() * poly_coeff.element_size() // 32
        return best_multiexp(_as_scalars(poly_coeff, n), SrsSlice(self._g, 0, n))

    def commit_lagrange(self, poly_lagrange) -> np.ndarray:
        """commit_lagrange = best_multiexp(poly, &self.g_lagrange[..poly.len()])"""
        n = poly_lagrange.shape[0] if not _is_device(poly_lagrange) else poly_lagrange.numel() * poly_lagrange.element_size() // 32
        assert n == self.n, "commit_lagrange: polynomial must have exactly n evaluations"
        return best_multiexp(_as_scalars(poly_lagrange, n), SrsSlice(self._gl, 0, n))

    def precompute(self, n_hint: int = 0, c: int = 0, lagrange: bool = True, coeff: bool = True) -> None:
        """registration-time window tables 2^(c w) * P for the two bases (mi355_srs_precompute): W x the HBM, fewer windows, no Horner tail."""
        if coeff:
            check(lib().mi355_srs_precompute(self._g, n_hint, c))
        if lagrange:
            check(lib().mi355_srs_precompute(self._gl, n_hint, c))

    def commit_many(self, polys, lagrange: bool = False) -> np.ndarray:
        """commit / commit_lagrange for a list of polynomials of equal length (all device tensors or all host arrays) in ONE pass
        (mi355_msm_g1_batch_dev / _host): [M, 12].  create_proof commits the advice columns of a phase one after the other
        [EXT halo2_proofs src/plonk/prover.rs, SURVEY 3.2 step 2]; this is that loop as one call."""
        out = np.zeros((len(polys), 12), dtype=np.uint64)
        if len(polys) == 0:
            return out
        dev = _is_device(polys[0])
        assert all(_is_device(p) == dev for p in polys), "commit_many: mix of host and device polynomials"
        if dev:
            n = polys[0].numel() * polys[0].element_size() // 32
            assert all(p.numel() * p.element_size() // 32 == n for p in polys), "commit_many: polynomials must have equal length"
            arr = (C.c_void_p * len(polys))(*[p.data_ptr() for p in polys])
            fn = lib().mi355_msm_g1_batch_dev
        else:
            polys = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
            n = polys[0].shape[0]
            assert all(p.shape == (n, 4) for p in polys), "commit_many: polynomials must have equal length"
            arr = (C.c_void_p * len(polys))(*[p.ctypes.data for p in polys])
            fn = lib().mi355_msm_g1_batch_host
        assert n <= self.n and (not lagrange or n == self.n)
        check(fn(self._gl if lagrange else self._g, 0, arr, len(polys), n, ptr(out)))
        return out

    def downsize(self, k: int) -> None:
        """ParamsKZG::downsize(k) [REF integration/tests/integration.rs:17-22]: keep g[..2^k], rebuild g_lagrange = g_to_lagrange(g, k)
        (a size-2^k inverse DFT over G1 points, on the device).  The coefficient basis keeps its registration (and window tables)."""
        assert k <= self.k, "downsize: k must not exceed the current degree"
        if k == self.k:
            return
        w_inv = pow(pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD), R_MOD - 2, R_MOD)
        h = C.c_uint64()
        check(lib().mi355_srs_downsize(self._g, k, ptr(fr(w_inv)), ptr(fr(pow(1 << k, R_MOD - 2, R_MOD))), C.byref(h)))
        check(lib().mi355_srs_release(self._gl))
        self._gl = h.value
        self.k, self.n = k, 1 << k

    def read_g(self, lagrange: bool = False) -> np.ndarray:
        """the first n points of g (or g_lagrange) back on the host: [n, 8] (what ParamsKZG::write serialises)."""
        out = np.zeros((self.n, 8), dtype=np.uint64)
        check(lib().mi355_srs_read_host(self._gl if lagrange else self._g, 0, self.n, ptr(out)))
        return out

    def g_slice(self, offset: int, n: int) -> SrsSlice:
        return SrsSlice(self._g, offset, n)

    def g_lagrange_slice(self, offset: int, n: int) -> SrsSlice:
        return SrsSlice(self._gl, offset, n)

    def write(self, path: str) -> None:
        """ParamsKZG::write (SerdeFormat::RawBytes): a file prover::load_params / params_from_file accepts."""
        write_params(path, self.k, self.read_g(), self.read_g(lagrange=True), self.g2, self.s_g2)

    def release(self) -> None:
        for h in (self._g, self._gl):
            check(lib().mi355_srs_release(h))
        self._owner = None


# ------------------------------------------------------------------------------------------ wire codec of G1 (proof / .vkey bytes)
def g1_to_bytes(g1) -> bytes:
    """compressed encoding written to transcripts and .vkey files [EXT-recalled halo2curves derive/curve.rs; pinned by fixture KAT A4]:
    32 bytes little-endian x, bit 254 = parity of canonical y, identity = 32 zero bytes.  Accepts G1Affine (8 limbs) or a normalised
    G1 (12 limbs) as returned by best_multiexp.  Host-side (the transcript lives on the host, as in the reference)."""
    v = [int(t) for t in g1]
    if len(v) == 12:
        assert v[8:] == [0, 0, 0, 0] or sum(l << (64 * i) for i, l in enumerate(v[8:])) == (1 << 256) % P_MOD, "normalise first"
        if v[8:] == [0, 0, 0, 0]:
            return bytes(32)
    rinv = pow(1 << 256, -1, P_MOD)
    x = sum(l << (64 * i) for i, l in enumerate(v[0:4])) * rinv % P_MOD
    y = sum(l << (64 * i) for i, l in enumerate(v[4:8])) * rinv % P_MOD
    if x == 0 and y == 0:
        return bytes(32)
    return (x | ((y & 1) << 254)).to_bytes(32, "little")


def g1_from_bytes(b: bytes) -> np.ndarray:
    """inverse of g1_to_bytes -> G1Affine limbs (Montgomery); raises ValueError for x not on the curve (y^2 = x^3 + 3, p = 3 mod 4)."""
    v = int.from_bytes(b, "little")
    sign, x = (v >> 254) & 1, v & ((1 << 254) - 1)
    if x == 0 and sign == 0:
        return np.zeros(8, dtype=np.uint64)
    if x >= P_MOD:
        raise ValueError("x out of range")
    y2 = (x * x * x + 3) % P_MOD
    y = pow(y2, (P_MOD + 1) // 4, P_MOD)
    if y * y % P_MOD != y2:
        raise ValueError("not on the curve")
    if (y & 1) != sign:
        y = P_MOD - y
    m = lambda t: [((t << 256) % P_MOD >> (64 * i)) & _M64 for i in range(4)]
    return np.array(m(x) + m(y), dtype=np.uint64)


# ------------------------------------------------------------------------------------------ params files (SerdeFormat::RawBytes)
def params_file_size(k: int) -> int:
    """`u32 LE k | g[2^k] x 64 B | g_lagrange[2^k] x 64 B | g2 128 B | s_g2 128 B` [EXT-recalled ParamsKZG::write_custom, RawBytes];
    params26 = 8 589 934 852 bytes.  `prover::load_params` rejects any other length."""
    return 4 + 2 * (1 << k) * 64 + 256


def write_params(path: str, k: int, g: np.ndarray, g_lagrange: np.ndarray, g2: bytes = bytes(128), s_g2: bytes = bytes(128)) -> None:
    g = np.ascontiguousarray(g, dtype=np.uint64); g_lagrange = np.ascontiguousarray(g_lagrange, dtype=np.uint64)
    assert g.shape == (1 << k, 8) and g_lagrange.shape == (1 << k, 8) and len(g2) == 128 and len(s_g2) == 128
    with open(path, "wb") as f:
        f.write(int(k).to_bytes(4, "little")); f.write(g.tobytes()); f.write(g_lagrange.tobytes()); f.write(g2); f.write(s_g2)


def read_params(path: str):
    """-> (k, g [n,8], g_lagrange [n,8], g2 bytes, s_g2 bytes); memory-mapped so a params26 file is not copied twice."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        k = int.from_bytes(f.read(4), "little")
    if not (0 < k <= FR_S) or size != params_file_size(k):
        raise ValueError(f"{path}: not a RawBytes params file (k={k}, {size} bytes, expected {params_file_size(k) if 0 < k <= FR_S else 'n/a'})")
    n = 1 << k
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    g = mm[4: 4 + n * 64].view(np.uint64).reshape(n, 8)
    gl = mm[4 + n * 64: 4 + 2 * n * 64].view(np.uint64).reshape(n, 8)
    tail = bytes(mm[4 + 2 * n * 64:])
    return k, g, gl, tail[:128], tail[128:]


def params_from_file(path: str, validate: bool = False, downsize_to: int = 0) -> "ParamsKZG":
    """Prover::load_params(dir, degree) [REF bin/src/trace_prover.rs:35-36] for one degree: the file is streamed into HBM by the library
    (mi355_srs_load_params_file: pinned double-buffered reads, exact-length rule, optional on-device point validation) and both bases
    are registered; downsize_to < k reproduces load_params on a larger file (g truncated, g_lagrange rebuilt on the device)."""
    k, hg, hl = C.c_uint32(), C.c_uint64(), C.c_uint64()
    g2 = (C.c_uint8 * 128)(); s_g2 = (C.c_uint8 * 128)()
    check(lib().mi355_srs_load_params_file(path.encode(), 1 if validate else 0, C.byref(k), C.byref(hg), C.byref(hl), g2, s_g2))
    p = ParamsKZG(k.value, hg.value, hl.value)
    p.g2, p.s_g2 = bytes(g2), bytes(s_g2)
    if downsize_to and downsize_to < p.k:
        p.downsize(downsize_to)
    return p


class _Scalars:
    """adapter giving device tensors the `.shape[0]` best_multiexp expects"""

    def __init__(self, t, n):
        self._t, self.shape = t, (n,)
        self.is_cuda = True

    def data_ptr(self):
        return self._t.data_ptr()


def _as_scalars(x, n):
    return _Scalars(x, n) if _is_device(x) else np.ascontiguousarray(x, dtype=np.uint64)
