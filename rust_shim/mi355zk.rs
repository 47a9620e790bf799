//! mi355zk.rs -- the Rust side of the drop-in: goes into the fork of `halo2_proofs` (scroll-tech/halo2 @ e5ddf67,
//! the crate scroll-prover's GPU images already replace wholesale [REF docker/chain-prover/gpu/Dockerfile:7-8]) as
//! `halo2_proofs/src/mi355zk.rs`, with `build.rs` emitting `cargo:rustc-link-lib=dylib=mi355zk`.
//!
//! NOT compiled in this repository's container (no rustc/cargo).  Its behaviour -- who owns a handle, when it
//! is released, what clone / downsize do, which slices are never registered -- is replayed call for call through the same C-ABI by
//! the compiled test `tests/cpp/test_shim_replay.cpp` (`pytest -m gpu tests/test_cpp_mirror.py`).
//!
//! Ownership model (round 2; the round-1 address-keyed map is gone -- a freed and re-used Vec address could select a stale basis):
//!   * a registration is a `GpuBasis(u64)`; `Drop` releases it (`mi355_srs_release`);
//!   * `ParamsKZG` carries `gpu_g` / `gpu_g_lagrange: Option<Arc<GpuBasis>>`, filled by `setup` / `read_custom`; `clone()` shares
//!     the Arcs (no second upload, no second 48 GiB window table);
//!   * `downsize(k)`: `g.truncate(n)` -> `mi355_srs_register_prefix` (a view that shares memory and tables), `g_lagrange` is rebuilt
//!     on the device (`mi355_srs_downsize` + `mi355_srs_read_host`) -- the `load_params_map` pattern
//!     [REF integration/tests/integration.rs:12-22], [REF bin/src/trace_prover.rs:35-36];
//!   * `commit` / `commit_lagrange` use the handle of `self`; the generic `best_multiexp(coeffs, bases)` only sees slices and
//!     therefore goes through `mi355_msm_g1_adhoc_host` -- nothing is ever registered by address.
//!
//! Resident polynomials (round 3): `DevicePoly` owns one `mi355_buf_alloc` block (Drop -> `mi355_buf_free`, which recycles the block without a
//! device synchronisation).  create_proof uploads each witness column ONCE (`DevicePoly::from_slice`; the DMA overlaps the commitment of the
//! previous column because `mi355_buf_upload` does not hold the device lock while it runs) and then commits, transforms, combines, evaluates
//! and opens it through the `*_dev` entry points -- host memory sees the 96-byte commitments and 32-byte evaluations only.  Measured with the
//! compiled stand-in `tests/cpp/test_create_proof_replay.cpp`: one layer-4 proof's GPU side 1.51 s resident against 6.37 s through `*_host`.
//!
//! Threading: every entry point may be called from any rayon worker; locks are per device.  `mi355_msm_set_normalise` /
//! `mi355_msm_set_window_bits` act on the CALLING THREAD only -- set them on the thread that issues the MSM.
//!
//! Layout contract asserted at start-up: size_of::<Fr>() == 32, size_of::<G1Affine>() == 64,
//! size_of::<G1>() == 96 and Fr::one() serialises to R = 2^256 mod r in little-endian u64 limbs (fixture KAT A1).
#![allow(non_camel_case_types)]
use std::any::TypeId;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::{Arc, Once};

use halo2curves::bn256::{Fr, G1Affine, G1};

pub const MI355_OK: c_int = 0;

extern "C" {
    pub fn mi355_init(device_id: c_int) -> c_int;
    pub fn mi355_init_multi(device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn mi355_last_error() -> *const c_char;
    pub fn mi355_srs_register_host(bases_affine_host: *const c_void, n: u64, handle_out: *mut u64) -> c_int;
    pub fn mi355_srs_register_prefix(parent_handle: u64, n: u64, handle_out: *mut u64) -> c_int;
    pub fn mi355_srs_release(handle: u64) -> c_int;
    pub fn mi355_srs_precompute(handle: u64, n_hint: u64, c: c_int) -> c_int;
    pub fn mi355_srs_downsize(g_handle: u64, k: u32, omega_inv: *const c_void, n_inv: *const c_void, g_lagrange_handle_out: *mut u64) -> c_int;
    pub fn mi355_srs_read_host(handle: u64, offset: u64, n: u64, out_affine_host: *mut c_void) -> c_int;
    pub fn mi355_g1_fft_host(points_jac_host: *mut c_void, log_n: u32, omega: *const c_void) -> c_int;
    pub fn mi355_msm_g1_host(srs: u64, base_offset: u64, scalars_host: *const c_void, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_msm_g1_batch_host(srs: u64, base_offset: u64, scalars_host: *const *const c_void, batch: u32, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_msm_g1_adhoc_host(bases: *const c_void, scalars: *const c_void, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_g1_batch_normalize_host(g1_points_host: *const c_void, affine_out_host: *mut c_void, n: u64) -> c_int;
    pub fn mi355_ntt_fr_host(data_host: *mut c_void, log_n: u32, omega: *const c_void) -> c_int;
    pub fn mi355_intt_fr_host(data_host: *mut c_void, log_n: u32, omega_inv: *const c_void, divisor: *const c_void) -> c_int;
    pub fn mi355_coeff_to_extended_host(dst: *mut c_void, coeffs: *const c_void, log_n: u32, log_ext: u32,
                                        g_coset: *const c_void, g_coset_inv: *const c_void, extended_omega: *const c_void) -> c_int;
    pub fn mi355_extended_to_coeff_host(data: *mut c_void, log_ext: u32, g_coset: *const c_void, g_coset_inv: *const c_void,
                                        extended_omega_inv: *const c_void, extended_ifft_divisor: *const c_void) -> c_int;
    // ---- resident buffers and the `*_dev` half of the ABI (include/mi355zk.h)
    pub fn mi355_buf_alloc(bytes: u64, device_slot: c_int, dev_ptr_out: *mut *mut c_void) -> c_int;
    pub fn mi355_buf_free(dev_ptr: *mut c_void) -> c_int;
    pub fn mi355_buf_upload(dst_dev: *mut c_void, src_host: *const c_void, bytes: u64) -> c_int;
    pub fn mi355_buf_download(dst_host: *mut c_void, src_dev: *const c_void, bytes: u64) -> c_int;
    pub fn mi355_buf_upload_packed(dst_dev: *mut c_void, src_host: *const c_void, n: u64, width_bytes: u32) -> c_int;
    pub fn mi355_buf_upload_sparse(dst_dev: *mut c_void, n: u64, idx_host: *const u32, vals_host: *const c_void, count: u64) -> c_int;
    pub fn mi355_host_compact_nonzero(src_host: *const c_void, n: u64, idx_out: *mut u32, vals_out: *mut c_void, count_out: *mut u64, threads: c_int) -> c_int;
    pub fn mi355_buf_copy(dst_dev: *mut c_void, src_dev: *const c_void, bytes: u64) -> c_int;
    pub fn mi355_buf_zero(dst_dev: *mut c_void, bytes: u64) -> c_int;
    pub fn mi355_host_alloc(bytes: u64, host_ptr_out: *mut *mut c_void) -> c_int;
    pub fn mi355_host_free(host_ptr: *mut c_void) -> c_int;
    pub fn mi355_mem_info(device_slot: c_int, free_bytes: *mut u64, total_bytes: *mut u64, live_buf_bytes: *mut u64, pooled_bytes: *mut u64, workspace_bytes: *mut u64) -> c_int;
    pub fn mi355_msm_g1_dev(srs: u64, base_offset: u64, scalars_dev: *const c_void, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_msm_g1_batch_dev(srs: u64, base_offset: u64, scalars_dev: *const *const c_void, batch: u32, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_intt_fr_dev(data_dev: *mut c_void, log_n: u32, omega_inv: *const c_void, divisor: *const c_void) -> c_int;
    pub fn mi355_ntt_fr_batch_host(data_host: *const *mut c_void, batch: u32, log_n: u32, omega: *const c_void, divisor: *const c_void) -> c_int;
    pub fn mi355_ntt_fr_batch_dev(data_dev: *const *mut c_void, batch: u32, log_n: u32, omega: *const c_void, divisor: *const c_void) -> c_int;
    pub fn mi355_coset_ntt_fr_batch_dev(dst_dev: *const *mut c_void, coeffs_dev: *const *const c_void, batch: u32, log_n: u32,
                                        coset_factor: *const c_void, omega: *const c_void) -> c_int;
    pub fn mi355_extended_to_coeff_dev(data_dev: *mut c_void, log_ext: u32, g_coset: *const c_void, g_coset_inv: *const c_void,
                                       extended_omega_inv: *const c_void, extended_ifft_divisor: *const c_void) -> c_int;
    pub fn mi355_fr_gate_eval_dev(dst_dev: *mut c_void, polys_dev: *const *const c_void, n_polys: u32, coeffs_fr_host: *const c_void,
                                  term_len: *const u32, n_terms: u32, factor_poly: *const u32, factor_rot: *const i32, n: u64, accumulate: c_int) -> c_int;
    pub fn mi355_fr_interleave_dev(dst_dev: *mut c_void, parts_dev: *const *const c_void, q_parts: u32, n: u64) -> c_int;
    pub fn mi355_fr_batch_invert_dev(data_dev: *mut c_void, n: u64) -> c_int;
    pub fn mi355_fr_prefix_product_dev(dst_dev: *mut c_void, src_dev: *const c_void, n: u64, total_out_host: *mut c_void) -> c_int;
    pub fn mi355_fr_prefix_sum_dev(dst_dev: *mut c_void, src_dev: *const c_void, n: u64, total_out_host: *mut c_void) -> c_int;
    pub fn mi355_fr_kate_division_dev(dst_dev: *mut c_void, poly_dev: *const c_void, n: u64, z: *const c_void) -> c_int;
    pub fn mi355_eval_polynomial_batch_dev(polys_dev: *const *const c_void, batch: u32, n: u64, points: *const c_void, out_fr_host: *mut c_void) -> c_int;
    pub fn mi355_eval_polynomial_dev(poly_dev: *const c_void, n: u64, point: *const c_void, out_fr_host: *mut c_void) -> c_int;
}

/// Offload threshold: below this the PCIe copy + launch latency lose against rayon (env MI355_MSM_MIN_LOGN / MI355_NTT_MIN_LOGN).
fn min_log(var: &str, default: u32) -> u32 { std::env::var(var).ok().and_then(|v| v.parse().ok()).unwrap_or(default) }

static INIT: Once = Once::new();
static mut AVAILABLE: bool = false;

/// MI355_DEVICES="0,1,2,3,4,5,6,7" binds several GPUs to this one prover process (bases are sharded by point range, every MSM fans
/// out behind the same call, partial sums meet in one ncclAllGather); MI355_DEVICE / default 0 binds one.  Never panics: on any
/// failure the caller keeps the CPU path.
pub fn available() -> bool {
    INIT.call_once(|| unsafe {
        assert_eq!(std::mem::size_of::<Fr>(), 32);
        assert_eq!(std::mem::size_of::<G1Affine>(), 64);
        assert_eq!(std::mem::size_of::<G1>(), 96);
        let one: [u64; 4] = std::mem::transmute(Fr::one());
        assert_eq!(one, [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]);
        let rc = match std::env::var("MI355_DEVICES") {
            Ok(list) => {
                let ids: Vec<c_int> = list.split(',').filter_map(|s| s.trim().parse().ok()).collect();
                if ids.is_empty() { mi355_init(0) } else { mi355_init_multi(ids.as_ptr(), ids.len() as c_int) }
            }
            Err(_) => mi355_init(std::env::var("MI355_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)),
        };
        AVAILABLE = rc == MI355_OK;
        if !AVAILABLE { log::warn!("mi355zk unavailable: {:?}; using the CPU path", std::ffi::CStr::from_ptr(mi355_last_error())); }
    });
    unsafe { AVAILABLE }
}

/// One registered basis in HBM.  Owned through `Arc` by every `ParamsKZG` that shares it; released when the last owner drops.
#[derive(Debug)]
pub struct GpuBasis(pub u64);
impl Drop for GpuBasis {
    fn drop(&mut self) { unsafe { let _ = mi355_srs_release(self.0); } }
}

impl GpuBasis {
    /// Called by ParamsKZG::{setup, read_custom} once `g` / `g_lagrange` are final.
    pub fn register(bases: &[G1Affine]) -> Option<Arc<GpuBasis>> {
        if !available() || bases.is_empty() { return None; }
        let mut h = 0u64;
        if unsafe { mi355_srs_register_host(bases.as_ptr() as *const c_void, bases.len() as u64, &mut h) } != MI355_OK { return None; }
        // registration-time window tables (W x the basis in HBM; MI355_SRS_PRECOMPUTE=0 disables): all windows share one bucket set
        if std::env::var("MI355_SRS_PRECOMPUTE").map(|v| v != "0").unwrap_or(true) { unsafe { let _ = mi355_srs_precompute(h, 0, 0); } }
        Some(Arc::new(GpuBasis(h)))
    }
    /// `&g[..n]` after `g.truncate(n)`: a view that shares device memory and window tables with `self`.
    pub fn prefix(self: &Arc<Self>, n: usize) -> Option<Arc<GpuBasis>> {
        let mut h = 0u64;
        if unsafe { mi355_srs_register_prefix(self.0, n as u64, &mut h) } != MI355_OK { return None; }
        Some(Arc::new(GpuBasis(h)))
    }
    /// best_multiexp(coeffs, &basis[..coeffs.len()]) on the registered basis.
    pub fn multiexp(&self, coeffs: &[Fr]) -> Option<G1> {
        if (coeffs.len() as u64) < (1u64 << min_log("MI355_MSM_MIN_LOGN", 14)) { return None; }
        let mut out = std::mem::MaybeUninit::<G1>::uninit();
        let rc = unsafe { mi355_msm_g1_host(self.0, 0, coeffs.as_ptr() as *const c_void, coeffs.len() as u64, out.as_mut_ptr() as *mut c_void) };
        if rc == MI355_OK { Some(unsafe { out.assume_init() }) } else { None }
    }
    /// One call for the `polys.iter().map(|p| params.commit_lagrange(p, blind))` loops of create_proof (advice / instance / lookup /
    /// permutation columns of one phase): equal-length polynomials over this basis.  None -> caller runs the per-polynomial loop.
    pub fn multiexp_many(&self, polys: &[&[Fr]]) -> Option<Vec<G1>> {
        if polys.is_empty() { return Some(vec![]); }
        let n = polys[0].len();
        assert!(polys.iter().all(|p| p.len() == n));
        if (n as u64) < (1u64 << min_log("MI355_MSM_MIN_LOGN", 14)) { return None; }
        let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr() as *const c_void).collect();
        let mut out: Vec<G1> = Vec::with_capacity(polys.len());
        let rc = unsafe { mi355_msm_g1_batch_host(self.0, 0, ptrs.as_ptr(), polys.len() as u32, n as u64, out.as_mut_ptr() as *mut c_void) };
        if rc != MI355_OK { return None; }
        unsafe { out.set_len(polys.len()); }
        Some(out)
    }
    /// `g_to_lagrange(&g[..2^k], k)` for `ParamsKZG::downsize` / `setup`: a size-2^k inverse FFT over curve points (minutes on the
    /// CPU).  Returns the new `g_lagrange` Vec and its registration.  None -> run the CPU code.
    pub fn g_to_lagrange(&self, k: u32, omega_inv: &Fr, n_inv: &Fr) -> Option<(Vec<G1Affine>, Arc<GpuBasis>)> {
        let n = 1usize << k;
        let mut hl = 0u64;
        if unsafe { mi355_srs_downsize(self.0, k, omega_inv as *const Fr as *const c_void, n_inv as *const Fr as *const c_void, &mut hl) } != MI355_OK { return None; }
        let gl = Arc::new(GpuBasis(hl));
        let mut out: Vec<G1Affine> = Vec::with_capacity(n);
        if unsafe { mi355_srs_read_host(hl, 0, n as u64, out.as_mut_ptr() as *mut c_void) } != MI355_OK { return None; }
        unsafe { out.set_len(n); }
        if std::env::var("MI355_SRS_PRECOMPUTE").map(|v| v != "0").unwrap_or(true) { unsafe { let _ = mi355_srs_precompute(hl, 0, 0); } }
        Some((out, gl))
    }
}

/// One polynomial (or any vector of `Fr`) resident in HBM: a `mi355_buf_alloc` block.  `Drop` hands the block back to the library's pool
/// (no `hipFree`, no device synchronisation; work already queued on it stays valid).  `slot` picks the device of an `MI355_DEVICES` process.
#[derive(Debug)]
pub struct DevicePoly { ptr: *mut c_void, len: usize, slot: c_int }
unsafe impl Send for DevicePoly {}
impl Drop for DevicePoly {
    fn drop(&mut self) { if !self.ptr.is_null() { unsafe { let _ = mi355_buf_free(self.ptr); } } }
}
impl DevicePoly {
    pub fn zeroed(len: usize, slot: c_int) -> Option<DevicePoly> {
        if !available() { return None; }
        let mut p: *mut c_void = std::ptr::null_mut();
        if unsafe { mi355_buf_alloc((len * 32) as u64, slot, &mut p) } != MI355_OK { return None; }
        let d = DevicePoly { ptr: p, len, slot };
        if unsafe { mi355_buf_zero(d.ptr, (len * 32) as u64) } != MI355_OK { return None; }
        Some(d)
    }
    /// The witness upload: the only bulk host -> device traffic of a proof.  Returns when `v` may be reused; the copy overlaps whatever the
    /// device is computing for other threads.
    pub fn from_slice(v: &[Fr], slot: c_int) -> Option<DevicePoly> {
        if !available() || v.is_empty() { return None; }
        let mut p: *mut c_void = std::ptr::null_mut();
        if unsafe { mi355_buf_alloc((v.len() * 32) as u64, slot, &mut p) } != MI355_OK { return None; }
        let d = DevicePoly { ptr: p, len: v.len(), slot };
        if unsafe { mi355_buf_upload(d.ptr, v.as_ptr() as *const c_void, (v.len() * 32) as u64) } != MI355_OK { return None; }
        Some(d)
    }
    /// The same upload for a column that is mostly zeros (selectors, padding rows, sparse lookup inputs): only the non-zero cells cross PCIe, as (index, value) pairs.
    /// `scratch` (idx, vals) is reused across columns; below `min_zero_fraction` of zeros the plain upload is cheaper and is used instead.
    pub fn from_slice_sparse(v: &[Fr], slot: c_int, scratch: &mut (Vec<u32>, Vec<Fr>), threads: c_int, min_zero_fraction: f64) -> Option<DevicePoly> {
        if !available() || v.is_empty() || v.len() > (1usize << 32) { return None; }
        scratch.0.resize(v.len(), 0); scratch.1.resize(v.len(), Fr::zero());
        let mut count: u64 = 0;
        if unsafe { mi355_host_compact_nonzero(v.as_ptr() as *const c_void, v.len() as u64, scratch.0.as_mut_ptr(), scratch.1.as_mut_ptr() as *mut c_void, &mut count, threads) } != MI355_OK { return None; }
        if (count as f64) > (1.0 - min_zero_fraction) * v.len() as f64 { return DevicePoly::from_slice(v, slot); }
        let mut p: *mut c_void = std::ptr::null_mut();
        if unsafe { mi355_buf_alloc((v.len() * 32) as u64, slot, &mut p) } != MI355_OK { return None; }
        let d = DevicePoly { ptr: p, len: v.len(), slot };
        if unsafe { mi355_buf_upload_sparse(d.ptr, v.len() as u64, scratch.0.as_ptr(), scratch.1.as_ptr() as *const c_void, count) } != MI355_OK { return None; }
        Some(d)
    }
    /// A column whose kind bounds its cells (selector bits, bytes, range-checked limbs below 2^lookup_bits, 64-bit words): the caller hands the CANONICAL values as
    /// `width`-byte little-endian integers and `width` bytes per cell cross the link.
    pub fn from_packed(values_le: &[u8], n: usize, width: u32, slot: c_int) -> Option<DevicePoly> {
        if !available() || n == 0 || values_le.len() != n * width as usize { return None; }
        let mut p: *mut c_void = std::ptr::null_mut();
        if unsafe { mi355_buf_alloc((n * 32) as u64, slot, &mut p) } != MI355_OK { return None; }
        let d = DevicePoly { ptr: p, len: n, slot };
        if unsafe { mi355_buf_upload_packed(d.ptr, values_le.as_ptr() as *const c_void, n as u64, width) } != MI355_OK { return None; }
        Some(d)
    }
    pub fn len(&self) -> usize { self.len }
    pub fn as_ptr(&self) -> *const c_void { self.ptr }
    pub fn as_mut_ptr(&mut self) -> *mut c_void { self.ptr }
    /// Back to a `Vec<Fr>` (tests, and the few places that still want the values on the CPU).
    pub fn to_vec(&self) -> Option<Vec<Fr>> {
        let mut out: Vec<Fr> = Vec::with_capacity(self.len);
        if unsafe { mi355_buf_download(out.as_mut_ptr() as *mut c_void, self.ptr, (self.len * 32) as u64) } != MI355_OK { return None; }
        unsafe { out.set_len(self.len); }
        Some(out)
    }
    /// `EvaluationDomain::lagrange_to_coeff`, in place.
    pub fn lagrange_to_coeff(&mut self, k: u32, omega_inv: &Fr, ifft_divisor: &Fr) -> bool {
        if k > 28 || self.len != 1usize << k { return false; }   // a short block would be written past its end
        unsafe { mi355_intt_fr_dev(self.ptr, k, omega_inv as *const Fr as *const c_void, ifft_divisor as *const Fr as *const c_void) == MI355_OK }
    }
    /// `eval_polynomial(self, point)`.
    pub fn eval(&self, point: &Fr) -> Option<Fr> {
        let mut out = std::mem::MaybeUninit::<Fr>::uninit();
        let rc = unsafe { mi355_eval_polynomial_dev(self.ptr, self.len as u64, point as *const Fr as *const c_void, out.as_mut_ptr() as *mut c_void) };
        if rc == MI355_OK { Some(unsafe { out.assume_init() }) } else { None }
    }
}
impl GpuBasis {
    /// commit / commit_lagrange of a resident polynomial: the scalars never leave HBM (with several devices each shard's slice crosses xGMI).
    pub fn multiexp_dev(&self, poly: &DevicePoly) -> Option<G1> {
        // (a polynomial longer than the registered basis is rejected by the library: MI355_EBADARG -> None -> CPU path)
        let mut out = std::mem::MaybeUninit::<G1>::uninit();
        let rc = unsafe { mi355_msm_g1_dev(self.0, 0, poly.as_ptr(), poly.len() as u64, out.as_mut_ptr() as *mut c_void) };
        if rc == MI355_OK { Some(unsafe { out.assume_init() }) } else { None }
    }
    /// The per-column commit loop of a phase over resident columns, as one pass.
    pub fn multiexp_many_dev(&self, polys: &[&DevicePoly]) -> Option<Vec<G1>> {
        if polys.is_empty() { return Some(vec![]); }
        let n = polys[0].len();
        assert!(polys.iter().all(|p| p.len() == n));
        let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr()).collect();
        let mut out: Vec<G1> = Vec::with_capacity(polys.len());
        if unsafe { mi355_msm_g1_batch_dev(self.0, 0, ptrs.as_ptr(), polys.len() as u32, n as u64, out.as_mut_ptr() as *mut c_void) } != MI355_OK { return None; }
        unsafe { out.set_len(polys.len()); }
        Some(out)
    }
}
/// `polys.iter_mut().for_each(|p| domain.lagrange_to_coeff(p))` (divisor = Some(n^-1)) or a loop of `best_fft` (None) over resident
/// polynomials as ONE call: with several devices the independent transforms run concurrently where their buffers live.
pub fn fft_many_dev(polys: &mut [&mut DevicePoly], k: u32, omega: &Fr, divisor: Option<&Fr>) -> bool {
    if k > 28 || polys.iter().any(|p| p.len() != 1usize << k) { return false; }   // as fft_many: every operand is exactly 2^k long
    let ptrs: Vec<*mut c_void> = polys.iter_mut().map(|p| p.as_mut_ptr()).collect();
    let d = divisor.map(|d| d as *const Fr as *const c_void).unwrap_or(std::ptr::null());
    unsafe { mi355_ntt_fr_batch_dev(ptrs.as_ptr(), ptrs.len() as u32, k, omega as *const Fr as *const c_void, d) == MI355_OK }
}
/// The same loop over HOST polynomials (a shim that has not adopted `DevicePoly`): uploads, transforms and downloads of neighbouring
/// items overlap inside the library, and the items are dealt over the bound devices.
pub fn fft_many(polys: &mut [&mut [Fr]], k: u32, omega: &Fr, divisor: Option<&Fr>) -> bool {
    if polys.iter().any(|p| p.len() != 1usize << k) { return false; }
    let ptrs: Vec<*mut c_void> = polys.iter_mut().map(|p| p.as_mut_ptr() as *mut c_void).collect();
    let d = divisor.map(|d| d as *const Fr as *const c_void).unwrap_or(std::ptr::null());
    unsafe { mi355_ntt_fr_batch_host(ptrs.as_ptr(), ptrs.len() as u32, k, omega as *const Fr as *const c_void, d) == MI355_OK }
}
/// One term list of `evaluate_h` on one coset part: `dst[i] (+)= sum_j coeffs[j] * prod polys[p][(i + rot) mod n]` in ONE launch.
/// `terms[j]` = (coefficient, [(index into polys, rotation in elements)]).  At most 16 terms / 48 factors / 16 factors per term per call (split and accumulate).
pub fn gate_eval_dev(dst: &mut DevicePoly, polys: &[&DevicePoly], terms: &[(Fr, Vec<(u32, i32)>)], accumulate: bool) -> bool {
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr()).collect();
    let coeffs: Vec<Fr> = terms.iter().map(|t| t.0).collect();
    let term_len: Vec<u32> = terms.iter().map(|t| t.1.len() as u32).collect();
    let fp: Vec<u32> = terms.iter().flat_map(|t| t.1.iter().map(|f| f.0)).collect();
    let fr: Vec<i32> = terms.iter().flat_map(|t| t.1.iter().map(|f| f.1)).collect();
    let n = dst.len() as u64;
    // the kernel reads polys[p][(i + rot) mod n] for every i < n: n must be a power of two and no operand may be shorter than dst
    if n == 0 || !n.is_power_of_two() || polys.iter().any(|p| (p.len() as u64) < n) { return false; }
    if fp.iter().any(|&p| p as usize >= polys.len()) { return false; }
    unsafe { mi355_fr_gate_eval_dev(dst.as_mut_ptr(), ptrs.as_ptr(), ptrs.len() as u32, coeffs.as_ptr() as *const c_void, term_len.as_ptr(),
                                    terms.len() as u32, fp.as_ptr(), fr.as_ptr(), n, accumulate as c_int) == MI355_OK }
}

/// The same with raw device pointers (operands that are slices of a larger block: the quotient pieces inside `h`, or proving-key polynomials held elsewhere).
/// The caller guarantees every pointer covers `dst.len()` elements.
pub fn gate_eval_ptrs(dst: &mut DevicePoly, ptrs: &[*const c_void], terms: &[(Fr, Vec<(u32, i32)>)], accumulate: bool) -> bool {
    let coeffs: Vec<Fr> = terms.iter().map(|t| t.0).collect();
    let term_len: Vec<u32> = terms.iter().map(|t| t.1.len() as u32).collect();
    let fp: Vec<u32> = terms.iter().flat_map(|t| t.1.iter().map(|f| f.0)).collect();
    let fr: Vec<i32> = terms.iter().flat_map(|t| t.1.iter().map(|f| f.1)).collect();
    let n = dst.len() as u64;
    if n == 0 || !n.is_power_of_two() || fp.iter().any(|&p| p as usize >= ptrs.len()) { return false; }
    unsafe { mi355_fr_gate_eval_dev(dst.as_mut_ptr(), ptrs.as_ptr(), ptrs.len() as u32, coeffs.as_ptr() as *const c_void, term_len.as_ptr(),
                                    terms.len() as u32, fp.as_ptr(), fr.as_ptr(), n, accumulate as c_int) == MI355_OK }
}

/// Replacement body of the generic `best_multiexp` for C = G1Affine when the bases are just a slice (not `self.g` of a ParamsKZG):
/// ad-hoc upload, nothing registered, nothing cached.  Returns None -> caller runs the original CPU code.
pub fn multiexp_g1(coeffs: &[Fr], bases: &[G1Affine]) -> Option<G1> {
    assert_eq!(coeffs.len(), bases.len());            // same panic as the reference
    if !available() || (coeffs.len() as u64) < (1u64 << min_log("MI355_MSM_MIN_LOGN", 14)) { return None; }
    let mut out = std::mem::MaybeUninit::<G1>::uninit();
    let rc = unsafe { mi355_msm_g1_adhoc_host(bases.as_ptr() as *const c_void, coeffs.as_ptr() as *const c_void, coeffs.len() as u64, out.as_mut_ptr() as *mut c_void) };
    if rc == MI355_OK { Some(unsafe { out.assume_init() }) } else { None }
}

/// `G1::batch_normalize(p, q)` for long vectors (g_to_lagrange / setup outputs; the dozen commitments of a proof stay on the CPU).
/// Returns false -> caller runs the original CPU code.
pub fn batch_normalize_g1(p: &[G1], q: &mut [G1Affine]) -> bool {
    assert_eq!(p.len(), q.len());                     // same panic as the reference
    if !available() || (p.len() as u64) < (1u64 << min_log("MI355_NORMALIZE_MIN_LOGN", 12)) { return false; }
    unsafe { mi355_g1_batch_normalize_host(p.as_ptr() as *const c_void, q.as_mut_ptr() as *mut c_void, p.len() as u64) == MI355_OK }
}

/// Replacement body of `best_fft` for G = Scalar = Fr (the G = curve-point instantiation keeps the CPU code).
pub fn fft_fr<S: 'static, G: 'static>(a: &mut [G], omega: &S, log_n: u32) -> bool {
    if TypeId::of::<S>() != TypeId::of::<Fr>() || TypeId::of::<G>() != TypeId::of::<Fr>() { return false; }
    if !available() || log_n < min_log("MI355_NTT_MIN_LOGN", 16) { return false; }
    unsafe { mi355_ntt_fr_host(a.as_mut_ptr() as *mut c_void, log_n, omega as *const S as *const c_void) == MI355_OK }
}
