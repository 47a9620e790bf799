//! mi355zk.rs -- the Rust side of the drop-in: goes into the fork of `halo2_proofs` (scroll-tech/halo2 @ e5ddf67,
//! the crate scroll-prover's GPU images already replace wholesale [REF docker/chain-prover/gpu/Dockerfile:7-8]) as
//! `halo2_proofs/src/mi355zk.rs`, with `build.rs` emitting `cargo:rustc-link-lib=dylib=mi355zk`.
//!
//! NOT compiled in this repository's container (no rustc/cargo): this is the binding a
//! maintainer adds; every `extern "C"` item mirrors include/mi355zk.h one-to-one.
//!
//! Layout contract asserted at start-up: size_of::<Fr>() == 32, size_of::<G1Affine>() == 64,
//! size_of::<G1>() == 96 and Fr::one() serialises to R = 2^256 mod r in little-endian u64 limbs (fixture KAT A1).
#![allow(non_camel_case_types)]
use std::any::TypeId;
use std::collections::HashMap;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::{Mutex, Once};

use halo2curves::bn256::{Fr, G1Affine, G1};

pub const MI355_OK: c_int = 0;

extern "C" {
    pub fn mi355_init(device_id: c_int) -> c_int;
    pub fn mi355_last_error() -> *const c_char;
    pub fn mi355_srs_register_host(bases_affine_host: *const c_void, n: u64, handle_out: *mut u64) -> c_int;
    pub fn mi355_srs_release(handle: u64) -> c_int;
    pub fn mi355_srs_precompute(handle: u64, n_hint: u64, c: c_int) -> c_int;
    pub fn mi355_srs_downsize(g_handle: u64, k: u32, omega_inv: *const c_void, n_inv: *const c_void, g_lagrange_handle_out: *mut u64) -> c_int;
    pub fn mi355_srs_read_host(handle: u64, offset: u64, n: u64, out_affine_host: *mut c_void) -> c_int;
    pub fn mi355_g1_fft_host(points_jac_host: *mut c_void, log_n: u32, omega: *const c_void) -> c_int;
    pub fn mi355_msm_g1_host(srs: u64, base_offset: u64, scalars_host: *const c_void, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_msm_g1_batch_host(srs: u64, base_offset: u64, scalars_host: *const *const c_void, batch: u32, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_msm_g1_adhoc_host(bases: *const c_void, scalars: *const c_void, n: u64, out_g1_host: *mut c_void) -> c_int;
    pub fn mi355_ntt_fr_host(data_host: *mut c_void, log_n: u32, omega: *const c_void) -> c_int;
    pub fn mi355_intt_fr_host(data_host: *mut c_void, log_n: u32, omega_inv: *const c_void, divisor: *const c_void) -> c_int;
    pub fn mi355_coeff_to_extended_host(dst: *mut c_void, coeffs: *const c_void, log_n: u32, log_ext: u32,
                                        g_coset: *const c_void, g_coset_inv: *const c_void, extended_omega: *const c_void) -> c_int;
    pub fn mi355_extended_to_coeff_host(data: *mut c_void, log_ext: u32, g_coset: *const c_void, g_coset_inv: *const c_void,
                                        extended_omega_inv: *const c_void, extended_ifft_divisor: *const c_void) -> c_int;
}

/// Offload threshold: below this the PCIe copy + launch latency lose against rayon (env MI355_MSM_MIN_LOGN / MI355_NTT_MIN_LOGN).
fn min_log(var: &str, default: u32) -> u32 { std::env::var(var).ok().and_then(|v| v.parse().ok()).unwrap_or(default) }

static INIT: Once = Once::new();
static mut AVAILABLE: bool = false;

/// One device per process; MI355_DEVICE selects it (default 0).  Never panics: on any failure the caller keeps the CPU path.
pub fn available() -> bool {
    INIT.call_once(|| unsafe {
        assert_eq!(std::mem::size_of::<Fr>(), 32);
        assert_eq!(std::mem::size_of::<G1Affine>(), 64);
        assert_eq!(std::mem::size_of::<G1>(), 96);
        let one: [u64; 4] = std::mem::transmute(Fr::one());
        assert_eq!(one, [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]);
        let dev = std::env::var("MI355_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        AVAILABLE = mi355_init(dev) == MI355_OK;
        if !AVAILABLE { log::warn!("mi355zk unavailable: {:?}; using the CPU path", std::ffi::CStr::from_ptr(mi355_last_error())); }
    });
    unsafe { AVAILABLE }
}

lazy_static::lazy_static! {
    /// (pointer, len) of a `Vec<G1Affine>` basis inside a long-lived ParamsKZG -> registered handle.  The reference keeps
    /// params in a process-wide map borrowed by every prover [REF integration/src/prove.rs:11-17], so pointer identity is stable.
    static ref SRS: Mutex<HashMap<(usize, usize), u64>> = Mutex::new(HashMap::new());
}

fn srs_handle(bases: &[G1Affine]) -> Option<(u64, u64)> {
    // a slice `&params.g[..n]` of a registered basis resolves to (handle, offset)
    let mut map = SRS.lock().unwrap();
    let (p, l) = (bases.as_ptr() as usize, bases.len());
    for (&(bp, bl), &h) in map.iter() {
        if p >= bp && p + l * 64 <= bp + bl * 64 { return Some((h, ((p - bp) / 64) as u64)); }
    }
    let mut h = 0u64;
    let rc = unsafe { mi355_srs_register_host(bases.as_ptr() as *const c_void, l as u64, &mut h) };
    if rc != MI355_OK { return None; }
    // registration-time window tables (W x the basis in HBM; MI355_SRS_PRECOMPUTE=0 disables): all windows share one bucket set
    if std::env::var("MI355_SRS_PRECOMPUTE").map(|v| v != "0").unwrap_or(true) { unsafe { let _ = mi355_srs_precompute(h, 0, 0); } }
    map.insert((p, l), h);
    Some((h, 0))
}

/// Called by ParamsKZG::{setup, read_custom, downsize} right after `g` / `g_lagrange` are final (register the FULL vectors).
pub fn register_basis(bases: &[G1Affine]) { if available() { let _ = srs_handle(bases); } }

/// Replacement body of `best_multiexp` for C = G1Affine.  Returns None -> caller runs the original CPU code.
pub fn multiexp_g1(coeffs: &[Fr], bases: &[G1Affine]) -> Option<G1> {
    assert_eq!(coeffs.len(), bases.len());            // same panic as the reference
    if !available() || (coeffs.len() as u64) < (1u64 << min_log("MI355_MSM_MIN_LOGN", 14)) { return None; }
    let mut out = std::mem::MaybeUninit::<G1>::uninit();
    let rc = unsafe {
        match srs_handle(bases) {
            Some((h, off)) => mi355_msm_g1_host(h, off, coeffs.as_ptr() as *const c_void, coeffs.len() as u64, out.as_mut_ptr() as *mut c_void),
            None => mi355_msm_g1_adhoc_host(bases.as_ptr() as *const c_void, coeffs.as_ptr() as *const c_void, coeffs.len() as u64, out.as_mut_ptr() as *mut c_void),
        }
    };
    if rc == MI355_OK { Some(unsafe { out.assume_init() }) } else { None }
}

/// One call for the `polys.iter().map(|p| params.commit_lagrange(p, blind))` loops of create_proof (advice / instance / lookup /
/// permutation columns of one phase): equal-length polynomials over one registered basis.  The blinding term is added by the caller as
/// in the reference.  None -> caller runs the original per-polynomial loop.
pub fn multiexp_g1_many(polys: &[&[Fr]], bases: &[G1Affine]) -> Option<Vec<G1>> {
    if polys.is_empty() { return Some(vec![]); }
    let n = polys[0].len();
    assert!(polys.iter().all(|p| p.len() == n) && n <= bases.len());
    if !available() || (n as u64) < (1u64 << min_log("MI355_MSM_MIN_LOGN", 14)) { return None; }
    let (h, off) = srs_handle(&bases[..n])?;
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr() as *const c_void).collect();
    let mut out: Vec<G1> = Vec::with_capacity(polys.len());
    let rc = unsafe { mi355_msm_g1_batch_host(h, off, ptrs.as_ptr(), polys.len() as u32, n as u64, out.as_mut_ptr() as *mut c_void) };
    if rc != MI355_OK { return None; }
    unsafe { out.set_len(polys.len()); }
    Some(out)
}

/// `g_to_lagrange(g, k)` for `ParamsKZG::downsize` / `setup`: a size-2^k inverse FFT over curve points, minutes on the CPU.
/// `g` must be (a prefix of) a registered basis; returns the new `g_lagrange` (and registers it).  None -> run the CPU code.
pub fn g_to_lagrange(g: &[G1Affine], k: u32, omega_inv: &Fr, n_inv: &Fr) -> Option<Vec<G1Affine>> {
    if !available() { return None; }
    let n = 1usize << k;
    let (h, off) = srs_handle(&g[..n])?;
    if off != 0 { return None; }
    let mut hl = 0u64;
    let rc = unsafe { mi355_srs_downsize(h, k, omega_inv as *const Fr as *const c_void, n_inv as *const Fr as *const c_void, &mut hl) };
    if rc != MI355_OK { return None; }
    let mut out: Vec<G1Affine> = Vec::with_capacity(n);
    let rc = unsafe { mi355_srs_read_host(hl, 0, n as u64, out.as_mut_ptr() as *mut c_void) };
    unsafe { let _ = mi355_srs_release(hl); }          // the caller's Vec is registered by address on first use, like every basis
    if rc != MI355_OK { return None; }
    unsafe { out.set_len(n); }
    Some(out)
}

/// Replacement body of `best_fft` for G = Scalar = Fr (the G = curve-point instantiation keeps the CPU code).
pub fn fft_fr<S: 'static, G: 'static>(a: &mut [G], omega: &S, log_n: u32) -> bool {
    if TypeId::of::<S>() != TypeId::of::<Fr>() || TypeId::of::<G>() != TypeId::of::<Fr>() { return false; }
    if !available() || log_n < min_log("MI355_NTT_MIN_LOGN", 16) { return false; }
    unsafe { mi355_ntt_fr_host(a.as_mut_ptr() as *mut c_void, log_n, omega as *const S as *const c_void) == MI355_OK }
}
