//! create_proof_resident.rs -- the resident route R1-R7 of rust_shim/CALL_SITES.md as Rust, against the `DevicePoly` / `GpuBasis` API of
//! rust_shim/mi355zk.rs.  It goes next to `halo2_proofs/src/plonk/prover.rs` in the fork of scroll-tech/halo2 @ e5ddf67
//! [REF Cargo.lock:1886-1888] and is what `create_proof` calls between its transcript operations when `mi355zk::available()`.
//!
//! NOT compiled in this repository's container (no rustc / cargo).  It is the line-by-line twin of
//! `mi355zk::halo2::create_proof_gpu_side` in include/mi355zk_create_proof.hpp, which IS compiled and run
//! (tests/cpp/test_create_proof_replay.cpp, `pytest -m gpu tests/test_cpp_mirror.py`, `bench.py` `proof_mix`): each block below names the C++
//! lines it mirrors ("hpp: step N"), and tests/test_shim_matches_header.py holds the `extern "C"` block it relies on to include/mi355zk.h.
//!
//! What stays in prover.rs: the transcript (every `write_point` / `write_scalar` / `squeeze_challenge` happens on the host between the
//! calls below, on the 96-byte / 32-byte results), witness synthesis (`parallel_syn`), the blinding rows, and the compilation of the circuit's
//! `Expression` graph into term lists (`GateSlice`: <= 16 terms, <= 48 factors, <= 24 polynomials per launch -- `GraphEvaluator` already
//! holds the calculation nodes; an intermediate node with more than one use becomes a `GateSlice` with `dst = Tmp(i)`).
#![allow(dead_code)]
use std::os::raw::c_void;
use std::sync::Arc;

use halo2curves::bn256::{Fr, G1};

use crate::mi355zk::{self, DevicePoly, GpuBasis};

/// One operand of a term: a polynomial of the proof (by index into `Resident::polys`), of the proving key, or an intermediate, read at `rot`.
#[derive(Clone, Copy)]
pub enum Operand { Witness(usize), Fixed(usize), Sigma(usize), Identity, LActive, L0, Tmp(usize) }
/// `dst (+)= sum_j coeff_j * prod_k operand_jk(omega^rot_jk X)`: one `mi355_fr_gate_eval_dev` launch (hpp: `Launch`).
pub struct GateSlice { pub to_tmp: Option<usize>, pub terms: Vec<(Fr, Vec<(Operand, i32)>)> }

/// The proving key's polynomials as halo2's `ProvingKey` keeps them, resident in HBM (hpp: `ProvingKeyDevice`): coefficients (for the openings)
/// and the Q coset parts of the extended domain (for evaluate_h).  `cosets[..][q]` is empty when the HBM budget asked for on-the-fly cosets.
pub struct ResidentPk {
    pub fixed: Vec<DevicePoly>, pub sigma: Vec<DevicePoly>, pub sigma_lagrange: Vec<DevicePoly>, pub identity_lagrange: DevicePoly,
    pub l_active: DevicePoly, pub l0: DevicePoly, pub identity: DevicePoly,
    pub fixed_cosets: Vec<Vec<DevicePoly>>, pub sigma_cosets: Vec<Vec<DevicePoly>>, pub identity_cosets: Vec<DevicePoly>,
    pub l_active_cosets: Vec<DevicePoly>, pub l0_cosets: Vec<DevicePoly>,
}
pub struct Domain { pub k: u32, pub extended_k: u32, pub omega: Fr, pub omega_inv: Fr, pub ifft_divisor: Fr, pub g_coset: Fr, pub g_coset_inv: Fr,
                    pub extended_omega: Fr, pub extended_omega_inv: Fr, pub extended_ifft_divisor: Fr }

/// R1 (hpp: steps 1-3, the uploader thread).  Called from the rayon worker that finished synthesising `column`: the DMA runs without the device
/// lock, so column i + 1 crosses PCIe while column i is being committed.
pub fn upload_column(column: &[Fr], slot: i32) -> Option<DevicePoly> { DevicePoly::from_slice(column, slot) }

/// R2 (hpp: `commit_one` / `commit_many`): `params.commit_lagrange` of one column, or of a whole phase's columns as ONE pass (the many-column
/// layers 0 and 3 commit 32 columns per call).  None -> the caller commits on the CPU as before.
pub fn commit_columns(g_lagrange: &Arc<GpuBasis>, cols: &[&DevicePoly]) -> Option<Vec<G1>> {
    if cols.len() == 1 { g_lagrange.multiexp_dev(cols[0]).map(|c| vec![c]) } else { g_lagrange.multiexp_many_dev(cols) }
}

/// R3 (hpp: step 4).  The grand product of one permutation chunk, built on the device from Lagrange values: `num` and `den` are the two
/// products prod_j (c_j + beta delta^j X + gamma) and prod_j (c_j + beta sigma_j + gamma), each produced by `gate_eval_dev` launches
/// (`slices`, writing `tmp`); then z[i + 1] = z[i] num[i] / den[i].  Returns z (Lagrange values), to be committed with R2.
pub fn permutation_product(slices: &[GateSlice], resolve: &dyn Fn(Operand) -> *const c_void, tmp: &mut [DevicePoly], n: usize, num: usize, den: usize) -> Option<DevicePoly> {
    for s in slices { run_slice(s, tmp[s.to_tmp?].as_mut_ptr(), n, None, false, resolve, tmp)?; }
    unsafe {
        if mi355zk::mi355_fr_batch_invert_dev(tmp[den].as_mut_ptr(), n as u64) != 0 { return None; }
        if mi355_fr_vec_op_dev(2, tmp[num].as_mut_ptr(), tmp[num].as_ptr(), tmp[den].as_ptr(), n as u64) != 0 { return None; }
        let mut z = DevicePoly::zeroed(n, 0)?;
        if mi355zk::mi355_fr_prefix_product_dev(z.as_mut_ptr(), tmp[num].as_ptr(), n as u64, std::ptr::null_mut()) != 0 { return None; }
        Some(z)
    }
}
/// R3, lookups (mv_lookup): phi[i + 1] = phi[i] + m[i] / (a[i] + beta); `t` holds a + beta on entry (one `gate_eval_dev` with a constant term).
pub fn lookup_running_sum(t: &mut DevicePoly, m: &DevicePoly, n: usize) -> Option<DevicePoly> {
    unsafe {
        if mi355zk::mi355_fr_batch_invert_dev(t.as_mut_ptr(), n as u64) != 0 { return None; }
        if mi355_fr_vec_op_dev(2, t.as_mut_ptr(), t.as_ptr(), m.as_ptr(), n as u64) != 0 { return None; }
        let mut phi = DevicePoly::zeroed(n, 0)?;
        if mi355zk::mi355_fr_prefix_sum_dev(phi.as_mut_ptr(), t.as_ptr(), n as u64, std::ptr::null_mut()) != 0 { return None; }
        Some(phi)
    }
}

/// R4 (hpp: step 6): every witness polynomial to coefficients in one batched call.
pub fn all_to_coeff(polys: &mut [&mut DevicePoly], d: &Domain) -> bool { mi355zk::fft_many_dev(polys, d.k, &d.omega_inv, Some(&d.ifft_divisor)) }

/// R5 (hpp: step 7): the quotient.  Per coset part q: the coset evaluations of every witness polynomial in ONE call, the expression slices
/// (intermediates into `tmp`, everything else accumulated into the part of h, the part's constant 1 / ((zeta omega_ext^q)^n - 1) multiplied
/// into the coefficients), then the parts interleaved into the extended domain's order and the 2^(k + e) inverse.  Returns h(X): Q n coefficients.
pub fn quotient(polys: &[&DevicePoly], pk: &ResidentPk, slices: &[GateSlice], d: &Domain, q_parts: usize, part_factor: &dyn Fn(usize) -> Fr,
                vanishing_inv: &dyn Fn(usize) -> Fr, n_tmp: usize) -> Option<DevicePoly> {
    let n = 1usize << d.k;
    let mut parts: Vec<DevicePoly> = (0..polys.len()).map(|_| DevicePoly::zeroed(n, 0)).collect::<Option<_>>()?;
    let mut tmp: Vec<DevicePoly> = (0..n_tmp).map(|_| DevicePoly::zeroed(n, 0)).collect::<Option<_>>()?;
    let mut hparts: Vec<DevicePoly> = (0..q_parts).map(|_| DevicePoly::zeroed(n, 0)).collect::<Option<_>>()?;
    for q in 0..q_parts {
        let factor = part_factor(q);
        let dst: Vec<*mut c_void> = parts.iter_mut().map(|p| p.as_mut_ptr()).collect();
        let src: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr()).collect();
        if unsafe { mi355zk::mi355_coset_ntt_fr_batch_dev(dst.as_ptr(), src.as_ptr(), src.len() as u32, d.k, &factor as *const Fr as *const c_void,
                                                          &d.omega as *const Fr as *const c_void) } != 0 { return None; }
        let tq_inv = vanishing_inv(q);
        let mut first = true;
        for s in slices {
            let resolve = |o: Operand| -> *const c_void { match o {
                Operand::Witness(i) => parts[i].as_ptr(), Operand::Fixed(i) => pk.fixed_cosets[i][q].as_ptr(), Operand::Sigma(i) => pk.sigma_cosets[i][q].as_ptr(),
                Operand::Identity => pk.identity_cosets[q].as_ptr(), Operand::LActive => pk.l_active_cosets[q].as_ptr(), Operand::L0 => pk.l0_cosets[q].as_ptr(),
                Operand::Tmp(i) => tmp[i].as_ptr() } };
            match s.to_tmp {
                Some(t) => { let p = tmp[t].as_mut_ptr(); run_slice(s, p, n, None, false, &resolve, &tmp)?; }
                None => { run_slice(s, hparts[q].as_mut_ptr(), n, Some(tq_inv), !first, &resolve, &tmp)?; first = false; }
            }
        }
    }
    let mut h = DevicePoly::zeroed(q_parts * n, 0)?;
    let pp: Vec<*const c_void> = hparts.iter().map(|p| p.as_ptr()).collect();
    unsafe {
        if mi355zk::mi355_fr_interleave_dev(h.as_mut_ptr(), pp.as_ptr(), q_parts as u32, n as u64) != 0 { return None; }
        if mi355zk::mi355_extended_to_coeff_dev(h.as_mut_ptr(), d.extended_k, &d.g_coset as *const Fr as *const c_void, &d.g_coset_inv as *const Fr as *const c_void,
                                                 &d.extended_omega_inv as *const Fr as *const c_void, &d.extended_ifft_divisor as *const Fr as *const c_void) != 0 { return None; }
    }
    Some(h)   // R7: parts, tmp, hparts drop here: their blocks return to the pool for the next proof
}

/// R6 (hpp: steps 8-10).  The quotient pieces are slices of `h` (`params.commit` on each), the evaluations of every queried
/// (polynomial, rotation) pair come back with ONE synchronisation, the multi-open combination is one fused launch per 16 polynomials,
/// each opening quotient one `kate_division` + one commitment.
pub fn evaluate_all(polys: &[*const c_void], points: &[Fr], n: usize) -> Option<Vec<Fr>> {
    let mut out: Vec<Fr> = Vec::with_capacity(polys.len());
    if unsafe { mi355zk::mi355_eval_polynomial_batch_dev(polys.as_ptr(), polys.len() as u32, n as u64, points.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void) } != 0 { return None; }
    unsafe { out.set_len(polys.len()); }
    Some(out)
}
pub fn open_combination(all: &[&DevicePoly], v: Fr, z: &[Fr], g: &Arc<GpuBasis>, n: usize) -> Option<(DevicePoly, Vec<G1>)> {
    let mut lin = DevicePoly::zeroed(n, 0)?;
    let mut pw = Fr::one();
    for (c, chunk) in all.chunks(16).enumerate() {
        let terms: Vec<(Fr, Vec<(u32, i32)>)> = (0..chunk.len()).map(|i| { let t = (pw, vec![(i as u32, 0)]); pw *= v; t }).collect();
        if !mi355zk::gate_eval_dev(&mut lin, chunk, &terms, c != 0) { return None; }
    }
    let mut commitments = Vec::new();
    for zj in z {
        let mut quot = DevicePoly::zeroed(n, 0)?;
        if unsafe { mi355zk::mi355_fr_kate_division_dev(quot.as_mut_ptr(), lin.as_ptr(), n as u64, zj as *const Fr as *const c_void) } != 0 { return None; }
        commitments.push(g.multiexp_dev(&quot)?);
    }
    Some((lin, commitments))
}

/// What stays resident for a process that holds several layers (hpp: `plan_residency`).  `shapes[i]` = (k, fixed + sigma + 3 proving-key
/// polynomials, witness polynomials, quotient parts Q); returns for each layer whether its proving key keeps its Q coset parts in HBM, then which degrees
/// get window tables on (Lagrange basis, coefficient basis).  Cosets first (a recomputed part costs one coset transform per polynomial, ~4.8 us per MiB at
/// every k), then tables (8 % per commitment), the Lagrange bases before the coefficient bases, 8 % of the device left unplanned.
pub fn plan_residency(shapes: &[(u32, u32, u32, u32)], hbm_gib: f64) -> (Vec<bool>, Vec<(u32, bool, bool)>) {
    let gib = |k: u32| (32u64 << k) as f64 / (1u64 << 30) as f64;
    let budget = hbm_gib * 0.92;
    let mut degrees: Vec<u32> = shapes.iter().map(|s| s.0).collect(); degrees.sort_unstable(); degrees.dedup();
    let srs: f64 = degrees.iter().map(|&k| 4.0 * gib(k)).sum();
    let keys: f64 = shapes.iter().map(|s| gib(s.0) * (2.0 * (s.1 as f64 - 3.0) + 4.0)).sum();
    let working = shapes.iter().map(|s| gib(s.0) * (2.0 * s.2 as f64 + 3.0 * s.3 as f64 + 8.0) + (1u64 << s.0) as f64 * 286.0 / (1u64 << 30) as f64 + 0.5).fold(0.0, f64::max);
    let mut used = srs + keys + working;
    let mut order: Vec<usize> = (0..shapes.len()).collect();
    order.sort_by(|&a, &b| (gib(shapes[a].0) * (shapes[a].1 * shapes[a].3) as f64).partial_cmp(&(gib(shapes[b].0) * (shapes[b].1 * shapes[b].3) as f64)).unwrap());
    let (mut resident, mut lean_tmp) = (vec![false; shapes.len()], 0.0f64);
    for i in order {
        let c = gib(shapes[i].0) * (shapes[i].1 * shapes[i].3) as f64;
        if used + c + lean_tmp <= budget { resident[i] = true; used += c; } else { lean_tmp = lean_tmp.max(gib(shapes[i].0) * shapes[i].1 as f64); }
    }
    used += lean_tmp;
    let mut tables: Vec<(u32, bool, bool)> = degrees.iter().rev().map(|&k| (k, false, false)).collect();
    for pass in 0..2 { for t in tables.iter_mut() {
        let w = if t.0 >= 24 { 12.0 } else { 15.0 };
        let bytes = 2.0 * gib(t.0) * w;
        if used + bytes <= budget { used += bytes; if pass == 0 { t.1 = true } else { t.2 = true } }
    } }
    (resident, tables)
}

/// one `GateSlice` through `mi355_fr_gate_eval_dev` (hpp: `detail::run_launch`): operands are deduplicated into the launch's polynomial list
fn run_slice(s: &GateSlice, dst: *mut c_void, n: usize, scale: Option<Fr>, accumulate: bool, resolve: &dyn Fn(Operand) -> *const c_void, _tmp: &[DevicePoly]) -> Option<()> {
    let mut ptrs: Vec<*const c_void> = Vec::new();
    let (mut coeffs, mut term_len, mut fp, mut fr) = (Vec::<Fr>::new(), Vec::<u32>::new(), Vec::<u32>::new(), Vec::<i32>::new());
    for (c, factors) in &s.terms {
        coeffs.push(match scale { Some(k) => *c * k, None => *c });
        term_len.push(factors.len() as u32);
        for (o, rot) in factors {
            let p = resolve(*o);
            let idx = match ptrs.iter().position(|x| *x == p) { Some(i) => i, None => { ptrs.push(p); ptrs.len() - 1 } };
            fp.push(idx as u32); fr.push(*rot);
        }
    }
    let rc = unsafe { mi355zk::mi355_fr_gate_eval_dev(dst, ptrs.as_ptr(), ptrs.len() as u32, coeffs.as_ptr() as *const c_void, term_len.as_ptr(), term_len.len() as u32,
                                                      fp.as_ptr(), fr.as_ptr(), n as u64, accumulate as i32) };
    if rc == 0 { Some(()) } else { None }
}

extern "C" {
    // the two element-wise entry points this file needs beyond the block in mi355zk.rs (same header, include/mi355zk.h)
    pub fn mi355_fr_vec_op_dev(op: std::os::raw::c_int, dst_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, n: u64) -> std::os::raw::c_int;
}
