//! create_proof_resident.rs -- the resident route R1-R8 of rust_shim/CALL_SITES.md as Rust, against the `DevicePoly` / `GpuBasis` API of
//! rust_shim/mi355zk.rs.  It goes next to `halo2_proofs/src/plonk/prover.rs` in the fork of scroll-tech/halo2 @ e5ddf67
//! [REF Cargo.lock:1886-1888] and is what `create_proof` calls between its transcript operations when `mi355zk::available()`.
//!
//! NOT compiled and NEVER TYPE-CHECKED: this repository's container has no rustc / cargo.  It is the twin of
//! `mi355zk::plonk::create_proof` in include/mi355zk_plonk.hpp, which IS compiled and run for all seven layers (tests/cpp/test_plonk_replay.cpp,
//! `pytest -m gpu tests/test_plonk_protocol.py`, `bench.py` `proof_mix`) and whose proof bytes equal a CPU restatement of halo2's create_proof
//! (oracle/plonk.py): each block below names the step of that function it mirrors ("hpp: step N").  tests/test_shim_matches_header.py holds the
//! `extern "C"` block it relies on to include/mi355zk.h.  Closures never borrow a vector that is mutated while they live: device pointers
//! are collected into plain `Vec<*const c_void>` first.
//!
//! What stays in prover.rs: the transcript (every `write_point` / `write_scalar` / `squeeze_challenge` happens on the host between the calls below, on
//! the 96-byte / 32-byte results: include/mi355zk_transcript.hpp holds the C++ counterparts: Poseidon, Keccak / EVM, Blake2b), witness synthesis (`parallel_syn`), the rng (blinding rows, the
//! z / phi blinding values, the random polynomial), and the compilation of the circuit's `Expression` graph into `Launch`es.  In the C++ twin the graph
//! arrives as the snark-verifier PlonkProtocol JSON the reference ships with its proofs ([REF release-v0.13.1/chunk.protocol]); in the fork it is
//! `pk.vk.cs` itself -- the same tree (gates, permutation, lookups folded with y), so the compiler of mi355zk_plonk.hpp (`Compiler`: cost model per
//! Product node, temporaries, common-prefix groups) ports node for node onto `Expression<F>`.
#![allow(dead_code)]
use std::os::raw::{c_int, c_void};
use std::sync::Arc;

use halo2curves::bn256::{Fr, G1};

use crate::mi355zk::{self, DevicePoly, GpuBasis};

/// One operand of a term (hpp: `Atom`): a polynomial of the proof or of the proving key (by its index in the protocol's numbering), one of the fixed
/// polynomials the numerator names through CommonPolynomial leaves (l_0, l_last, l_active, X: `pk.common`), or a temporary; read at rotation `rot`.
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum Atom { Poly(usize, i32), Common(usize), Tmp(usize) }
/// `dst (+)= sum_j coeff_j * prod_k atom_jk`: one `mi355_fr_gate_eval_dev` launch (hpp: `Launch`).  `dst`: Some(t) = TMP[t], None = the quotient part.
pub struct Launch { pub dst: Option<usize>, pub accumulate: bool, pub terms: Vec<(Fr, Vec<Atom>)> }

/// The proving key as halo2's `ProvingKey` keeps it, resident in HBM (hpp: `ProvingKey`): per preprocessed / common polynomial its coefficients (for the
/// openings), its Q coset parts (for evaluate_h; empty when the HBM plan asked for on-the-fly cosets) and, where step 4 reads them, its Lagrange values.
pub struct ResidentPk {
    pub num_pre: usize,
    pub pre_coeff: Vec<DevicePoly>, pub pre_lagrange: Vec<Option<DevicePoly>>, pub pre_cosets: Vec<Vec<DevicePoly>>,
    pub common_coeff: Vec<DevicePoly>, pub common_cosets: Vec<Vec<DevicePoly>>, pub identity_lagrange: DevicePoly, pub identity_common: usize,
}
pub struct Domain { pub k: u32, pub extended_k: u32, pub omega: Fr, pub omega_inv: Fr, pub ifft_divisor: Fr, pub g_coset: Fr, pub g_coset_inv: Fr,
                    pub extended_omega: Fr, pub extended_omega_inv: Fr, pub extended_ifft_divisor: Fr }
fn p(x: &Fr) -> *const c_void { x as *const Fr as *const c_void }

/// R1 (hpp: steps 2-3, the uploader threads).  Called from the rayon worker that finished synthesising `column`: the DMA runs without the device lock, so
/// column i + 1 crosses PCIe while column i is being committed.  A column whose KIND bounds its cells (selector bits, bytes, range-checked limbs, 64-bit
/// words) goes through `DevicePoly::from_packed` (4-20x less link time, profiles/r05_narrow_uploads.json); scanning a plain column for zeros costs what the
/// sparse form saves, so `from_slice_sparse` is for producers that already hold (index, value) pairs.
pub fn upload_column(column: &[Fr], slot: i32) -> Option<DevicePoly> { DevicePoly::from_slice(column, slot) }

/// R2 (hpp: `commit_one` / `commit_many`): `params.commit_lagrange` of one column, or of a whole phase's columns as ONE pass (the many-column layers commit
/// up to 32 columns per call).  None -> the caller commits on the CPU as before.  The caller writes each point to the transcript in order.
pub fn commit_columns(basis: &Arc<GpuBasis>, cols: &[&DevicePoly]) -> Option<Vec<G1>> {
    if cols.len() == 1 { basis.multiexp_dev(cols[0]).map(|c| vec![c]) } else { basis.multiexp_many_dev(cols) }
}

/// R3 (hpp: step 4, permutation).  One chunk's grand product from Lagrange values: `launches` leave prod_j (c_j + beta delta^j X + gamma) in TMP[0] and
/// prod_j (c_j + beta sigma_j + gamma) in TMP[1]; z[i + 1] = z[i] TMP[0][i] / TMP[1][i], scaled by `carry` = the previous chunk's value at the l_last row
/// (the z_c(X) - z_(c-1)(omega^last X) link of the protocol), the last `blind.len()` rows overwritten with the prover's blinding values.
/// Returns (z, its value at row `usable` = the next chunk's carry).
pub fn permutation_chunk(launches: &[Launch], operand: &dyn Fn(Atom) -> *const c_void, tmp: &mut [DevicePoly], n: usize, usable: usize, carry: Option<Fr>, blind: &[Fr]) -> Option<(DevicePoly, Fr)> {
    let tmp_ptr: Vec<*mut c_void> = tmp.iter_mut().map(|t| t.as_mut_ptr()).collect();
    for l in launches { run_launch(l, tmp_ptr[l.dst?], n, None, l.accumulate, operand)?; }
    unsafe {
        if mi355zk::mi355_fr_batch_invert_dev(tmp_ptr[1], n as u64) != 0 { return None; }
        if mi355_fr_vec_op_dev(2, tmp_ptr[0], tmp_ptr[0], tmp_ptr[1], n as u64) != 0 { return None; }
        let mut z = DevicePoly::zeroed(n, 0)?;
        if mi355zk::mi355_fr_prefix_product_dev(z.as_mut_ptr(), tmp_ptr[0], n as u64, std::ptr::null_mut()) != 0 { return None; }
        if let Some(c) = carry { if mi355_fr_vec_axpy_dev(z.as_mut_ptr(), std::ptr::null(), z.as_ptr(), p(&c), n as u64) != 0 { return None; } }
        let mut at_last = Fr::zero();
        if mi355zk::mi355_buf_download(&mut at_last as *mut Fr as *mut c_void, (z.as_ptr() as *const u8).add(32 * usable) as *const c_void, 32) != 0 { return None; }
        if !blind.is_empty() && mi355zk::mi355_buf_upload((z.as_mut_ptr() as *mut u8).add(32 * (usable + 1)) as *mut c_void, blind.as_ptr() as *const c_void, 32 * blind.len() as u64) != 0 { return None; }
        Some((z, at_last))
    }
}
/// R3, lookups (hpp: step 4, mv-lookup): `launches` leave T + beta in TMP[0], I + beta in TMP[1] (from the protocol's own table / input expressions), their
/// product in TMP[2] and (T + beta) - m (I + beta) in TMP[3]; phi[i + 1] = phi[i] + TMP[3][i] / TMP[2][i]  (= 1 / (I + beta) - m / (T + beta)).
pub fn lookup_running_sum(launches: &[Launch], operand: &dyn Fn(Atom) -> *const c_void, tmp: &mut [DevicePoly], n: usize, usable: usize, blind: &[Fr]) -> Option<DevicePoly> {
    let tmp_ptr: Vec<*mut c_void> = tmp.iter_mut().map(|t| t.as_mut_ptr()).collect();
    for l in launches { run_launch(l, tmp_ptr[l.dst?], n, None, l.accumulate, operand)?; }
    unsafe {
        if mi355zk::mi355_fr_batch_invert_dev(tmp_ptr[2], n as u64) != 0 { return None; }
        if mi355_fr_vec_op_dev(2, tmp_ptr[3], tmp_ptr[3], tmp_ptr[2], n as u64) != 0 { return None; }
        let mut phi = DevicePoly::zeroed(n, 0)?;
        if mi355zk::mi355_fr_prefix_sum_dev(phi.as_mut_ptr(), tmp_ptr[3], n as u64, std::ptr::null_mut()) != 0 { return None; }
        if mi355_synchronize() != 0 { return None; }
        if !blind.is_empty() && mi355zk::mi355_buf_upload((phi.as_mut_ptr() as *mut u8).add(32 * (usable + 1)) as *mut c_void, blind.as_ptr() as *const c_void, 32 * blind.len() as u64) != 0 { return None; }
        Some(phi)
    }
}

/// R4 (hpp: step 5): the random polynomial of the vanishing argument -- coefficients drawn by the caller's rng, uploaded like a column, ONE `params.commit`
/// on the coefficient basis, one evaluation in step 9, no transform.
pub fn commit_random(g: &Arc<GpuBasis>, random_coeffs: &[Fr]) -> Option<(DevicePoly, G1)> { let d = DevicePoly::from_slice(random_coeffs, 0)?; let c = g.multiexp_dev(&d)?; Some((d, c)) }

/// R5 (hpp: step 6): every witness polynomial to coefficients in one batched call.
pub fn all_to_coeff(polys: &mut [&mut DevicePoly], d: &Domain) -> bool { mi355zk::fft_many_dev(polys, d.k, &d.omega_inv, Some(&d.ifft_divisor)) }

/// R6 (hpp: step 7): the quotient.  Per coset part q: the coset evaluations of every witness polynomial the plan reads in ONE call, the compiled launches
/// (temporaries into `tmp`, everything else accumulated into the part of h with the part's constant 1 / ((zeta omega_ext^q)^n - 1) multiplied into the
/// coefficients), then the parts interleaved into the extended domain's order and the 2^(k + e) inverse.  `witness[i]` = (protocol index, coefficients).
/// Returns h(X): Q n coefficients.
pub fn quotient(witness: &[(usize, &DevicePoly)], pk: &ResidentPk, plan: &[Launch], d: &Domain, q_parts: usize, part_factor: &dyn Fn(usize) -> Fr,
                vanishing_inv: &dyn Fn(usize) -> Fr, n_tmp: usize) -> Option<DevicePoly> {
    let n = 1usize << d.k;
    let mut parts: Vec<DevicePoly> = (0..witness.len()).map(|_| DevicePoly::zeroed(n, 0)).collect::<Option<_>>()?;
    let mut tmp: Vec<DevicePoly> = (0..n_tmp).map(|_| DevicePoly::zeroed(n, 0)).collect::<Option<_>>()?;
    let mut hparts: Vec<DevicePoly> = (0..q_parts).map(|_| DevicePoly::zeroed(n, 0)).collect::<Option<_>>()?;
    // raw pointers first: the closure below must not borrow `parts` / `tmp` while launches write through them
    let part_ptr: Vec<*mut c_void> = parts.iter_mut().map(|x| x.as_mut_ptr()).collect();
    let tmp_ptr: Vec<*mut c_void> = tmp.iter_mut().map(|x| x.as_mut_ptr()).collect();
    let h_ptr: Vec<*mut c_void> = hparts.iter_mut().map(|x| x.as_mut_ptr()).collect();
    let src: Vec<*const c_void> = witness.iter().map(|w| w.1.as_ptr()).collect();
    for q in 0..q_parts {
        let factor = part_factor(q);
        if unsafe { mi355zk::mi355_coset_ntt_fr_batch_dev(part_ptr.as_ptr() as *const *mut c_void, src.as_ptr(), src.len() as u32, d.k, p(&factor), p(&d.omega)) } != 0 { return None; }
        let operand = |a: Atom| -> *const c_void { match a {
            Atom::Tmp(t) => tmp_ptr[t] as *const c_void,
            Atom::Common(c) => pk.common_cosets[c][q].as_ptr(),
            Atom::Poly(i, _) if i < pk.num_pre => pk.pre_cosets[i][q].as_ptr(),
            Atom::Poly(i, _) => part_ptr[witness.iter().position(|w| w.0 == i).expect("the plan reads a polynomial that was not handed in")] as *const c_void,
        } };
        let tq_inv = vanishing_inv(q);
        let mut first = true;
        for l in plan {
            match l.dst {
                Some(t) => run_launch(l, tmp_ptr[t], n, None, l.accumulate, &operand)?,
                None => { run_launch(l, h_ptr[q], n, Some(tq_inv), !first, &operand)?; first = false; }
            }
        }
    }
    let mut h = DevicePoly::zeroed(q_parts * n, 0)?;
    let pp: Vec<*const c_void> = h_ptr.iter().map(|x| *x as *const c_void).collect();
    unsafe {
        if mi355zk::mi355_fr_interleave_dev(h.as_mut_ptr(), pp.as_ptr(), q_parts as u32, n as u64) != 0 { return None; }
        if mi355zk::mi355_extended_to_coeff_dev(h.as_mut_ptr(), d.extended_k, p(&d.g_coset), p(&d.g_coset_inv), p(&d.extended_omega_inv), p(&d.extended_ifft_divisor)) != 0 { return None; }
    }
    Some(h)   // parts, tmp, hparts drop here: their blocks return to the pool for the next proof
}

/// R7 (hpp: steps 8-9).  The quotient pieces are slices of `h` (`params.commit` on each); the protocol's `evaluations` -- in ITS order -- and the Q pieces at x
/// come back with ONE synchronisation.
pub fn evaluate_all(polys: &[*const c_void], points: &[Fr], n: usize) -> Option<Vec<Fr>> {
    let mut out: Vec<Fr> = Vec::with_capacity(polys.len());
    if unsafe { mi355zk::mi355_eval_polynomial_batch_dev(polys.as_ptr(), polys.len() as u32, n as u64, points.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void) } != 0 { return None; }
    unsafe { out.set_len(polys.len()); }
    Some(out)
}

/// R8 (hpp: step 10): SHPLONK as halo2's ProverSHPLONK runs it.  `sets[i]` = one rotation set: its points x omega^rot, its polynomials (coefficient
/// pointers, first-appearance order of the queries) and, per polynomial, the interpolated remainder's coefficients (host, <= 4).  Per set: A_i = sum_j
/// y^j P_ij (fused launches of 16; `y_pows[i][j]` = y^j), N_i = A_i - R_i (only the lowest |points| coefficients change), N_i / prod (X - point) by one in-place
/// `kate_division` per point; H = sum_i v^i Q_i.  The caller commits H, squeezes u, then `linearised` builds L = sum_i v^i zd_i (A_i - r_i(u))
/// - Z_T(u) H scaled by 1 / zd_0 and returns L / (X - u) for the second commitment.  (Ascending powers: the order the reference's released proofs satisfy,
/// tests/test_plonk_protocol.py::test_reference_released_proofs_verify.)
pub struct RotationSet { pub points: Vec<Fr>, pub polys: Vec<*const c_void>, pub remainder_sum: Vec<Fr> }
pub fn shplonk_quotient(sets: &[RotationSet], y_pows: &[Vec<Fr>], v: Fr, n: usize) -> Option<(Vec<DevicePoly>, DevicePoly)> {
    let (mut combos, mut h, mut work) = (Vec::new(), DevicePoly::zeroed(n, 0)?, DevicePoly::zeroed(n, 0)?);
    let mut v_pow = Fr::one();
    for (i, s) in sets.iter().enumerate() {
        let mut a = DevicePoly::zeroed(n, 0)?;
        for (c, chunk) in s.polys.chunks(16).enumerate() {
            let terms: Vec<(Fr, Vec<(u32, i32)>)> = (0..chunk.len()).map(|j| (y_pows[i][c * 16 + j], vec![(j as u32, 0)])).collect();
            if !mi355zk::gate_eval_ptrs(&mut a, chunk, &terms, c != 0) { return None; }
        }
        let m = s.points.len();
        unsafe {
            if mi355zk::mi355_buf_copy(work.as_mut_ptr(), a.as_ptr(), 32 * n as u64) != 0 { return None; }
            let mut low = vec![Fr::zero(); m];
            if mi355zk::mi355_buf_download(low.as_mut_ptr() as *mut c_void, work.as_ptr(), 32 * m as u64) != 0 { return None; }
            for t in 0..m { low[t] -= s.remainder_sum[t]; }
            if mi355zk::mi355_buf_upload(work.as_mut_ptr(), low.as_ptr() as *const c_void, 32 * m as u64) != 0 { return None; }
            for t in 0..m {   // dst == poly + 1 element: the quotient replaces coefficients t + 1 .. in place
                let base = work.as_mut_ptr() as *mut u8;
                if mi355zk::mi355_fr_kate_division_dev(base.add(32 * (t + 1)) as *mut c_void, base.add(32 * t) as *const c_void, (n - t) as u64, p(&s.points[t])) != 0 { return None; }
            }
            if mi355_fr_vec_axpy_dev(h.as_mut_ptr(), h.as_ptr(), (work.as_ptr() as *const u8).add(32 * m) as *const c_void, p(&v_pow), (n - m) as u64) != 0 { return None; }
        }
        v_pow *= v;
        combos.push(a);
    }
    Some((combos, h))
}
pub fn shplonk_linearised(combos: &[DevicePoly], h: &DevicePoly, coeffs: &[Fr], h_coeff: Fr, constant: Fr, u: Fr, n: usize) -> Option<DevicePoly> {
    let mut l = DevicePoly::zeroed(n, 0)?;
    let mut ptrs: Vec<*const c_void> = combos.iter().map(|c| c.as_ptr()).collect(); ptrs.push(h.as_ptr());
    let mut terms: Vec<(Fr, Vec<(u32, i32)>)> = coeffs.iter().enumerate().map(|(i, c)| (*c, vec![(i as u32, 0)])).collect();
    terms.push((h_coeff, vec![(combos.len() as u32, 0)]));
    if !mi355zk::gate_eval_ptrs(&mut l, &ptrs, &terms, false) { return None; }
    unsafe {
        let mut l0 = Fr::zero();
        if mi355zk::mi355_buf_download(&mut l0 as *mut Fr as *mut c_void, l.as_ptr(), 32) != 0 { return None; }
        l0 -= constant;
        if mi355zk::mi355_buf_upload(l.as_mut_ptr(), &l0 as *const Fr as *const c_void, 32) != 0 { return None; }
        let mut w = DevicePoly::zeroed(n, 0)?;
        if mi355zk::mi355_fr_kate_division_dev(w.as_mut_ptr(), l.as_ptr(), n as u64, p(&u)) != 0 { return None; }
        Some(w)   // n - 1 coefficients, the top one stays zero
    }
}

/// What stays resident for a process that holds several layers (hpp: `plan_residency`, `pk_sizes`).  `layers[i]` = (k, proving-key
/// polynomials incl. the common ones, Lagrange forms kept, witness polynomials incl. the instance, plan temporaries, quotient parts Q); returns for each
/// layer whether its proving key keeps its Q coset parts in HBM, then which degrees get window tables (Lagrange basis, coefficient basis).  Cosets first (a
/// recomputed part costs one coset transform per polynomial), smaller keys first, then tables (~8 % per commitment), 8 % of the device left unplanned.
pub fn plan_residency(layers: &[(u32, u32, u32, u32, u32, u32)], hbm_gib: f64) -> (Vec<bool>, Vec<(u32, bool, bool)>) {
    let gib = |k: u32| (32u64 << k) as f64 / (1u64 << 30) as f64;
    let budget = hbm_gib * 0.92;
    let mut degrees: Vec<u32> = layers.iter().map(|s| s.0).collect(); degrees.sort_unstable(); degrees.dedup();
    let srs: f64 = degrees.iter().map(|&k| 4.0 * gib(k)).sum();
    let keys: f64 = layers.iter().map(|s| gib(s.0) * (s.1 + s.2) as f64).sum();
    let working = layers.iter().map(|s| gib(s.0) * (2.0 * s.3 as f64 - 1.0 + s.4 as f64 + 3.0 * s.5 as f64) + (1u64 << s.0) as f64 * 286.0 / (1u64 << 30) as f64 + 0.5).fold(0.0, f64::max);
    let mut used = srs + keys + working;
    let cosets = |s: &(u32, u32, u32, u32, u32, u32)| gib(s.0) * (s.1 * s.5) as f64;
    let mut order: Vec<usize> = (0..layers.len()).collect();
    order.sort_by(|&a, &b| cosets(&layers[a]).partial_cmp(&cosets(&layers[b])).unwrap());
    let (mut resident, mut lean_tmp) = (vec![false; layers.len()], 0.0f64);
    for i in order {
        let c = cosets(&layers[i]);
        if used + c + lean_tmp <= budget { resident[i] = true; used += c; } else { lean_tmp = lean_tmp.max(gib(layers[i].0) * layers[i].1 as f64); }
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

/// one `Launch` through `mi355_fr_gate_eval_dev` (hpp: `run_launch`): operands are deduplicated into the launch's polynomial list
fn run_launch(l: &Launch, dst: *mut c_void, n: usize, scale: Option<Fr>, accumulate: bool, operand: &dyn Fn(Atom) -> *const c_void) -> Option<()> {
    let mut ptrs: Vec<*const c_void> = Vec::new();
    let (mut coeffs, mut term_len, mut fp, mut fr) = (Vec::<Fr>::new(), Vec::<u32>::new(), Vec::<u32>::new(), Vec::<i32>::new());
    for (c, factors) in &l.terms {
        coeffs.push(match scale { Some(k) => *c * k, None => *c });
        term_len.push(factors.len() as u32);
        for a in factors {
            let ptr = operand(*a);
            let idx = match ptrs.iter().position(|x| *x == ptr) { Some(i) => i, None => { ptrs.push(ptr); ptrs.len() - 1 } };
            fp.push(idx as u32); fr.push(match a { Atom::Poly(_, r) => *r, _ => 0 });
        }
    }
    let rc = unsafe { mi355zk::mi355_fr_gate_eval_dev(dst, ptrs.as_ptr(), ptrs.len() as u32, coeffs.as_ptr() as *const c_void, term_len.as_ptr(), term_len.len() as u32,
                                                      fp.as_ptr(), fr.as_ptr(), n as u64, accumulate as i32) };
    if rc == 0 { Some(()) } else { None }
}

extern "C" {
    // the element-wise entry points this file needs beyond the block in mi355zk.rs (same header, include/mi355zk.h)
    pub fn mi355_fr_vec_op_dev(op: c_int, dst_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, n: u64) -> c_int;
    pub fn mi355_fr_vec_axpy_dev(dst_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, scalar: *const c_void, n: u64) -> c_int;
    pub fn mi355_synchronize() -> c_int;
}
