// mi355zk_create_proof.hpp -- the GPU side of halo2_proofs::plonk::create_proof as ONE callable flow over resident polynomials
// (mi355_buf_* / DevicePoly), on top of the C-ABI (include/mi355zk.h).  Included by mi355zk_halo2.hpp.
//
// What the reference reaches: gen_halo2_chunk_proof / gen_batch_proof / gen_bundle_proof [REF integration/src/prove.rs:37,67,95-97] run
// create_proof once per layer: layer 0 (k = 20 inner circuit [REF integration/src/mock.rs:22],
// [REF integration/src/capacity_checker.rs:90-92]), layers 1 / 2 (chunk compression [REF integration/configs/layer1.config:3-10],
// [REF integration/configs/layer2.config:3-10]), layers 3 / 4 (batch [REF integration/configs/layer3.config:3-8],
// [REF integration/configs/layer4.config:3-10]), layers 5 / 6 (bundle [REF integration/configs/layer5.config:3-6],
// [REF integration/configs/layer6.config:3-10]).  The step order is SURVEY.md 3.2 [EXT-recalled halo2_proofs plonk/prover.rs]:
//
//   1  instance column -> coefficients                              6  lagrange_to_coeff of every witness polynomial
//   2  advice columns: upload, commit_lagrange                      7  quotient: coset parts of every polynomial, evaluate_h, extended_to_coeff
//   3  lookup multiplicities m (mv-lookup): upload, commit          8  commit the Q quotient pieces
//   4  permutation grand products z, lookup running sums phi:      9  evaluations at x * omega^rot of every queried (polynomial, rotation)
//      built ON the device (batch inversion, scans), commit        10 SHPLONK-shaped multi-open: linear combination, 2 x kate_division, 2 commits
//
// The Rust twin of this file is rust_shim/create_proof_resident.rs (same steps, same calls, against rust_shim/mi355zk.rs's DevicePoly).
//
// WHAT IS REAL AND WHAT IS A STAND-IN.  Rust, the circuits and the traces are absent from this image, so the circuit is a
// synthetic one with the layer's COUNTS (columns, lookups, permutation chunks, degree: `layer_shape`) and the operand SHAPES of halo2's
// evaluate_h: custom gates that multiply rotated advice and fixed columns, the permutation argument
//   l_active(X) (z(omega X) prod_j (c_j + beta sigma_j + gamma) - z(X) prod_j (c_j + beta delta^j X + gamma)),  l_0(X) (1 - z(X)),
// and the log-derivative lookup argument  l_active(X) ((phi(omega X) - phi(X)) (a(X) + beta) - m(X)).
// The witness SATISFIES these constraints (synthesize_witness computes the dependent columns), so the quotient h(X) is a genuine polynomial and
// the caller can check  h(x) (x^n - 1) == sum_g y^g gate_g(x)  at a random x from the evaluations alone -- as a verifier would
// (tests/cpp/test_create_proof_replay.cpp does, with the CPU oracle).  The proving key's fixed / sigma / l_* polynomials are RESIDENT in
// extended-coset form (ProvingKeyDevice), as halo2's ProvingKey holds them, and evaluate_h reads them in every proof.
// Transcript hashing, witness synthesis of real circuits and the exact SHPLONK rotation-set structure stay on the host side / out of scope.
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <thread>

namespace mi355zk {
namespace halo2 {

// ------------------------------------------------------------------------------------------------ shapes of the seven layers
struct CircuitShape {
  int layer = 4;
  uint32_t k = 26;
  uint32_t advice = 3;         // phase-0 witness columns (basic-gate advice + lookup advice of halo2-base's FlexGate / RangeChip)
  uint32_t fixed = 2;          // fixed columns the gates read (constants + selectors); resident in the proving key
  uint32_t lookups = 1;        // lookup arguments: one multiplicity column m and one running sum phi each
  uint32_t perm_columns = 5;   // columns under the permutation argument; one sigma polynomial each (proving key)
  uint32_t chunk_len = 2;      // columns per grand product z (halo2: cs.degree() - 2)
  uint32_t degree = 5;         // quotient pieces Q = degree - 1 (all compression fixtures: quotient.num_chunk = 4)
  const char *source = "";
  uint32_t Q() const { return degree - 1; }
  uint32_t perm_z() const { return (perm_columns + chunk_len - 1) / chunk_len; }
  uint32_t witness_polys() const { return 1 + advice + lookups + perm_z() + lookups; }
  uint32_t commitments() const { return advice + lookups + perm_z() + lookups + Q() + 2; }
};
// Counts per layer.  Layers 2 and 4 are pinned by the fixtures' `num_witness` and proof word counts ([1,1,3] -> 11 G1, [3,1,4] -> 14 G1);
// 1, 3, 5, 6 follow their configs with the same rule (perm columns = advice + fixed + instance, chunk_len 2); layer 0 has no fixture and no
// config in the reference (hundreds of columns [EXT-recalled]): its numbers are a GUESS, overridable from the command line of the replay.
inline CircuitShape layer_shape(int layer) {
  CircuitShape s; s.layer = layer;
  switch (layer) {
    case 0: s.k = 20; s.advice = 800; s.fixed = 120; s.lookups = 60; s.perm_columns = 150; s.chunk_len = 6; s.degree = 9; s.source = "guess (SURVEY 3.3 row 0: O(10^3) commitments, capacity 10^6 rows)"; break;
    case 1: s.k = 24; s.advice = 17; s.fixed = 2; s.lookups = 2; s.perm_columns = 19; s.source = "layer1.config: 15 advice + 2 lookup advice + 1 fixed"; break;
    case 2: s.k = 25; s.advice = 1; s.fixed = 2; s.lookups = 1; s.perm_columns = 4; s.source = "layer2.config; full_proof_1.json num_witness [1,1,3]"; break;
    case 3: s.k = 21; s.advice = 93; s.fixed = 3; s.lookups = 8; s.perm_columns = 96; s.source = "layer3.config: 85 advice + 8 lookup advice + 2 fixed"; break;
    case 4: s.k = 26; s.advice = 3; s.fixed = 2; s.lookups = 1; s.perm_columns = 5; s.source = "layer4.config; full_proof_batch_agg_1.json num_witness [3,1,4]"; break;
    case 5: s.k = 21; s.advice = 5; s.fixed = 2; s.lookups = 1; s.perm_columns = 7; s.source = "layer5.config: 4 advice + 1 lookup advice + 1 fixed"; break;
    case 6: s.k = 26; s.advice = 2; s.fixed = 2; s.lookups = 1; s.perm_columns = 4; s.source = "layer6.config: 1 advice + 1 lookup advice + 1 fixed; proof.data 51 words"; break;
    default: throw std::invalid_argument("layer_shape: layers 0..6");
  }
  return s;
}

namespace detail {
inline Fr fr_add(const Fr &a, const Fr &b) { return from_fe(zk::Fr::add(to_fe(a), to_fe(b))); }
inline Fr fr_sub(const Fr &a, const Fr &b) { return from_fe(zk::Fr::sub(to_fe(a), to_fe(b))); }
inline Fr fr_neg(const Fr &a) { return from_fe(zk::Fr::neg(to_fe(a))); }
inline Fr fr_one() { return fr_from_u64(1); }
inline uint32_t log2_u32(uint32_t v) { uint32_t l = 0; while ((1u << l) < v) l++; return l; }
inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
}  // namespace detail

// ------------------------------------------------------------------------------------------------ the expression plan (what evaluate_h runs)
enum PolyKind : uint8_t { P_INSTANCE = 0, P_ADVICE, P_M, P_Z, P_PHI, P_FIXED, P_SIGMA, P_ID, P_LACTIVE, P_L0, P_TMP, P_KINDS };
struct PolyRef { PolyKind kind; uint32_t idx; bool operator<(const PolyRef &o) const { return kind != o.kind ? kind < o.kind : idx < o.idx; } bool operator==(const PolyRef &o) const { return kind == o.kind && idx == o.idx; } };
struct Factor { PolyRef p; int32_t rot; };
struct Term { Fr coeff; std::vector<Factor> f; };                      // coeff * prod_k f_k(omega^rot_k X); f empty: the constant
struct Launch { bool to_tmp; uint32_t tmp; std::vector<Term> terms; };   // one mi355_fr_gate_eval_dev call: TMP[tmp] = sum(terms) or quotient += sum(terms)
struct Challenges { Fr theta, beta, gamma, y, x, v, z0, z1; };          // what the transcript would squeeze; drawn by the caller (the transcript stays on the host)
struct Query { PolyRef p; int32_t rot; bool operator<(const Query &o) const { return p == o.p ? rot < o.rot : p < o.p; } };
struct ExpressionPlan {
  std::vector<Launch> quotient;                     // in order; TMP launches precede the quotient launches that read them
  std::vector<std::vector<Launch>> perm_product;    // per chunk p, on the LAGRANGE domain: TMP[0] = prod u_j, TMP[1] = prod v_j (step 4)
  std::vector<Query> queries;                       // every (witness / fixed / sigma polynomial, rotation) the expressions read: step 9 evaluates exactly these
  uint32_t gates = 0, terms = 0;
  Fr delta;                                         // the coset generator of the identity permutation (stand-in for halo2curves' DELTA)
};
constexpr uint32_t PLAN_MAX_TERMS = 16, PLAN_MAX_FACTORS = 48, PLAN_MAX_POLYS = 24;   // per launch (mi355_fr_gate_eval_dev)

// which column sits at position e of the permutation: the instance column first, then the advice columns, wrapping around (a real circuit
// also permutes its constant columns; their VALUES would come from the proving key's fixed_values, which this stand-in drops after synthesis)
inline PolyRef perm_column(const CircuitShape &s, uint32_t e) {
  e %= 1 + s.advice;
  if (e == 0) return {P_INSTANCE, 0};
  return {P_ADVICE, e - 1};
}
inline int32_t gate_rot(uint32_t i) { return (int32_t)(i % 3) - 1; }

inline ExpressionPlan build_plan(const CircuitShape &s, const Challenges &ch) {
  using namespace detail;
  ExpressionPlan P; P.delta = fr_from_u64(7);
  const Fr one = fr_one(), minus_one = fr_neg(one);
  Fr ypow = one;
  Launch cur{false, 0, {}};
  std::set<PolyRef> cur_polys; uint32_t cur_factors = 0;
  auto flush = [&]() { if (!cur.terms.empty()) { P.quotient.push_back(cur); cur = Launch{false, 0, {}}; cur_polys.clear(); cur_factors = 0; } };
  auto add_gate = [&](std::vector<Term> g) {                       // one gate's terms, scaled by y^gate; a gate never straddles two launches
    std::set<PolyRef> polys = cur_polys; uint32_t nf = cur_factors;
    for (auto &t : g) { nf += (uint32_t)t.f.size(); for (auto &f : t.f) polys.insert(f.p); }
    if (cur.terms.size() + g.size() > PLAN_MAX_TERMS || nf > PLAN_MAX_FACTORS || polys.size() > PLAN_MAX_POLYS) { flush(); polys.clear(); nf = 0; for (auto &t : g) { nf += (uint32_t)t.f.size(); for (auto &f : t.f) polys.insert(f.p); } }
    for (auto &t : g) { t.coeff = fr_mul(t.coeff, ypow); cur.terms.push_back(t); }
    cur_polys = polys; cur_factors = nf;
    ypow = fr_mul(ypow, ch.y); P.gates++; P.terms += (uint32_t)g.size();
  };
  const PolyRef inst{P_INSTANCE, 0}, a0{P_ADVICE, 0};
  // gate 0: the instance column is tied to the first advice column:  inst - f_0 a_0 a_0(omega X) - 7 a_0(omega^-1 X)
  add_gate({{one, {{inst, 0}}}, {minus_one, {{{P_FIXED, 0}, 0}, {a0, 0}, {a0, 1}}}, {fr_neg(fr_from_u64(7)), {{a0, -1}}}});
  // custom gates, one per dependent advice column i >= 2:  a_i - f a_0(omega^s X) a_1 - c_i f a_(i-1)(omega^-1 X)
  for (uint32_t i = 2; i < s.advice; i++) {
    const PolyRef f{P_FIXED, i % s.fixed}, a1{P_ADVICE, 1}, ai{P_ADVICE, i}, ap{P_ADVICE, i - 1};
    add_gate({{one, {{ai, 0}}}, {minus_one, {{f, 0}, {a0, gate_rot(i)}, {a1, 0}}}, {fr_neg(fr_from_u64(3 + i)), {{f, 0}, {ap, -1}}}});
  }
  // log-derivative lookups: l_active ((phi(omega X) - phi(X)) (a + beta) - m)
  for (uint32_t l = 0; l < s.lookups; l++) {
    const PolyRef a{P_ADVICE, l % s.advice}, m{P_M, l}, phi{P_PHI, l}, la{P_LACTIVE, 0};
    add_gate({{one, {{la, 0}, {phi, 1}, {a, 0}}}, {ch.beta, {{la, 0}, {phi, 1}}}, {minus_one, {{la, 0}, {phi, 0}, {a, 0}}}, {fr_neg(ch.beta), {{la, 0}, {phi, 0}}}, {minus_one, {{la, 0}, {m, 0}}}});
  }
  flush();
  // permutation argument, chunk by chunk: the sums u_j, v_j are intermediates (halo2's GraphEvaluator keeps them as calculation nodes)
  P.perm_product.resize(s.perm_z());
  Fr dpow = one;
  for (uint32_t p = 0; p < s.perm_z(); p++) {
    const uint32_t cl = std::min(s.chunk_len, s.perm_columns - p * s.chunk_len);
    std::vector<Factor> us, vs;
    for (uint32_t j = 0; j < cl; j++) {
      const uint32_t e = p * s.chunk_len + j;
      const PolyRef c = perm_column(s, e);
      Launch lu{true, 2 * j, {{one, {{c, 0}}}, {fr_mul(ch.beta, dpow), {{{P_ID, 0}, 0}}}, {ch.gamma, {}}}};          // u_j = c + beta delta^e X + gamma
      Launch lv{true, 2 * j + 1, {{one, {{c, 0}}}, {ch.beta, {{{P_SIGMA, e}, 0}}}, {ch.gamma, {}}}};                 // v_j = c + beta sigma_e + gamma
      P.quotient.push_back(lu); P.quotient.push_back(lv);
      P.perm_product[p].push_back(lu); P.perm_product[p].push_back(lv);
      us.push_back({{P_TMP, 2 * j}, 0}); vs.push_back({{P_TMP, 2 * j + 1}, 0});
      dpow = fr_mul(dpow, P.delta);
    }
    // step 4 (Lagrange domain): TMP[2 cl] = prod u, TMP[2 cl + 1] = prod v
    P.perm_product[p].push_back(Launch{true, 2 * cl, {{one, us}}});
    P.perm_product[p].push_back(Launch{true, 2 * cl + 1, {{one, vs}}});
    const PolyRef z{P_Z, p}, la{P_LACTIVE, 0}, l0{P_L0, 0};
    std::vector<Factor> t1{{la, 0}, {z, 1}}, t2{{la, 0}, {z, 0}};
    t1.insert(t1.end(), vs.begin(), vs.end()); t2.insert(t2.end(), us.begin(), us.end());
    add_gate({{one, t1}, {minus_one, t2}});
    add_gate({{one, {{l0, 0}}}, {minus_one, {{l0, 0}, {z, 0}}}});
    flush();
  }
  std::set<Query> q;
  for (const auto &L : P.quotient) for (const auto &t : L.terms) for (const auto &f : t.f)
    if (f.p.kind != P_TMP && f.p.kind != P_ID && f.p.kind != P_LACTIVE && f.p.kind != P_L0 && f.p.kind != P_INSTANCE) q.insert({f.p, f.rot});
  P.queries.assign(q.begin(), q.end());
  return P;
}

// ------------------------------------------------------------------------------------------------ the proving key's resident polynomials
// halo2's ProvingKey keeps, per fixed column and per permutation column, the Lagrange values, the coefficients AND the extended-coset
// evaluations, plus l_0 / l_last / l_active_row on the extended domain [EXT-recalled halo2_proofs plonk.rs ProvingKey, permutation::ProvingKey];
// evaluate_h reads the coset forms in every proof.  Here the extended domain is kept as Q coset parts of 2^k (the scroll fork's
// coeff_to_extended_part), part q on device slot q % D.  resident_cosets = false keeps only coefficients and recomputes a part's cosets at the
// start of that part (1 / Q of the memory, one more coset transform per polynomial and part): the HBM-budget fallback.
struct ProvingKeyDevice {
  CircuitShape shape; bool resident_cosets = true; int devices = 1;
  std::vector<DevicePoly> fixed_coeff, sigma_coeff, sigma_lagrange, fixed_lagrange;
  DevicePoly id_coeff, id_lagrange, l_active_coeff, l0_coeff;
  std::vector<std::vector<DevicePoly>> fixed_cosets, sigma_cosets;   // [column][part]
  std::vector<DevicePoly> id_cosets, l_active_cosets, l0_cosets;     // [part]
  uint64_t bytes = 0;
  const DevicePoly &coeff(const PolyRef &r) const {
    switch (r.kind) { case P_FIXED: return fixed_coeff[r.idx]; case P_SIGMA: return sigma_coeff[r.idx]; case P_ID: return id_coeff; case P_LACTIVE: return l_active_coeff; case P_L0: return l0_coeff; default: throw std::invalid_argument("not a proving-key polynomial"); }
  }
  const DevicePoly *coset(const PolyRef &r, uint32_t q) const {
    if (!resident_cosets) return nullptr;
    switch (r.kind) { case P_FIXED: return &fixed_cosets[r.idx][q]; case P_SIGMA: return &sigma_cosets[r.idx][q]; case P_ID: return &id_cosets[q]; case P_LACTIVE: return &l_active_cosets[q]; case P_L0: return &l0_cosets[q]; default: return nullptr; }
  }
};
inline Fr coset_factor(const EvaluationDomain &dom, uint32_t q) { return detail::fr_mul(dom.g_coset, detail::fr_pow(dom.extended_omega, q)); }

// A host column whose storage is either ordinary memory or page-locked memory of the library (mi355_host_alloc): the choice is the caller's, per column
// vector, at run time.  A first copy out of pageable memory crosses PCIe at ~34 GB/s, out of page-locked memory at ~56 GB/s; for the many-column
// layers that is the difference between steps 2-3 being bound by the link or by their commitments.
template <class T> struct ColumnAllocator {
  using value_type = T; bool pinned = false;
  using propagate_on_container_move_assignment = std::true_type;   // `column = Column(alloc)` must carry the page-locked choice with it
  using propagate_on_container_copy_assignment = std::true_type;
  using propagate_on_container_swap = std::true_type;
  ColumnAllocator() = default; explicit ColumnAllocator(bool p) : pinned(p) {}
  template <class U> ColumnAllocator(const ColumnAllocator<U> &o) : pinned(o.pinned) {}
  T *allocate(size_t n) {
    if (!pinned) return static_cast<T *>(::operator new(n * sizeof(T)));
    void *p = nullptr; check(mi355_host_alloc(n * sizeof(T), &p)); return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t) noexcept { if (pinned) (void)mi355_host_free(p); else ::operator delete(p); }
  template <class U> bool operator==(const ColumnAllocator<U> &o) const { return pinned == o.pinned; }
  template <class U> bool operator!=(const ColumnAllocator<U> &o) const { return pinned != o.pinned; }
};
using Column = std::vector<Fr, ColumnAllocator<Fr>>;

// witness-like values: 60 % zero, 20 % < 256, 10 % 64-bit, 10 % uniform; uniform = every element random
template <class Vec> inline void fill_column(Vec &v, uint64_t seed, bool uniform, int threads) {
  static const std::vector<Fr> small = [] { std::vector<Fr> t(256); for (uint64_t i = 0; i < 256; i++) t[i] = detail::fr_from_u64(i); return t; }();
  const uint64_t n = v.size();
  auto work = [&](uint64_t lo, uint64_t hi, uint64_t sd) {
    std::mt19937_64 g(sd);
    for (uint64_t i = lo; i < hi; i++) {
      const uint64_t u = g() % 10;
      if (uniform || u == 9) v[i] = Fr{{g(), g(), g(), g() & ((uint64_t(1) << 60) - 1)}};
      else if (u < 6) v[i] = Fr{{0, 0, 0, 0}};
      else if (u < 8) v[i] = small[g() & 255];
      else v[i] = detail::fr_mul(small[(g() & 254) + 1], Fr{{g(), g() & 0xffff, 0, 0}});
    }
  };
  if (n < (1u << 16) || threads <= 1) { work(0, n, seed); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(work, n * t / threads, n * (t + 1) / threads, seed * 1000003 + t);
  for (auto &t : th) t.join();
}

namespace detail {
// one Launch through mi355_fr_gate_eval_dev; resolve(PolyRef) -> device pointer of the operand in the domain the launch runs on
template <class Resolve> inline void run_launch(const Launch &L, void *dst, uint64_t n, const Fr &scale, bool accumulate, Resolve resolve) {
  std::vector<const void *> polys; std::map<PolyRef, uint32_t> slot;
  std::vector<Fr> coeffs; std::vector<uint32_t> tl, fp; std::vector<int32_t> fr;
  for (const auto &t : L.terms) {
    coeffs.push_back(fr_mul(t.coeff, scale)); tl.push_back((uint32_t)t.f.size());
    for (const auto &f : t.f) {
      auto it = slot.find(f.p);
      if (it == slot.end()) { it = slot.emplace(f.p, (uint32_t)polys.size()).first; polys.push_back(resolve(f.p)); }
      fp.push_back(it->second); fr.push_back(f.rot);
    }
  }
  check(mi355_fr_gate_eval_dev(dst, polys.empty() ? nullptr : polys.data(), (uint32_t)polys.size(), coeffs.data(), tl.data(), (uint32_t)tl.size(), fp.data(), fr.data(), n, accumulate ? 1 : 0));
}
inline DevicePoly ones_vector(uint64_t n, int slot) {   // all-ones through the fused kernel's constant term
  DevicePoly d(n, slot); const Fr one = fr_one(); const uint32_t tl = 0;
  check(mi355_fr_gate_eval_dev(d.p, nullptr, 0, one.data(), &tl, 1, nullptr, nullptr, n, 0));
  return d;
}
inline DevicePoly clone(const DevicePoly &s, int slot) { DevicePoly d(s.n, slot); check(mi355_buf_copy(d.p, s.p, s.n * 32)); return d; }
}  // namespace detail

// keygen_pk's device side: fixed / sigma columns (synthetic values), the identity polynomial X, l_active (all rows but the last) and l_0, each as
// Lagrange values -> coefficients -> Q coset parts.  Set-up time, outside any proof.
inline ProvingKeyDevice keygen_device(const CircuitShape &s, const EvaluationDomain &dom, uint64_t seed, bool resident_cosets, int devices, int threads) {
  using namespace detail;
  ProvingKeyDevice pk; pk.shape = s; pk.resident_cosets = resident_cosets; pk.devices = std::max(1, devices);
  const uint64_t n = dom.n; const uint32_t Q = s.Q();
  auto to_coeff = [&](const DevicePoly &lag) { DevicePoly c = clone(lag, 0); check(mi355_intt_fr_dev(c.p, dom.k, dom.omega_inv.data(), dom.ifft_divisor.data())); return c; };
  auto cosets_of = [&](const DevicePoly &coeff) {
    std::vector<DevicePoly> parts;
    if (!resident_cosets) return parts;
    for (uint32_t q = 0; q < Q; q++) {
      const int slot = (int)(q % (uint32_t)pk.devices);
      DevicePoly part(n, slot); const Fr f = coset_factor(dom, q);
      if (slot == 0) check(mi355_coset_ntt_fr_dev(part.p, coeff.p, dom.k, f.data(), dom.omega.data()));
      else { check(mi355_buf_copy(part.p, coeff.p, n * 32)); check(mi355_coset_ntt_fr_dev(part.p, part.p, dom.k, f.data(), dom.omega.data())); }
      parts.push_back(std::move(part));
    }
    return parts;
  };
  std::vector<Fr> host(n);
  for (uint32_t f = 0; f < s.fixed; f++) {
    fill_column(host, seed + 11 * f, false, threads);
    pk.fixed_lagrange.push_back(DevicePoly::from_host(host, 0));
    pk.fixed_coeff.push_back(to_coeff(pk.fixed_lagrange.back()));
    pk.fixed_cosets.push_back(cosets_of(pk.fixed_coeff.back()));
  }
  for (uint32_t e = 0; e < s.perm_columns; e++) {
    fill_column(host, seed + 100003 + 13 * e, true, threads);
    pk.sigma_lagrange.push_back(DevicePoly::from_host(host, 0));
    pk.sigma_coeff.push_back(to_coeff(pk.sigma_lagrange.back()));
    pk.sigma_cosets.push_back(cosets_of(pk.sigma_coeff.back()));
  }
  { std::vector<Fr>().swap(host); }
  const Fr one = fr_one(), zero = fr_zero();
  pk.id_coeff = DevicePoly(n, 0); check(mi355_buf_zero(pk.id_coeff.p, n * 32)); check(mi355_buf_upload(pk.id_coeff.at(1), one.data(), 32));   // X
  pk.id_lagrange = clone(pk.id_coeff, 0); check(mi355_ntt_fr_dev(pk.id_lagrange.p, dom.k, dom.omega.data()));                                 // omega^row
  pk.id_cosets = cosets_of(pk.id_coeff);
  { DevicePoly la = ones_vector(n, 0); check(mi355_synchronize()); check(mi355_buf_upload(la.at(n - 1), zero.data(), 32)); pk.l_active_coeff = to_coeff(la); }
  pk.l_active_cosets = cosets_of(pk.l_active_coeff);
  { DevicePoly l0(n, 0); check(mi355_buf_zero(l0.p, n * 32)); check(mi355_synchronize()); check(mi355_buf_upload(l0.at(0), one.data(), 32)); pk.l0_coeff = to_coeff(l0); }
  pk.l0_cosets = cosets_of(pk.l0_coeff);
  check(mi355_synchronize());
  const uint64_t per = n * 32, npk = s.fixed + s.perm_columns + 3;
  pk.bytes = per * (s.fixed * 2 + s.perm_columns * 2 + 4) + (resident_cosets ? per * npk * Q : 0);
  return pk;
}

// The witness a real prover synthesises on the CPU before create_proof starts: instance, advice columns, lookup multiplicities, all Lagrange
// values in host memory.  Dependent columns are computed so that every gate of build_plan holds on every row (the device is used as a
// calculator here, outside any timed region; fixed_lagrange is dropped afterwards).
struct Witness { Column instance; std::vector<Column> advice, m; };
inline Witness synthesize_witness(const CircuitShape &s, const EvaluationDomain &dom, ProvingKeyDevice &pk, uint64_t seed, int threads, bool pinned = false) {
  using namespace detail;
  Witness w; const uint64_t n = dom.n;
  const ColumnAllocator<Fr> alloc(pinned);
  w.instance = Column(alloc);
  for (uint32_t i = 0; i < s.advice; i++) w.advice.emplace_back(alloc);
  for (uint32_t l = 0; l < s.lookups; l++) w.m.emplace_back(alloc);
  std::vector<DevicePoly> adv(s.advice);
  const Fr one = fr_one();
  for (uint32_t i = 0; i < std::min<uint32_t>(2, s.advice); i++) { w.advice[i].resize(n); fill_column(w.advice[i], seed + 7 * i, i == 0, threads); adv[i] = DevicePoly(n, 0); check(mi355_buf_upload(adv[i].p, w.advice[i].data(), n * 32)); }
  auto resolve = [&](const PolyRef &r) -> const void * { return r.kind == P_ADVICE ? adv[r.idx].p : pk.fixed_lagrange[r.idx].p; };
  for (uint32_t i = 2; i < s.advice; i++) {   // a_i = f a_0(omega^s X) a_1 + c_i f a_(i-1)(omega^-1 X)
    const PolyRef f{P_FIXED, i % s.fixed};
    Launch L{true, 0, {{one, {{f, 0}, {{P_ADVICE, 0}, gate_rot(i)}, {{P_ADVICE, 1}, 0}}}, {fr_from_u64(3 + i), {{f, 0}, {{P_ADVICE, i - 1}, -1}}}}};
    adv[i] = DevicePoly(n, 0);
    run_launch(L, adv[i].p, n, one, false, resolve);
    w.advice[i].resize(n); check(mi355_buf_download(w.advice[i].data(), adv[i].p, n * 32));
    if (i >= 3) adv[i - 1].release();   // only a_0, a_1 and the previous column are read again
  }
  { // instance = f_0 a_0 a_0(omega X) + 7 a_0(omega^-1 X)
    Launch L{true, 0, {{one, {{{P_FIXED, 0}, 0}, {{P_ADVICE, 0}, 0}, {{P_ADVICE, 0}, 1}}}, {fr_from_u64(7), {{{P_ADVICE, 0}, -1}}}}};
    DevicePoly inst(n, 0); run_launch(L, inst.p, n, one, false, resolve); w.instance.resize(n); check(mi355_buf_download(w.instance.data(), inst.p, n * 32));
  }
  for (uint32_t l = 0; l < s.lookups; l++) {   // multiplicities: small counts
    w.m[l].resize(n);
    std::mt19937_64 g(seed + 5000 + l);
    static const std::vector<Fr> small = [] { std::vector<Fr> t(8); for (uint64_t i = 0; i < 8; i++) t[i] = fr_from_u64(i); return t; }();
    for (auto &x : w.m[l]) { const uint64_t u = g(); x = (u & 3) ? small[0] : small[(u >> 8) & 7]; }
  }
  pk.fixed_lagrange.clear();   // a ProvingKey keeps fixed_values too, but nothing on the proof path reads them
  check(mi355_synchronize());
  return w;
}

// ------------------------------------------------------------------------------------------------ what stays resident
// A prover process holds several layers at once (a chunk prover the degrees {20, 24, 25}, a batch prover {21, 26} [REF bin/src/trace_prover.rs:35-36]): the
// SRS of every degree, every layer's proving key, and the working set of the ONE proof that runs.  Everything resident does not fit 288 GiB; this is the
// rule of section 7c as code.  What each optional resident buys per proof: window tables of a basis ~8 % of every commitment on it (W x the basis of HBM);
// the Q coset parts of a proving key one coset transform per polynomial and part (layer 4: 0.33 s of 1.65 s measured).  So cosets are kept before tables,
// the keys with the most transforms per byte first, then tables go to the Lagrange bases (they carry most commitments), largest saving per byte first.
struct LayerResidency { CircuitShape shape; bool cosets_resident = false; bool table_lagrange = false, table_coeff = false; };
struct ResidencyPlan {
  std::vector<LayerResidency> layers; double srs_gib = 0, keys_gib = 0, tables_gib = 0, working_gib = 0, total_gib = 0, budget_gib = 0; bool fits = false;
};
inline double gib_per_poly(uint32_t k) { return (double)(uint64_t(32) << k) / (1024.0 * 1024 * 1024); }
inline double working_set_gib(const CircuitShape &s) {   // as measured in profiles/r04_replay_layers.md: polynomials + one part of each, temporaries, h twice, openings, NTT scratch, MSM workspace
  const double per = gib_per_poly(s.k), n = (double)(uint64_t(1) << s.k);
  return per * (2.0 * s.witness_polys() + 2 * s.chunk_len + 2 * s.Q() + 3) + per * s.Q() + per + n * 13 * 22 / (1024.0 * 1024 * 1024) + 0.5;
}
inline ResidencyPlan plan_residency(const std::vector<CircuitShape> &shapes, double hbm_gib, double reserve_fraction = 0.08) {
  ResidencyPlan P; P.budget_gib = hbm_gib * (1.0 - reserve_fraction);
  std::set<uint32_t> degrees;
  for (const auto &s : shapes) { P.layers.push_back({s}); degrees.insert(s.k); P.working_gib = std::max(P.working_gib, working_set_gib(s)); }
  for (uint32_t k : degrees) P.srs_gib += 2 * 2 * gib_per_poly(k);                                   // two bases of 64-byte points
  auto key_base = [](const CircuitShape &s) { return gib_per_poly(s.k) * (s.fixed * 2 + s.perm_columns * 2 + 4); };
  auto key_cosets = [](const CircuitShape &s) { return gib_per_poly(s.k) * (s.fixed + s.perm_columns + 3) * s.Q(); };
  auto key_lean_tmp = [](const CircuitShape &s) { return gib_per_poly(s.k) * (s.fixed + s.perm_columns + 3); };
  for (const auto &s : shapes) P.keys_gib += key_base(s);
  double used = P.srs_gib + P.keys_gib + P.working_gib, lean_tmp = 0;   // lean_tmp: one part's worth of temporaries for the largest key whose cosets are recomputed
  // 1. cosets: every part of a resident key saves one transform per proof; a transform costs ~ 2^k, a part holds 32 * 2^k bytes: the saving per byte is the
  //    same for every layer, so the order only matters when not all fit -- smaller keys first keeps more layers fully resident
  std::vector<size_t> order(shapes.size()); for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key_cosets(shapes[a]) < key_cosets(shapes[b]); });
  for (size_t i : order) {
    const double c = key_cosets(shapes[i]);
    if (used + c + lean_tmp <= P.budget_gib) { P.layers[i].cosets_resident = true; used += c; P.keys_gib += c; }
    else lean_tmp = std::max(lean_tmp, key_lean_tmp(shapes[i]));
  }
  used += lean_tmp;
  // 2. window tables: W x 64 bytes per point (W = 12 at k >= 24, 15 below); Lagrange bases first, the largest degree first (its commitments are the longest)
  std::vector<uint32_t> ks(degrees.rbegin(), degrees.rend());
  for (int pass = 0; pass < 2; pass++) for (uint32_t k : ks) {
    const double t = 2 * gib_per_poly(k) * (k >= 24 ? 12 : 15);
    if (used + t > P.budget_gib) continue;
    used += t; P.tables_gib += t;
    for (auto &L : P.layers) if (L.shape.k == k) (pass == 0 ? L.table_lagrange : L.table_coeff) = true;
  }
  P.total_gib = used; P.fits = used <= P.budget_gib;
  return P;
}

// ------------------------------------------------------------------------------------------------ create_proof, GPU side
struct ProofOptions { int devices = 1; int threads = 8; uint32_t commit_batch = 0 /* 0: by column count */; int upload_threads = 1 /* one thread moves pageable memory at the link's rate; more only delay the first column */; int early_intt = -1 /* -1: by column count, 0 / 1: off / on */; };
struct CommitRecord { PolyRef p; int piece; G1 c; };   // piece >= 0: quotient piece; p.kind == P_KINDS with piece -1 - j: SHPLONK quotient j
struct ProofGpuSide {
  std::vector<CommitRecord> commitments;              // in transcript order
  std::vector<Fr> evals;                              // parallel to plan.queries, then the Q quotient pieces at x
  double step_ms[11] = {0}; double total_ms = 0;
  uint64_t peak_hbm_bytes = 0, hbm_total_bytes = 0;
  uint32_t intt = 0, coset_ntt = 0, gate_launches = 0;
  // what the caller may inspect afterwards (coefficient forms, resident): witness polynomials by kind, quotient pieces, the opened combination
  std::map<PolyRef, DevicePoly> coeff;
  DevicePoly h, lin, quot[2];
};

inline ProofGpuSide create_proof_gpu_side(uint64_t h_g, uint64_t h_g_lagrange, const EvaluationDomain &dom, const ProvingKeyDevice &pk, const ExpressionPlan &plan,
                                          const Witness &wit, const Challenges &ch, const ProofOptions &opt) {
  using namespace detail;
  using Clock = std::chrono::steady_clock;
  const CircuitShape &s = pk.shape;
  const uint32_t k = dom.k, Q = s.Q(); const uint64_t n = dom.n;
  if (dom.extended_k != k + log2_u32(Q) || (Q & (Q - 1))) throw std::invalid_argument("create_proof_gpu_side: the domain's extended_k must be k + log2(Q), Q a power of two");
  ProofGpuSide R;
  const int D = std::max(1, std::min<int>(opt.devices, (int)Q));
  auto ms_since = [](Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); };
  const auto t_start = Clock::now(); auto t = t_start;
  auto lap = [&](int step) { R.step_ms[step] += ms_since(t); t = Clock::now(); };
  auto commit_one = [&](uint64_t basis, const DevicePoly &p, PolyRef ref, int piece, const void *ptr = nullptr) {
    G1 out; check(mi355_msm_g1_dev(basis, 0, ptr ? ptr : p.p, n, out.data())); R.commitments.push_back({ref, piece, out});
  };
  auto commit_many = [&](uint64_t basis, const std::vector<PolyRef> &refs, std::map<PolyRef, DevicePoly> &store) {
    const uint32_t B = opt.commit_batch ? opt.commit_batch : 32;   // resident polynomials: nothing to wait for, 32 per pass
    for (size_t base = 0; base < refs.size(); base += B) {
      const uint32_t cnt = (uint32_t)std::min<size_t>(B, refs.size() - base);
      std::vector<const void *> ptrs(cnt); std::vector<G1> outs(cnt);
      for (uint32_t i = 0; i < cnt; i++) ptrs[i] = store.at(refs[base + i]).p;
      check(mi355_msm_g1_batch_dev(basis, 0, ptrs.data(), cnt, n, outs.data()));
      for (uint32_t i = 0; i < cnt; i++) R.commitments.push_back({refs[base + i], -1000, outs[i]});
    }
  };
  std::map<PolyRef, DevicePoly> &poly = R.coeff;   // Lagrange values until step 6, coefficients afterwards
  // ---- steps 1-3: the witness crosses PCIe on a second host thread (a rayon worker in the real caller); commitments as the columns arrive
  std::vector<std::pair<PolyRef, const Column *>> uploads;
  // order: what is committed first crosses PCIe first; the instance column is not committed and is read from step 4 on, so it goes last
  // (at k = 26 every column in front of the first commitment is 40 ms of idle device)
  for (uint32_t i = 0; i < s.advice; i++) uploads.push_back({{P_ADVICE, i}, &wit.advice[i]});
  for (uint32_t l = 0; l < s.lookups; l++) uploads.push_back({{P_M, l}, &wit.m[l]});
  uploads.push_back({{P_INSTANCE, 0}, &wit.instance});
  for (const auto &u : uploads) poly[u.first];   // every entry exists before the uploader starts: the map's structure does not change under the readers
  // opt.upload_threads host threads (rayon workers in the real caller) share the columns round-robin: a copy from pageable memory is staged by the
  // calling thread, and one thread alone moves ~33 GB/s of the link's ~55 (mi355_buf_upload takes no device lock, so the copies also run under the
  // commitments of earlier columns)
  std::mutex mu; std::condition_variable cv; std::vector<char> arrived(uploads.size(), 0); std::string upload_error;
  const size_t UT = (size_t)std::max(1, std::min<int>(opt.upload_threads, (int)uploads.size()));
  auto upload_worker = [&](size_t first) {
    try {
      for (size_t i = first; i < uploads.size(); i += UT) {
        DevicePoly d(uploads[i].second->size(), 0); check(mi355_buf_upload(d.p, uploads[i].second->data(), uploads[i].second->size() * 32));
        { std::lock_guard<std::mutex> lk(mu); poly.at(uploads[i].first) = std::move(d); arrived[i] = 1; }
        cv.notify_all();
      }
    } catch (const std::exception &e) { { std::lock_guard<std::mutex> lk(mu); upload_error = e.what(); std::fill(arrived.begin(), arrived.end(), 1); } cv.notify_all(); }
  };
  struct Joiner { std::vector<std::thread> th; void join() { for (auto &t : th) if (t.joinable()) t.join(); } ~Joiner() { join(); } } uploaders;
  for (size_t w = 0; w < UT; w++) uploaders.th.emplace_back(upload_worker, w);
  auto wait_for = [&](size_t i) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return arrived[i] != 0; }); if (!upload_error.empty()) throw Error(MI355_EHIP, "witness upload: " + upload_error); };
  // columns per batched commitment: as many as cross PCIe in ~20 ms (1 GiB of scalars), at most 32 -- a batch is only committed once its last column has
  // arrived, so with big columns (layer 1: 17 x 512 MiB) a 32-column batch would wait for the whole witness before the first kernel (162 ms of idle device)
  const uint32_t batch_cap = opt.commit_batch ? opt.commit_batch : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(32, (uint64_t(1) << 25) / n));
  const bool batch_cols = s.advice >= 16 && batch_cap > 1;
  // With many columns steps 2-3 are bound by PCIe (layer 0: 28.9 GB of witness at ~34 GB/s from pageable memory = 0.85 s against 0.42 s of MSM kernels),
  // so the device idles half of that time.  lagrange_to_coeff of a column needs nothing but the column: the inverse transforms of the columns already
  // committed run in those gaps, into coefficient copies (step 4 still reads the Lagrange values), and step 6 only transforms what is left.
  const bool early_intt = opt.early_intt < 0 ? s.advice >= 8 : opt.early_intt != 0;
  std::map<PolyRef, DevicePoly> coeff_early;
  auto to_coeff_early = [&](const std::vector<PolyRef> &refs) {
    if (!early_intt) return;
    std::vector<void *> ptrs;
    for (const auto &r : refs) { DevicePoly c = clone(poly.at(r), 0); ptrs.push_back(c.p); coeff_early[r] = std::move(c); }
    check(mi355_ntt_fr_batch_dev(ptrs.data(), (uint32_t)ptrs.size(), k, dom.omega_inv.data(), dom.ifft_divisor.data())); R.intt += (uint32_t)ptrs.size();
  };
  {                                                                                     // steps 2, 3
    std::vector<PolyRef> pending;
    for (size_t i = 0; i + 1 < uploads.size(); i++) {
      wait_for(i);
      if (!batch_cols) { commit_one(h_g_lagrange, poly.at(uploads[i].first), uploads[i].first, -1000); to_coeff_early({uploads[i].first}); }
      else { pending.push_back(uploads[i].first); if (pending.size() == batch_cap || i + 2 == uploads.size() || uploads[i + 1].first.kind != uploads[i].first.kind) { commit_many(h_g_lagrange, pending, poly); to_coeff_early(pending); pending.clear(); } }
    }
  }
  lap(2);
  wait_for(uploads.size() - 1);
  uploaders.join();
  DevicePoly inst_lagrange = clone(poly.at({P_INSTANCE, 0}), 0);                        // the permutation argument reads the instance VALUES in step 4
  check(mi355_intt_fr_dev(poly.at({P_INSTANCE, 0}).p, k, dom.omega_inv.data(), dom.ifft_divisor.data())); R.intt++;   // step 1 (placed here: its column arrives last)
  lap(1);
  // ---- step 4: grand products and running sums, built on the device from the Lagrange values
  {
    std::vector<PolyRef> made;
    const uint32_t ntmp = 2 * s.chunk_len + 2;
    std::vector<DevicePoly> tmp; for (uint32_t i = 0; i < ntmp; i++) tmp.emplace_back(n, 0);
    auto resolve = [&](const PolyRef &r) -> const void * {
      switch (r.kind) {
        case P_INSTANCE: return inst_lagrange.p; case P_TMP: return tmp[r.idx].p; case P_ID: return pk.id_lagrange.p; case P_SIGMA: return pk.sigma_lagrange[r.idx].p;
        default: return poly.at(r).p;
      }
    };
    for (uint32_t p = 0; p < s.perm_z(); p++) {
      const auto &Ls = plan.perm_product[p];
      const uint32_t cl = (uint32_t)(Ls.size() - 2) / 2;
      DevicePoly z(n, 0);
      for (const auto &L : Ls) { run_launch(L, tmp[L.tmp].p, n, fr_one(), false, resolve); R.gate_launches++; }   // u_j, v_j, prod u, prod v
      check(mi355_fr_batch_invert_dev(tmp[2 * cl + 1].p, n));
      check(mi355_fr_vec_op_dev(2, tmp[2 * cl].p, tmp[2 * cl].p, tmp[2 * cl + 1].p, n));
      check(mi355_fr_prefix_product_dev(z.p, tmp[2 * cl].p, n, nullptr));                                          // z[i + 1] = z[i] prod u / prod v
      poly[{P_Z, p}] = std::move(z); made.push_back({P_Z, p});
      if (s.perm_z() < 8) { commit_one(h_g_lagrange, poly.at({P_Z, p}), {P_Z, p}, -1000); made.pop_back(); }
    }
    for (uint32_t l = 0; l < s.lookups; l++) {                                         // phi: phi[i + 1] = phi[i] + m[i] / (a[i] + beta)
      const PolyRef a{P_ADVICE, l % s.advice};
      Launch L{true, 0, {{fr_one(), {{a, 0}}}, {ch.beta, {}}}};
      run_launch(L, tmp[0].p, n, fr_one(), false, resolve); R.gate_launches++;
      check(mi355_fr_batch_invert_dev(tmp[0].p, n));
      check(mi355_fr_vec_op_dev(2, tmp[0].p, tmp[0].p, poly.at({P_M, l}).p, n));
      DevicePoly phi(n, 0);
      check(mi355_fr_prefix_sum_dev(phi.p, tmp[0].p, n, nullptr));
      poly[{P_PHI, l}] = std::move(phi); made.push_back({P_PHI, l});
      if (s.perm_z() < 8) { commit_one(h_g_lagrange, poly.at({P_PHI, l}), {P_PHI, l}, -1000); made.pop_back(); }
    }
    if (!made.empty()) commit_many(h_g_lagrange, made, poly);
    inst_lagrange.release();
  }
  lap(4);
  // ---- step 6: every witness polynomial to coefficients, one batched call
  {
    std::vector<void *> ptrs;
    for (auto &kv : poly) if (kv.first.kind != P_INSTANCE && !coeff_early.count(kv.first)) ptrs.push_back(kv.second.p);
    if (!ptrs.empty()) check(mi355_ntt_fr_batch_dev(ptrs.data(), (uint32_t)ptrs.size(), k, dom.omega_inv.data(), dom.ifft_divisor.data()));
    R.intt += (uint32_t)ptrs.size();
    for (auto &kv : coeff_early) poly.at(kv.first) = std::move(kv.second);   // the Lagrange values of those columns go back to the pool
    coeff_early.clear();
  }
  lap(6);
  // ---- step 7: the quotient, coset part by coset part; part q on device q % D (pk cosets of that part live there)
  R.h = DevicePoly(Q * n, 0);
  {
    std::vector<PolyRef> wrefs; for (auto &kv : poly) wrefs.push_back(kv.first);
    const uint32_t NP = (uint32_t)wrefs.size();
    std::vector<std::map<PolyRef, DevicePoly>> part_on(D), coeff_on(D), pkpart_on(D);
    std::vector<std::vector<DevicePoly>> tmp_on(D); std::vector<DevicePoly> hq_on(D);
    std::vector<DevicePoly> hpart; for (uint32_t q = 0; q < Q; q++) hpart.emplace_back(n, 0);   // part q of h: the evaluations at zeta omega_ext^(q + Q i)
    for (int d = 0; d < D; d++) {
      for (const auto &r : wrefs) part_on[d][r] = DevicePoly(n, d);
      for (uint32_t i = 0; i < 2 * s.chunk_len; i++) tmp_on[d].emplace_back(n, d);
      if (d > 0) { hq_on[d] = DevicePoly(n, d); for (const auto &r : wrefs) { coeff_on[d][r] = DevicePoly(n, d); check(mi355_buf_copy(coeff_on[d][r].p, poly.at(r).p, n * 32)); } }
    }
    std::vector<std::string> errs(D);
    std::vector<uint32_t> launches(D, 0), cosets(D, 0);
    auto do_parts = [&](int d) {
      try {
        for (uint32_t q = (uint32_t)d; q < Q; q += (uint32_t)D) {
          const Fr factor = coset_factor(dom, q);
          std::vector<void *> dst(NP); std::vector<const void *> src(NP);
          for (uint32_t i = 0; i < NP; i++) { dst[i] = part_on[d].at(wrefs[i]).p; src[i] = d == 0 ? poly.at(wrefs[i]).p : coeff_on[d].at(wrefs[i]).p; }
          check(mi355_coset_ntt_fr_batch_dev(dst.data(), src.data(), NP, k, factor.data(), dom.omega.data())); cosets[d] += NP;
          if (!pk.resident_cosets) {   // the HBM-lean proving key: this part's fixed / sigma / l_* cosets are recomputed from the coefficients
            std::set<PolyRef> need;
            for (const auto &L : plan.quotient) for (const auto &tm : L.terms) for (const auto &f : tm.f) if (f.p.kind >= P_FIXED && f.p.kind <= P_L0) need.insert(f.p);
            for (const auto &r : need) {
              auto it = pkpart_on[d].find(r); if (it == pkpart_on[d].end()) it = pkpart_on[d].emplace(r, DevicePoly(n, d)).first;
              if (d == 0) check(mi355_coset_ntt_fr_dev(it->second.p, pk.coeff(r).p, k, factor.data(), dom.omega.data()));
              else { check(mi355_buf_copy(it->second.p, pk.coeff(r).p, n * 32)); check(mi355_coset_ntt_fr_dev(it->second.p, it->second.p, k, factor.data(), dom.omega.data())); }
              cosets[d]++;
            }
          }
          auto resolve = [&](const PolyRef &r) -> const void * {
            if (r.kind == P_TMP) return tmp_on[d][r.idx].p;
            if (r.kind >= P_FIXED && r.kind <= P_L0) { const DevicePoly *c = pk.coset(r, q); return c ? c->p : pkpart_on[d].at(r).p; }
            return part_on[d].at(r).p;
          };
          // 1 / ((zeta omega_ext^q)^n - 1): the vanishing polynomial is constant on a coset part; it rides on the coefficients
          const Fr tq_inv = fr_inv(fr_sub(fr_pow(factor, n), fr_one()));
          void *hq = d == 0 ? hpart[q].p : hq_on[d].p;
          bool first = true;
          for (const auto &L : plan.quotient) {
            if (L.to_tmp) run_launch(L, tmp_on[d][L.tmp].p, n, fr_one(), false, resolve);
            else { run_launch(L, hq, n, tq_inv, !first, resolve); first = false; }
            launches[d]++;
          }
          if (d != 0) check(mi355_buf_copy(hpart[q].p, hq, n * 32));
        }
      } catch (const std::exception &e) { errs[d] = e.what(); }
    };
    { std::vector<std::thread> th; for (int d = 1; d < D; d++) th.emplace_back(do_parts, d); do_parts(0); for (auto &x : th) x.join(); }
    for (int d = 0; d < D; d++) { if (!errs[d].empty()) throw Error(MI355_EHIP, "quotient part on device slot " + std::to_string(d) + ": " + errs[d]); R.gate_launches += launches[d]; R.coset_ntt += cosets[d]; }
    // the parts interleave into the extended domain's natural order (index q + Q i), which extended_to_coeff inverts: h(X), Q n coefficients
    { std::vector<const void *> pp(Q); for (uint32_t q = 0; q < Q; q++) pp[q] = hpart[q].p; check(mi355_fr_interleave_dev(R.h.p, pp.data(), Q, n)); }
    check(mi355_extended_to_coeff_dev(R.h.p, dom.extended_k, dom.g_coset.data(), dom.g_coset_inv.data(), dom.extended_omega_inv.data(), dom.extended_ifft_divisor.data()));
  }
  lap(7);
  for (uint32_t q = 0; q < Q; q++) commit_one(h_g, R.h, {P_KINDS, 0}, (int)q, R.h.at((uint64_t)q * n));   // step 8
  lap(8);
  auto coeff_of = [&](const PolyRef &r) -> const DevicePoly & { return (r.kind >= P_FIXED && r.kind <= P_L0) ? pk.coeff(r) : poly.at(r); };
  {                                                                                     // step 9: every queried (polynomial, rotation), one synchronisation
    std::vector<const void *> ptrs; std::vector<Fr> pts;
    for (const auto &qr : plan.queries) {
      Fr pt = ch.x;
      if (qr.rot > 0) pt = fr_mul(pt, fr_pow(dom.omega, (uint64_t)qr.rot)); else if (qr.rot < 0) pt = fr_mul(pt, fr_pow(dom.omega_inv, (uint64_t)(-(int64_t)qr.rot)));
      ptrs.push_back(coeff_of(qr.p).p); pts.push_back(pt);
    }
    for (uint32_t q = 0; q < Q; q++) { ptrs.push_back(R.h.at((uint64_t)q * n)); pts.push_back(ch.x); }
    R.evals.resize(ptrs.size());
    check(mi355_eval_polynomial_batch_dev(ptrs.data(), (uint32_t)ptrs.size(), n, pts.data(), R.evals.data()));
  }
  lap(9);
  {                                                                                     // step 10
    R.lin = DevicePoly(n, 0);
    std::vector<const void *> all;
    for (auto &kv : poly) all.push_back(kv.second.p);
    for (const auto &c : pk.fixed_coeff) all.push_back(c.p);
    for (const auto &c : pk.sigma_coeff) all.push_back(c.p);
    for (uint32_t q = 0; q < Q; q++) all.push_back(R.h.at((uint64_t)q * n));
    Fr pw = fr_one();
    for (size_t base = 0; base < all.size(); base += 16) {   // sum_i v^i p_i(X): one fused launch per 16 polynomials
      const uint32_t cnt = (uint32_t)std::min<size_t>(16, all.size() - base);
      std::vector<Fr> cs(cnt); std::vector<uint32_t> tl(cnt, 1), fp(cnt); std::vector<int32_t> fr(cnt, 0);
      for (uint32_t i = 0; i < cnt; i++) { cs[i] = pw; pw = fr_mul(pw, ch.v); fp[i] = i; }
      check(mi355_fr_gate_eval_dev(R.lin.p, all.data() + base, cnt, cs.data(), tl.data(), cnt, fp.data(), fr.data(), n, base ? 1 : 0));
      R.gate_launches++;
    }
    for (int j = 0; j < 2; j++) {
      R.quot[j] = DevicePoly(n, 0);
      check(mi355_buf_zero(R.quot[j].p, n * 32));
      check(mi355_fr_kate_division_dev(R.quot[j].p, R.lin.p, n, (j ? ch.z1 : ch.z0).data()));   // n - 1 coefficients, the top one stays zero
      commit_one(h_g, R.quot[j], {P_KINDS, 0}, -1 - j);
    }
  }
  check(mi355_synchronize());
  lap(10);
  R.total_ms = ms_since(t_start);
  { uint64_t fr = 0, tot = 0; check(mi355_mem_info(0, &fr, &tot, nullptr, nullptr, nullptr)); R.peak_hbm_bytes = tot - fr; R.hbm_total_bytes = tot; }
  return R;
}

}  // namespace halo2
}  // namespace mi355zk
