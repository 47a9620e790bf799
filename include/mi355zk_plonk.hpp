// mi355zk_plonk.hpp -- halo2_proofs::plonk::create_proof for a PlonkProtocol, over resident polynomials on the MI355X, through the C-ABI (include/mi355zk.h).
//
// What the reference reaches: gen_halo2_chunk_proof / gen_batch_proof / gen_bundle_proof [REF integration/src/prove.rs:37,67,95-97] run create_proof once per
// layer; the constraint system of a layer is the PlonkProtocol the proof carries ([REF release-v0.13.1/chunk.protocol],
// [REF integration/tests/test_data/full_proof_batch_agg_1.json]; mi355zk_plonk_protocol.hpp).  This header is the prover for ANY such protocol:
//
//   keygen          commit_lagrange of every fixed / sigma column (= the verifying key, serialised like [REF release-v0.13.1/vk_chunk.vkey]); coefficient and
//                   extended-coset forms of the proving key resident in HBM (Q coset parts of 2^k, the scroll fork's coeff_to_extended_part), plus the
//                   common polynomials the numerator names (l_0, l_last = Lagrange(-7), l_active = 1 - l_last - sum Lagrange(-6..-1), X)
//   compile         quotient.numerator -> launches of mi355_fr_gate_eval_dev (sums of products of rotated polynomials).  A cost model decides per Product node
//                   whether to distribute it or to materialise a factor as a temporary (one extra pass over HBM against re-evaluating its factors in every
//                   term); challenges are scalars by then.  Nothing in the plan comes from the recognised argument structure -- the tree is compiled as it
//                   stands, so the quotient the device computes is the fixture's numerator and not a re-derivation of it.
//   create_proof    SURVEY 3.2's steps with halo2's transcript: instance values and advice commitments in, theta out; m commitments in, beta / gamma out; grand
//                   products z (chunks linked through z_(c-1)(w^last X)), running sums phi, the RANDOM polynomial of the vanishing argument (step 5) in, y out;
//                   the quotient part by part, its Q pieces in, x out; the evaluations in the protocol's order; SHPLONK over the rotation sets the protocol's
//                   `queries` imply (per set: interpolated remainders, division by the set's vanishing polynomial; one combined quotient commitment, the
//                   linearised polynomial at u, its kate_division) -- two closing commitments, as the fixtures' two trailing G1 words.
//   the proof       bytes in the reference's layout: 32-byte compressed G1 commitments in transcript order, canonical little-endian
//                   Fr evaluations, two compressed SHPLONK points: 896 B for layer 2 ([1,1,3] witness polynomials, Q = 4, 17 evaluations), 1 312 B for layer 4.
//
// Checked by oracle/plonk.py (tests only): a verifier that walks the JSON tree itself, and a CPU restatement of this prover whose proof bytes must be IDENTICAL.
// The Rust twin of the flow is rust_shim/create_proof_resident.rs.  Host code here is the transcript, the plan compiler and a few field operations per rotation
// set; everything proportional to 2^k runs on the device.
#pragma once
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "mi355zk_plonk_circuit.hpp"
#include "mi355zk_transcript.hpp"

namespace mi355zk {
namespace plonk {

using halo2::check;
using halo2::DevicePoly;
using halo2::Error;
using halo2::EvaluationDomain;
using halo2::G1;

// ------------------------------------------------------------------------------------------------ the plan
enum AtomKind : uint8_t { A_POLY = 0, A_COMMON, A_TMP };
struct Atom { AtomKind kind; uint32_t idx; int32_t rot; bool operator<(const Atom &o) const { return kind != o.kind ? kind < o.kind : idx != o.idx ? idx < o.idx : rot < o.rot; } };
struct Term { Fr coeff; std::vector<Atom> f; };
using SoP = std::vector<Term>;
struct Launch { int dst; bool accumulate; std::vector<Term> terms; };     // dst >= 0: TMP[dst]; dst == -1: the quotient accumulator of the part
constexpr uint32_t PLAN_MAX_TERMS = 16, PLAN_MAX_FACTORS = 48, PLAN_MAX_POLYS = 24, PLAN_MAX_TERM_LEN = 16;   // per launch (mi355_fr_gate_eval_dev)

struct CommonRegistry {   // the fixed polynomials the numerator names through CommonPolynomial leaves, deduplicated by subtree
  std::vector<CommonLinear> defs; std::map<std::string, uint32_t> by_key;
  uint32_t id_of(const Expr &e, bool create) {
    std::string key; expr_key(e, key);
    auto it = by_key.find(key);
    if (it != by_key.end()) return it->second;
    if (!create) throw std::invalid_argument("plan compiler: common polynomial " + key + " is not in the proving key");
    CommonLinear d; common_linear_terms(e, fr_one(), d);
    defs.push_back(d); by_key.emplace(key, (uint32_t)defs.size() - 1);
    return (uint32_t)defs.size() - 1;
  }
};

struct Compiler {
  CommonRegistry &reg; bool create_commons; std::vector<Fr> ch;
  std::vector<Launch> out; uint32_t tmp_base = 0, tmp_next = 0, tmp_max = 0;
  uint32_t constraints = 0, terms_total = 0;
  static constexpr double PASS = 3.0;       // one more pass over HBM (write + read 32 B per row), in units of one factor of the ALU-bound fused kernel
  Compiler(CommonRegistry &r, bool create, std::vector<Fr> challenges) : reg(r), create_commons(create), ch(std::move(challenges)) {}

  static double F(const SoP &S) { double f = 0; for (const auto &t : S) f += (double)t.f.size() + 0.5; return f; }
  static bool is_unit(const SoP &S) { return S.size() == 1 && S[0].f.size() == 1 && S[0].coeff == fr_one(); }
  static size_t maxlen(const SoP &S) { size_t m = 0; for (const auto &t : S) m = std::max(m, t.f.size()); return m; }
  static SoP scaled(SoP S, const Fr &c) { for (auto &t : S) t.coeff = fr_mul(t.coeff, c); return S; }
  static void merge_scalars(SoP &S) {   // one constant term per sum
    int first = -1;
    for (size_t i = 0; i < S.size();) { if (!S[i].f.empty()) { i++; continue; } if (first < 0) { first = (int)i++; continue; } S[first].coeff = fr_add(S[first].coeff, S[i].coeff); S.erase(S.begin() + (long)i); }
    if (first >= 0 && fr_is_zero(S[first].coeff) && S.size() > 1) S.erase(S.begin() + first);
  }
  // TMP[t] = S, split to the launch limits
  void emit(int dst, const SoP &S, bool accumulate_first = false) {
    Launch cur{dst, accumulate_first, {}}; std::set<Atom> polys; uint32_t nf = 0; bool any = false;
    auto key = [](const Atom &a) { return Atom{a.kind, a.idx, 0}; };
    for (const auto &t : S) {
      if (t.f.size() > PLAN_MAX_TERM_LEN) throw std::invalid_argument("plan compiler: a term exceeds 16 factors");
      std::set<Atom> np = polys; for (const auto &a : t.f) np.insert(key(a));
      if (cur.terms.size() + 1 > PLAN_MAX_TERMS || nf + t.f.size() > PLAN_MAX_FACTORS || np.size() > PLAN_MAX_POLYS) {
        out.push_back(cur); any = true; cur = Launch{dst, true, {}}; polys.clear(); nf = 0; np.clear(); for (const auto &a : t.f) np.insert(key(a));
      }
      cur.terms.push_back(t); polys = np; nf += (uint32_t)t.f.size();
    }
    if (!cur.terms.empty() || !any) out.push_back(cur);
  }
  SoP materialize(const SoP &S) {
    if (is_unit(S)) return S;
    const uint32_t t = tmp_base + tmp_next++; tmp_max = std::max(tmp_max, t + 1);
    emit((int)t, S);
    return SoP{Term{fr_one(), {Atom{A_TMP, t, 0}}}};
  }
  SoP expand(const SoP &A, const SoP &B) {
    SoP R; R.reserve(A.size() * B.size());
    for (const auto &a : A) for (const auto &b : B) { Term t{fr_mul(a.coeff, b.coeff), a.f}; t.f.insert(t.f.end(), b.f.begin(), b.f.end()); R.push_back(std::move(t)); }
    return R;
  }
  SoP product(SoP A, SoP B) {
    for (;;) {
      const double nA = (double)A.size(), nB = (double)B.size(), inf = 1e30;
      const bool fits = maxlen(A) + maxlen(B) <= PLAN_MAX_TERM_LEN;
      const double c_expand = fits ? nB * F(A) + nA * F(B) : inf;
      const double c_matA = is_unit(A) ? inf : F(A) + PASS + F(B) + nB;
      const double c_matB = is_unit(B) ? inf : F(B) + PASS + F(A) + nA;
      if (c_expand <= c_matA && c_expand <= c_matB) return expand(A, B);
      if (c_matA <= c_matB) A = materialize(A); else B = materialize(B);
    }
  }
  Fr scalar_of(const SoP &S) const { if (S.size() != 1 || !S[0].f.empty()) throw std::invalid_argument("plan compiler: the base of DistributePowers must be a scalar"); return S[0].coeff; }
  SoP compile(const Expr &e) {
    bool common = false;
    if ((e.kind == Expr::IDENTITY || e.kind == Expr::LAGRANGE || e.kind == Expr::SUM || e.kind == Expr::NEG) && is_common_linear(e, &common) && common)
      return SoP{Term{fr_one(), {Atom{A_COMMON, reg.id_of(e, create_commons), 0}}}};
    switch (e.kind) {
      case Expr::CONSTANT: return SoP{Term{e.c, {}}};
      case Expr::CHALLENGE: return SoP{Term{ch.at((size_t)e.i), {}}};
      case Expr::POLY: return SoP{Term{fr_one(), {Atom{A_POLY, (uint32_t)e.i, e.rot}}}};
      case Expr::NEG: return scaled(compile(e.kids[0]), fr_neg(fr_one()));
      case Expr::SCALED: return scaled(compile(e.kids[0]), e.c);
      case Expr::SUM: { SoP a = compile(e.kids[0]), b = compile(e.kids[1]); a.insert(a.end(), b.begin(), b.end()); merge_scalars(a); return a; }
      case Expr::PROD: return product(compile(e.kids[0]), compile(e.kids[1]));
      case Expr::DPOW: {   // Horner in the base: ((e_0 b + e_1) b + e_2) ...
        const Fr b = scalar_of(compile(e.kids.back()));
        SoP acc = compile(e.kids[0]);
        for (size_t i = 1; i + 1 < e.kids.size(); i++) { acc = scaled(std::move(acc), b); SoP x = compile(e.kids[i]); acc.insert(acc.end(), x.begin(), x.end()); merge_scalars(acc); }
        return acc;
      }
      default: throw std::invalid_argument("plan compiler: unexpected node");
    }
  }
  // the whole numerator: constraint i of DistributePowers(constraints, y) carries y^(m - 1 - i); constraints that needed temporaries are flushed at once so that the
  // next one may reuse them
  // Constraints that needed temporaries share an accumulate launch while their temporaries fit TMP_GROUP blocks: each constraint is compiled with ids from 0 and
  // RELOCATED behind the ids of the constraints already waiting; everything waiting is flushed (its terms emitted) before ids are reused.
  // COMMON PREFIXES.  Many constraints are  p * (...)  with the same leading polynomial p -- every assigned gate of the inner circuit carries the one selector, every
  // permutation chunk and lookup carries l_active.  Distributed, p is loaded, re-sliced and multiplied into every term; instead the bracketed parts of such a group
  // (>= 16 constraints) accumulate, with their powers of y, into ONE long-lived temporary G_p, and the quotient receives the single term p * G_p at the end:
  // one factor fewer in every term of the group for one more pass over HBM per group.
  static constexpr uint32_t TMP_GROUP = 12, GROUP_BASE = 32;   // waiting constraints share up to TMP_GROUP local temporaries (ids < GROUP_BASE); the long-lived G_p live at GROUP_BASE + g
  static uint32_t prefix_min() { const char *e = std::getenv("MI355_PLAN_PREFIX_MIN"); const long v = e ? std::atol(e) : 16; return v <= 0 ? 0xffffffffu : (uint32_t)v; }   // 0: never group (A/B, profiles/r05_gate_eval.md)
  uint32_t prefix_groups = 0;
  uint32_t tmps_used() const { std::set<int> u; for (const auto &L : out) if (L.dst >= 0) u.insert(L.dst); return (uint32_t)u.size(); }
  void compile_numerator(const Expr &num) {
    const size_t m = num.kids.size() - 1; const Fr y = scalar_of(compile(num.kids.back()));
    std::vector<Fr> ypow(m, fr_one()); for (size_t i = 1; i < m; i++) ypow[i] = fr_mul(ypow[i - 1], y);
    // pass 0: which constraints share a leading polynomial
    auto prefix_of = [&](const Expr &c, Atom &a) -> bool {
      if (c.kind != Expr::PROD) return false;
      const Expr &p = c.kids[0]; bool common = false;
      if (p.kind == Expr::POLY) { a = Atom{A_POLY, (uint32_t)p.i, p.rot}; return true; }
      if (is_common_linear(p, &common) && common) { a = Atom{A_COMMON, reg.id_of(p, create_commons), 0}; return true; }
      return false;
    };
    std::map<Atom, uint32_t> count; std::map<Atom, int> group_tmp;
    for (size_t i = 0; i < m; i++) { Atom a; if (prefix_of(num.kids[i], a)) count[a]++; }
    for (const auto &kv : count) if (kv.second >= prefix_min()) { group_tmp[kv.first] = (int)(GROUP_BASE + prefix_groups++); tmp_max = std::max(tmp_max, GROUP_BASE + prefix_groups); }
    // pass 1
    std::map<int, SoP> pending; std::set<int> started; uint32_t waiting_tmps = 0;
    auto relocate = [](std::vector<Term> &ts, uint32_t off) { for (auto &t : ts) for (auto &a : t.f) if (a.kind == A_TMP && a.idx < GROUP_BASE) a.idx += off; };
    auto flush = [&]() {
      for (auto &kv : pending) { if (kv.second.empty()) continue; emit(kv.first, kv.second, kv.first < 0 || started.count(kv.first) > 0); started.insert(kv.first); kv.second.clear(); }
      waiting_tmps = 0;
    };
    for (size_t i = 0; i < m; i++) {
      Atom pa; const bool grouped = prefix_of(num.kids[i], pa) && group_tmp.count(pa);
      const size_t first_launch = out.size();
      tmp_base = 0; tmp_next = 0;
      SoP s = scaled(compile(grouped ? num.kids[i].kids[1] : num.kids[i]), ypow[m - 1 - i]);
      constraints++; terms_total += (uint32_t)s.size();
      const uint32_t used = tmp_next;
      if (used > GROUP_BASE) throw std::invalid_argument("plan compiler: one constraint needs more than 32 temporaries");
      std::vector<Launch> mine(out.begin() + (long)first_launch, out.end()); out.resize(first_launch);   // this constraint's temporaries
      if (used > 0 && waiting_tmps > 0 && waiting_tmps + used > TMP_GROUP) flush();
      if (used > 0) {
        if (waiting_tmps > 0) { for (auto &L : mine) { L.dst += (int)waiting_tmps; relocate(L.terms, waiting_tmps); } relocate(s, waiting_tmps); }
        waiting_tmps += used; tmp_max = std::max(tmp_max, waiting_tmps);
      }
      out.insert(out.end(), mine.begin(), mine.end());
      SoP &dst = pending[grouped ? group_tmp[pa] : -1];
      dst.insert(dst.end(), s.begin(), s.end());
    }
    flush();                                                                                                    // every G_p is complete ...
    for (const auto &kv : group_tmp) pending[-1].push_back(Term{fr_one(), {kv.first, Atom{A_TMP, (uint32_t)kv.second, 0}}});
    flush();                                                                                                    // ... before the quotient reads p * G_p
  }
};

// what the fused kernel was asked to do, summed over every launch of this process: the ALGORITHMIC side of its roofline (profiles/r05_gate_eval.md compares it with
// the FETCH_SIZE / WRITE_SIZE and SQ counters of the same run).  bytes = 32 B x rows x (distinct operand polynomials + dst written + dst read when accumulating)
// distinct_rows: (polynomial, rotation) pairs a launch names, times its rows -- what a launch would load if every operand were re-sliced ONCE and kept in registers across its
// terms; hot_repeat_rows: the loads beyond the first of operands that appear in >= 4 terms of one launch (selectors, l_active, a prefix polynomial): what a register cache of a
// few hot operands could save (measure the operand cache instead of arguing it away -- profiles/r06_gate_operand_reuse.md)
struct GateStats { std::atomic<uint64_t> launches{0}, bytes{0}, factor_rows{0}, term_rows{0}, distinct_rows{0}, hot_repeat_rows{0}; };
inline GateStats &gate_stats() { static GateStats g; return g; }
inline void gate_eval(void *dst, const void *const *polys, uint32_t n_polys, const Fr *coeffs, const uint32_t *term_len, uint32_t n_terms, const uint32_t *factor_poly, const int32_t *factor_rot, uint64_t n, int accumulate) {
  check(mi355_fr_gate_eval_dev(dst, polys, n_polys, coeffs, term_len, n_terms, factor_poly, factor_rot, n, accumulate));
  uint64_t nf = 0; for (uint32_t j = 0; j < n_terms; j++) nf += term_len[j];
  GateStats &g = gate_stats(); g.launches++; g.bytes += 32 * n * (uint64_t)(n_polys + 1 + (accumulate ? 1 : 0)); g.factor_rows += nf * n; g.term_rows += (uint64_t)n_terms * n;
  { std::map<std::pair<uint32_t, int32_t>, uint32_t> uses; uint32_t f = 0;
    for (uint32_t j = 0; j < n_terms; j++) { std::set<std::pair<uint32_t, int32_t>> in_term; for (uint32_t q = 0; q < term_len[j]; q++, f++) if (in_term.insert({factor_poly[f], factor_rot[f]}).second) uses[{factor_poly[f], factor_rot[f]}]++; }
    uint64_t hot = 0; for (const auto &kv : uses) if (kv.second >= 4) hot += kv.second - 1;
    g.distinct_rows += (uint64_t)uses.size() * n; g.hot_repeat_rows += hot * n; }
}
// one Launch through mi355_fr_gate_eval_dev; resolve(Atom) -> device pointer of the operand on the domain the launch runs on
template <class Resolve> inline void run_launch(const Launch &L, void *dst, uint64_t n, const Fr &scale, bool accumulate, Resolve resolve) {
  std::vector<const void *> polys; std::map<Atom, uint32_t> slot;
  std::vector<Fr> coeffs; std::vector<uint32_t> tl, fp; std::vector<int32_t> fr;
  for (const auto &t : L.terms) {
    coeffs.push_back(fr_mul(t.coeff, scale)); tl.push_back((uint32_t)t.f.size());
    for (const auto &f : t.f) {
      const Atom key{f.kind, f.idx, 0};
      auto it = slot.find(key);
      if (it == slot.end()) { it = slot.emplace(key, (uint32_t)polys.size()).first; polys.push_back(resolve(key)); }
      fp.push_back(it->second); fr.push_back(f.rot);
    }
  }
  if (L.terms.empty()) { if (!accumulate) check(mi355_buf_zero(dst, n * 32)); return; }
  gate_eval(dst, polys.empty() ? nullptr : polys.data(), (uint32_t)polys.size(), coeffs.data(), tl.data(), (uint32_t)tl.size(), fp.empty() ? nullptr : fp.data(), fr.empty() ? nullptr : fr.data(), n, accumulate ? 1 : 0);
}
inline DevicePoly clone(const DevicePoly &s, int slot) { DevicePoly d(s.n, slot); check(mi355_buf_copy(d.p, s.p, s.n * 32)); return d; }
inline Fr part_factor(const EvaluationDomain &dom, uint32_t q) { return fr_mul(dom.g_coset, fr_pow(dom.extended_omega, q)); }

// ------------------------------------------------------------------------------------------------ the proving key, resident
struct ProvingKey {
  const Protocol *P = nullptr; std::unique_ptr<EvaluationDomain> dom; bool resident_cosets = true; int devices = 1;
  std::vector<DevicePoly> pre_lagrange, pre_coeff; std::vector<std::vector<DevicePoly>> pre_cosets;          // [polynomial][part]; Lagrange values only where step 4 reads them
  CommonRegistry commons; std::vector<DevicePoly> common_lagrange, common_coeff; std::vector<std::vector<DevicePoly>> common_cosets;
  uint32_t identity_common = 0;
  std::vector<uint8_t> vk;                                                                                  // u32 BE k | u32 BE fixed columns | compressed commitments
  uint64_t bytes = 0;
  const DevicePoly &coeff(const Atom &a) const { return a.kind == A_COMMON ? common_coeff.at(a.idx) : pre_coeff.at(a.idx); }
  const DevicePoly *coset(const Atom &a, uint32_t q) const { if (!resident_cosets) return nullptr; return a.kind == A_COMMON ? &common_cosets.at(a.idx)[q] : &pre_cosets.at(a.idx)[q]; }
};
// which preprocessed polynomials step 4 reads as Lagrange values: permuted fixed columns, every sigma, whatever the lookups' table / input expressions name
inline std::set<uint32_t> lagrange_needed(const Protocol &P) {
  std::set<uint32_t> s;
  for (const auto &c : P.perm) for (const auto &col : c.columns) { s.insert(col.sigma); if (P.is_pre(col.column)) s.insert(col.column); }
  for (const auto &l : P.lookups) { std::vector<std::pair<int32_t, int32_t>> r; collect_polys(*l.table, r); collect_polys(*l.input, r); for (const auto &x : r) if (P.is_pre((uint32_t)x.first)) s.insert((uint32_t)x.first); }
  return s;
}
// HBM a layer's prover needs, from the protocol alone: a dry compile gives the plan's temporaries and the common polynomials
struct PkSizes { uint32_t polys, lagrange, commons, plan_tmps; double base_bytes, coset_bytes, lean_tmp_bytes, working_bytes; };
inline PkSizes pk_sizes(const Protocol &P) {
  const double per = (double)P.n * 32; PkSizes s;
  CommonRegistry reg; { Expr id; id.kind = Expr::IDENTITY; reg.id_of(id, true); }
  Compiler cmp(reg, true, std::vector<Fr>(4, fr_one())); cmp.compile_numerator(P.numerator);
  s.commons = (uint32_t)reg.defs.size(); s.plan_tmps = cmp.tmps_used();
  s.polys = P.num_pre + s.commons; s.lagrange = (uint32_t)lagrange_needed(P).size() + 1;
  s.base_bytes = per * (s.polys + s.lagrange); s.coset_bytes = per * s.polys * P.Q; s.lean_tmp_bytes = per * s.polys;
  uint32_t NW = 1; for (auto w : P.num_witness) NW += w;                     // instance + witness polynomials
  uint32_t max_chunk = 0; for (const auto &c : P.perm) max_chunk = std::max<uint32_t>(max_chunk, (uint32_t)c.columns.size());
  // the peak is step 7: every polynomial's coefficients + one coset part of each (the random polynomial has none), the plan's temporaries, h as Q parts and as one
  // vector; step 4 (Lagrange values + 2 + 2 chunk temporaries) and step 10 (one combination per rotation set + H, L, work, the combined quotient) stay below it when
  // the early coefficient copies of a many-column layer are counted (one more copy of every advice column).  Plus the scratch of the 2^(k + e) inverse transform,
  // the MSM workspace (~22 B per entry, up to 13 windows) and fixed overheads.
  const double step7 = per * (2.0 * NW - 1 + s.plan_tmps + 2.0 * P.Q), step4 = per * (NW + (P.num_advice() >= 8 ? P.num_advice() : 0) + 3.0 + 2.0 * max_chunk), step10 = per * (NW + P.Q + 10.0);
  s.working_bytes = std::max(step7, std::max(step4, step10)) + per * P.Q + (double)P.n * 13 * 22 + 0.5 * 1024.0 * 1024 * 1024;
  return s;
}


// ------------------------------------------------------------------------------------------------ what stays resident
// A prover process holds several layers at once (a chunk prover the degrees {20, 24, 25}, a batch prover {21, 26} [REF bin/src/trace_prover.rs:35-36]): the
// SRS of every degree, every layer's proving key, and the working set of the ONE proof that runs.  Everything resident does not fit 288 GiB; this is the
// rule of section 7c as code.  What each optional resident buys per proof: window tables of a basis ~8 % of every commitment on it (W x the basis of HBM);
// the Q coset parts of a proving key one coset transform per polynomial and part.  So cosets are kept before tables, the subset of keys that saves the most proof time and fits (round 6; round 5: smaller keys first, so that more layers stay
// fully resident), then tables go to the Lagrange bases (they carry most commitments), the largest degree first.
// one coset transform of 2^k on an MI355X, batched (0.109 ms at 2^20, 0.228 at 2^21, 0.479 at 2^22, 2.03 at 2^24, 4.3 at 2^25, 9.0 at 2^26)
inline double coset_transform_ms(uint32_t k) {
  static const double t[7] = {0.109, 0.228, 0.479, 1.0, 2.03, 4.3, 9.0};
  return k < 20 ? 0.109 / (double)(1u << (20 - k)) : k <= 26 ? t[k - 20] : 9.0 * (double)(1u << (k - 26));
}
struct LayerResidency { const Protocol *P; PkSizes sz; bool cosets_resident = false, table_lagrange = false, table_coeff = false; };
struct ResidencyPlan { std::vector<LayerResidency> layers; double srs_gib = 0, keys_gib = 0, tables_gib = 0, working_gib = 0, total_gib = 0, budget_gib = 0; bool fits = false; };
inline ResidencyPlan plan_residency(const std::vector<const Protocol *> &protos, double hbm_gib, double reserve_fraction = 0.08) {
  const double GiB = 1024.0 * 1024 * 1024;
  ResidencyPlan R; R.budget_gib = hbm_gib * (1.0 - reserve_fraction);
  std::set<uint32_t> degrees;
  for (const Protocol *p : protos) { LayerResidency L{p, pk_sizes(*p)}; R.working_gib = std::max(R.working_gib, L.sz.working_bytes / GiB); degrees.insert(p->k); R.keys_gib += L.sz.base_bytes / GiB; R.layers.push_back(L); }
  for (uint32_t k : degrees) R.srs_gib += 2.0 * (double)(uint64_t(64) << k) / GiB;                       // two bases of 64-byte points
  // Which keys keep their coset parts (round 6): the subset that SAVES THE MOST PROOF TIME and fits -- a resident key saves one coset transform per polynomial, part and proof
  // (coset_transform_ms: measured, section 5).  Round 5 took "smaller keys first"; by value a chunk prover would rather keep layer 1's 80 GiB (160 transforms of 2^24 = 0.33 s)
  // than layer 0's 68.5 GiB (2 192 transforms of 2^20 = 0.24 s) -- but that plan ran OUT OF MEMORY inside layer 0's quotient on the device (288 GiB), because a lean layer 0
  // adds its recomputed part (8.6 GiB) to the LARGEST working set of the process.  So the accounting of a multi-layer process is the measured one (chunk prover: 276 GiB real):
  //   * the buffer pool keeps the high-water mark of the largest per-layer demand = that layer's blocks + (if ITS key is lean) one part's cosets, plus ~10 % that another
  //     layer's block sizes cannot reuse;
  //   * the MSM workspace and the transform scratch are sized by the LARGEST degree / extended domain of the process, whichever layer has the largest working set;
  //   * the folded coset shift keeps one table per coset factor for degrees up to 2^24 (bigger ones are only built into spare memory: lib_ntt.hip).
  // At most 7 layers: all subsets are tried.
  const size_t NL = R.layers.size();
  std::vector<double> core(NL), msm_ws(NL), ntt_scr(NL);
  double max_msm = 0, max_scr = 0, fold_tables = 0;
  for (size_t i = 0; i < NL; i++) {
    const Protocol &P = *R.layers[i].P;
    msm_ws[i] = (double)P.n * 13 * 22 / GiB; ntt_scr[i] = (double)P.n * 32 * P.Q / GiB;
    core[i] = R.layers[i].sz.working_bytes / GiB - msm_ws[i] - ntt_scr[i];
    max_msm = std::max(max_msm, msm_ws[i]); max_scr = std::max(max_scr, ntt_scr[i]);
  }
  if (NL > 1) for (uint32_t k : degrees) if (k <= 24) { uint32_t q = 0; for (const auto &L : R.layers) if (L.P->k == k) q = std::max(q, L.P->Q); fold_tables += (double)q * 36 * (double)(uint64_t(1) << k) / GiB; }
  auto demand = [&](uint32_t mask, double &mem, double &val) {          // GiB a process needs with the keys of `mask` resident; val = ms of transforms saved per round
    mem = 0; val = 0; double pool = 0;
    for (size_t i = 0; i < NL; i++) {
      const LayerResidency &L = R.layers[i];
      const bool res = mask >> i & 1;
      if (res) { mem += L.sz.coset_bytes / GiB; val += (double)L.sz.polys * L.P->Q * coset_transform_ms(L.P->k); }
      pool = std::max(pool, core[i] + (res ? 0.0 : L.sz.lean_tmp_bytes / GiB));
    }
    return (NL > 1 ? 1.10 : 1.0) * pool + max_msm + max_scr + fold_tables;
  };
  uint32_t best_mask = 0; double best_val = -1, best_mem = 0, best_work = 0;
  for (uint32_t mask = 0; mask < (1u << NL); mask++) {
    double mem, val; const double work = demand(mask, mem, val);
    if (R.srs_gib + R.keys_gib + mem + work > R.budget_gib) continue;
    if (val > best_val + 1e-9 || (val > best_val - 1e-9 && mem + work < best_mem + best_work)) { best_val = val; best_mask = mask; best_mem = mem; best_work = work; }
  }
  if (best_val < 0) { double mem, val; best_work = demand(0, mem, val); best_mem = 0; best_mask = 0; }   // not even every key lean fits: the caller sees fits == false
  for (size_t i = 0; i < NL; i++) if (best_mask >> i & 1) R.layers[i].cosets_resident = true;
  R.working_gib = best_work; R.keys_gib += best_mem;
  double used = R.srs_gib + R.keys_gib + R.working_gib;
  std::vector<uint32_t> ks(degrees.rbegin(), degrees.rend());
  for (int pass = 0; pass < 2; pass++) for (uint32_t k : ks) {
    const double t = (double)(uint64_t(64) << k) * (k >= 24 ? 12 : 15) / GiB;                               // W x 64 bytes per point
    if (used + t > R.budget_gib) continue;
    used += t; R.tables_gib += t;
    for (auto &L : R.layers) if (L.P->k == k) (pass == 0 ? L.table_lagrange : L.table_coeff) = true;
  }
  R.total_gib = used; R.fits = used <= R.budget_gib;
  return R;
}

inline std::unique_ptr<ProvingKey> keygen(const Protocol &P, const Circuit &C, uint64_t h_g_lagrange, bool resident_cosets, int devices) {
  auto pk = std::make_unique<ProvingKey>(); pk->P = &P; pk->resident_cosets = resident_cosets; pk->devices = std::max(1, devices);
  pk->dom = std::make_unique<EvaluationDomain>(P.Q + 1, P.k);
  const EvaluationDomain &dom = *pk->dom;
  if (dom.extended_k != P.extended_k) throw std::invalid_argument("keygen: extended domain mismatch");
  const uint64_t n = P.n; const uint32_t Q = P.Q;
  auto to_coeff = [&](const DevicePoly &lag) { DevicePoly c = clone(lag, 0); check(mi355_intt_fr_dev(c.p, dom.k, dom.omega_inv.data(), dom.ifft_divisor.data())); return c; };
  auto cosets_of = [&](const DevicePoly &coeff) {
    std::vector<DevicePoly> parts;
    if (!resident_cosets) return parts;
    for (uint32_t q = 0; q < Q; q++) {
      const int slot = (int)(q % (uint32_t)pk->devices);
      DevicePoly part(n, slot); const Fr f = part_factor(dom, q);
      if (slot == 0) check(mi355_coset_ntt_fr_dev(part.p, coeff.p, dom.k, f.data(), dom.omega.data()));
      else { check(mi355_buf_copy(part.p, coeff.p, n * 32)); check(mi355_coset_ntt_fr_dev(part.p, part.p, dom.k, f.data(), dom.omega.data())); }
      parts.push_back(std::move(part));
    }
    return parts;
  };
  const std::set<uint32_t> keep = lagrange_needed(P);
  pk->pre_lagrange.resize(P.num_pre); pk->pre_coeff.resize(P.num_pre); pk->pre_cosets.resize(P.num_pre);
  pk->vk.resize(8 + 32 * (size_t)P.num_pre);
  { const uint32_t nf = P.num_pre - P.num_sigma(); for (int i = 0; i < 4; i++) { pk->vk[i] = (uint8_t)(P.k >> (24 - 8 * i)); pk->vk[4 + i] = (uint8_t)(nf >> (24 - 8 * i)); } }
  std::vector<Fr> sig;
  for (uint32_t p = 0; p < P.num_pre; p++) {
    DevicePoly lag(n, 0);
    int j = -1; for (size_t t = 0; t < C.pcols.size(); t++) if (C.pcols[t].sigma == p) j = (int)t;
    if (j >= 0) { C.sigma_column((uint32_t)j, sig); check(mi355_buf_upload(lag.p, sig.data(), n * 32)); }
    else check(mi355_buf_upload(lag.p, C.pre[p].data(), n * 32));
    G1 c; check(mi355_msm_g1_dev(h_g_lagrange, 0, lag.p, n, c.data()));
    halo2::G1Affine a; std::memcpy(a.data(), c.data(), 64); const halo2::G1Bytes b = halo2::g1_to_bytes(a);
    std::memcpy(pk->vk.data() + 8 + 32 * (size_t)p, b.data(), 32);
    pk->pre_coeff[p] = to_coeff(lag);
    pk->pre_cosets[p] = cosets_of(pk->pre_coeff[p]);
    if (keep.count(p)) pk->pre_lagrange[p] = std::move(lag);
  }
  // the common polynomials: enumerate them with a structure-only compile (dummy challenges), then build each from its definition on the Lagrange domain
  { Expr id; id.kind = Expr::IDENTITY; pk->identity_common = pk->commons.id_of(id, true); }
  { Compiler cmp(pk->commons, true, std::vector<Fr>(4, fr_one())); cmp.compile_numerator(P.numerator); }
  const Fr one = fr_one();
  for (const auto &d : pk->commons.defs) {
    DevicePoly lag(n, 0); check(mi355_buf_zero(lag.p, n * 32));
    if (!fr_is_zero(d.x_coeff)) { check(mi355_synchronize()); check(mi355_buf_upload(lag.at(1), d.x_coeff.data(), 32)); check(mi355_ntt_fr_dev(lag.p, dom.k, dom.omega.data())); }   // x_coeff X -> x_coeff omega^row
    if (!fr_is_zero(d.constant)) {
      const void *pp[1] = {lag.p}; const Fr cs[2] = {one, d.constant}; const uint32_t tl[2] = {1, 0}, fp[1] = {0}; const int32_t fr_[1] = {0};
      gate_eval(lag.p, pp, 1, cs, tl, 2, fp, fr_, n, 0);
    }
    for (const auto &sp : d.lagrange) {
      const uint64_t row = (uint64_t)(((int64_t)sp.first % (int64_t)n + (int64_t)n) % (int64_t)n);
      Fr cur; check(mi355_buf_download(cur.data(), lag.at(row), 32)); cur = fr_add(cur, sp.second); check(mi355_buf_upload(lag.at(row), cur.data(), 32));
    }
    pk->common_coeff.push_back(to_coeff(lag));
    pk->common_cosets.push_back(cosets_of(pk->common_coeff.back()));
    pk->common_lagrange.push_back(std::move(lag));
  }
  for (size_t i = 0; i < pk->common_lagrange.size(); i++) if (i != pk->identity_common) pk->common_lagrange[i].release();   // step 4 reads X only
  check(mi355_synchronize());
  const PkSizes sz = pk_sizes(P);
  pk->bytes = (uint64_t)(sz.base_bytes + (resident_cosets ? sz.coset_bytes : 0));
  return pk;
}

// ------------------------------------------------------------------------------------------------ create_proof
struct ProofOptions { int devices = 1; int threads = 8; uint32_t commit_batch = 0 /* 0: by column count */; int upload_threads = 1; int early_intt = -1 /* -1: by column count */;
                      bool sparse_uploads = false /* columns that are at least half zeros cross PCIe as (index, value) pairs */;
                      bool packed_multiplicities = false /* the lookup multiplicities cross PCIe as the 4-byte counts they are (mi355_buf_upload_packed); their blinding rows follow as 32-byte words */;
                      TranscriptKind transcript = TranscriptKind::ByLayer /* the reference's choice for the protocol's layer (reference_transcript below): Poseidon for 0-5, Evm for 6; or name one */; };
// the transcript the reference proves a layer with: Poseidon for every proof the next layer verifies in-circuit (layers 0-5, [REF integration/src/prove.rs:30-43,67,95-97] -> snark-verifier-sdk
// gen_snark_shplonk), Keccak in the EVM layout for layer 6 (gen_evm_proof_shplonk: what the released verifier contract reads).  Files without a layer number are the reference's fixtures (layers 2, 4).
inline TranscriptKind reference_transcript(const Protocol &P) { return P.layer == 6 ? TranscriptKind::Evm : TranscriptKind::Poseidon; }
// The first scalar of the transcript.  A verifier built from a PlonkProtocol (snark-verifier's PlonkVerifier, the next layer's in-circuit verifier, the EVM contract) absorbs the
// protocol's `transcript_initial_state`, never the key's bytes -- so when the protocol file carries one (the reference's own files do) the prover MUST start from it, or its
// proof cannot pass that verifier whatever else is right.  Protocols generated here carry none (halo2 derives it by hashing a Rust Debug string of the
// pinned key, which does not exist outside Rust): for those keys the scalar is a convention of this repository, the Blake2b hash of the .vkey bytes (vk_transcript_repr).
inline Fr vk_transcript_scalar(const Protocol &P, const std::vector<uint8_t> &vk_bytes) { return P.has_initial_state ? P.initial_state : vk_transcript_repr(vk_bytes); }
inline const char *transcript_name(TranscriptKind k) { return k == TranscriptKind::Poseidon ? "poseidon" : k == TranscriptKind::Evm ? "evm" : "blake2b"; }
struct ProofResult {
  std::vector<uint8_t> proof;
  double step_ms[11] = {0}; double total_ms = 0;
  uint64_t peak_hbm_bytes = 0, hbm_total_bytes = 0;
  uint64_t sparse_columns = 0, packed_columns = 0, witness_link_bytes = 0;
  uint32_t msm = 0, intt = 0, coset_ntt = 0, gate_launches = 0, evals = 0, plan_launches = 0, plan_terms = 0, plan_tmps = 0, plan_constraints = 0, plan_prefix_groups = 0, rotation_sets = 0;
};

namespace detail {
// coefficients (low first) of the polynomial of degree < m through (points[i], values[i]); m <= 8, host arithmetic
inline std::vector<Fr> interpolate(const std::vector<Fr> &pts, const std::vector<Fr> &vals) {
  const size_t m = pts.size(); std::vector<Fr> out(m, fr_zero());
  for (size_t i = 0; i < m; i++) {
    std::vector<Fr> num{fr_one()}; Fr den = fr_one();
    for (size_t j = 0; j < m; j++) {
      if (j == i) continue;
      std::vector<Fr> nx(num.size() + 1, fr_zero());
      for (size_t t = 0; t < num.size(); t++) { nx[t + 1] = fr_add(nx[t + 1], num[t]); nx[t] = fr_sub(nx[t], fr_mul(pts[j], num[t])); }
      num.swap(nx); den = fr_mul(den, fr_sub(pts[i], pts[j]));
    }
    const Fr s = fr_mul(vals[i], fr_inv(den));
    for (size_t t = 0; t < num.size(); t++) out[t] = fr_add(out[t], fr_mul(s, num[t]));
  }
  return out;
}
inline Fr horner(const std::vector<Fr> &c, const Fr &x) { Fr acc = fr_zero(); for (size_t i = c.size(); i-- > 0;) acc = fr_add(fr_mul(acc, x), c[i]); return acc; }
}  // namespace detail

inline ProofResult create_proof(uint64_t h_g, uint64_t h_g_lagrange, const ProvingKey &pk, const Circuit &wit, const ProofOptions &opt) {
  using Clock = std::chrono::steady_clock;
  const Protocol &P = *pk.P; const EvaluationDomain &dom = *pk.dom;
  const uint32_t k = P.k, Q = P.Q; const uint64_t n = P.n, u = P.usable;
  const uint32_t A = P.num_advice(), NL = (uint32_t)P.lookups.size(), NZ = (uint32_t)P.perm.size();
  ProofResult R;
  // resident coset parts were placed by keygen at q % pk.devices: the gate kernel on device d must find part q on d, so the proof runs on the key's device count
  // (a differing opt.devices would read a coset that sits on another device)
  if (pk.resident_cosets && pk.devices != std::max(1, opt.devices)) throw std::invalid_argument("create_proof: ProofOptions::devices (" + std::to_string(opt.devices) + ") differs from the proving key's (" + std::to_string(pk.devices) + ") while its coset parts are resident");
  const int D = std::max(1, std::min<int>(opt.devices, (int)Q));
  auto ms_since = [](Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); };
  const auto t_start = Clock::now(); auto tl = t_start;
  auto lap = [&](int step) { R.step_ms[step] += ms_since(tl); tl = Clock::now(); };
  Transcript T(opt.transcript == TranscriptKind::ByLayer ? reference_transcript(P) : opt.transcript);
  T.common_scalar(vk_transcript_scalar(P, pk.vk));
  for (const auto &v : wit.instances) T.common_scalar(v);
  std::map<uint32_t, DevicePoly> poly;   // protocol index -> Lagrange values until step 6, coefficients afterwards
  auto commit_one = [&](uint64_t basis, const void *ptr) { G1 out; check(mi355_msm_g1_dev(basis, 0, ptr, n, out.data())); R.msm++; T.write_point(out); };
  auto commit_many = [&](uint64_t basis, const std::vector<uint32_t> &refs) {
    const uint32_t B = opt.commit_batch ? opt.commit_batch : 32;
    for (size_t base = 0; base < refs.size(); base += B) {
      const uint32_t cnt = (uint32_t)std::min<size_t>(B, refs.size() - base);
      std::vector<const void *> ptrs(cnt); std::vector<G1> outs(cnt);
      for (uint32_t i = 0; i < cnt; i++) ptrs[i] = poly.at(refs[base + i]).p;
      check(mi355_msm_g1_batch_dev(basis, 0, ptrs.data(), cnt, n, outs.data())); R.msm += cnt;
      for (uint32_t i = 0; i < cnt; i++) T.write_point(outs[i]);
    }
  };
  // ---- steps 1-3: the witness crosses PCIe on other host threads (rayon workers in the real caller); commitments as the columns arrive
  std::vector<std::pair<uint32_t, const Column *>> uploads;
  for (uint32_t i = 0; i < A; i++) uploads.push_back({P.phase0[0] + i, &wit.advice[i]});
  for (uint32_t l = 0; l < NL; l++) uploads.push_back({P.phase0[1] + l, &wit.m[l]});
  uploads.push_back({P.random_poly, &wit.random_poly});                                   // step 5's polynomial (coefficients): needed last, crosses last
  for (const auto &up : uploads) poly[up.first];
  poly[P.inst0];
  for (const auto &c : P.perm) poly[c.z];                                                 // every entry exists before the uploaders start: the map's structure does not change under them
  for (const auto &l : P.lookups) poly[l.phi];
  std::mutex mu; std::condition_variable cv; std::vector<char> arrived(uploads.size(), 0); std::string upload_error;
  const size_t UT = (size_t)std::max(1, std::min<int>(opt.upload_threads, (int)uploads.size()));
  std::atomic<uint64_t> sparse_cols{0}, packed_cols{0}, link_bytes{0};
  auto upload_worker = [&](size_t first) {
    try {
      std::vector<uint32_t> sidx; std::vector<Fr> svals;   // this worker's scratch for the sparse form
      for (size_t i = first; i < uploads.size(); i += UT) {
        const uint64_t len = uploads[i].second->size();
        DevicePoly d(len, 0);
        uint64_t nz = len;
        if (opt.sparse_uploads && len >= (1u << 12) && uploads[i].first != P.random_poly) {
          sidx.resize(len); svals.resize(len);
          check(mi355_host_compact_nonzero(uploads[i].second->data(), len, sidx.data(), svals.data(), &nz, std::max(1, opt.threads / (int)UT)));
        }
        const bool is_m = uploads[i].first >= P.phase0[1] && uploads[i].first < P.phase0[1] + NL;
        if (opt.packed_multiplicities && is_m && wit.m_counts.size() == NL) {
          // a column whose KIND bounds its cells: counts below 2^32 on the rows up to l_last, then the prover's blinding values
          check(mi355_buf_upload_packed(d.p, wit.m_counts[uploads[i].first - P.phase0[1]].data(), len, 4));
          if (P.blind) check(mi355_buf_upload(d.at(u + 1), uploads[i].second->data() + (u + 1), P.blind * 32));
          packed_cols++; link_bytes += len * 4 + P.blind * 32;
        }
        else if (2 * nz <= len) { check(mi355_buf_upload_sparse(d.p, len, sidx.data(), svals.data(), nz)); sparse_cols++; link_bytes += nz * 36; }
        else { check(mi355_buf_upload(d.p, uploads[i].second->data(), len * 32)); link_bytes += len * 32; }
        { std::lock_guard<std::mutex> lk(mu); poly.at(uploads[i].first) = std::move(d); arrived[i] = 1; }
        cv.notify_all();
      }
    } catch (const std::exception &e) { { std::lock_guard<std::mutex> lk(mu); upload_error = e.what(); std::fill(arrived.begin(), arrived.end(), 1); } cv.notify_all(); }
  };
  {                                                                                     // step 1's input, made NOW: the instance column (public values, zero below).  Its few words must
    // cross PCIe before the uploader threads fill the copy stream: that stream is first-in first-out, and queued behind the witness this 1 KiB upload waited for the random polynomial
    // of step 5 (2 GiB at k = 26) -- 33 ms of host time in step 1 of every k = 26 proof until round 6 (17 ms at k = 25).  No device-wide synchronisation either: mi355_buf_upload
    // orders itself behind the zeroing of its block, and the compute stream behind the copy.
    DevicePoly inst(n, 0); check(mi355_buf_zero(inst.p, n * 32));
    if (!wit.instances.empty()) check(mi355_buf_upload(inst.p, wit.instances.data(), wit.instances.size() * 32));
    poly.at(P.inst0) = std::move(inst);
  }
  struct Joiner { std::vector<std::thread> th; void join() { for (auto &t : th) if (t.joinable()) t.join(); } ~Joiner() { join(); } } uploaders;
  for (size_t w = 0; w < UT; w++) uploaders.th.emplace_back(upload_worker, w);
  auto wait_for = [&](size_t i) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return arrived[i] != 0; }); if (!upload_error.empty()) throw Error(MI355_EHIP, "witness upload: " + upload_error); };
  const uint32_t batch_cap = opt.commit_batch ? opt.commit_batch : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(32, (uint64_t(1) << 25) / n));
  const bool batch_cols = A >= 16 && batch_cap > 1;
  const bool early_intt = opt.early_intt < 0 ? A >= 8 : opt.early_intt != 0;
  std::map<uint32_t, DevicePoly> coeff_early;
  auto to_coeff_early = [&](const std::vector<uint32_t> &refs) {
    if (!early_intt) return;
    std::vector<void *> ptrs;
    for (uint32_t r : refs) { DevicePoly c = clone(poly.at(r), 0); ptrs.push_back(c.p); coeff_early[r] = std::move(c); }
    check(mi355_ntt_fr_batch_dev(ptrs.data(), (uint32_t)ptrs.size(), k, dom.omega_inv.data(), dom.ifft_divisor.data())); R.intt += (uint32_t)ptrs.size();
  };
  auto commit_phase = [&](size_t lo, size_t hi) {                                       // uploads[lo, hi): one phase's columns, committed in order as they arrive
    std::vector<uint32_t> pending;
    for (size_t i = lo; i < hi; i++) {
      wait_for(i);
      if (!batch_cols) { commit_one(h_g_lagrange, poly.at(uploads[i].first).p); to_coeff_early({uploads[i].first}); }
      else { pending.push_back(uploads[i].first); if (pending.size() == batch_cap || i + 1 == hi) { commit_many(h_g_lagrange, pending); to_coeff_early(pending); pending.clear(); } }
    }
  };
  std::vector<Fr> ch;
  commit_phase(0, A);                                                                   // step 2
  for (uint32_t i = 0; i < P.num_challenge[0]; i++) ch.push_back(T.squeeze_challenge()); // theta
  commit_phase(A, A + NL);                                                              // step 3 (the multiplicities are the caller's: they do not depend on theta for the lookups halo2 compresses)
  for (uint32_t i = 0; i < P.num_challenge[1]; i++) ch.push_back(T.squeeze_challenge()); // beta, gamma
  const Fr beta = ch.at(1), gamma = ch.at(2);
  lap(2);
  DevicePoly inst_lagrange = clone(poly.at(P.inst0), 0);
  check(mi355_intt_fr_dev(poly.at(P.inst0).p, k, dom.omega_inv.data(), dom.ifft_divisor.data())); R.intt++;
  lap(1);
  // ---- step 4: grand products and running sums, built on the device from the Lagrange values
  {
    uint32_t max_chunk = 0; for (const auto &c : P.perm) max_chunk = std::max<uint32_t>(max_chunk, (uint32_t)c.columns.size());
    Compiler cmp(const_cast<CommonRegistry &>(pk.commons), false, ch);
    std::vector<DevicePoly> tmp;
    auto tmp_at = [&](uint32_t i) -> DevicePoly & { while (tmp.size() <= i) tmp.emplace_back(n, 0); return tmp[i]; };
    auto resolve = [&](const Atom &a) -> const void * {
      if (a.kind == A_TMP) return tmp_at(a.idx).p;
      if (a.kind == A_COMMON) { if (a.idx != pk.identity_common) throw std::invalid_argument("step 4 reads no common polynomial but X"); return pk.common_lagrange[a.idx].p; }
      if (P.is_pre(a.idx)) { if (!pk.pre_lagrange[a.idx].p) throw std::invalid_argument("step 4: Lagrange values of a preprocessed polynomial were not kept"); return pk.pre_lagrange[a.idx].p; }
      if (P.is_instance(a.idx)) return inst_lagrange.p;
      return poly.at(a.idx).p;
    };
    auto run_all = [&]() { for (const auto &L : cmp.out) { run_launch(L, tmp_at((uint32_t)L.dst).p, n, fr_one(), L.accumulate, resolve); R.gate_launches++; } cmp.out.clear(); };
    Fr carry = fr_one();
    for (uint32_t c = 0; c < NZ; c++) {
      const PermChunk &chunk = P.perm[c]; const uint32_t cl = (uint32_t)chunk.columns.size();
      SoP us{Term{fr_one(), {}}}, vs{Term{fr_one(), {}}};
      cmp.tmp_base = 2; cmp.tmp_next = 0;
      for (uint32_t j = 0; j < cl; j++) {
        const PermColumn &pc = chunk.columns[j];
        SoP uj{Term{fr_one(), {Atom{A_POLY, pc.column, 0}}}, Term{fr_mul(beta, pc.delta_pow), {Atom{A_COMMON, pk.identity_common, 0}}}, Term{gamma, {}}};
        SoP vj{Term{fr_one(), {Atom{A_POLY, pc.column, 0}}}, Term{beta, {Atom{A_POLY, pc.sigma, 0}}}, Term{gamma, {}}};
        us = cmp.product(std::move(us), cmp.materialize(uj)); vs = cmp.product(std::move(vs), cmp.materialize(vj));
      }
      cmp.emit(0, us); cmp.emit(1, vs);                                                  // TMP[0] = prod_j u_j, TMP[1] = prod_j v_j
      run_all();
      check(mi355_fr_batch_invert_dev(tmp_at(1).p, n));
      check(mi355_fr_vec_op_dev(2, tmp_at(0).p, tmp_at(0).p, tmp_at(1).p, n));
      DevicePoly z(n, 0);
      check(mi355_fr_prefix_product_dev(z.p, tmp_at(0).p, n, nullptr));                  // z[0] = 1, z[i + 1] = z[i] prod u / prod v
      if (c > 0) check(mi355_fr_vec_axpy_dev(z.p, nullptr, z.p, carry.data(), n));       // chunk c starts where chunk c - 1 ended: z_c(1) = z_(c-1)(w^last)
      if (c + 1 < NZ) check(mi355_buf_download(carry.data(), z.at(u), 32));
      if (P.blind) check(mi355_buf_upload(z.at(u + 1), wit.z_blind.at(c).data(), P.blind * 32));
      poly.at(chunk.z) = std::move(z);
    }
    for (uint32_t l = 0; l < NL; l++) {                                                 // phi[i + 1] = phi[i] + 1 / (I + beta) - m / (T + beta)
      const Lookup &lk = P.lookups[l];
      cmp.tmp_base = 4; cmp.tmp_next = 0;
      SoP tb = cmp.compile(*lk.table); tb.push_back(Term{beta, {}}); Compiler::merge_scalars(tb);
      SoP ib = cmp.compile(*lk.input); ib.push_back(Term{beta, {}}); Compiler::merge_scalars(ib);
      cmp.emit(0, tb); cmp.emit(1, ib);
      const Atom t0{A_TMP, 0, 0}, t1{A_TMP, 1, 0}, mm{A_POLY, lk.m, 0};
      cmp.emit(2, SoP{Term{fr_one(), {t0, t1}}});
      cmp.emit(3, SoP{Term{fr_one(), {t0}}, Term{fr_neg(fr_one()), {mm, t1}}});
      run_all();
      check(mi355_fr_batch_invert_dev(tmp_at(2).p, n));
      check(mi355_fr_vec_op_dev(2, tmp_at(3).p, tmp_at(3).p, tmp_at(2).p, n));
      DevicePoly phi(n, 0);
      check(mi355_fr_prefix_sum_dev(phi.p, tmp_at(3).p, n, nullptr));
      if (P.blind) check(mi355_buf_upload(phi.at(u + 1), wit.phi_blind.at(l).data(), P.blind * 32));
      poly.at(lk.phi) = std::move(phi);
    }
    std::vector<uint32_t> made; for (uint32_t c = 0; c < NZ; c++) made.push_back(P.perm[c].z); for (uint32_t l = 0; l < NL; l++) made.push_back(P.lookups[l].phi);
    // small domains: one pass whatever the count (round 6: one reduction tail per batch); big ones keep one commitment per pass -- a batch of two at 2^24 doubles the sorter's
    // workspace (8.8 GB) for no gain, and a multi-layer prover process has no HBM to spare
    if (made.size() >= 8 || n <= (uint64_t(1) << 22)) commit_many(h_g_lagrange, made); else for (uint32_t r : made) commit_one(h_g_lagrange, poly.at(r).p);
    inst_lagrange.release();
  }
  lap(4);
  // ---- step 5: the random polynomial of the vanishing argument: one commitment on the coefficient basis, one evaluation later, no transform
  wait_for(uploads.size() - 1);
  uploaders.join();
  commit_one(h_g, poly.at(P.random_poly).p);
  for (uint32_t i = 0; i < P.num_challenge[2]; i++) ch.push_back(T.squeeze_challenge()); // y
  lap(5);
  // ---- step 6: every witness polynomial to coefficients, one batched call
  {
    std::vector<void *> ptrs;
    for (auto &kv : poly) if (kv.first != P.inst0 && kv.first != P.random_poly && !coeff_early.count(kv.first)) ptrs.push_back(kv.second.p);
    if (!ptrs.empty()) check(mi355_ntt_fr_batch_dev(ptrs.data(), (uint32_t)ptrs.size(), k, dom.omega_inv.data(), dom.ifft_divisor.data()));
    R.intt += (uint32_t)ptrs.size();
    for (auto &kv : coeff_early) poly.at(kv.first) = std::move(kv.second);
    coeff_early.clear();
  }
  lap(6);
  // ---- step 7: the quotient, coset part by coset part; part q on device q % D (the proving key's cosets of that part live there)
  DevicePoly h(Q * n, 0);
  {
    Compiler cmp(const_cast<CommonRegistry &>(pk.commons), false, ch);
    cmp.compile_numerator(P.numerator);
    R.plan_launches = (uint32_t)cmp.out.size(); R.plan_terms = cmp.terms_total; R.plan_tmps = cmp.tmps_used(); R.plan_constraints = cmp.constraints; R.plan_prefix_groups = cmp.prefix_groups;
    std::set<uint32_t> wset; std::set<Atom> pkset;
    for (const auto &L : cmp.out) for (const auto &t : L.terms) for (const auto &f : t.f) { if (f.kind == A_POLY && !P.is_pre(f.idx)) wset.insert(f.idx); else if (f.kind != A_TMP) pkset.insert(Atom{f.kind, f.idx, 0}); }
    const std::vector<uint32_t> wrefs(wset.begin(), wset.end()); const uint32_t NP = (uint32_t)wrefs.size();
    std::vector<std::map<uint32_t, DevicePoly>> part_on(D), coeff_on(D); std::vector<std::map<Atom, DevicePoly>> pkpart_on(D);
    std::vector<std::vector<DevicePoly>> tmp_on(D); std::vector<DevicePoly> hq_on(D);
    std::vector<DevicePoly> hpart; for (uint32_t q = 0; q < Q; q++) hpart.emplace_back(n, 0);
    for (int d = 0; d < D; d++) {
      for (uint32_t r : wrefs) part_on[d][r] = DevicePoly(n, d);
      tmp_on[d].resize(cmp.tmp_max);
      for (const auto &L : cmp.out) if (L.dst >= 0 && !tmp_on[d][(size_t)L.dst].p) tmp_on[d][(size_t)L.dst] = DevicePoly(n, d);   // only the ids the plan writes
      if (d > 0) { hq_on[d] = DevicePoly(n, d); for (uint32_t r : wrefs) { coeff_on[d][r] = DevicePoly(n, d); check(mi355_buf_copy(coeff_on[d][r].p, poly.at(r).p, n * 32)); } }
    }
    std::vector<std::string> errs(D); std::vector<uint32_t> launches(D, 0), cosets(D, 0);
    auto do_parts = [&](int d) {
      try {
        for (uint32_t q = (uint32_t)d; q < Q; q += (uint32_t)D) {
          const Fr factor = part_factor(dom, q);
          std::vector<void *> dst(NP); std::vector<const void *> src(NP);
          for (uint32_t i = 0; i < NP; i++) { dst[i] = part_on[d].at(wrefs[i]).p; src[i] = d == 0 ? poly.at(wrefs[i]).p : coeff_on[d].at(wrefs[i]).p; }
          check(mi355_coset_ntt_fr_batch_dev(dst.data(), src.data(), NP, k, factor.data(), dom.omega.data())); cosets[d] += NP;
          if (!pk.resident_cosets) {   // the HBM-lean proving key: this part's cosets are recomputed from the coefficients, in ONE batched call (round 6: a lean many-column key
            std::vector<void *> kd; std::vector<const void *> ks;   // used to issue one transform per polynomial -- 2 192 single 2^20 launches per layer-0 proof)
            for (const auto &a : pkset) {
              auto it = pkpart_on[d].find(a); if (it == pkpart_on[d].end()) it = pkpart_on[d].emplace(a, DevicePoly(n, d)).first;
              if (d != 0) check(mi355_buf_copy(it->second.p, pk.coeff(a).p, n * 32));   // another device: copy first, then in place
              kd.push_back(it->second.p); ks.push_back(d == 0 ? pk.coeff(a).p : it->second.p);
            }
            if (!kd.empty()) check(mi355_coset_ntt_fr_batch_dev(kd.data(), ks.data(), (uint32_t)kd.size(), k, factor.data(), dom.omega.data()));
            cosets[d] += (uint32_t)kd.size();
          }
          auto resolve = [&](const Atom &a) -> const void * {
            if (a.kind == A_TMP) return tmp_on[d].at(a.idx).p;
            if (a.kind == A_COMMON || P.is_pre(a.idx)) { const DevicePoly *c = pk.coset(a, q); return c ? c->p : pkpart_on[d].at(a).p; }
            return part_on[d].at(a.idx).p;
          };
          // 1 / ((zeta w_ext^q)^n - 1): the vanishing polynomial is constant on a coset part; it rides on the coefficients
          const Fr tq_inv = fr_inv(fr_sub(fr_pow(factor, n), fr_one()));
          void *hq = d == 0 ? hpart[q].p : hq_on[d].p;
          bool first = true;
          for (const auto &L : cmp.out) {
            if (L.dst >= 0) run_launch(L, tmp_on[d].at((size_t)L.dst).p, n, fr_one(), L.accumulate, resolve);
            else { run_launch(L, hq, n, tq_inv, !first, resolve); first = false; }
            launches[d]++;
          }
          if (d != 0) check(mi355_buf_copy(hpart[q].p, hq, n * 32));
        }
      } catch (const std::exception &e) { errs[d] = e.what(); }
    };
    { std::vector<std::thread> th; for (int d = 1; d < D; d++) th.emplace_back(do_parts, d); do_parts(0); for (auto &x : th) x.join(); }
    for (int d = 0; d < D; d++) { if (!errs[d].empty()) throw Error(MI355_EHIP, "quotient part on device slot " + std::to_string(d) + ": " + errs[d]); R.gate_launches += launches[d]; R.coset_ntt += cosets[d]; }
    { std::vector<const void *> pp(Q); for (uint32_t q = 0; q < Q; q++) pp[q] = hpart[q].p; check(mi355_fr_interleave_dev(h.p, pp.data(), Q, n)); }
    check(mi355_extended_to_coeff_dev(h.p, dom.extended_k, dom.g_coset.data(), dom.g_coset_inv.data(), dom.extended_omega_inv.data(), dom.extended_ifft_divisor.data()));
  }
  lap(7);
  if (n > (uint64_t(1) << 22)) { for (uint32_t q = 0; q < Q; q++) commit_one(h_g, h.at((uint64_t)q * n)); }   // step 8, big domains: one commitment per pass (see step 4)
  else {                                                                                // step 8: the Q pieces of h, one batched pass (at k = 20 / 21 one reduction tail instead of Q)
    std::vector<const void *> ptrs(Q); std::vector<G1> outs(Q);
    for (uint32_t q = 0; q < Q; q++) ptrs[q] = h.at((uint64_t)q * n);
    check(mi355_msm_g1_batch_dev(h_g, 0, ptrs.data(), Q, n, outs.data())); R.msm += Q;
    for (uint32_t q = 0; q < Q; q++) T.write_point(outs[q]);
  }
  const Fr x = T.squeeze_challenge();
  lap(8);
  auto coeff_ptr = [&](uint32_t p) -> const void * { return P.is_pre(p) ? pk.pre_coeff.at(p).p : poly.at(p).p; };
  auto rot_point = [&](int32_t rot) { Fr pt = x; if (rot > 0) pt = fr_mul(pt, fr_pow(dom.omega, (uint64_t)rot)); else if (rot < 0) pt = fr_mul(pt, fr_pow(dom.omega_inv, (uint64_t)(-(int64_t)rot))); return pt; };
  std::map<PolyRot, Fr> evals;
  {                                                                                     // step 9: the evaluations, in the protocol's order; plus the Q pieces at x (for the combined quotient's value)
    std::vector<const void *> ptrs; std::vector<Fr> pts;
    for (const auto &e : P.evaluations) { ptrs.push_back(coeff_ptr(e.poly)); pts.push_back(rot_point(e.rot)); }
    for (uint32_t q = 0; q < Q; q++) { ptrs.push_back(h.at((uint64_t)q * n)); pts.push_back(x); }
    std::vector<Fr> ev(ptrs.size());
    check(mi355_eval_polynomial_batch_dev(ptrs.data(), (uint32_t)ptrs.size(), n, pts.data(), ev.data()));
    for (size_t i = 0; i < P.evaluations.size(); i++) { evals[P.evaluations[i]] = ev[i]; T.write_scalar(ev[i]); }
    R.evals = (uint32_t)P.evaluations.size();
    const Fr xn = fr_pow(x, n); Fr f = fr_one(), hx = fr_zero();
    for (uint32_t q = 0; q < Q; q++) { hx = fr_add(hx, fr_mul(f, ev[P.evaluations.size() + q])); f = fr_mul(f, xn); }
    evals[{P.quotient_poly, 0}] = hx;
  }
  lap(9);
  {                                                                                     // step 10: SHPLONK
    // the vanishing argument opens ONE combined quotient polynomial h_0 + x^n h_1 + ...
    DevicePoly hcomb(n, 0);
    {
      const Fr xn = fr_pow(x, n); std::vector<const void *> pp(Q); std::vector<Fr> cs(Q); std::vector<uint32_t> tl_(Q, 1), fp(Q); std::vector<int32_t> fr_(Q, 0);
      Fr f = fr_one(); for (uint32_t q = 0; q < Q; q++) { pp[q] = h.at((uint64_t)q * n); cs[q] = f; f = fr_mul(f, xn); fp[q] = q; }
      gate_eval(hcomb.p, pp.data(), Q, cs.data(), tl_.data(), Q, fp.data(), fr_.data(), n, 0); R.gate_launches++;
    }
    auto opened = [&](uint32_t p) -> const void * { return p == P.quotient_poly ? hcomb.p : coeff_ptr(p); };
    const Fr ys = T.squeeze_challenge(), v = T.squeeze_challenge();
    const std::vector<RotationSet> sets = rotation_sets(P.queries); const size_t M = sets.size(); R.rotation_sets = (uint32_t)M;
    std::vector<DevicePoly> Acomb; std::vector<std::vector<Fr>> points(M); std::vector<std::vector<std::vector<Fr>>> rcoef(M);
    DevicePoly H(n, 0), work(n, 0); check(mi355_buf_zero(H.p, n * 32));
    Fr vpow_i = fr_one();
    for (size_t i = 0; i < M; i++) {
      const RotationSet &s = sets[i]; const size_t np = s.polys.size(), m = s.rots.size();
      for (int32_t r : s.rots) points[i].push_back(rot_point(r));
      // A_i = sum_j y^j P_ij (the j-th polynomial of a set carries y^j: what the reference's released proofs satisfy, tests/test_plonk_protocol.py::test_reference_released_proofs_verify),
      // and the same combination of the interpolated remainders (m coefficients, on the host)
      std::vector<Fr> ypow(np, fr_one()); for (size_t j = 1; j < np; j++) ypow[j] = fr_mul(ypow[j - 1], ys);
      DevicePoly Ai(n, 0);
      for (size_t base = 0; base < np; base += 16) {
        const uint32_t cnt = (uint32_t)std::min<size_t>(16, np - base);
        std::vector<const void *> pp(cnt); std::vector<Fr> cs(cnt); std::vector<uint32_t> tl_(cnt, 1), fp(cnt); std::vector<int32_t> fr_(cnt, 0);
        for (uint32_t j = 0; j < cnt; j++) { pp[j] = opened(s.polys[base + j]); cs[j] = ypow[base + j]; fp[j] = j; }
        gate_eval(Ai.p, pp.data(), cnt, cs.data(), tl_.data(), cnt, fp.data(), fr_.data(), n, base ? 1 : 0); R.gate_launches++;
      }
      std::vector<Fr> rsum(m, fr_zero());
      for (size_t j = 0; j < np; j++) {
        std::vector<Fr> vals; for (int32_t r : s.rots) vals.push_back(evals.at({s.polys[j], r}));
        rcoef[i].push_back(detail::interpolate(points[i], vals));
        for (size_t t = 0; t < m; t++) rsum[t] = fr_add(rsum[t], fr_mul(ypow[j], rcoef[i][j][t]));
      }
      // N_i = A_i - R_i: only the lowest m coefficients change; then N_i / prod (X - point), one kate_division per point, in place (shifting up by one each time)
      check(mi355_buf_copy(work.p, Ai.p, n * 32));
      { std::vector<Fr> low(m); check(mi355_buf_download(low.data(), work.p, m * 32)); for (size_t t = 0; t < m; t++) low[t] = fr_sub(low[t], rsum[t]); check(mi355_buf_upload(work.p, low.data(), m * 32)); }
      for (size_t t = 0; t < m; t++) check(mi355_fr_kate_division_dev(work.at(t + 1), work.at(t), n - t, points[i][t].data()));
      // H += v^i Q_i (the i-th rotation set carries v^i); Q_i has n - m coefficients at work[m ..]
      check(mi355_fr_vec_axpy_dev(H.p, H.p, work.at(m), vpow_i.data(), n - m));
      vpow_i = fr_mul(vpow_i, v);
      Acomb.push_back(std::move(Ai));
    }
    commit_one(h_g, H.p);
    const Fr uu = T.squeeze_challenge();
    std::vector<Fr> super; for (size_t i = 0; i < M; i++) for (const auto &pt : points[i]) if (std::find(super.begin(), super.end(), pt) == super.end()) super.push_back(pt);
    Fr zt = fr_one(); for (const auto &pt : super) zt = fr_mul(zt, fr_sub(uu, pt));
    // L = sum_i v^i zd_i (A_i - r_i(u)) - Z_T(u) H, scaled by 1 / zd_0;  L(u) = 0
    std::vector<Fr> vpow(M, fr_one()); for (size_t i = 1; i < M; i++) vpow[i] = fr_mul(vpow[i - 1], v);
    std::vector<Fr> lc(M); Fr zd0 = fr_one(), cst = fr_zero();
    for (size_t i = 0; i < M; i++) {
      Fr zd = fr_one(); for (const auto &pt : super) if (std::find(points[i].begin(), points[i].end(), pt) == points[i].end()) zd = fr_mul(zd, fr_sub(uu, pt));
      if (i == 0) zd0 = zd;
      const size_t np = sets[i].polys.size(); Fr ri = fr_zero();
      Fr yp = fr_one(); for (size_t j = 0; j < np; j++) { ri = fr_add(ri, fr_mul(yp, detail::horner(rcoef[i][j], uu))); yp = fr_mul(yp, ys); }
      lc[i] = fr_mul(vpow[i], zd); cst = fr_add(cst, fr_mul(lc[i], ri));
    }
    const Fr zi = fr_inv(zd0);
    DevicePoly Lx(n, 0);
    {
      if (M + 1 > PLAN_MAX_TERMS) throw std::invalid_argument("SHPLONK: more than 15 rotation sets");
      std::vector<const void *> pp; std::vector<Fr> cs; for (size_t i = 0; i < M; i++) { pp.push_back(Acomb[i].p); cs.push_back(fr_mul(lc[i], zi)); }
      pp.push_back(H.p); cs.push_back(fr_neg(fr_mul(zt, zi)));
      std::vector<uint32_t> tl_(M + 1, 1), fp(M + 1); std::vector<int32_t> fr_(M + 1, 0); for (size_t i = 0; i <= M; i++) fp[i] = (uint32_t)i;
      gate_eval(Lx.p, pp.data(), (uint32_t)(M + 1), cs.data(), tl_.data(), (uint32_t)(M + 1), fp.data(), fr_.data(), n, 0); R.gate_launches++;
      Fr l0; check(mi355_buf_download(l0.data(), Lx.p, 32)); l0 = fr_sub(l0, fr_mul(cst, zi)); check(mi355_buf_upload(Lx.p, l0.data(), 32));
    }
    check(mi355_buf_zero(work.p, n * 32));
    check(mi355_fr_kate_division_dev(work.p, Lx.p, n, uu.data()));                       // n - 1 coefficients, the top one stays zero
    commit_one(h_g, work.p);
  }
  check(mi355_synchronize());
  lap(10);
  R.total_ms = ms_since(t_start);
  R.proof = std::move(T.proof); R.sparse_columns = sparse_cols.load(); R.packed_columns = packed_cols.load(); R.witness_link_bytes = link_bytes.load();
  { uint64_t fr_ = 0, tot = 0; check(mi355_mem_info(0, &fr_, &tot, nullptr, nullptr, nullptr)); R.peak_hbm_bytes = tot - fr_; R.hbm_total_bytes = tot; }
  return R;
}

}  // namespace plonk
}  // namespace mi355zk
