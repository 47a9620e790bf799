// mi355zk_plonk_circuit.hpp -- a circuit instance for a PlonkProtocol: fixed columns, copy constraints and a witness that SATISFY the protocol's constraints.
//
// In the reference this is the CPU side that runs before create_proof: keygen assigns the fixed columns and the permutation, `synthesize` fills the advice
// columns [EXT-recalled halo2_proofs plonk/keygen.rs, plonk/prover.rs WitnessCollection; the circuits are zkevm-circuits' / aggregator's, absent from the
// checkout].  No circuit of the reference can be synthesised here (no Rust, no traces), so this builder makes an instance for the protocol it is given --
// the reference's own constraint systems of layers 2 and 4, the halo2-base rule for layers 1 / 3 / 5 / 6, the stated-shape stand-in for layer 0 -- by reading
// the recognised structure (mi355zk_plonk_protocol.hpp):
//   vertical gates   q (a + a(wX) a(w^2 X) - a(w^3 X)), halo2-base's FlexGate: blocks of four rows, three free inputs, the output computed, q = 1 on the first
//                    row of every used block (a densely used prefix of the column, the rest zero: how a real halo2-base column looks)
//   assigned gates   s (P - t): t := P row by row, columns in index order (the synthetic inner circuit)
//   lookups          inputs are drawn FROM the table (range table 0 .. 2^lookup_bits - 1, or W-column tuples), the multiplicities m counted
//   copy constraints 2-cycles between free cells of different kinds (gate inputs, lookup-advice cells, fixed constants, instance cells); sigma follows
//   blinding         the last `blind` rows of every advice / m column are random, as halo2 leaves them; z / phi blinding values and the random
//                    polynomial of the vanishing argument are drawn here too (in halo2 the prover draws them from its rng during create_proof)
// Everything here is HOST code and set-up (untimed).  `dump` writes the instance for the CPU restatement of the prover (oracle/plonk.py, tests only).
#pragma once
#include <functional>
#include <memory>
#include <random>
#include <set>
#include <thread>

#include "mi355zk_plonk_protocol.hpp"

namespace mi355zk {
namespace plonk {

using halo2::Column;
using halo2::ColumnAllocator;

struct CopyPair { uint32_t ja; uint64_t ra; uint32_t jb; uint64_t rb; };   // (permutation position, row) <-> (permutation position, row); value flows a -> b

struct Circuit {
  const Protocol *pr = nullptr;
  std::vector<Column> pre;            // Lagrange values of the preprocessed polynomials; the sigma columns are produced on demand (sigma_column)
  std::vector<Fr> instances;
  std::vector<Column> advice, m;
  std::vector<std::vector<uint32_t>> m_counts;   // the multiplicities as the integers they are (rows below the l_last row; the blinding rows of `m` are random field elements)
  std::vector<std::vector<Fr>> z_blind, phi_blind;
  Column random_poly;                 // coefficients
  std::vector<CopyPair> pairs; std::vector<PermColumn> pcols; std::vector<Fr> omega_pow;
  uint64_t gates_active = 0, lookup_rows = 0;
  int threads = 8;

  bool is_sigma(uint32_t p) const { for (const auto &c : pcols) if (c.sigma == p) return true; return false; }
  // sigma_j[row] = delta^j' omega^row' for the image (j', row') of (j, row) under the permutation
  void sigma_column(uint32_t j, std::vector<Fr> &out) const {
    const uint64_t n = pr->n; out.resize(n);
    parallel(n, [&](uint64_t lo, uint64_t hi) { for (uint64_t r = lo; r < hi; r++) out[r] = fr_mul(pcols[j].delta_pow, omega_pow[r]); });
    for (const auto &p : pairs) {
      if (p.ja == j) out[p.ra] = fr_mul(pcols[p.jb].delta_pow, omega_pow[p.rb]);
      if (p.jb == j) out[p.rb] = fr_mul(pcols[p.ja].delta_pow, omega_pow[p.ra]);
    }
  }
  template <class F> void parallel(uint64_t n, F f) const {
    if (n < (1u << 14) || threads <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(f, n * t / threads, n * (t + 1) / threads);
    for (auto &t : th) t.join();
  }
};

namespace detail {
struct Rng {   // splitmix64
  uint64_t s; explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  Fr uniform() { return Fr{{next(), next(), next(), next() & ((uint64_t(1) << 60) - 1)}}; }   // a reduced Montgomery representation of a uniform element
};
inline const std::vector<Fr> &small_table() { static const std::vector<Fr> t = [] { std::vector<Fr> v(1 << 16); for (uint64_t i = 0; i < v.size(); i++) v[i] = fr_u64(i); return v; }(); return t; }
inline Fr fr_small(uint64_t v) { return v < (1u << 16) ? small_table()[v] : fr_u64(v); }
// witness-like values: 60 % zero, 20 % < 256, 10 % 64-bit, 10 % uniform
inline Fr witness_like(Rng &g) {
  const uint64_t u = g.next() % 10;
  if (u < 6) return fr_zero();
  if (u < 8) return fr_small(g.next() & 255);
  if (u < 9) return fr_u64(g.next());
  return g.uniform();
}
// a table index, biased like range-checked limbs: mostly small, some anywhere in the table
inline uint32_t table_index(Rng &g, uint32_t rows) { const uint64_t u = g.next(); return (u & 3) == 0 ? (uint32_t)((u >> 8) % rows) : (uint32_t)((u >> 8) % std::min<uint32_t>(rows, 256)); }

// row-wise evaluation of a gate's P over host columns (assigned gates)
struct RowEval {
  struct Op { uint8_t kind; const Fr *col = nullptr; int32_t rot = 0; Fr c{}; };   // 0 push column cell, 1 push constant, 2 add, 3 mul, 4 neg
  std::vector<Op> prog;
  void compile(const Expr &e, const std::function<const Fr *(uint32_t)> &column) {
    switch (e.kind) {
      case Expr::CONSTANT: prog.push_back({1, nullptr, 0, e.c}); return;
      case Expr::POLY: prog.push_back({0, column((uint32_t)e.i), e.rot, {}}); return;
      case Expr::NEG: compile(e.kids[0], column); prog.push_back({4}); return;
      case Expr::SUM: compile(e.kids[0], column); compile(e.kids[1], column); prog.push_back({2}); return;
      case Expr::PROD: compile(e.kids[0], column); compile(e.kids[1], column); prog.push_back({3}); return;
      case Expr::SCALED: compile(e.kids[0], column); prog.push_back({1, nullptr, 0, e.c}); prog.push_back({3}); return;
      default: throw std::invalid_argument("circuit builder: a gate's P may hold constants, columns, sums and products only");
    }
  }
  Fr at(uint64_t row, uint64_t n) const {
    Fr st[24]; int sp = 0;
    for (const auto &o : prog) {
      switch (o.kind) {
        case 0: st[sp++] = o.col[(row + n + (uint64_t)(int64_t)o.rot) & (n - 1)]; break;
        case 1: st[sp++] = o.c; break;
        case 2: sp--; st[sp - 1] = fr_add(st[sp - 1], st[sp]); break;
        case 3: sp--; st[sp - 1] = fr_mul(st[sp - 1], st[sp]); break;
        default: st[sp - 1] = fr_neg(st[sp - 1]); break;
      }
      if (sp >= 23) throw std::invalid_argument("circuit builder: expression too deep");
    }
    return st[0];
  }
};
}  // namespace detail

struct CircuitOptions { uint64_t seed = 1; int threads = 8; bool pinned = false; double fill = 0.9; double assign_density = 1.0 /* fraction of the usable rows on which the assigned gates' selector is on: the other cells of every dependent column stay zero */;
                        // the PROVER's randomness, separable from the witness (in halo2 it comes from the rng handed to create_proof): blind_seed != 0 draws the blinding rows, the z / phi
                        // blinding values and the random polynomial from their own stream (same witness, other proof bytes); zero_blinding leaves the blinding rows and values zero -- a proof
                        // the verifier still accepts (it cannot see the choice; zero knowledge is what is lost).  tests/test_plonk_protocol.py::test_gpu_prover_invisible_choices
                        uint64_t blind_seed = 0; bool zero_blinding = false; };

inline std::unique_ptr<Circuit> build_circuit(const Protocol &P, const CircuitOptions &opt) {
  using namespace detail;
  auto C = std::make_unique<Circuit>(); C->pr = &P; C->threads = std::max(1, opt.threads);
  const uint64_t n = P.n, u = P.usable; const uint32_t NI = std::min<uint64_t>(P.num_instance[0], u);
  const ColumnAllocator<Fr> alloc(opt.pinned);
  const uint32_t A = P.num_advice(), L = (uint32_t)P.lookups.size();
  auto adv_index = [&](uint32_t poly) -> int { return poly >= P.phase0[0] && poly < P.phase0[0] + A ? (int)(poly - P.phase0[0]) : -1; };
  // ---- roles
  for (const auto &ch : P.perm) for (const auto &c : ch.columns) C->pcols.push_back(c);
  const uint32_t NP = (uint32_t)C->pcols.size();
  std::map<uint32_t, uint32_t> perm_pos; for (uint32_t j = 0; j < NP; j++) perm_pos[C->pcols[j].column] = j;
  enum AdvKind { PLAIN, VERT, ASSIGN, LOOKUP_IN };
  struct AdvRole { AdvKind kind = PLAIN; const Gate *gate = nullptr; int lookup = -1; };
  std::vector<AdvRole> role(A);
  for (const auto &g : P.gates) {
    if (!g.assignable) throw std::invalid_argument("circuit builder: a gate is not of the form selector * (P - target)");
    const int t = adv_index(g.target.poly); if (t < 0) throw std::invalid_argument("circuit builder: gate target is not an advice column");
    std::vector<std::pair<int32_t, int32_t>> reads; collect_polys(*g.p, reads);
    bool vertical = g.target.rot > 0; bool self = false;
    for (const auto &r : reads) { if ((uint32_t)r.first == g.target.poly) { self = true; if (r.second < 0 || r.second >= g.target.rot) vertical = false; } else vertical = false; }
    if (vertical && self) role[t] = {VERT, &g, -1};
    else {
      if (g.target.rot != 0 || self) throw std::invalid_argument("circuit builder: an assigned gate must target rotation 0 of a column it does not read");
      for (const auto &r : reads) if ((uint32_t)r.first >= P.wit0 && (uint32_t)r.first >= g.target.poly) throw std::invalid_argument("circuit builder: an assigned gate may only read earlier columns");
      role[t] = {ASSIGN, &g, -1};
    }
  }
  struct LookupRole { std::vector<uint32_t> table; std::vector<int> input; std::vector<int> selector; int group = -1; };   // selector: preprocessed index or -1
  std::vector<LookupRole> lk(L);
  auto exprs_of = [](const Expr *e) { std::vector<const Expr *> v; if (e->kind == Expr::DPOW) for (size_t i = 0; i + 1 < e->kids.size(); i++) v.push_back(&e->kids[i]); else v.push_back(e); return v; };
  for (uint32_t l = 0; l < L; l++) {
    for (const Expr *t : exprs_of(P.lookups[l].table)) { if (t->kind != Expr::POLY || t->rot != 0 || !P.is_pre((uint32_t)t->i)) throw std::invalid_argument("circuit builder: table expressions must be fixed columns"); lk[l].table.push_back((uint32_t)t->i); }
    for (const Expr *in : exprs_of(P.lookups[l].input)) {
      const Expr *col = in; int sel = -1;
      if (in->kind == Expr::PROD && in->kids[0].kind == Expr::POLY && P.is_pre((uint32_t)in->kids[0].i)) { sel = in->kids[0].i; col = &in->kids[1]; }
      const int a = col->kind == Expr::POLY && col->rot == 0 ? adv_index((uint32_t)col->i) : -1;
      if (a < 0) throw std::invalid_argument("circuit builder: lookup inputs must be advice columns, optionally under a fixed selector");
      lk[l].input.push_back(a); lk[l].selector.push_back(sel);
      if (sel < 0) { if (role[a].kind == PLAIN) role[a] = {LOOKUP_IN, nullptr, (int)l}; else if (role[a].kind != LOOKUP_IN) throw std::invalid_argument("circuit builder: an unselected lookup input must be a free column"); }
    }
    if (lk[l].table.size() != lk[l].input.size()) throw std::invalid_argument("circuit builder: lookup width mismatch");
  }
  // lookups reading the same input columns share one index stream
  std::vector<std::vector<int>> groups;
  for (uint32_t l = 0; l < L; l++) { for (size_t g = 0; g < groups.size(); g++) if (lk[groups[g][0]].input == lk[l].input && lk[groups[g][0]].table == lk[l].table) { lk[l].group = (int)g; groups[g].push_back((int)l); } if (lk[l].group < 0) { lk[l].group = (int)groups.size(); groups.push_back({(int)l}); } }
  for (uint32_t l = 0; l < L; l++) for (int a : lk[l].input) if (role[a].kind == LOOKUP_IN && lk[role[a].lookup].group != lk[l].group) throw std::invalid_argument("circuit builder: a lookup-input column is shared by lookups with different tuples");
  const uint32_t bits = P.lookup_bits ? std::min<uint32_t>(P.lookup_bits, P.k - 1) : std::min<uint32_t>(16, P.k - 1);
  const uint32_t table_rows = (uint32_t)std::min<uint64_t>(uint64_t(1) << bits, u);
  // ---- columns
  C->pre.resize(P.num_pre, Column(alloc));
  for (uint32_t a = 0; a < A; a++) { C->advice.emplace_back(alloc); C->advice.back().assign(n, fr_zero()); }
  for (uint32_t l = 0; l < L; l++) { C->m.emplace_back(alloc); C->m.back().assign(n, fr_zero()); }
  auto pre_col = [&](uint32_t p) -> Column & { if (C->pre[p].empty()) C->pre[p].assign(n, fr_zero()); return C->pre[p]; };
  Rng top(opt.seed * 0x9E3779B97F4A7C15ull + 12345);
  { Rng g(top.next()); C->instances.resize(NI); for (auto &v : C->instances) v = h2d::from_fe(zk::Fr::from_canonical(h2d::to_fe(Fr{{g.next(), g.next() & 0xffffff, 0, 0}}))); }   // 88-bit values, like the accumulator limbs of the fixtures' instances
  // table columns: T_j[i] = i (j + 1) for i < table_rows, zero below
  std::set<uint32_t> table_polys; for (const auto &r : lk) for (uint32_t j = 0; j < r.table.size(); j++) table_polys.insert(r.table[j]);
  for (const auto &r : lk) for (uint32_t j = 0; j < r.table.size(); j++) { Column &t = pre_col(r.table[j]); C->parallel(table_rows, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) t[i] = fr_small(i * (j + 1)); }); }
  // coefficient columns: fixed columns a gate's P reads that are neither tables nor under the permutation -- small non-zero constants on every row
  { std::set<uint32_t> coeff; for (const auto &g : P.gates) { std::vector<std::pair<int32_t, int32_t>> reads; collect_polys(*g.p, reads); for (const auto &r : reads) if (P.is_pre((uint32_t)r.first) && !table_polys.count((uint32_t)r.first) && !perm_pos.count((uint32_t)r.first)) coeff.insert((uint32_t)r.first); }
    for (uint32_t p : coeff) { Column &c = pre_col(p); const uint64_t sd = top.next(); C->parallel(n, [&](uint64_t lo, uint64_t hi) { Rng rg(sd + lo * 7919); for (uint64_t i = lo; i < hi; i++) c[i] = fr_small(1 + (rg.next() & 0xff)); }); } }
  // index streams of the lookup groups (which table row each active row reads)
  std::vector<std::vector<uint32_t>> idx(groups.size());
  for (size_t g = 0; g < groups.size(); g++) idx[g].assign(n, 0);
  // ---- vertical-gate columns and plain columns: free cells
  std::vector<int> vert, plain; for (uint32_t a = 0; a < A; a++) { if (role[a].kind == VERT) vert.push_back((int)a); else if (role[a].kind == PLAIN) plain.push_back((int)a); }
  const uint32_t G = (uint32_t)vert.size();
  const uint64_t B = u / 4, Bact = std::max<uint64_t>(1, (uint64_t)(opt.fill * (double)B));
  C->gates_active = Bact * G;
  for (uint32_t gi = 0; gi < G; gi++) {
    Column &a = C->advice[vert[gi]]; const uint64_t sd = top.next();
    C->parallel(Bact, [&](uint64_t lo, uint64_t hi) { Rng g(sd + lo * 7919); for (uint64_t b = lo; b < hi; b++) { a[4 * b] = witness_like(g); a[4 * b + 1] = fr_small(g.next() & 0xffff); a[4 * b + 2] = witness_like(g); } });
    Column &q = pre_col(role[vert[gi]].gate->selector);
    C->parallel(Bact, [&](uint64_t lo, uint64_t hi) { for (uint64_t b = lo; b < hi; b++) q[4 * b] = fr_one(); });
  }
  for (int a : plain) { Column &c = C->advice[a]; const uint64_t sd = top.next(); C->parallel(u, [&](uint64_t lo, uint64_t hi) { Rng g(sd + lo * 7919); for (uint64_t i = lo; i < hi; i++) c[i] = witness_like(g); }); }
  // ---- lookup inputs
  for (size_t g = 0; g < groups.size(); g++) {
    const LookupRole &r = lk[groups[g][0]];
    const bool in_place = r.selector[0] >= 0;
    const uint64_t sd = top.next();
    if (!in_place) {   // whole columns hold table tuples
      C->parallel(u, [&](uint64_t lo, uint64_t hi) { Rng rg(sd + lo * 7919); for (uint64_t i = lo; i < hi; i++) idx[g][i] = table_index(rg, table_rows); });
      for (size_t j = 0; j < r.input.size(); j++) { Column &c = C->advice[r.input[j]]; C->parallel(u, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) c[i] = fr_small((uint64_t)idx[g][i] * (j + 1)); }); }
      C->lookup_rows += u;
    } else {           // halo2-base with a single basic-gate column: q_lookup marks the range-checked cells (the middle input of every second used block)
      if (r.input.size() != 1 || role[r.input[0]].kind != VERT) throw std::invalid_argument("circuit builder: in-place lookups are supported on a vertical-gate column");
      Column &a = C->advice[r.input[0]]; Column &q = pre_col((uint32_t)r.selector[0]);
      C->parallel(Bact, [&](uint64_t lo, uint64_t hi) { Rng rg(sd + lo * 7919); for (uint64_t b = lo; b < hi; b++) { if (b & 1) continue; const uint32_t v = table_index(rg, table_rows); a[4 * b + 1] = fr_small(v); q[4 * b + 1] = fr_one(); idx[g][4 * b + 1] = v; } });
      C->lookup_rows += Bact / 2;
    }
  }
  // ---- fixed constants under the permutation
  std::vector<uint32_t> fixed_perm; for (uint32_t j = 0; j < NP; j++) if (P.is_pre(C->pcols[j].column)) fixed_perm.push_back(j);
  const uint64_t NC = std::min<uint64_t>(256, Bact / 3);
  for (uint32_t j : fixed_perm) { Column &f = pre_col(C->pcols[j].column); Rng g(top.next()); for (uint64_t r = 0; r < NC; r++) f[r] = fr_u64(g.next()); }
  // ---- copy constraints: 2-cycles between free cells
  auto cell = [&](uint32_t j, uint64_t r) -> Fr & {
    const uint32_t p = C->pcols[j].column;
    if (P.is_pre(p)) return C->pre[p][r];
    if (P.is_instance(p)) return C->instances.at(r);
    return C->advice[adv_index(p)][r];
  };
  auto pos_of_adv = [&](int a) -> int { auto it = perm_pos.find(P.phase0[0] + (uint32_t)a); return it == perm_pos.end() ? -1 : (int)it->second; };
  int inst_pos = -1; for (uint32_t j = 0; j < NP; j++) if (P.is_instance(C->pcols[j].column)) inst_pos = (int)j;
  if (G) {
    for (uint32_t gi = 0; gi < G; gi++) {
      const int ja = pos_of_adv(vert[gi]), jb = pos_of_adv(vert[(gi + 1) % G]);
      if (ja >= 0 && jb >= 0) for (uint64_t b = 0; b < Bact; b += 3) C->pairs.push_back({(uint32_t)ja, 4 * b, (uint32_t)jb, 4 * (Bact - 1 - b) + 2});                  // input 0 of block b <-> input 2 of a block of the next column
    }
    const uint64_t FS = (fixed_perm.size() + G - 1) / G;   // fixed columns that share one gate column take interleaved blocks: no gate input is fed twice
    for (size_t fi = 0; fi < fixed_perm.size(); fi++) { const int jb = pos_of_adv(vert[fi % G]); if (jb >= 0) for (uint64_t r = 0; r < NC && 3 * (r * FS + fi / G) + 1 < Bact; r++) C->pairs.push_back({fixed_perm[fi], r, (uint32_t)jb, 4 * (3 * (r * FS + fi / G) + 1)}); }   // a constant feeds a gate input
    if (inst_pos >= 0) { const int jb = pos_of_adv(vert[G - 1]); if (jb >= 0) for (uint64_t r = 0; r < NI && 3 * r + 2 < Bact; r++) C->pairs.push_back({(uint32_t)inst_pos, r, (uint32_t)jb, 4 * (3 * r + 2)}); }                                  // a public input feeds a gate input
    uint32_t li = 0;
    for (uint32_t a = 0; a < A; a++) if (role[a].kind == LOOKUP_IN && li < G) {   // a range-checked cell is copied into the middle input of a gate
      const int ja = pos_of_adv((int)a), jb = pos_of_adv(vert[li % G]); li++;
      if (ja >= 0 && jb >= 0) for (uint64_t b = 0; b < std::min(Bact, u); b++) C->pairs.push_back({(uint32_t)ja, b, (uint32_t)jb, 4 * b + 1});
    }
  } else {
    for (size_t t = 0; t + 1 < plain.size(); t += 2) { const int ja = pos_of_adv(plain[t]), jb = pos_of_adv(plain[t + 1]); if (ja >= 0 && jb >= 0) for (uint64_t r = 0; r < u; r += 4) C->pairs.push_back({(uint32_t)ja, r, (uint32_t)jb, u - 1 - r}); }
    if (inst_pos >= 0 && !plain.empty()) { const int jb = pos_of_adv(plain[0]); if (jb >= 0) for (uint64_t r = 0; r < NI && 4 * r + 1 < u; r++) C->pairs.push_back({(uint32_t)inst_pos, r, (uint32_t)jb, 4 * r + 1}); }
  }
  for (const auto &p : C->pairs) cell(p.jb, p.rb) = cell(p.ja, p.ra);
  // ---- outputs of the vertical gates, then the assigned columns in index order
  auto host_column = [&](uint32_t poly) -> const Fr * {
    if (P.is_pre(poly)) return pre_col(poly).data();
    if (P.is_instance(poly)) throw std::invalid_argument("circuit builder: gates reading the instance column are not supported");
    return C->advice[adv_index(poly)].data();
  };
  for (uint32_t gi = 0; gi < G; gi++) {
    const Gate &g = *role[vert[gi]].gate;
    if (g.target.rot != 3) throw std::invalid_argument("circuit builder: vertical gates span four rows (halo2-base's basic gate)");
    RowEval ev; ev.compile(*g.p, host_column);
    Column &a = C->advice[vert[gi]]; C->parallel(Bact, [&](uint64_t lo, uint64_t hi) { for (uint64_t b = lo; b < hi; b++) a[4 * b + 3] = ev.at(4 * b, n); });
  }
  // every advice column's blinding rows are drawn BEFORE the assigned gates run: their P may read rotated cells of earlier columns across the wrap-around
  Rng blind_top(opt.blind_seed * 0xD1B54A32D192ED03ull + 777);
  auto blind_stream = [&]() { const uint64_t own = top.next(), other = blind_top.next(); return opt.blind_seed ? other : own; };   // `top` advances either way: the witness does not depend on blind_seed
  for (uint32_t a = 0; a < A; a++) { Rng g(blind_stream()); for (uint64_t r = u + 1; r < n; r++) C->advice[a][r] = opt.zero_blinding ? fr_zero() : g.uniform(); }
  std::set<uint32_t> assign_selectors;
  for (uint32_t a = 0; a < A; a++) if (role[a].kind == ASSIGN) {
    const Gate &g = *role[a].gate;
    Column &sel = pre_col(g.selector);
    if (assign_selectors.insert(g.selector).second) {
      const uint64_t thr = opt.assign_density >= 1.0 ? ~uint64_t(0) : (uint64_t)(opt.assign_density * 18446744073709551615.0);
      C->parallel(u, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) { Rng h(i * 0x9E3779B97F4A7C15ull + g.selector); if (h.next() <= thr) sel[i] = fr_one(); } });
    }
    RowEval ev; ev.compile(*g.p, host_column);
    Column &t = C->advice[a];
    C->parallel(u, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) if (!fr_is_zero(sel[i])) t[i] = ev.at(i, n); });
  }
  // ---- multiplicities
  for (uint32_t l = 0; l < L; l++) {
    std::vector<uint32_t> cnt(table_rows, 0);
    const auto &ix = idx[lk[l].group];
    for (uint64_t i = 0; i < u; i++) cnt[ix[i]]++;       // rows a selector switches off read the all-zero tuple = table row 0, which idx holds there
    Column &mc = C->m[l];
    C->parallel(table_rows, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) mc[i] = fr_small(cnt[i]); });
    cnt.resize(n, 0); C->m_counts.push_back(std::move(cnt));
    Rng g(blind_stream()); for (uint64_t r = u + 1; r < n; r++) mc[r] = opt.zero_blinding ? fr_zero() : g.uniform();
  }
  // ---- whatever preprocessed polynomial nothing assigned stays all-zero (materialised), the permutation's bookkeeping, the prover's randomness
  for (uint32_t p = 0; p < P.num_pre; p++) { bool sig = false; for (const auto &c : C->pcols) sig = sig || c.sigma == p; if (!sig) (void)pre_col(p); }
  C->omega_pow.resize(n);
  { const int T = C->threads; std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back([&, t]() { const uint64_t lo = n * t / T, hi = n * (t + 1) / T; Fr w = fr_pow(P.omega, lo); for (uint64_t i = lo; i < hi; i++) { C->omega_pow[i] = w; w = fr_mul(w, P.omega); } }); for (auto &t : th) t.join(); }
  { Rng g(blind_stream()); C->z_blind.assign(P.perm.size(), std::vector<Fr>(P.blind)); for (auto &v : C->z_blind) for (auto &x : v) x = opt.zero_blinding ? fr_zero() : g.uniform(); C->phi_blind.assign(L, std::vector<Fr>(P.blind)); for (auto &v : C->phi_blind) for (auto &x : v) x = opt.zero_blinding ? fr_zero() : g.uniform(); }
  { C->random_poly = Column(alloc); C->random_poly.resize(n); const uint64_t sd = blind_stream();   /* the random polynomial stays random under zero_blinding (its commitment must not be the identity) */ C->parallel(n, [&](uint64_t lo, uint64_t hi) { Rng g(sd + lo * 7919); for (uint64_t i = lo; i < hi; i++) C->random_poly[i] = g.uniform(); }); }
  return C;
}

// the instance for the CPU restatement (oracle/plonk.py ProofInputs.load): raw little-endian Montgomery words, as the columns sit in memory
inline void dump_circuit(const Circuit &C, const std::string &dir, const Fr &tau, const std::string &protocol_path) {
  const Protocol &P = *C.pr;
  auto wr = [&](const std::string &name, const std::vector<const void *> &blocks, const std::vector<size_t> &bytes) {
    std::ofstream f(dir + "/" + name, std::ios::binary); if (!f) throw std::invalid_argument("cannot write " + dir + "/" + name);
    for (size_t i = 0; i < blocks.size(); i++) f.write(static_cast<const char *>(blocks[i]), (std::streamsize)bytes[i]);
  };
  std::vector<std::vector<Fr>> sig; std::vector<const void *> b; std::vector<size_t> s;
  for (uint32_t p = 0; p < P.num_pre; p++) {
    int j = -1; for (size_t t = 0; t < C.pcols.size(); t++) if (C.pcols[t].sigma == p) j = (int)t;
    if (j >= 0) { sig.emplace_back(); C.sigma_column((uint32_t)j, sig.back()); }
  }
  { size_t si = 0; for (uint32_t p = 0; p < P.num_pre; p++) { if (C.is_sigma(p)) b.push_back(sig[si++].data()); else b.push_back(C.pre[p].data()); s.push_back(P.n * 32); } }
  wr("pre.bin", b, s);
  wr("instance.bin", {C.instances.data()}, {C.instances.size() * 32});
  b.clear(); s.clear(); for (const auto &c : C.advice) { b.push_back(c.data()); s.push_back(P.n * 32); } wr("advice.bin", b, s);
  b.clear(); s.clear(); for (const auto &c : C.m) { b.push_back(c.data()); s.push_back(P.n * 32); } wr("m.bin", b, s);
  b.clear(); s.clear(); for (const auto &c : C.z_blind) { b.push_back(c.data()); s.push_back(c.size() * 32); } wr("z_blind.bin", b, s);
  b.clear(); s.clear(); for (const auto &c : C.phi_blind) { b.push_back(c.data()); s.push_back(c.size() * 32); } wr("phi_blind.bin", b, s);
  wr("random.bin", {C.random_poly.data()}, {P.n * 32});
  { std::ifstream in(protocol_path, std::ios::binary); std::ofstream out(dir + "/protocol.json", std::ios::binary); out << in.rdbuf(); }
  const Fr tc = fr_to_canonical(tau); char hex[65];
  std::snprintf(hex, sizeof hex, "%016llx%016llx%016llx%016llx", (unsigned long long)tc[3], (unsigned long long)tc[2], (unsigned long long)tc[1], (unsigned long long)tc[0]);
  std::ofstream man(dir + "/manifest.json");
  man << "{\"layer\": " << P.layer << ", \"k\": " << P.k << ", \"tau\": \"" << hex << "\", \"blind\": " << P.blind << ", \"num_instance\": " << C.instances.size()
      << ", \"copy_pairs\": " << C.pairs.size() << ", \"gates_active\": " << C.gates_active << ", \"lookup_rows\": " << C.lookup_rows << "}\n";
}

}  // namespace plonk
}  // namespace mi355zk
