// mi355zk_plonk_protocol.hpp -- the constraint system create_proof runs on, read from a snark-verifier `PlonkProtocol` JSON.
//
// What the reference holds: every proof the `prover` crate emits carries the PlonkProtocol of its circuit -- [REF release-v0.13.1/chunk.protocol]
// (layer 2, k = 25; == [REF integration/tests/test_data/chunk_chunk_0.protocol]) and the base64 `protocol` of
// [REF integration/tests/test_data/full_proof_batch_agg_1.json] (layer 4, k = 26).  `quotient.numerator` is the circuit's whole constraint system as an
// expression tree (Sum / Product / Negated / Scaled / Polynomial{poly, rotation} / Challenge / Constant / CommonPolynomial{Identity, Lagrange(i)} /
// DistributePowers), `queries` the opened (polynomial, rotation) pairs, `evaluations` the order of the evaluations in the proof, `num_witness` /
// `num_challenge` the commitment phases.  This header parses that file and RECOGNISES in the numerator what halo2 reads off its ConstraintSystem when
// it builds the argument polynomials [EXT-recalled halo2_proofs plonk/permutation/prover.rs, plonk/mv_lookup/prover.rs]:
//   permutation chunks   l_active (z(wX) prod_j (c_j + beta sigma_j + gamma) - z(X) prod_j (c_j + beta delta^j X + gamma))
//   lookups              l_active ((T + beta)(I + beta)(phi(wX) - phi(X)) - ((T + beta) - m (I + beta))),  T / I theta-compressed table / input
//   gates                everything else, as  selector * (P - target)
// The quotient itself is NOT built from the recognised structure: mi355zk_plonk.hpp compiles the tree as it stands.  Host-only code, no device calls.
// Polynomial numbering (snark-verifier): [preprocessed | instance | witness phase 0 (advice) | phase 1 (lookup m) | phase 2 (z.., phi.., random) | quotient].
#pragma once
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>

#include "mi355zk_halo2.hpp"

namespace mi355zk {
namespace plonk {

using halo2::Fr;
namespace h2d = halo2::detail;
inline Fr fr_add(const Fr &a, const Fr &b) { return h2d::from_fe(zk::Fr::add(h2d::to_fe(a), h2d::to_fe(b))); }
inline Fr fr_sub(const Fr &a, const Fr &b) { return h2d::from_fe(zk::Fr::sub(h2d::to_fe(a), h2d::to_fe(b))); }
inline Fr fr_neg(const Fr &a) { return h2d::from_fe(zk::Fr::neg(h2d::to_fe(a))); }
inline Fr fr_mul(const Fr &a, const Fr &b) { return h2d::fr_mul(a, b); }
inline Fr fr_inv(const Fr &a) { return h2d::fr_inv(a); }
inline Fr fr_pow(const Fr &a, uint64_t e) { return h2d::fr_pow(a, e); }
inline Fr fr_u64(uint64_t v) { return h2d::fr_from_u64(v); }
inline Fr fr_one() { return fr_u64(1); }
inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
inline bool fr_is_zero(const Fr &a) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
inline Fr fr_to_canonical(const Fr &a) { return h2d::from_fe(zk::Fr::to_canonical(h2d::to_fe(a))); }

// ------------------------------------------------------------------------------------------------ a small JSON reader (integers stay exact 64-bit)
namespace json {
struct Value {
  enum Type { NUL, BOOL, INT, STR, ARR, OBJ } type = NUL;
  bool b = false; bool negative = false; uint64_t u = 0; std::string s;
  std::vector<Value> arr; std::vector<std::pair<std::string, Value>> obj;
  int64_t i64() const { return negative ? -(int64_t)u : (int64_t)u; }
  const Value *find(const std::string &k) const { for (const auto &kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
  const Value &at(const std::string &k) const { const Value *v = find(k); if (!v) throw std::invalid_argument("json: missing key " + k); return *v; }
};
struct Parser {
  const std::string &t; size_t p = 0;
  explicit Parser(const std::string &text) : t(text) {}
  void ws() { while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\t' || t[p] == '\r')) p++; }
  [[noreturn]] void fail(const char *what) { throw std::invalid_argument(std::string("json: ") + what + " at offset " + std::to_string(p)); }
  Value parse() {
    ws(); if (p >= t.size()) fail("unexpected end");
    Value v; const char c = t[p];
    if (c == '{') {
      v.type = Value::OBJ; p++; ws();
      if (t[p] == '}') { p++; return v; }
      for (;;) { ws(); Value k = parse(); if (k.type != Value::STR) fail("object key"); ws(); if (t[p++] != ':') fail("':'"); v.obj.emplace_back(k.s, parse()); ws(); if (t[p] == ',') { p++; continue; } if (t[p] == '}') { p++; return v; } fail("',' or '}'"); }
    }
    if (c == '[') {
      v.type = Value::ARR; p++; ws();
      if (t[p] == ']') { p++; return v; }
      for (;;) { v.arr.push_back(parse()); ws(); if (t[p] == ',') { p++; continue; } if (t[p] == ']') { p++; return v; } fail("',' or ']'"); }
    }
    if (c == '"') { v.type = Value::STR; p++; while (p < t.size() && t[p] != '"') { if (t[p] == '\\') p++; v.s.push_back(t[p++]); } p++; return v; }
    if (c == 'n') { p += 4; return v; }
    if (c == 't') { v.type = Value::BOOL; v.b = true; p += 4; return v; }
    if (c == 'f') { v.type = Value::BOOL; p += 5; return v; }
    if (c == '-' || (c >= '0' && c <= '9')) {
      v.type = Value::INT; if (c == '-') { v.negative = true; p++; }
      while (p < t.size() && t[p] >= '0' && t[p] <= '9') v.u = v.u * 10 + (uint64_t)(t[p++] - '0');
      if (p < t.size() && (t[p] == '.' || t[p] == 'e' || t[p] == 'E')) fail("non-integer number");
      return v;
    }
    fail("unexpected character");
  }
};
inline Value parse_file(const std::string &path) {
  std::ifstream f(path); if (!f) throw std::invalid_argument("cannot open " + path);
  std::stringstream ss; ss << f.rdbuf(); const std::string text = ss.str();
  Parser P(text); return P.parse();
}
}  // namespace json

// ------------------------------------------------------------------------------------------------ the expression tree
struct Expr {
  enum Kind : uint8_t { CONSTANT, IDENTITY, LAGRANGE, POLY, CHALLENGE, NEG, SUM, PROD, SCALED, DPOW } kind = CONSTANT;
  Fr c{};                       // CONSTANT, SCALED (the scalar)
  int32_t i = 0, rot = 0;       // POLY: (i, rot); CHALLENGE / LAGRANGE: i
  std::vector<Expr> kids;       // NEG / SCALED: 1; SUM / PROD: 2; DPOW: the expressions, then the base
  bool is_poly(int32_t p, int32_t r) const { return kind == POLY && i == p && rot == r; }
  bool is_challenge(int32_t j) const { return kind == CHALLENGE && i == j; }
};
inline Fr fr_from_json(const json::Value &v) { if (v.type != json::Value::ARR || v.arr.size() != 4) throw std::invalid_argument("field element: four limbs expected"); return Fr{{v.arr[0].u, v.arr[1].u, v.arr[2].u, v.arr[3].u}}; }
inline Expr parse_expr(const json::Value &v) {
  if (v.type != json::Value::OBJ || v.obj.size() != 1) throw std::invalid_argument("expression: one-key object expected");
  const std::string &k = v.obj[0].first; const json::Value &a = v.obj[0].second;
  Expr e;
  if (k == "Constant") { e.kind = Expr::CONSTANT; e.c = fr_from_json(a); }
  else if (k == "CommonPolynomial") { if (a.type == json::Value::STR) e.kind = Expr::IDENTITY; else { e.kind = Expr::LAGRANGE; e.i = (int32_t)a.at("Lagrange").i64(); } }
  else if (k == "Polynomial") { e.kind = Expr::POLY; e.i = (int32_t)a.at("poly").i64(); e.rot = (int32_t)a.at("rotation").i64(); }
  else if (k == "Challenge") { e.kind = Expr::CHALLENGE; e.i = (int32_t)a.i64(); }
  else if (k == "Negated") { e.kind = Expr::NEG; e.kids.push_back(parse_expr(a)); }
  else if (k == "Sum" || k == "Product") { e.kind = k == "Sum" ? Expr::SUM : Expr::PROD; e.kids.push_back(parse_expr(a.arr.at(0))); e.kids.push_back(parse_expr(a.arr.at(1))); }
  else if (k == "Scaled") { e.kind = Expr::SCALED; e.kids.push_back(parse_expr(a.arr.at(0))); e.c = fr_from_json(a.arr.at(1)); }
  else if (k == "DistributePowers") { e.kind = Expr::DPOW; for (const auto &x : a.arr.at(0).arr) e.kids.push_back(parse_expr(x)); e.kids.push_back(parse_expr(a.arr.at(1))); if (e.kids.size() < 2) throw std::invalid_argument("DistributePowers: empty"); }
  else throw std::invalid_argument("expression: unknown node " + k);
  return e;
}
// structural key of a subtree (deduplicates the common polynomials)
inline void expr_key(const Expr &e, std::string &out) {
  char buf[96];
  switch (e.kind) {
    case Expr::CONSTANT: std::snprintf(buf, sizeof buf, "C%llx.%llx.%llx.%llx", (unsigned long long)e.c[0], (unsigned long long)e.c[1], (unsigned long long)e.c[2], (unsigned long long)e.c[3]); out += buf; return;
    case Expr::IDENTITY: out += "X"; return;
    case Expr::LAGRANGE: std::snprintf(buf, sizeof buf, "L%d", e.i); out += buf; return;
    case Expr::POLY: std::snprintf(buf, sizeof buf, "p%d@%d", e.i, e.rot); out += buf; return;
    case Expr::CHALLENGE: std::snprintf(buf, sizeof buf, "c%d", e.i); out += buf; return;
    default: break;
  }
  static const char *nm[] = {"", "", "", "", "", "N(", "S(", "P(", "K(", "D("};
  out += nm[e.kind];
  for (const auto &kd : e.kids) { expr_key(kd, out); out += ","; }
  out += ")";
}
// a subtree made of constants, CommonPolynomials, sums and negations only: ONE fixed polynomial of the proving key (l_0, l_last, l_active, X)
inline bool is_common_linear(const Expr &e, bool *has_common = nullptr) {
  switch (e.kind) {
    case Expr::CONSTANT: return true;
    case Expr::IDENTITY: case Expr::LAGRANGE: if (has_common) *has_common = true; return true;
    case Expr::NEG: return is_common_linear(e.kids[0], has_common);
    case Expr::SUM: return is_common_linear(e.kids[0], has_common) && is_common_linear(e.kids[1], has_common);
    default: return false;
  }
}
struct CommonLinear { Fr constant = fr_zero(), x_coeff = fr_zero(); std::map<int32_t, Fr> lagrange; };   // constant + x_coeff X + sum_i lagrange[i] L_i(X)
inline void common_linear_terms(const Expr &e, const Fr &scale, CommonLinear &out) {
  switch (e.kind) {
    case Expr::CONSTANT: out.constant = fr_add(out.constant, fr_mul(scale, e.c)); return;
    case Expr::IDENTITY: out.x_coeff = fr_add(out.x_coeff, scale); return;
    case Expr::LAGRANGE: { Fr &v = out.lagrange.emplace(e.i, fr_zero()).first->second; v = fr_add(v, scale); return; }
    case Expr::NEG: common_linear_terms(e.kids[0], fr_neg(scale), out); return;
    case Expr::SUM: common_linear_terms(e.kids[0], scale, out); common_linear_terms(e.kids[1], scale, out); return;
    default: throw std::invalid_argument("not a common-linear subtree");
  }
}
inline void collect_polys(const Expr &e, std::vector<std::pair<int32_t, int32_t>> &out) {
  if (e.kind == Expr::POLY) { out.emplace_back(e.i, e.rot); return; }
  for (const auto &kd : e.kids) collect_polys(kd, out);
}
inline void collect_lagrange(const Expr &e, std::vector<int32_t> &out) { if (e.kind == Expr::LAGRANGE) out.push_back(e.i); for (const auto &kd : e.kids) collect_lagrange(kd, out); }
inline void flatten_product(const Expr &e, std::vector<const Expr *> &out) { if (e.kind == Expr::PROD) { flatten_product(e.kids[0], out); flatten_product(e.kids[1], out); } else out.push_back(&e); }

// ------------------------------------------------------------------------------------------------ the protocol and the structure recognised in it
struct PolyRot { uint32_t poly; int32_t rot; bool operator<(const PolyRot &o) const { return poly != o.poly ? poly < o.poly : rot < o.rot; } bool operator==(const PolyRot &o) const { return poly == o.poly && rot == o.rot; } };
struct PermColumn { uint32_t column, sigma; Fr delta_pow; };
struct PermChunk { uint32_t z; std::vector<PermColumn> columns; };
struct Lookup { uint32_t phi, m; const Expr *table, *input; };          // table / input: the theta-compressed DistributePowers nodes (or a single expression)
struct Gate { const Expr *expr; uint32_t selector; const Expr *p; PolyRot target; bool assignable; };   // selector * (p - target) when assignable

struct Protocol {
  uint32_t k = 0; uint64_t n = 0; Fr omega{}, omega_inv{}, n_inv{};
  uint32_t num_pre = 0; std::vector<uint32_t> num_instance, num_witness, num_challenge;
  uint32_t inst0 = 0, wit0 = 0, quotient_poly = 0, random_poly = 0, Q = 0, extended_k = 0, lookup_bits = 0; int layer = -1;
  std::vector<uint32_t> phase0;
  std::vector<PolyRot> evaluations, queries;
  Expr numerator;
  int32_t last_rot = 0; uint32_t blind = 0; uint64_t usable = 0;   // l_last = Lagrange(last_rot); rows usable+1 .. n-1 are blinding rows
  std::vector<PermChunk> perm; std::vector<Lookup> lookups; std::vector<Gate> gates;
  std::string source;
  // the scalar a verifier of THIS protocol file absorbs first (snark-verifier's PlonkProtocol::transcript_initial_state = halo2's hash of the verifying key): the
  // reference's protocol files carry it [REF release-v0.13.1/chunk.protocol "transcript_initial_state"]; generated protocols do not (see vk_transcript_scalar)
  bool has_initial_state = false; Fr initial_state{};
  Protocol() = default;
  Protocol(const Protocol &) = delete;               // `perm` / `lookups` / `gates` point into `numerator`
  Protocol &operator=(const Protocol &) = delete;

  uint32_t num_sigma() const { uint32_t s = 0; for (const auto &c : perm) s += (uint32_t)c.columns.size(); return s; }
  uint32_t num_advice() const { return num_witness.at(0); }
  uint32_t commitments() const { uint32_t s = Q + 2; for (auto w : num_witness) s += w; return s; }
  bool is_pre(uint32_t p) const { return p < num_pre; }
  bool is_instance(uint32_t p) const { return p >= inst0 && p < wit0; }

  void load(const std::string &path) {
    const json::Value file = json::parse_file(path);
    const json::Value &root = file.find("protocol") ? file.at("protocol") : file;   // the golden fixtures wrap the protocol with their provenance
    const json::Value &d = root.at("domain");
    k = (uint32_t)d.at("k").u; n = uint64_t(1) << k;
    omega = fr_from_json(d.at("gen")); omega_inv = fr_from_json(d.at("gen_inv")); n_inv = fr_from_json(d.at("n_inv"));
    { const halo2::EvaluationDomain dom(2, k); if (!(dom.omega == omega) || !(dom.omega_inv == omega_inv) || !(dom.ifft_divisor == n_inv)) throw std::invalid_argument("protocol: domain constants differ from EvaluationDomain::new"); }
    if (const json::Value *np = root.find("num_preprocessed")) num_pre = (uint32_t)np->u; else num_pre = (uint32_t)root.at("preprocessed").arr.size();
    for (const auto &v : root.at("num_instance").arr) num_instance.push_back((uint32_t)v.u);
    for (const auto &v : root.at("num_witness").arr) num_witness.push_back((uint32_t)v.u);
    for (const auto &v : root.at("num_challenge").arr) num_challenge.push_back((uint32_t)v.u);
    if (num_instance.size() != 1 || num_witness.size() != 3 || num_challenge.size() != 3 || num_challenge[0] != 1 || num_challenge[1] != 2 || num_challenge[2] != 1)
      throw std::invalid_argument("protocol: expected one instance column and the phases [advice | m | z, phi, random] with challenges [theta | beta, gamma | y]");
    inst0 = num_pre; wit0 = inst0 + (uint32_t)num_instance.size();
    uint32_t off = wit0; for (auto w : num_witness) { phase0.push_back(off); off += w; }
    quotient_poly = off; random_poly = off - 1;
    for (const auto &v : root.at("evaluations").arr) evaluations.push_back({(uint32_t)v.at("poly").u, (int32_t)v.at("rotation").i64()});
    for (const auto &v : root.at("queries").arr) queries.push_back({(uint32_t)v.at("poly").u, (int32_t)v.at("rotation").i64()});
    const json::Value &q = root.at("quotient");
    Q = (uint32_t)q.at("num_chunk").u;
    if (Q < 2 || (Q & (Q - 1)) || Q > 8) throw std::invalid_argument("protocol: quotient.num_chunk must be 2, 4 or 8");
    extended_k = k; while ((1u << (extended_k - k)) < Q) extended_k++;
    numerator = parse_expr(q.at("numerator"));
    if (const json::Value *lb = root.find("lookup_bits")) lookup_bits = (uint32_t)lb->u;
    if (const json::Value *ly = root.find("layer")) layer = (int)ly->i64();
    if (const json::Value *sc = root.find("source")) source = sc->s;
    if (const json::Value *is = root.find("transcript_initial_state")) if (is->type == json::Value::ARR) { initial_state = fr_from_json(*is); has_initial_state = true; }   // JSON null (Option::None) = absent
    recognise();
  }

 private:
  void recognise() {
    if (numerator.kind != Expr::DPOW || !numerator.kids.back().is_challenge(3)) throw std::invalid_argument("protocol: numerator must be DistributePowers(constraints, y)");
    std::vector<int32_t> lag; collect_lagrange(numerator, lag);
    last_rot = 0; for (auto l : lag) last_rot = std::min(last_rot, l);
    if (last_rot >= -1) throw std::invalid_argument("protocol: no l_last in the numerator");
    blind = (uint32_t)(-last_rot - 1); usable = n - (uint64_t)(-last_rot);
    for (size_t ci = 0; ci + 1 < numerator.kids.size(); ci++) {
      const Expr &c = numerator.kids[ci];
      if (c.kind != Expr::PROD) throw std::invalid_argument("protocol: constraint " + std::to_string(ci) + " is not a product");
      const Expr &a = c.kids[0], &b = c.kids[1];
      if (a.kind == Expr::LAGRANGE) continue;   // l_0 / l_last boundary constraints: implied by how z and phi are built
      bool common = false;
      if (is_common_linear(a, &common) && common) {
        if (b.kind != Expr::SUM || b.kids[0].kind != Expr::PROD || b.kids[1].kind != Expr::NEG) throw std::invalid_argument("protocol: unrecognised l_active constraint");
        const Expr &l0 = b.kids[0].kids[0], &l1 = b.kids[0].kids[1], &neg = b.kids[1].kids[0];
        if (l0.kind == Expr::POLY && l0.rot == 1) {           // permutation chunk
          PermChunk ch; ch.z = (uint32_t)l0.i;
          if (neg.kind != Expr::PROD || !neg.kids[0].is_poly(l0.i, 0)) throw std::invalid_argument("protocol: permutation chunk: z(X) side");
          std::vector<const Expr *> fs, fi; flatten_product(l1, fs); flatten_product(neg.kids[1], fi);
          if (fs.size() != fi.size()) throw std::invalid_argument("protocol: permutation chunk: factor counts differ");
          for (size_t j = 0; j < fs.size(); j++) {
            const Expr &s = *fs[j], &d = *fi[j];   // Sum(Sum(c, Product(beta, sigma)), gamma)   Sum(Sum(c, Product(Product(beta, delta^j), X)), gamma)
            auto bad = [&]() { throw std::invalid_argument("protocol: permutation factor " + std::to_string(j) + " has an unexpected shape"); };
            if (s.kind != Expr::SUM || d.kind != Expr::SUM || !s.kids[1].is_challenge(2) || !d.kids[1].is_challenge(2) || s.kids[0].kind != Expr::SUM || d.kids[0].kind != Expr::SUM) bad();
            const Expr &cs = s.kids[0].kids[0], &bs = s.kids[0].kids[1], &cd = d.kids[0].kids[0], &bd = d.kids[0].kids[1];
            if (cs.kind != Expr::POLY || cs.rot != 0 || !cd.is_poly(cs.i, 0) || bs.kind != Expr::PROD || !bs.kids[0].is_challenge(1) || bs.kids[1].kind != Expr::POLY || bs.kids[1].rot != 0) bad();
            if (bd.kind != Expr::PROD || bd.kids[1].kind != Expr::IDENTITY || bd.kids[0].kind != Expr::PROD || !bd.kids[0].kids[0].is_challenge(1) || bd.kids[0].kids[1].kind != Expr::CONSTANT) bad();
            ch.columns.push_back({(uint32_t)cs.i, (uint32_t)bs.kids[1].i, bd.kids[0].kids[1].c});
          }
          perm.push_back(ch);
        } else {                                              // lookup
          auto bad = [&]() { throw std::invalid_argument("protocol: unrecognised lookup constraint"); };
          if (l0.kind != Expr::PROD || l1.kind != Expr::SUM || l1.kids[0].kind != Expr::POLY || l1.kids[0].rot != 1 || l1.kids[1].kind != Expr::NEG || !l1.kids[1].kids[0].is_poly(l1.kids[0].i, 0)) bad();
          const Expr &tb = l0.kids[0], &ib = l0.kids[1];
          if (tb.kind != Expr::SUM || ib.kind != Expr::SUM || !tb.kids[1].is_challenge(1) || !ib.kids[1].is_challenge(1)) bad();
          if (neg.kind != Expr::SUM || neg.kids[1].kind != Expr::NEG || neg.kids[1].kids[0].kind != Expr::PROD || neg.kids[1].kids[0].kids[0].kind != Expr::POLY) bad();
          std::string k1, k2, k3, k4; expr_key(tb, k1); expr_key(neg.kids[0], k2); expr_key(ib, k3); expr_key(neg.kids[1].kids[0].kids[1], k4);
          if (k1 != k2 || k3 != k4) bad();
          lookups.push_back({(uint32_t)l1.kids[0].i, (uint32_t)neg.kids[1].kids[0].kids[0].i, &tb.kids[0], &ib.kids[0]});
        }
        continue;
      }
      Gate g{&c, 0, nullptr, {0, 0}, false};                  // a custom gate; selector * (P - target) is what the witness builder can satisfy by assignment
      if (a.kind == Expr::POLY && a.rot == 0 && is_pre((uint32_t)a.i) && b.kind == Expr::SUM && b.kids[1].kind == Expr::NEG && b.kids[1].kids[0].kind == Expr::POLY && (uint32_t)b.kids[1].kids[0].i >= wit0) {
        g.selector = (uint32_t)a.i; g.p = &b.kids[0]; g.target = {(uint32_t)b.kids[1].kids[0].i, b.kids[1].kids[0].rot}; g.assignable = true;
      }
      gates.push_back(g);
    }
    if (perm.empty()) throw std::invalid_argument("protocol: no permutation argument recognised");
    if (num_witness[1] != lookups.size() || num_witness[2] != perm.size() + lookups.size() + 1) throw std::invalid_argument("protocol: num_witness does not match the recognised arguments");
    for (size_t c = 0; c < perm.size(); c++) if (perm[c].z != phase0[2] + c) throw std::invalid_argument("protocol: grand products out of order");
    for (size_t l = 0; l < lookups.size(); l++) if (lookups[l].phi != phase0[2] + perm.size() + l || lookups[l].m != phase0[1] + l) throw std::invalid_argument("protocol: lookup polynomials out of order");
  }
};

// SHPLONK's rotation sets [EXT-recalled halo2_proofs poly/kzg/multiopen/shplonk.rs construct_intermediate_sets]: polynomials opened at the same SET of rotations
// share one set.  Order: first appearance in `queries`, for the sets and inside a set (snark-verifier's query_sets; the order under which all of the reference's stored proofs verify).
struct RotationSet { std::vector<int32_t> rots; std::vector<uint32_t> polys; };
inline std::vector<RotationSet> rotation_sets(const std::vector<PolyRot> &queries) {
  std::vector<uint32_t> order; std::map<uint32_t, std::vector<int32_t>> rots;
  for (const auto &q : queries) { auto it = rots.find(q.poly); if (it == rots.end()) { it = rots.emplace(q.poly, std::vector<int32_t>{}).first; order.push_back(q.poly); } if (std::find(it->second.begin(), it->second.end(), q.rot) == it->second.end()) it->second.push_back(q.rot); }
  std::vector<RotationSet> sets;
  for (uint32_t p : order) {
    std::vector<int32_t> key = rots[p]; std::sort(key.begin(), key.end());
    bool placed = false;
    for (auto &s : sets) { std::vector<int32_t> sk = s.rots; std::sort(sk.begin(), sk.end()); if (sk == key) { s.polys.push_back(p); placed = true; break; } }
    if (!placed) sets.push_back({rots[p], {p}});
  }
  return sets;
}

}  // namespace plonk
}  // namespace mi355zk
