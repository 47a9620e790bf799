// test_create_proof_replay.cpp -- a COMPILED caller that runs halo2_proofs::plonk::create_proof's GPU side for one layer of scroll-prover's
// proof stack through the C-ABI and then checks everything it produced.  The flow itself is mi355zk::halo2::create_proof_gpu_side
// (include/mi355zk_create_proof.hpp: steps 1-10 of SURVEY.md 3.2 over resident polynomials, the proving key's cosets resident in HBM); this
// program is the set-up around it (synthetic SRS with a known tau, keygen, witness synthesis, the HBM budget) and the checker after it.
// The reference reaches create_proof once per layer [REF integration/src/prove.rs:36-43,67,95-97]; Rust is absent from this image.
//
//   layers 0-6: counts from mi355zk::halo2::layer_shape (configs [REF integration/configs/layer1.config:3-10] ... layer6.config, fixtures
//   test_data/full_proof_1.json and full_proof_batch_agg_1.json; layer 0 is a stated guess, overridable: --advice / --fixed / --lookups / --perm / --chunk / --degree)
//
// Checks, all AFTER the clock stops, all with the CPU oracle (oracle/, TEST INFRASTRUCTURE) on downloaded data:
//   (1) every commitment == p(tau) G for the polynomial it commits to (Horner over the downloaded coefficients + one scalar multiple);
//   (2) every evaluation == Horner over the downloaded coefficients;
//   (3) SEMANTICS of the quotient: at the challenge x,  h(x) (x^n - 1) == sum_g y^g gate_g(x), the right-hand side recomputed from the
//       EVALUATIONS alone through the same expression plan (custom gates, permutation argument with its (c + beta sigma + gamma) products,
//       log-derivative lookups; l_active / l_0 / instance evaluated by the checker as a verifier would) -- the witness satisfies the constraints,
//       so this fails if any coset transform, gate launch, the part interleave, the 2^(k + e) inverse or the division by the vanishing
//       polynomial is wrong;
//   (4) both multi-open quotients with the trapdoor, G1 operations only:  commit(q_j) (tau - z_j) == commit(lin) - lin(z_j) G.
// --host-api additionally replays the MSM / NTT / evaluation counts through the host-pointer entry points (what a shim without DevicePoly pays).
// Prints one JSON line; exit code 0 = all checks passed, 2 = no GPU (mi355_init failed).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mi355zk_halo2.hpp"

extern "C" {   // oracle (checker only)
void orc_eval_polynomial_mt(void *out, const void *poly, uint64_t n, const void *point, int threads);
void orc_g1_mul(void *out_jac, const void *p_affine, const void *scalar_mont);
void orc_g1_generator(void *out_affine);
void orc_g1_to_affine(void *o, const void *p);
void orc_g1_add_affine(void *o_jac, const void *p_jac, const void *q_affine);
void orc_f_add(int w, void *o, const void *a, const void *b);
void orc_f_sub(int w, void *o, const void *a, const void *b);
void orc_f_mul(int w, void *o, const void *a, const void *b);
void orc_f_neg(int w, void *o, const void *a);
void orc_f_pow(int w, void *o, const void *a, const uint64_t *e);
}

using namespace mi355zk::halo2;
namespace h2d = mi355zk::halo2::detail;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

static int failures = 0;
#define CK(expr) do { int _rc = (expr); if (_rc != MI355_OK) { std::printf("FAILED %s:%d  %s -> %d [%s]\n", __FILE__, __LINE__, #expr, _rc, mi355_last_error()); failures++; } } while (0)
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

// oracle field arithmetic (Fr = selector 1)
static Fr o_add(const Fr &a, const Fr &b) { Fr r; orc_f_add(1, r.data(), a.data(), b.data()); return r; }
static Fr o_sub(const Fr &a, const Fr &b) { Fr r; orc_f_sub(1, r.data(), a.data(), b.data()); return r; }
static Fr o_mul(const Fr &a, const Fr &b) { Fr r; orc_f_mul(1, r.data(), a.data(), b.data()); return r; }
static Fr o_neg(const Fr &a) { Fr r; orc_f_neg(1, r.data(), a.data()); return r; }
static Fr o_pow(const Fr &a, uint64_t e) { uint64_t ee[4] = {e, 0, 0, 0}; Fr r; orc_f_pow(1, r.data(), a.data(), ee); return r; }
static Fr o_eval(const std::vector<Fr> &c, const Fr &x, int threads) { Fr r; orc_eval_polynomial_mt(r.data(), c.data(), c.size(), x.data(), threads); return r; }

static G1Affine field_commit(const std::vector<Fr> &coeffs, const Fr &tau, int threads) {
  const Fr ev = o_eval(coeffs, tau, threads);
  G1Affine gen, out; G1 j; orc_g1_generator(gen.data()); orc_g1_mul(j.data(), gen.data(), ev.data()); orc_g1_to_affine(out.data(), j.data());
  return out;
}
static bool is_identity(const G1 &g) { for (auto w : g) if (w) return false; return true; }
static bool commit_matches(const G1 &got, const G1Affine &want) {
  bool ident = true; for (auto w : want) ident = ident && w == 0;
  if (ident) return is_identity(got);
  return std::memcmp(got.data(), want.data(), 64) == 0;   // normalised Jacobian: (x, y, R)
}
static std::vector<Fr> download(const void *p, uint64_t n) { std::vector<Fr> v(n); CK(mi355_buf_download(v.data(), p, n * 32)); return v; }
static Fr rot_point(const EvaluationDomain &dom, const Fr &x, int32_t rot) {
  if (rot > 0) return o_mul(x, o_pow(dom.omega, (uint64_t)rot));
  if (rot < 0) return o_mul(x, o_pow(dom.omega_inv, (uint64_t)(-(int64_t)rot)));
  return x;
}
static const char *kind_name(PolyKind k) { static const char *n[] = {"instance", "advice", "m", "z", "phi", "fixed", "sigma", "id", "l_active", "l_0", "tmp", "h/shplonk"}; return n[k]; }

int main(int argc, char **argv) {
  int layer_id = 4, devices = 1, threads = (int)std::thread::hardware_concurrency(), proofs = 2; long k_override = -1; bool host_api = false, do_check = true;
  std::string tables = "auto", pk_mode = "auto"; int upload_threads = 1, early_intt = -1; bool pinned_witness = false;
  long o_advice = -1, o_fixed = -1, o_lookups = -1, o_perm = -1, o_chunk = -1, o_degree = -1;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> long { return i + 1 < argc ? std::atol(argv[++i]) : 0; };
    auto nexts = [&]() -> std::string { return i + 1 < argc ? std::string(argv[++i]) : std::string(); };
    if (a == "--layer") layer_id = (int)next(); else if (a == "--k") k_override = next(); else if (a == "--devices") devices = (int)next();
    else if (a == "--threads") threads = (int)next(); else if (a == "--host-api") host_api = true; else if (a == "--no-check") do_check = false;
    else if (a == "--no-tables") tables = "off"; else if (a == "--tables") tables = nexts(); else if (a == "--pk-cosets") pk_mode = nexts(); else if (a == "--proofs") proofs = (int)next();
    else if (a == "--upload-threads") upload_threads = (int)next(); else if (a == "--early-intt") early_intt = (int)next(); else if (a == "--pinned-witness") pinned_witness = true;
    else if (a == "--advice") o_advice = next(); else if (a == "--fixed") o_fixed = next(); else if (a == "--lookups") o_lookups = next();
    else if (a == "--perm") o_perm = next(); else if (a == "--chunk") o_chunk = next(); else if (a == "--degree") o_degree = next();
    else { std::printf("usage: %s [--layer 0..6] [--k K] [--advice A --fixed F --lookups L --perm P --chunk C --degree D] [--devices D] [--threads T] [--proofs N] [--upload-threads U] [--early-intt 0|1] [--pinned-witness]\n"
                       "          [--tables auto|on|lagrange|off] [--pk-cosets auto|resident|on-the-fly] [--host-api] [--no-check]\n", argv[0]); return 1; }
  }
  if (threads < 1) threads = 1; if (threads > 16) threads = 16;   // the GPU box's container gets 16 of the host's CPUs
  const char *tq = std::getenv("MI355_REPLAY_THREADS"); if (tq) threads = std::max(1, std::atoi(tq));
  if (proofs < 1) proofs = 1;
  CircuitShape S;
  try { S = layer_shape(layer_id); } catch (const std::exception &e) { std::printf("%s\n", e.what()); return 1; }
  if (k_override > 0) S.k = (uint32_t)k_override;
  if (o_advice > 0) S.advice = (uint32_t)o_advice; if (o_fixed > 0) S.fixed = (uint32_t)o_fixed; if (o_lookups >= 0) S.lookups = (uint32_t)o_lookups;
  if (o_perm > 0) S.perm_columns = (uint32_t)o_perm; if (o_chunk > 0) S.chunk_len = (uint32_t)o_chunk; if (o_degree > 0) S.degree = (uint32_t)o_degree;
  if (S.chunk_len > 6 || S.chunk_len + 2 > S.degree || ((S.degree - 1) & (S.degree - 2)) || S.degree < 4 || S.degree > 9) { std::printf("unsupported shape: degree - 1 must be a power of two in 4..8, chunk_len <= min(6, degree - 2)\n"); return 1; }
  const uint32_t k = S.k, Q = S.Q(); const uint64_t n = uint64_t(1) << k;
  {
    std::vector<int> ids(devices); for (int d = 0; d < devices; d++) ids[d] = d;
    if (std::getenv("MI355_ALLOW_DUP_DEVICES")) for (auto &d : ids) d = 0;
    const int rc = devices == 1 ? mi355_init(0) : mi355_init_multi(ids.data(), devices);
    if (rc != MI355_OK) { std::printf("mi355_init failed (%d): %s\n", rc, mi355_last_error()); return 2; }
  }
  int rc_main = 0;
  try {
  const EvaluationDomain dom(S.degree, k);   // quotient degree = degree - 1 = Q: extended_k = k + log2 Q
  const Fr tau = h2d::fr_from_u64(0x5343524F4C4C0001ull + layer_id);
  // ---- ParamsKZG (set-up time): synthetic SRS on the device, both bases registered
  uint64_t hg = 0, hl = 0;
  {
    DevicePoly g(2 * n, 0), gl(2 * n, 0);   // 64 bytes per point
    check(mi355_srs_setup_dev(g.p, gl.p, k, tau.data(), dom.omega.data()));
    check(mi355_srs_register_dev(g.p, n, 1, &hg)); check(mi355_srs_register_dev(gl.p, n, 1, &hl));
    check(mi355_synchronize());
  }
  check(mi355_buf_trim());   // the two staging blocks of the set-up go back to HIP
  // ---- HBM budget: what must live in HBM for this layer's prover, and which optional residents fit on top
  uint64_t hbm_free = 0, hbm_total = 0; check(mi355_mem_info(0, &hbm_free, &hbm_total, nullptr, nullptr, nullptr));
  const double per = (double)n * 32, GiB = 1024.0 * 1024 * 1024;
  const uint32_t npk = S.fixed + S.perm_columns + 3, NPW = S.witness_polys();
  const double pk_base = per * (S.fixed * 2 + S.perm_columns * 2 + 4), pk_cosets = per * npk * Q, pk_lean_tmp = per * npk;
  // a proof's own blocks: coefficients + one part of every witness polynomial, the permutation temporaries, h as parts and as one vector, the opened
  // combination and its two quotients (the step-4 temporaries are pooled and reused by the parts); the NTT scratch of the 2^(k + e) inverse;
  // the MSM workspace (digit plane, two record streams, sorted stream: ~22 B per entry, up to 13 windows); fixed overheads
  const double working = per * (2.0 * NPW + 2 * S.chunk_len + 2 * Q + 3) + per * Q + per + (double)n * 13 * 22 + 0.5 * GiB;
  const double table_one = (double)n * 64 * (k >= 24 ? 12 : 15);
  const double usable = 0.94 * (double)hbm_free;   // what is free after the SRS registration, minus allocator slack
  bool resident = true; int n_tables = 0;
  if (pk_mode == "on-the-fly") resident = false;
  else if (pk_mode == "auto" && pk_base + pk_cosets + working > usable) resident = false;
  // ---- keygen (device side) and the witness (host side, as create_proof receives it)
  ProvingKeyDevice pk = keygen_device(S, dom, 0xC0FFEE + layer_id, resident, devices, threads);
  Witness wit = synthesize_witness(S, dom, pk, 9000 + layer_id, threads, pinned_witness);
  check(mi355_buf_trim());
  // window tables (W x a basis) only where they fit next to the proving key that is now resident and the working set of a proof: the measured
  // free memory decides, the Lagrange basis first (commit_lagrange carries most commitments); otherwise the table-free schedule (+ ~8 % per MSM)
  uint64_t free_after_pk = 0; check(mi355_mem_info(0, &free_after_pk, nullptr, nullptr, nullptr, nullptr));
  const double room = 0.97 * (double)free_after_pk - working - (resident ? 0.0 : pk_lean_tmp);
  if (tables == "on") n_tables = 2; else if (tables == "lagrange") n_tables = 1; else if (tables == "off") n_tables = 0;
  else { const double margin = 0.08 * (double)hbm_total; n_tables = room >= 2 * table_one + margin ? 2 : room >= table_one + margin ? 1 : 0; }   // 8 % of the device stays unplanned: a table is worth 5 ms per commitment, an out-of-memory proof is worth nothing
  if (devices > 1 && tables == "auto") n_tables = 2;   // the estimate above is for one device; shards divide everything
  if (n_tables >= 1 && mi355_srs_precompute(hl, 0, 0) != MI355_OK) { std::printf("window tables for g_lagrange did not fit (%s): table-free schedule\n", mi355_last_error()); n_tables = 0; }
  if (n_tables >= 2 && mi355_srs_precompute(hg, 0, 0) != MI355_OK) { std::printf("window tables for g did not fit (%s): Lagrange basis only\n", mi355_last_error()); n_tables = 1; }
  Challenges ch;
  ch.theta = h2d::fr_from_u64(0x7468657461ull); ch.beta = h2d::fr_from_u64(0xBE7A0000BE7A0001ull); ch.gamma = h2d::fr_from_u64(0x6A6D6D6100000003ull);
  ch.y = h2d::fr_from_u64(0x7900000000000005ull); ch.x = h2d::fr_from_u64(0x1234567890ABCDEFull); ch.v = h2d::fr_from_u64(0xABCDEF0123456789ull);
  ch.z0 = h2d::fr_from_u64(0x1111); ch.z1 = h2d::fr_from_u64(0x1112);
  const ExpressionPlan plan = build_plan(S, ch);
  ProofOptions opt; opt.devices = devices; opt.threads = threads; opt.upload_threads = upload_threads; opt.early_intt = early_intt;
  // ---- the proofs: a prover process runs proof after proof; the first one grows the workspace arena and the buffer pool, the last one is reported
  ProofGpuSide R; double first_ms = 0;
  for (int it = 0; it < proofs; it++) {
    R = ProofGpuSide();   // the previous proof's polynomials go back to the pool
    R = create_proof_gpu_side(hg, hl, dom, pk, plan, wit, ch, opt);
    if (it == 0) first_ms = R.total_ms;
  }
  uint64_t live = 0, pooled = 0, ws = 0, fr_end = 0; check(mi355_mem_info(0, &fr_end, nullptr, &live, &pooled, &ws));

  // ---- checks (outside the timing)
  uint32_t checked = 0; bool semantic_ok = false, trapdoor_ok = false;
  if (do_check && failures == 0) {
    // (1) commitments
    std::map<PolyRef, std::vector<Fr>> host_coeff;
    auto coeffs_of = [&](const PolyRef &r) -> const std::vector<Fr> & {
      auto it = host_coeff.find(r);
      if (it == host_coeff.end()) it = host_coeff.emplace(r, download((r.kind >= P_FIXED && r.kind <= P_L0) ? pk.coeff(r).p : R.coeff.at(r).p, n)).first;
      return it->second;
    };
    std::vector<Fr> scratch;
    for (size_t c = 0; c < R.commitments.size(); c++) {
      const CommitRecord &cr = R.commitments[c];
      const std::vector<Fr> *cf;
      if (cr.p.kind != P_KINDS) cf = &coeffs_of(cr.p);
      else if (cr.piece >= 0) { scratch = download(R.h.at((uint64_t)cr.piece * n), n); cf = &scratch; }
      else { scratch = download(R.quot[-1 - cr.piece].p, n); cf = &scratch; }
      const bool ok = commit_matches(cr.c, field_commit(*cf, tau, threads));
      if (!ok) std::printf("commitment %zu (%s %u, piece %d) != p(tau) G\n", c, kind_name(cr.p.kind), cr.p.idx, cr.piece);
      EXPECT(ok); checked++;
      if (cr.p.kind != P_KINDS && S.k >= 24 && S.advice + S.perm_z() > 24) host_coeff.erase(cr.p);   // wide AND big: do not keep every column on the host
    }
    // (2) evaluations
    std::map<Query, Fr> ev;
    for (size_t e = 0; e < plan.queries.size(); e++) {
      const Query &qr = plan.queries[e];
      const Fr want = o_eval(coeffs_of(qr.p), rot_point(dom, ch.x, qr.rot), threads);
      const bool ok = want == R.evals[e];
      if (!ok) std::printf("evaluation %zu (%s %u, rotation %d) differs from Horner\n", e, kind_name(qr.p.kind), qr.p.idx, qr.rot);
      EXPECT(ok); checked++; ev[qr] = R.evals[e];
    }
    Fr hx = h2d::fr_zero();
    {
      const Fr xn = o_pow(ch.x, n); Fr xq = h2d::fr_one();
      for (uint32_t q = 0; q < Q; q++) {
        scratch = download(R.h.at((uint64_t)q * n), n);
        const Fr want = o_eval(scratch, ch.x, threads);
        EXPECT(want == R.evals[plan.queries.size() + q]); checked++;
        hx = o_add(hx, o_mul(xq, R.evals[plan.queries.size() + q])); xq = o_mul(xq, xn);
      }
    }
    // (3) h(x) (x^n - 1) == sum_g y^g gate_g(x), from the evaluations through the plan
    {
      const std::vector<Fr> la = download(pk.l_active_coeff.p, n), l0 = download(pk.l0_coeff.p, n), inst = download(R.coeff.at({P_INSTANCE, 0}).p, n);
      std::map<uint32_t, Fr> tmp;
      auto value = [&](const Factor &f) -> Fr {
        switch (f.p.kind) {
          case P_TMP: return tmp.at(f.p.idx);
          case P_ID: return rot_point(dom, ch.x, f.rot);
          case P_LACTIVE: return o_eval(la, rot_point(dom, ch.x, f.rot), threads);
          case P_L0: return o_eval(l0, rot_point(dom, ch.x, f.rot), threads);
          case P_INSTANCE: return o_eval(inst, rot_point(dom, ch.x, f.rot), threads);
          default: return ev.at({f.p, f.rot});
        }
      };
      Fr N = h2d::fr_zero();
      for (const auto &L : plan.quotient) {
        Fr acc = h2d::fr_zero();
        for (const auto &t : L.terms) { Fr v = t.coeff; for (const auto &f : t.f) v = o_mul(v, value(f)); acc = o_add(acc, v); }
        if (L.to_tmp) tmp[L.tmp] = acc; else N = o_add(N, acc);
      }
      const Fr t_x = o_sub(o_pow(ch.x, n), h2d::fr_one());
      semantic_ok = o_mul(hx, t_x) == N;
      if (!semantic_ok) std::printf("SEMANTIC CHECK FAILED: h(x) (x^n - 1) != sum_g y^g gate_g(x)\n");
      EXPECT(semantic_ok); checked++;
    }
    // (4) the multi-open quotients with the trapdoor, in the group
    {
      G1 c_lin; CK(mi355_msm_g1_dev(hg, 0, R.lin.p, n, c_lin.data()));
      const std::vector<Fr> lin = download(R.lin.p, n);
      EXPECT(commit_matches(c_lin, field_commit(lin, tau, threads)));
      G1Affine gen; orc_g1_generator(gen.data());
      trapdoor_ok = true;
      for (int j = 0; j < 2; j++) {
        const Fr z = j ? ch.z1 : ch.z0;
        Fr lz; CK(mi355_eval_polynomial_dev(R.lin.p, n, z.data(), lz.data()));
        EXPECT(lz == o_eval(lin, z, threads));
        const G1 &cq = R.commitments[R.commitments.size() - 2 + j].c;
        G1Affine lhs{}, rhs{};
        if (!is_identity(cq)) { G1Affine cqa; std::memcpy(cqa.data(), cq.data(), 64); G1 t; const Fr d = o_sub(tau, z); orc_g1_mul(t.data(), cqa.data(), d.data()); orc_g1_to_affine(lhs.data(), t.data()); }
        { G1 t; const Fr m = o_neg(lz); orc_g1_mul(t.data(), gen.data(), m.data()); if (!is_identity(c_lin)) { G1Affine cl; std::memcpy(cl.data(), c_lin.data(), 64); orc_g1_add_affine(t.data(), t.data(), cl.data()); } orc_g1_to_affine(rhs.data(), t.data()); }
        const bool ok = lhs == rhs;
        if (!ok) std::printf("TRAPDOOR CHECK FAILED for multi-open quotient %d\n", j);
        trapdoor_ok = trapdoor_ok && ok; EXPECT(ok); checked++;
      }
    }
  }
  // ---- the same MSM / NTT / evaluation counts through the host-pointer entry points (what a shim without DevicePoly pays)
  double host_ms = -1, host_fft_ms = -1, host_fft_batched_ms = -1;
  const uint32_t n_commit_lag = S.advice + 2 * S.lookups + S.perm_z(), n_commit_coef = Q + 2;
  if (host_api && failures == 0) {
    std::vector<Fr> hp(wit.advice[0].begin(), wit.advice[0].end()); std::vector<Fr> hext(Q * n); G1 out; std::vector<std::vector<Fr>> cols(std::min<uint32_t>(8, NPW), hp);
    const uint32_t n_intt = NPW, n_ntt = NPW * Q, n_ev = (uint32_t)plan.queries.size() + Q;
    const auto t1 = Clock::now();
    for (uint32_t i = 0; i < n_commit_lag; i++) CK(mi355_msm_g1_host(hl, 0, hp.data(), n, out.data()));
    for (uint32_t i = 0; i < n_commit_coef; i++) CK(mi355_msm_g1_host(hg, 0, hp.data(), n, out.data()));
    const auto t2 = Clock::now();
    for (uint32_t i = 0; i < n_intt; i++) CK(mi355_intt_fr_host(hp.data(), k, dom.omega_inv.data(), dom.ifft_divisor.data()));
    for (uint32_t i = 0; i < n_ntt; i++) CK(mi355_ntt_fr_host(hp.data(), k, dom.omega.data()));
    host_fft_ms = ms_since(t2);
    CK(mi355_extended_to_coeff_host(hext.data(), dom.extended_k, dom.g_coset.data(), dom.g_coset_inv.data(), dom.extended_omega_inv.data(), dom.extended_ifft_divisor.data()));
    Fr e; for (uint32_t i = 0; i < n_ev; i++) CK(mi355_eval_polynomial_host(hp.data(), n, tau.data(), e.data()));
    host_ms = ms_since(t1);
    const uint32_t B = (uint32_t)cols.size();
    std::vector<void *> ptrs(B); for (uint32_t i = 0; i < B; i++) ptrs[i] = cols[i].data();
    const auto t3 = Clock::now();
    for (uint32_t done = 0; done < n_intt; done += B) CK(mi355_ntt_fr_batch_host(ptrs.data(), std::min(B, n_intt - done), k, dom.omega_inv.data(), dom.ifft_divisor.data()));
    for (uint32_t done = 0; done < n_ntt; done += B) CK(mi355_ntt_fr_batch_host(ptrs.data(), std::min(B, n_ntt - done), k, dom.omega.data(), nullptr));
    host_fft_batched_ms = ms_since(t3);
  }
  std::printf("{\"replay\": \"create_proof_gpu_side (include/mi355zk_create_proof.hpp): steps 1-10 through the C-ABI, polynomials and proving-key cosets resident\", \"layer\": %d, \"k\": %u, \"devices\": %d, "
              "\"shape\": {\"advice\": %u, \"fixed\": %u, \"lookups\": %u, \"perm_columns\": %u, \"chunk_len\": %u, \"perm_z\": %u, \"degree\": %u, \"quotient_pieces\": %u, \"source\": \"%s\"}, "
              "\"window_tables\": %s, \"window_table_bases\": %d, \"pk_cosets\": \"%s\", "
              "\"upload_threads\": %d, \"pinned_witness\": %s, \"msm\": %zu, \"intt\": %u, \"coset_ntt\": %u, \"gate_launches\": %u, \"gates\": %u, \"gate_terms\": %u, \"evals\": %zu, \"resident_ms\": %.3f, \"first_proof_ms\": %.3f, \"proofs\": %d, "
              "\"host_api_ms\": %.3f, \"host_api_fft_ms\": %.3f, \"host_api_fft_batched_ms\": %.3f, "
              "\"step_ms\": {\"1_instance\": %.2f, \"2_3_advice_lookup_commits\": %.2f, \"4_products\": %.2f, \"5_random\": 0.0, \"6_to_coeff\": %.2f, \"7_quotient\": %.2f, \"8_commit_h\": %.2f, \"9_evals\": %.2f, \"10_shplonk\": %.2f}, "
              "\"hbm\": {\"total_gib\": %.1f, \"peak_used_gib\": %.1f, \"proving_key_gib\": %.1f, \"live_buffers_gib\": %.1f, \"pooled_gib\": %.1f, \"workspace_gib\": %.1f, \"planned\": {\"pk_base_gib\": %.1f, \"pk_cosets_gib\": %.1f, \"working_set_gib\": %.1f, \"one_table_gib\": %.1f, \"usable_gib\": %.1f}}, "
              "\"checked\": %u, \"semantic_check\": %s, \"trapdoor_check\": %s, \"ok\": %s}\n",
              layer_id, k, devices, S.advice, S.fixed, S.lookups, S.perm_columns, S.chunk_len, S.perm_z(), S.degree, Q, S.source,
              n_tables ? "true" : "false", n_tables, resident ? "resident" : "on-the-fly",
              upload_threads, pinned_witness ? "true" : "false", R.commitments.size(), R.intt, R.coset_ntt, R.gate_launches, plan.gates, plan.terms, R.evals.size(), R.total_ms, first_ms, proofs,
              host_ms, host_fft_ms, host_fft_batched_ms,
              R.step_ms[1], R.step_ms[2], R.step_ms[4], R.step_ms[6], R.step_ms[7], R.step_ms[8], R.step_ms[9], R.step_ms[10],
              hbm_total / GiB, (hbm_total - fr_end) / GiB, pk.bytes / GiB, live / GiB, pooled / GiB, ws / GiB, pk_base / GiB, pk_cosets / GiB, working / GiB, table_one / GiB, usable / GiB,
              checked, semantic_ok ? "true" : "false", trapdoor_ok ? "true" : "false", failures == 0 ? "true" : "false");
  R = ProofGpuSide(); pk = ProvingKeyDevice();
  CK(mi355_srs_release(hg)); CK(mi355_srs_release(hl));
  } catch (const std::exception &e) { std::printf("FAILED with exception: %s\n", e.what()); failures++; rc_main = 1; }
  CK(mi355_shutdown());
  if (failures) { std::printf("%d check(s) FAILED\n", failures); std::fflush(stdout); return 1; }
  std::printf("all checks passed\n");
  std::fflush(stdout);   // exit handlers of the HIP / sanitizer runtimes run after main: what was printed must not depend on them
  return rc_main;
}
