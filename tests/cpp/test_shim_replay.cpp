// test_shim_replay.cpp -- a compiled stand-in for what rust_shim/mi355zk.rs does under scroll-prover (Rust is absent from this image).
// Everything goes through the C-ABI with HOST pointers, the way the patched halo2_proofs calls it:
//
//   * ParamsKZG owns two host Vec<G1Affine> plus reference-counted GPU registrations (Arc<GpuBasis> in the shim; shared_ptr here);
//     Drop releases the handle (mi355_srs_release)                                   [REF bin/src/trace_prover.rs:35-43 params_map]
//   * load_params_map = read the largest params, then clone + downsize per degree    [REF integration/tests/integration.rs:12-22]
//       clone    -> shares the registration (Arc::clone), no second upload, no second window table
//       downsize -> g: mi355_srs_register_prefix (shares memory AND tables); g_lagrange: mi355_srs_downsize + mi355_srs_read_host
//   * commit / commit_lagrange on &g[..n] sub-slices -> mi355_msm_g1_host(handle, 0, poly, n)
//   * best_multiexp on a slice that is NOT a ParamsKZG basis -> mi355_msm_g1_adhoc_host (never registered: a freed-and-reused
//     address can therefore not select a stale basis -- the hazard of the round-1 address-keyed map)
//   * commits arrive concurrently from worker threads (rayon); one of them asks for un-normalised results (a per-thread option)
//
// Expected values come from the CPU oracle (TEST INFRASTRUCTURE, the checker).  Exit code 0 = all checks passed.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "mi355zk.h"

extern "C" {   // oracle
void orc_best_multiexp(void *out, const void *coeffs, const void *bases, uint64_t n, int threads);
void orc_g1_to_affine(void *o, const void *p);
void orc_g1_mul_generator_vec(void *out, const void *scalars_mont, uint64_t n, int threads);
void orc_g_to_lagrange(void *out, const void *g, uint32_t k, const void *omega_inv, const void *n_inv);
void orc_f_inv(int w, void *o, const void *a);
void orc_f_pow(int w, void *o, const void *a, const uint64_t *e);
void orc_f_from_canonical(int w, void *o, const void *a);
}

struct Fr { uint64_t l[4]; };
struct G1Affine { uint64_t l[8]; bool operator==(const G1Affine &o) const { return std::memcmp(l, o.l, 64) == 0; } };
struct G1 { uint64_t l[12]; };

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED %s:%d  %s   [%s]\n", __FILE__, __LINE__, #cond, mi355_last_error()); failures++; } } while (0)

static Fr rand_fr(std::mt19937_64 &g) { return Fr{{g(), g(), g(), g() & ((uint64_t(1) << 60) - 1)}}; }
static G1Affine to_affine(const G1 &p) { G1Affine a; orc_g1_to_affine(a.l, p.l); return a; }
static G1Affine oracle_msm(const Fr *c, const G1Affine *b, uint64_t n) { G1 j; orc_best_multiexp(j.l, c, b, n, 4); return to_affine(j); }

// ---- the shim's types -----------------------------------------------------------------------------------------------------------
struct GpuBasis {                       // rust: struct GpuBasis(u64); impl Drop for GpuBasis { mi355_srs_release }
  uint64_t handle = 0;
  ~GpuBasis() { if (handle) { if (mi355_srs_release(handle) != MI355_OK) { std::printf("release failed: %s\n", mi355_last_error()); failures++; } } }
};
static std::shared_ptr<GpuBasis> register_basis(const std::vector<G1Affine> &v, bool precompute) {
  auto b = std::make_shared<GpuBasis>();
  if (mi355_srs_register_host(v.data(), v.size(), &b->handle) != MI355_OK) return nullptr;
  if (precompute) (void)mi355_srs_precompute(b->handle, 0, 0);
  return b;
}
struct ParamsKZG {
  uint32_t k = 0;
  std::vector<G1Affine> g, g_lagrange;              // host-owned, as in the reference
  std::shared_ptr<GpuBasis> gpu_g, gpu_gl;          // Option<Arc<GpuBasis>>: clone() shares, Drop of the last owner releases

  static ParamsKZG read(uint32_t k, const std::vector<G1Affine> &g, const std::vector<G1Affine> &gl, bool precompute) {   // read_custom + registration hook
    ParamsKZG p; p.k = k; p.g = g; p.g_lagrange = gl;
    p.gpu_g = register_basis(p.g, precompute); p.gpu_gl = register_basis(p.g_lagrange, precompute);
    return p;
  }
  // ParamsKZG::downsize(k) with the shim's hook: truncate g, prefix view of its registration, g_lagrange rebuilt on the device
  bool downsize(uint32_t new_k, const Fr &omega_inv, const Fr &n_inv) {
    if (new_k >= k) return new_k == k;
    const uint64_t n = uint64_t(1) << new_k;
    g.resize(n);
    auto view = std::make_shared<GpuBasis>();
    if (mi355_srs_register_prefix(gpu_g->handle, n, &view->handle) != MI355_OK) return false;
    auto gl = std::make_shared<GpuBasis>();
    if (mi355_srs_downsize(view->handle, new_k, &omega_inv, &n_inv, &gl->handle) != MI355_OK) return false;
    g_lagrange.assign(n, G1Affine{});
    if (mi355_srs_read_host(gl->handle, 0, n, g_lagrange.data()) != MI355_OK) return false;
    gpu_g = view; gpu_gl = gl; k = new_k;           // the old Arcs drop here; a clone elsewhere may still hold them
    return true;
  }
  bool commit(const Fr *poly, uint64_t n, G1 &out) const { return mi355_msm_g1_host(gpu_g->handle, 0, poly, n, out.l) == MI355_OK; }
  bool commit_lagrange(const Fr *poly, uint64_t n, G1 &out) const { return n == g_lagrange.size() && mi355_msm_g1_host(gpu_gl->handle, 0, poly, n, out.l) == MI355_OK; }
};
// generic best_multiexp hook: the bases are just a slice, possibly a temporary -> never registered
static bool best_multiexp(const Fr *c, const G1Affine *b, uint64_t n, G1 &out) { return mi355_msm_g1_adhoc_host(b, c, n, out.l) == MI355_OK; }

static Fr fr_from_u64(uint64_t v) { Fr c{{v, 0, 0, 0}}, m; orc_f_from_canonical(1, m.l, c.l); return m; }
static void domain_consts(uint32_t k, Fr &omega_inv, Fr &n_inv) {
  // omega_k = 7^((r-1)/2^28) ^ (2^(28-k)); constants computed by the oracle's field code (host set-up, as EvaluationDomain::new does)
  const uint64_t e[4] = {0x9b9709143e1f593full, 0x181585d2833e8487ull, 0x131a029b85045b68ull, 0x30644e72eull};   // (r - 1) >> 28
  Fr seven = fr_from_u64(7), root, w;
  orc_f_pow(1, root.l, seven.l, e);
  w = root;
  for (uint32_t i = k; i < 28; i++) { const uint64_t two[4] = {2, 0, 0, 0}; Fr t; orc_f_pow(1, t.l, w.l, two); w = t; }
  orc_f_inv(1, omega_inv.l, w.l);
  Fr n = fr_from_u64(uint64_t(1) << k); orc_f_inv(1, n_inv.l, n.l);
}

int main(int argc, char **argv) {
  int dev = 0; if (argc > 1) dev = std::atoi(argv[1]);
  if (mi355_init(dev) != MI355_OK) { std::printf("mi355_init: %s\n", mi355_last_error()); return 2; }
  std::mt19937_64 rng(2024);
  const uint32_t K = 14; const uint64_t N = uint64_t(1) << K;
  // "params15": g = random independent points (what a real SRS looks like to the MSM), g_lagrange = g_to_lagrange(g) by the oracle
  std::vector<Fr> ks(N); for (auto &x : ks) x = rand_fr(rng);
  std::vector<G1Affine> g(N), gl(N);
  orc_g1_mul_generator_vec(g.data(), ks.data(), N, 8);
  { Fr wi, ni; domain_consts(K, wi, ni); orc_g_to_lagrange(gl.data(), g.data(), K, wi.l, ni.l); }

  // ---- load_params_map: largest degree read once, smaller degrees = clone + downsize
  std::map<uint32_t, ParamsKZG> params_map;
  params_map[K] = ParamsKZG::read(K, g, gl, true);
  EXPECT(params_map[K].gpu_g && params_map[K].gpu_gl);
  void *tab_parent = nullptr; int c_parent = 0, w_parent = 0;
  EXPECT(mi355_srs_pre_dev_ptr(params_map[K].gpu_g->handle, &tab_parent, &c_parent, &w_parent) == MI355_OK && tab_parent);
  for (uint32_t k : {13u, 11u}) {
    ParamsKZG p = params_map[K];                      // clone: host Vecs copied, registrations shared
    EXPECT(p.gpu_g.get() == params_map[K].gpu_g.get());
    Fr wi, ni; domain_consts(k, wi, ni);
    EXPECT(p.downsize(k, wi, ni));
    // the prefix view shares the parent's window table (no second table), its g_lagrange equals the oracle's g_to_lagrange(g[..2^k])
    void *tab = nullptr; int c = 0, w = 0;
    EXPECT(mi355_srs_pre_dev_ptr(p.gpu_g->handle, &tab, &c, &w) == MI355_OK && tab == tab_parent && c == c_parent);
    std::vector<G1Affine> want(uint64_t(1) << k);
    orc_g_to_lagrange(want.data(), g.data(), k, wi.l, ni.l);
    EXPECT(p.g_lagrange == want);
    params_map[k] = std::move(p);
  }

  // ---- commits on sub-slices &g[..n] and on the Lagrange bases of every degree
  std::vector<Fr> poly(N); for (auto &x : poly) x = rand_fr(rng);
  for (auto &kv : params_map) {
    const ParamsKZG &p = kv.second; const uint64_t n = uint64_t(1) << p.k;
    G1 out;
    EXPECT(p.commit(poly.data(), n, out) && to_affine(out) == oracle_msm(poly.data(), p.g.data(), n));
    EXPECT(p.commit(poly.data(), n / 2 + 5, out) && to_affine(out) == oracle_msm(poly.data(), p.g.data(), n / 2 + 5));   // &g[..n'] of a shorter poly
    EXPECT(p.commit_lagrange(poly.data(), n, out) && to_affine(out) == oracle_msm(poly.data(), p.g_lagrange.data(), n));
    EXPECT(!p.commit_lagrange(poly.data(), n - 1, out));
    EXPECT(mi355_msm_g1_host(p.gpu_g->handle, 1, poly.data(), n, out.l) == MI355_EBADARG);   // base_offset + n beyond the basis: best_multiexp panics
  }

  // ---- drop order: the parent params go away first, the downsized clones keep working (memory lives until the last sharer drops)
  params_map.erase(K);
  { const ParamsKZG &p = params_map[13]; G1 out; EXPECT(p.commit(poly.data(), 1 << 13, out) && to_affine(out) == oracle_msm(poly.data(), p.g.data(), 1 << 13)); }

  // ---- a temporary basis is freed and its address re-used with other contents: must never resolve to a stale registration
  {
    const uint64_t n = 1 << 13;
    auto *tmp = new std::vector<G1Affine>(g.begin(), g.begin() + n);
    const G1Affine *addr = tmp->data();
    G1 out; EXPECT(best_multiexp(poly.data(), tmp->data(), n, out) && to_affine(out) == oracle_msm(poly.data(), g.data(), n));
    delete tmp;
    std::vector<G1Affine> other(gl.begin(), gl.begin() + n);            // very likely the same allocation: same size class, just freed
    std::printf("address reuse: %s\n", other.data() == addr ? "same address" : "different address (allocator did not recycle it)");
    EXPECT(best_multiexp(poly.data(), other.data(), n, out) && to_affine(out) == oracle_msm(poly.data(), other.data(), n));
  }

  // ---- 8 worker threads commit concurrently; odd threads ask for un-normalised results (per-thread option must not leak)
  {
    const ParamsKZG &p = params_map[13]; const uint64_t n = 1 << 13;
    std::vector<std::vector<Fr>> polys(8, std::vector<Fr>(n));
    std::vector<G1Affine> want(8);
    for (int t = 0; t < 8; t++) { for (auto &x : polys[t]) x = rand_fr(rng); want[t] = oracle_msm(polys[t].data(), p.g.data(), n); }
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++) th.emplace_back([&, t]() {
      if (t & 1) mi355_msm_set_normalise(0);
      for (int rep = 0; rep < 6; rep++) {
        G1 out;
        if (!p.commit(polys[t].data(), n, out) || !(to_affine(out) == want[t])) bad++;
        const bool normalised = out.l[8] == 0xd35d438dc58f0d9dull && out.l[9] == 0x0a78eb28f5c70b3dull && out.l[10] == 0x666ea36f7879462cull && out.l[11] == 0x0e0a77c19a07df2full;   // z == R mod p
        if (!(t & 1) && !normalised) bad++;             // an even thread must never see another thread's normalise(0)
      }
    });
    for (auto &x : th) x.join();
    EXPECT(bad.load() == 0);
  }

  // ---- released handles are gone
  {
    uint64_t h = params_map[11].gpu_g->handle;
    params_map.clear();
    uint64_t len = 0; EXPECT(mi355_srs_len(h, &len) == MI355_EBADARG);
  }
  EXPECT(mi355_shutdown() == MI355_OK);
  if (failures) { std::printf("%d check(s) FAILED\n", failures); return 1; }
  std::printf("all checks passed\n");
  return 0;
}
