// test_halo2_mirror.cpp -- exercises include/mi355zk_halo2.hpp (the C++ mirror of the halo2_proofs operator surface) the way the
// upstream crates test theirs: results against the CPU oracle (oracle/liboracle_bn254.so -- the checker), domain constants against
// the reference's released protocol file (KAT A1).  `--host-only` runs the part that needs no GPU.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <set>

#include "mi355zk_plonk.hpp"
#include "../../scroll-prover_amd/csrc/slab_ranges.hpp"   // the bookkeeping behind mi355_buf_alloc's slabs (host-only code, model-checked below)

using namespace mi355zk::halo2;

extern "C" {   // oracle (TEST INFRASTRUCTURE): see oracle/bn254_oracle.c
void orc_best_multiexp(void *out, const void *coeffs, const void *bases, uint64_t n, int threads);
void orc_g1_to_affine(void *o, const void *p);
void orc_g1_mul(void *o, const void *p, const void *scalar_mont);
void orc_g1_generator(void *o);
void orc_best_fft(void *a, const void *omega, uint32_t log_n, int threads);
void orc_ifft(void *a, const void *omega_inv, uint32_t log_n, const void *divisor, int threads);
void orc_g_to_lagrange(void *out, const void *g, uint32_t k, const void *omega_inv, const void *n_inv);
void orc_best_fft_g1(void *a, const void *omega, uint32_t log_n);
void orc_g1_compress(uint8_t out[32], const void *p);
int orc_g1_decompress(void *o, const uint8_t in[32]);
void orc_eval_polynomial(void *out, const void *poly, uint64_t n, const void *point);
void orc_coeff_to_extended(void *dst, const void *coeffs, uint32_t k, uint32_t ext_k, const void *g, const void *gi, const void *ew, int threads);
}

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static Fr rand_fr(std::mt19937_64 &g) { Fr r{g(), g(), g(), g() & ((uint64_t(1) << 60) - 1)}; return r; }

int main(int argc, char **argv) {
  bool host_only = false; std::string protocol_dir;
  for (int i = 1; i < argc; i++) { if (std::strcmp(argv[i], "--host-only") == 0) host_only = true; else if (std::strcmp(argv[i], "--protocols") == 0 && i + 1 < argc) protocol_dir = argv[++i]; }
  // --- EvaluationDomain::new against [REF release-v0.13.1/chunk.protocol] domain {k: 25, gen, gen_inv, n_inv}
  EvaluationDomain d25(2, 25);
  const Fr gen{13338605924273364442ull, 11440449704248451096ull, 16859609365912477452ull, 3421252324365184758ull};
  const Fr gen_inv{2738242980467392064ull, 8765460162850139420ull, 6637814084492473216ull, 1260493707339115276ull};
  const Fr n_inv{0, 0, 0, 549755813888ull};
  EXPECT(d25.omega == gen); EXPECT(d25.omega_inv == gen_inv); EXPECT(d25.ifft_divisor == n_inv);
  EXPECT(EvaluationDomain(5, 26).extended_k == 28); EXPECT(EvaluationDomain(4, 10).extended_k == 12); EXPECT(EvaluationDomain(3, 10).extended_k == 11);
  bool threw = false; try { EvaluationDomain bad(9, 26); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw);
  threw = false; try { best_multiexp(std::vector<Fr>(3), std::vector<G1Affine>(4)); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw);
  // --- compressed G1 codec against the oracle (itself pinned by the .vkey fixtures): encode, decode, rejects
  {
    std::mt19937_64 rc(7); G1Affine Gc; orc_g1_generator(Gc.data());
    std::vector<G1Affine> bases(64);
    for (auto &pt : bases) { Fr sk = rand_fr(rc); G1 j; orc_g1_mul(j.data(), Gc.data(), sk.data()); orc_g1_to_affine(pt.data(), j.data()); }
    for (uint64_t i = 0; i < 64; i++) {
      G1Bytes want; orc_g1_compress(want.data(), bases[i].data());
      const G1Bytes got = g1_to_bytes(bases[i]);
      EXPECT(got == want);
      G1Affine back; EXPECT(g1_from_bytes(got, back) && back == bases[i]);
      G1Bytes flipped = got; flipped[31] ^= 0x40;                              // the other root: -P
      G1Affine neg, wneg; EXPECT(g1_from_bytes(flipped, neg)); EXPECT(orc_g1_decompress(wneg.data(), flipped.data()) == 1 && neg == wneg);
      EXPECT(neg[0] == bases[i][0] && neg != bases[i]);
    }
    G1Affine id{}; EXPECT(g1_to_bytes(id) == G1Bytes{});
    G1Affine out; EXPECT(g1_from_bytes(G1Bytes{}, out) && out == id);
    G1Bytes bad{}; for (auto &b : bad) b = 0xff; bad[31] = 0x3f;               // x >= p
    EXPECT(!g1_from_bytes(bad, out));
    int rejected = 0;                                                          // about half of all x are not on the curve
    for (uint8_t v = 1; v < 40; v++) { G1Bytes t{}; t[0] = v; G1Affine o1, o2; const bool a = g1_from_bytes(t, o1); const int b = orc_g1_decompress(o2.data(), t.data()); EXPECT(a == (b == 1)); if (a) EXPECT(o1 == o2); else rejected++; }
    EXPECT(rejected > 5);
  }
  // --- the plan compiler of mi355zk::plonk::create_proof (include/mi355zk_plonk.hpp), host logic only, on the protocols of all seven layers at full size
  // (--protocols DIR: layer0.json ... layer6.json written by scroll-prover_amd/protocols.py; layers 2 / 4 equal the reference's fixtures): every launch stays
  // inside the limits of mi355_fr_gate_eval_dev, temporaries are written before they are read, the counts are the fixtures'
  if (!protocol_dir.empty()) {
    using namespace mi355zk::plonk;
    const uint32_t want_commitments[7] = {953, 35, 11, 148, 14, 17, 11};   // layers 2, 4 (and 6 = layer 2's system): the released proofs' word counts
    const uint32_t want_evals[7] = {0, 125, 17, 652, 27, 42, 17};
    std::vector<std::unique_ptr<Protocol>> protos;
    for (int layer = 0; layer <= 6; layer++) {
      auto P = std::make_unique<Protocol>(); P->load(protocol_dir + "/layer" + std::to_string(layer) + ".json");
      EXPECT(P->commitments() == want_commitments[layer]);
      if (want_evals[layer]) EXPECT(P->evaluations.size() == want_evals[layer]);
      EXPECT(P->last_rot == -7 && P->blind == 6 && P->queries.size() == P->evaluations.size() + 1);
      for (const auto &c : P->perm) EXPECT(c.columns.size() + 2 <= P->Q + 1);                                        // chunks of degree - 2
      CommonRegistry reg; Compiler cmp(reg, true, {fr_u64(2), fr_u64(3), fr_u64(5), fr_u64(7)});
      cmp.compile_numerator(P->numerator);
      EXPECT(reg.defs.size() >= 3 && reg.defs.size() <= 4);                                                            // l_0, l_last, l_active (, X)
      std::set<uint32_t> written; uint32_t quotient_launches = 0;
      for (const auto &L : cmp.out) {
        std::set<Atom> polys; uint32_t nf = 0;
        EXPECT(L.terms.size() >= 1 && L.terms.size() <= PLAN_MAX_TERMS);
        for (const auto &t : L.terms) {
          EXPECT(t.f.size() <= PLAN_MAX_TERM_LEN); nf += (uint32_t)t.f.size();
          for (const auto &f : t.f) { polys.insert(Atom{f.kind, f.idx, 0}); if (f.kind == A_TMP) EXPECT(written.count(f.idx) == 1); }
        }
        EXPECT(nf <= PLAN_MAX_FACTORS && polys.size() <= PLAN_MAX_POLYS);
        if (L.dst >= 0) { EXPECT((uint32_t)L.dst < cmp.tmp_max); if (!L.accumulate) written.erase((uint32_t)L.dst); written.insert((uint32_t)L.dst); } else quotient_launches++;
      }
      EXPECT(quotient_launches >= 1 && cmp.constraints + 1 == P->numerator.kids.size());
      // the plan computes the numerator: run it on ONE row of scalars (every (polynomial, rotation) leaf, X and every Lagrange polynomial a random value) and compare with the
      // tree walked directly -- with and without the common-prefix groups
      for (const char *pm : {"16", "0", "2"}) {
        setenv("MI355_PLAN_PREFIX_MIN", pm, 1);
        std::mt19937_64 rg(1000 + layer);
        std::map<std::pair<int32_t, int32_t>, Fr> leaf; std::map<int32_t, Fr> lag; const Fr xval = rand_fr(rg);
        const std::vector<Fr> chv{rand_fr(rg), rand_fr(rg), rand_fr(rg), rand_fr(rg)};
        std::function<Fr(const Expr &)> walk = [&](const Expr &e) -> Fr {
          switch (e.kind) {
            case Expr::CONSTANT: return e.c;
            case Expr::IDENTITY: return xval;
            case Expr::LAGRANGE: { auto it = lag.find(e.i); if (it == lag.end()) it = lag.emplace(e.i, rand_fr(rg)).first; return it->second; }
            case Expr::POLY: { auto it = leaf.find({e.i, e.rot}); if (it == leaf.end()) it = leaf.emplace(std::make_pair(e.i, e.rot), rand_fr(rg)).first; return it->second; }
            case Expr::CHALLENGE: return chv[(size_t)e.i];
            case Expr::NEG: return fr_neg(walk(e.kids[0]));
            case Expr::SUM: return fr_add(walk(e.kids[0]), walk(e.kids[1]));
            case Expr::PROD: return fr_mul(walk(e.kids[0]), walk(e.kids[1]));
            case Expr::SCALED: return fr_mul(walk(e.kids[0]), e.c);
            default: { const Fr b = walk(e.kids.back()); Fr acc = walk(e.kids[0]); for (size_t i = 1; i + 1 < e.kids.size(); i++) acc = fr_add(fr_mul(acc, b), walk(e.kids[i])); return acc; }
          }
        };
        const Fr want = walk(P->numerator);
        CommonRegistry reg2; Compiler c2(reg2, true, chv); c2.compile_numerator(P->numerator);
        std::vector<Fr> common_val; for (const auto &d : reg2.defs) { Fr v = fr_add(d.constant, fr_mul(d.x_coeff, xval)); for (const auto &sp : d.lagrange) v = fr_add(v, fr_mul(sp.second, lag.at(sp.first))); common_val.push_back(v); }
        std::vector<Fr> tmpv(c2.tmp_max, fr_zero()); Fr got = fr_zero();
        for (const auto &L : c2.out) {
          Fr acc = fr_zero();
          for (const auto &t : L.terms) { Fr v = t.coeff; for (const auto &f : t.f) v = fr_mul(v, f.kind == A_TMP ? tmpv[f.idx] : f.kind == A_COMMON ? common_val[f.idx] : leaf.at({(int32_t)f.idx, f.rot})); acc = fr_add(acc, v); }
          if (L.dst < 0) got = fr_add(got, acc); else tmpv[(size_t)L.dst] = L.accumulate ? fr_add(tmpv[(size_t)L.dst], acc) : acc;
        }
        if (!(got == want)) std::printf("layer %d, MI355_PLAN_PREFIX_MIN=%s: the compiled plan does not compute the numerator\n", layer, pm);
        EXPECT(got == want);
      }
      unsetenv("MI355_PLAN_PREFIX_MIN");
      if (layer == 2) { EXPECT(cmp.constraints == 7); const auto sets = rotation_sets(P->queries); EXPECT(sets.size() == 3 && sets[0].rots.size() == 4 && sets[2].polys.size() == 10); }
      if (layer == 4) { EXPECT(cmp.constraints == 10); const auto sets = rotation_sets(P->queries); EXPECT(sets.size() == 4 && sets[2].rots.size() == 3 && sets[2].polys.size() == 1); }   // z_0 is opened at x, w x and w^-7 x
      protos.push_back(std::move(P));
    }
    // --- the residency rule of DESIGN.md section 9 as code: one layer alone keeps its cosets (and below k = 26 both tables); a chunk prover {0, 1, 2} and a batch
    // prover {3, 4} fit 288 GiB only by giving things up, cosets before tables
    auto one = [&](int l) { return plan_residency({protos[l].get()}, 288.0); };
    for (int l : {0, 1, 2, 3, 5}) { const ResidencyPlan R = one(l); EXPECT(R.fits && R.layers[0].cosets_resident && R.layers[0].table_lagrange && R.layers[0].table_coeff); }
    { const ResidencyPlan R = one(4); EXPECT(R.fits && R.layers[0].cosets_resident && !R.layers[0].table_coeff); EXPECT(R.total_gib <= 288.0 * 0.92 + 1e-9); }
    const ResidencyPlan C = plan_residency({protos[0].get(), protos[1].get(), protos[2].get()}, 288.0);
    EXPECT(C.fits && C.total_gib <= 265.0);
    int resident = 0; for (const auto &L : C.layers) resident += L.cosets_resident;
    EXPECT(resident >= 2);
    // round 6: WHICH keys stay is decided by the proof time they save among the subsets that FIT under the measured accounting of a multi-layer process.  By value alone layers
    // 1 + 2 would stay and layer 0 recompute -- that plan ran out of memory on the device (a lean layer 0 adds its recomputed part to the largest working set); what fits and
    // saves most is layers 0 + 2 resident, layer 1 lean
    EXPECT(C.layers[0].cosets_resident && !C.layers[1].cosets_resident && C.layers[2].cosets_resident);
    const ResidencyPlan B = plan_residency({protos[3].get(), protos[4].get()}, 288.0);
    EXPECT(B.fits && B.layers[0].cosets_resident && !B.layers[1].cosets_resident);   // with the reference's real layer-4 system (13 key polynomials: 42 GiB + 104 GiB of cosets) a batch prover
    EXPECT(!B.layers[1].table_coeff);                                                // keeps layer 3's cosets and recomputes layer 4's per part; the k = 26 bases stay table-free
    for (const ResidencyPlan *R : {&C, &B}) std::printf("residency plan: srs %.1f keys %.1f tables %.1f working %.1f total %.1f of %.1f GiB\n", R->srs_gib, R->keys_gib, R->tables_gib, R->working_gib, R->total_gib, R->budget_gib);
    const ResidencyPlan T = plan_residency({protos[4].get()}, 80.0);                  // a smaller card: the key's cosets no longer fit
    EXPECT(!T.layers[0].cosets_resident);
  }
  // --- the slab bookkeeping of mi355_buf_alloc (csrc/slab_ranges.hpp) against a byte map: random carve / give-back sequences with the block sizes of a prover process (2^20-,
  // 2^24-, 2^25-row layers scaled down by 2^12), slabs added on a miss exactly as lib_core.hip does.  Invariants after every step: live blocks never overlap each other or a free
  // range, every byte of every slab is either live or free, adjacent free ranges of one slab are merged, no range spans two slabs -- and once everything is back, every slab is one
  // range and take_whole_slabs returns all of them
  {
    std::mt19937_64 rs(99);
    for (int round = 0; round < 3; round++) {
      mi355zk::SlabRanges A; const size_t SLAB = 1 << 18; uintptr_t next_base = 1 << 20;
      const size_t sizes[] = {256, 256 * 3, 8192, 8192 * 3, 131072, 262144, 262144 * 2, 768};
      std::map<uintptr_t, size_t> live; size_t carved_total = 0;
      auto check_state = [&]() {
        size_t covered = 0, slab_total = 0; uintptr_t prev_end = 0; bool prev_free = false; const mi355zk::SlabRanges::Slab *prev_slab = nullptr;
        std::map<uintptr_t, std::pair<size_t, bool>> all;
        for (auto &kv : live) all[kv.first] = {kv.second, false};
        for (auto &kv : A.free_ranges) { EXPECT(!all.count(kv.first)); all[kv.first] = {kv.second, true}; }
        for (auto &kv : all) {
          const auto *s = A.slab_of(kv.first);
          EXPECT(s != nullptr); if (!s) continue;
          EXPECT(kv.first + kv.second.first <= s->base + s->bytes);                                   // no block or range leaves its slab
          EXPECT(kv.first >= prev_end);                                                                // no overlap
          if (kv.second.second && prev_free && prev_slab == s) EXPECT(kv.first != prev_end);           // adjacent free ranges of one slab are merged
          prev_end = kv.first + kv.second.first; prev_free = kv.second.second; prev_slab = s; covered += kv.second.first;
        }
        for (auto &s : A.slabs) slab_total += s.bytes;
        EXPECT(covered == slab_total);                                                                 // every byte is live or free
      };
      for (int step = 0; step < 4000; step++) {
        const bool alloc = live.empty() || (rs() % 100) < (step < 2500 ? 58 : 35);
        if (alloc) {
          const size_t want = sizes[rs() % (round == 0 ? 4 : 8)];
          uintptr_t p = A.carve(want);
          if (!p) { const size_t sb = std::max(want, SLAB); A.add_slab(next_base, sb); next_base += sb + ((rs() & 1) ? 0 : 4096); p = A.carve(want); }   // some slabs touch the previous one, some do not
          EXPECT(p != 0); if (!p) break;
          EXPECT(!live.count(p)); live[p] = want; carved_total += want;
        } else {
          auto it = live.begin(); std::advance(it, (long)(rs() % live.size()));
          A.insert(it->first, it->second); live.erase(it);
        }
        if (step % 16 == 0) check_state();
      }
      check_state();
      std::vector<uintptr_t> gone; const size_t before = A.slabs.size(); size_t expect_bytes = 0;
      { std::set<const void *> busy; for (auto &kv : live) busy.insert(A.slab_of(kv.first)); for (auto &s : A.slabs) if (!busy.count(&s)) expect_bytes += s.bytes; }
      EXPECT(A.take_whole_slabs(gone) == expect_bytes);                                                // exactly the slabs without a live block
      for (auto &kv : live) EXPECT(A.slab_of(kv.first) != nullptr);
      while (!live.empty()) { A.insert(live.begin()->first, live.begin()->second); live.erase(live.begin()); }
      check_state();
      gone.clear(); A.take_whole_slabs(gone);
      EXPECT(A.slabs.empty() && A.free_ranges.empty() && A.free_bytes() == 0);
      EXPECT(before > 0 && carved_total > 0);
    }
    // best fit takes the smallest range that holds the request, from its front; a request no range holds returns 0 and changes nothing
    mi355zk::SlabRanges B; B.add_slab(0x100000, 0x10000); B.add_slab(0x200000, 0x4000);
    EXPECT(B.carve(0x3000) == 0x200000); EXPECT(B.carve(0x1000) == 0x203000); EXPECT(B.carve(0x1000) == 0x100000);
    EXPECT(B.carve(0x20000) == 0 && B.free_bytes() == 0xF000);
    B.insert(0x203000, 0x1000); B.insert(0x200000, 0x3000); EXPECT(B.free_ranges.at(0x200000) == 0x4000);
    // segregation by size: a small request never goes into a dedicated slab (one made for a single large block), even when that slab is entirely free and nothing else is
    { mi355zk::SlabRanges S; S.small_limit = 0x1000; S.add_slab(0x100000, 0x40000, /*dedicated=*/true);
      EXPECT(S.carve(0x100) == 0);                                                       // -> the caller opens a shared slab
      S.add_slab(0x200000, 0x10000); EXPECT(S.carve(0x100) == 0x200000);
      const uintptr_t big = S.carve(0x40000); EXPECT(big == 0x100000);                   // the large block takes the dedicated slab whole
      S.insert(big, 0x40000); std::vector<uintptr_t> gone; EXPECT(S.take_whole_slabs(gone) == 0x40000 && gone.size() == 1 && gone[0] == 0x100000);   // and the slab goes back whole
      EXPECT(S.carve(0x2000) == 0x200100);                                               // at or above the limit: best fit anywhere
      // random traffic with the rule on: small blocks are always found in shared slabs
      std::mt19937_64 r2(7); mi355zk::SlabRanges R; R.small_limit = 4096; uintptr_t nb = 1 << 24; std::map<uintptr_t, size_t> live2;
      for (int step = 0; step < 4000; step++) {
        if (live2.empty() || r2() % 100 < 55) {
          const size_t want = (r2() % 3 == 0) ? (size_t)(1 << 18) * (1 + r2() % 3) : (size_t)256 * (1 + r2() % 8);
          uintptr_t q = R.carve(want);
          if (!q) { const size_t sb = std::max<size_t>(want, 1 << 17); R.add_slab(nb, sb, want >= (1u << 17)); nb += sb + 4096; q = R.carve(want); }
          EXPECT(q != 0); if (!q) break;
          if (want < 4096) { const auto *sl = R.slab_of(q); EXPECT(sl && !sl->dedicated); }
          live2[q] = want;
        } else { auto it = live2.begin(); std::advance(it, (long)(r2() % live2.size())); R.insert(it->first, it->second); live2.erase(it); }
      }
    }
    // two slabs whose addresses touch stay two ranges
    mi355zk::SlabRanges T; T.add_slab(0x1000, 0x1000); T.add_slab(0x2000, 0x1000);
    const uintptr_t t0 = T.carve(0x1000), t1 = T.carve(0x1000); T.insert(t0, 0x1000); T.insert(t1, 0x1000);
    EXPECT(T.free_ranges.size() == 2 && T.carve(0x2000) == 0);
  }
  if (host_only) {
    // without a GPU every compute entry point must fail loudly (no CPU fallback)
    threw = false; try { init(0); } catch (const Error &e) { threw = e.code == MI355_ENODEVICE; } 
    if (!threw) { std::printf("note: a GPU is visible; host-only run skips the no-device check\n"); }
    std::printf(failures ? "FAILED (%d)\n" : "host-only checks passed\n", failures);
    return failures ? 1 : 0;
  }
  init(0);
  std::mt19937_64 rng(20240924);
  // --- best_multiexp / ParamsKZG::commit against the oracle
  const uint32_t k = 10; const uint64_t n = 1 << k;
  G1Affine G; orc_g1_generator(G.data());
  std::vector<G1Affine> bases(n); std::vector<Fr> sc(n);
  for (uint64_t i = 0; i < n; i++) { Fr s = rand_fr(rng); G1 j; orc_g1_mul(j.data(), G.data(), s.data()); orc_g1_to_affine(bases[i].data(), j.data()); sc[i] = rand_fr(rng); }
  G1 want_j; orc_best_multiexp(want_j.data(), sc.data(), bases.data(), n, 4);
  G1Affine want; orc_g1_to_affine(want.data(), want_j.data());
  G1 got = best_multiexp(sc, bases);
  EXPECT(std::memcmp(got.data(), want.data(), 64) == 0);
  {
    std::vector<G1Affine> rev(bases.rbegin(), bases.rend());
    ParamsKZG params(k, bases, rev, /*window_tables=*/true);
    G1 c1 = params.commit(sc);
    EXPECT(std::memcmp(c1.data(), want.data(), 64) == 0);
    G1 c2 = params.commit_lagrange(sc);
    G1 w2; orc_best_multiexp(w2.data(), sc.data(), rev.data(), n, 4); G1Affine w2a; orc_g1_to_affine(w2a.data(), w2.data());
    EXPECT(std::memcmp(c2.data(), w2a.data(), 64) == 0);
    threw = false; try { params.commit_lagrange(std::vector<Fr>(n - 1)); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw);
    // commit_many == the loop of commits (three columns, one of them all-zero)
    std::vector<Fr> col2(n), col3(n, Fr{0, 0, 0, 0});
    for (auto &x : col2) x = rand_fr(rng);
    std::vector<G1> many = params.commit_many({&sc, &col2, &col3});
    EXPECT(many.size() == 3 && many[0] == c1 && many[1] == params.commit(col2) && many[2] == (G1{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}));
    std::vector<G1> many_l = params.commit_many({&col2, &sc}, /*lagrange=*/true);
    EXPECT(many_l[1] == c2 && many_l[0] == params.commit_lagrange(col2));
    G1 wj; orc_best_multiexp(wj.data(), col2.data(), bases.data(), n, 4); G1Affine wa; orc_g1_to_affine(wa.data(), wj.data());
    EXPECT(std::memcmp(many[1].data(), wa.data(), 64) == 0);
    std::vector<Fr> shorter(n - 1);
    threw = false; try { params.commit_many({&sc, &shorter}); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw);
    // resident polynomials (DevicePoly): the same commitments without the scalars crossing PCIe again; move semantics; evaluation
    {
      DevicePoly d = DevicePoly::from_host(sc);
      EXPECT(params.commit(d) == c1 && params.commit_lagrange(d) == c2);
      DevicePoly moved = std::move(d);
      EXPECT(d.p == nullptr && moved.n == n && moved.to_host() == sc);
      const Fr x = rand_fr(rng);
      EXPECT(moved.eval(x) == eval_polynomial(sc, x));
      threw = false; try { params.commit_lagrange(DevicePoly(n / 2)); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw);
    }
  }
  // --- best_fft / EvaluationDomain against the oracle (raw Montgomery bytes)
  EvaluationDomain dom(4, k);
  std::vector<Fr> a(n); for (auto &x : a) x = rand_fr(rng);
  std::vector<Fr> f = a, wf = a;
  dom.coeff_to_lagrange(f); orc_best_fft(wf.data(), dom.omega.data(), k, 4);
  EXPECT(f == wf);
  std::vector<Fr> back = f; dom.lagrange_to_coeff(back); EXPECT(back == a);
  std::vector<Fr> ext = dom.coeff_to_extended(a), wext(dom.extended_len());
  orc_coeff_to_extended(wext.data(), a.data(), k, dom.extended_k, dom.g_coset.data(), dom.g_coset_inv.data(), dom.extended_omega.data(), 4);
  EXPECT(ext == wext);
  std::vector<Fr> coeffs = dom.extended_to_coeff(ext);
  EXPECT(coeffs.size() == n * 3); EXPECT(std::equal(a.begin(), a.end(), coeffs.begin()));
  for (size_t i = n; i < coeffs.size(); i++) EXPECT((coeffs[i] == Fr{0, 0, 0, 0}));
  // --- eval_polynomial
  Fr x = rand_fr(rng), ev = eval_polynomial(a, x), wev; orc_eval_polynomial(wev.data(), a.data(), n, x.data());
  EXPECT(ev == wev);
  EXPECT(f[5] == eval_polynomial(a, mi355zk::halo2::detail::fr_pow(dom.omega, 5)));   // a'[i] = a(omega^i)
  // --- best_fft over G1 and ParamsKZG::downsize against the oracle
  {
    const uint32_t k5 = 5; const uint64_t n5 = 32;
    EvaluationDomain d5(3, k5);
    std::vector<G1> jac(n5), wj(n5);
    const Fr one_q = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};   // R mod p
    for (uint64_t i = 0; i < n5; i++) { std::memcpy(jac[i].data(), bases[i].data(), 64); std::memcpy(jac[i].data() + 8, one_q.data(), 32); }
    wj = jac; orc_best_fft_g1(wj.data(), d5.omega.data(), k5);
    best_fft(jac, d5.omega, k5);
    for (uint64_t i = 0; i < n5; i++) { G1Affine w; orc_g1_to_affine(w.data(), wj[i].data()); EXPECT(std::memcmp(jac[i].data(), w.data(), 64) == 0 && std::memcmp(jac[i].data() + 8, one_q.data(), 32) == 0); }
    // Curve::batch_normalize on the oracle's un-normalised DFT output (z != 1) plus an identity
    { std::vector<G1> pj = wj; pj[3] = G1{}; std::vector<G1Affine> qa(n5), wa(n5); batch_normalize(pj, qa);
      for (uint64_t i = 0; i < n5; i++) orc_g1_to_affine(wa[i].data(), pj[i].data());
      EXPECT(qa == wa); EXPECT(qa[3] == G1Affine{});
      threw = false; try { std::vector<G1Affine> shorter(n5 - 1); batch_normalize(pj, shorter); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw); }
    std::vector<G1Affine> rev(bases.rbegin(), bases.rend());
    ParamsKZG params(k, bases, rev);
    params.downsize(k5);
    EXPECT(params.k == k5 && params.n == n5);
    std::vector<G1Affine> want(n5), got = params.get_g_lagrange(), g5 = params.get_g();
    orc_g_to_lagrange(want.data(), bases.data(), k5, d5.omega_inv.data(), d5.ifft_divisor.data());
    EXPECT(got == want); EXPECT(std::equal(g5.begin(), g5.end(), bases.begin()));
    std::vector<Fr> s5(sc.begin(), sc.begin() + n5);
    G1 c = params.commit_lagrange(s5), wc; orc_best_multiexp(wc.data(), s5.data(), want.data(), n5, 1); G1Affine wca; orc_g1_to_affine(wca.data(), wc.data());
    EXPECT(std::memcmp(c.data(), wca.data(), 64) == 0);
    threw = false; try { params.downsize(k5 + 1); } catch (const std::invalid_argument &) { threw = true; } EXPECT(threw);
  }
  // --- ParamsKZG::read: a RawBytes params file written here, streamed into HBM by the library
  {
    const uint32_t k6 = 6; const uint64_t n6 = 64;
    const char *path = "/tmp/mi355_cpp_params6";
    FILE *f = std::fopen(path, "wb");
    const uint8_t hdr[4] = {6, 0, 0, 0}; std::vector<uint8_t> tail(256); for (size_t i = 0; i < 256; i++) tail[i] = (uint8_t)i;
    std::fwrite(hdr, 1, 4, f); std::fwrite(bases.data(), 64, n6, f); std::fwrite(bases.data() + n6, 64, n6, f); std::fwrite(tail.data(), 1, 256, f); std::fclose(f);
    auto p6 = ParamsKZG::read(path, /*validate=*/true);
    EXPECT(p6->k == k6 && p6->n == n6 && p6->g2[5] == 5 && p6->s_g2[0] == 128);
    std::vector<G1Affine> g6 = p6->get_g(), gl6 = p6->get_g_lagrange();
    EXPECT(std::equal(g6.begin(), g6.end(), bases.begin()) && std::equal(gl6.begin(), gl6.end(), bases.begin() + n6));
    std::vector<Fr> s6(sc.begin(), sc.begin() + n6);
    G1 c6 = p6->commit(s6), w6; orc_best_multiexp(w6.data(), s6.data(), bases.data(), n6, 1); G1Affine w6a; orc_g1_to_affine(w6a.data(), w6.data());
    EXPECT(std::memcmp(c6.data(), w6a.data(), 64) == 0);
    f = std::fopen(path, "ab"); std::fputc(0, f); std::fclose(f);                       // one byte too long: rejected like load_params
    threw = false; try { ParamsKZG::read(path); } catch (const Error &e) { threw = e.code == MI355_EBADARG; } EXPECT(threw);
    std::remove(path);
  }
  std::printf(failures ? "FAILED (%d)\n" : "all checks passed\n", failures);
  return failures ? 1 : 0;
}
