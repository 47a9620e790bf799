// lib_ntt.hip -- libmi355zk.so, the Fr translation unit: launch orchestration of ntt29.hpp (plan cache, <= 3 global passes), the
// EvaluationDomain wrappers (ifft, coset extension and its inverse), distribute_powers, the element-wise vector operations, the gate-shaped
// fused evaluation (mi355_fr_gate_eval_dev), eval_polynomial, and the batched / replicated entry points that spread independent transforms
// over the bound devices.  Host logic only; all arithmetic runs in the kernels.
// kernel headers first: lib_common.hpp defines the macro `g` (the calling thread's device context), a name the kernels use for locals
#include "ntt.hpp"
#include "ntt29.hpp"
#include "lib_common.hpp"

namespace mi355 {

int ntt_tu_init_device() {
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_strided<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_final<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return MI355_OK;
}

// ------------------------------------------------------------------------------------------------ NTT
constexpr uint32_t NTT_DIRECT_TW_MAX_LOG = 20;   // 2^20 x 36 B = 38 MB per table at most
std::string plan_key(uint32_t log_n, const void *omega) { std::string k((const char *)omega, 32); k.push_back((char)log_n); return k; }

int pow_table29(NttPlan &p, Tw29 *out, const fe_t &base, uint64_t step, uint32_t count) {
  uint4 *lo, *hi; uint32_t *top;
  CHK(dev_malloc((void **)&lo, (size_t)count * 16, "ntt twiddles")); p.owned.push_back(lo);
  CHK(dev_malloc((void **)&hi, (size_t)count * 16, "ntt twiddles")); p.owned.push_back(hi);
  CHK(dev_malloc((void **)&top, (size_t)count * 4, "ntt twiddles")); p.owned.push_back(top);
  hipLaunchKernelGGL(k_pow_table29, dim3(ceil_div(count, 256)), dim3(256), 0, g.stream, lo, hi, top, base, step, count);
  HIPCHK(hipGetLastError());
  out->lo = lo; out->hi = hi; out->top = top;
  return MI355_OK;
}

int launch_pow_table(fe_t *out, const fe_t &base, uint64_t step, uint32_t count) {
  hipLaunchKernelGGL(k_pow_table, dim3(ceil_div(count, 256)), dim3(256), 0, g.stream, out, base, step, count);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}

static void free_plan_tables(NttPlan &p) {
  for (void *q : p.owned) (void)hipFree(q);
  p.owned.clear();
}
// big twiddle tables (256 MB and more) are only built into HBM that is really spare: the table plus max(16 GiB, 1/12 of the device) must be free -- a multi-layer prover process
// that peaks at 274 GiB of 288 builds none of them and takes the table-free form of the same pass (the lo x hi product, or the separate coset shift)
static bool hbm_spare_for_table(uint64_t entries) {
  if (entries * 36 < (256ull << 20)) return true;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
  return fr >= entries * 36 + std::max<size_t>((size_t)16 << 30, tot / 12);
}
// three device arrays of one 29-bit table (SoA), all or nothing
static bool alloc_tw29(uint64_t cnt, uint4 **lo, uint4 **hi, uint32_t **top) {
  *lo = *hi = nullptr; *top = nullptr;
  if (hipMalloc((void **)lo, cnt * 16) == hipSuccess && hipMalloc((void **)hi, cnt * 16) == hipSuccess && hipMalloc((void **)top, cnt * 4) == hipSuccess) return true;
  (void)hipGetLastError();
  if (*lo) (void)hipFree(*lo); if (*hi) (void)hipFree(*hi); if (*top) (void)hipFree(*top);
  return false;
}
static int build_plan(NttPlan &p, uint32_t log_n, const void *omega) {
  if (log_n <= 8) { p.levels = 1; p.log_m[0] = log_n; }
  else if (log_n <= g.ntt_two_level_max_log) { p.levels = 2; p.log_m[0] = (log_n + 1) / 2; p.log_m[1] = log_n / 2; }   // two passes up to 2^18 (2^20 as an A/B knob: 1024-point columns, two adjacent columns per tile)
  else { p.levels = 3; p.log_m[0] = (log_n + 2) / 3; p.log_m[1] = (log_n + 1) / 3; p.log_m[2] = log_n / 3; }
  fe_t w; memcpy(&w, omega, 32);
  const uint64_t N = 1ull << log_n;
  uint32_t log_s = log_n;
  for (uint32_t l = 0; l < p.levels; l++) {
    const uint32_t lm = p.log_m[l];
    CHK(pow_table29(p, &p.tw29_m[l], w, N >> lm, std::max(1u, (1u << lm) >> 1)));
    if (l + 1 < p.levels) {
      // inter-level twiddles w_S^e, e < 2^log_s: ONE table when it is small enough to live in L2 (no lo x hi product per element),
      // otherwise the usual two half-size tables
      p.split[l] = log_s <= NTT_DIRECT_TW_MAX_LOG ? log_s : (log_s + 1) / 2;
      bool direct = false;
      if (log_s >= g.ntt_direct2_min_log && log_s <= g.ntt_direct2_max_log) {
        // big level: every twiddle w_S^(column k) once, in the order the pass reads them (36 B x 2^log_s: 2.4 GB at 2^26, read coalesced
        // next to the data by a pass that is ALU-bound); saves the lo x hi product per element.  HBM may be full of window tables: when the
        // allocation fails the level falls back to the lo x hi pair, which is functionally equivalent.
        const uint64_t cnt = 1ull << log_s; uint4 *lo, *hi; uint32_t *top;
        if (hbm_spare_for_table(cnt) && alloc_tw29(cnt, &lo, &hi, &top)) {
          p.owned.push_back(lo); p.owned.push_back(hi); p.owned.push_back(top);
          hipLaunchKernelGGL(k_pow_table29_2d, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, g.stream, lo, hi, top, Fr::pow_u64(w, N >> log_s), log_s - lm, cnt);
          HIPCHK(hipGetLastError());
          p.tw29_s_lo[l].lo = lo; p.tw29_s_lo[l].hi = hi; p.tw29_s_lo[l].top = top; p.tw29_s_hi[l] = p.tw29_s_lo[l]; p.direct2[l] = 1;
          direct = true;
        }
      }
      if (!direct) {
        CHK(pow_table29(p, &p.tw29_s_lo[l], w, N >> log_s, 1u << p.split[l]));
        CHK(pow_table29(p, &p.tw29_s_hi[l], w, (N >> log_s) << p.split[l], 1u << (log_s - p.split[l])));
      }
    }
    log_s -= lm;
  }
  return MI355_OK;
}
int get_plan(uint32_t log_n, const void *omega, NttPlan **out) {
  const std::string key = plan_key(log_n, omega);
  auto it = g.ntt_plans.find(key);
  if (it != g.ntt_plans.end()) { *out = &it->second; return MI355_OK; }
  NttPlan p; p.log_n = log_n;
  const int rc = build_plan(p, log_n, omega);
  if (rc != MI355_OK) { (void)hipStreamSynchronize(g.stream); free_plan_tables(p); return rc; }   // nothing of a half-built plan is kept
  g.ntt_plans[key] = p; *out = &g.ntt_plans[key];
  return MI355_OK;
}

// radix-4 register rounds (two DIF stages per LDS round trip): one work item per 4 elements.  (The radix-2 and radix-8 instantiations and the raw-scratch modes were A/B paths
// of rounds 3-4 -- results in HISTORY.md section 5 -- and left the library in round 6.)
#define NTT29_LAUNCH(KERN, BLOCKS, TILE, LDS, ...) hipLaunchKernelGGL(KERN<2>, dim3(BLOCKS), dim3(std::max(64u, std::min(512u, (TILE) / 4))), LDS, s, __VA_ARGS__)

uint32_t cols_for(uint32_t log_m) { uint32_t lc = 3; while (lc > 0 && log_m + lc > g.ntt_tile_log) lc--; return lc; }

// an inverse transform's divisor (the three post-scaling constants equal) is folded into the inter-level twiddles of the last strided
// pass: one table of 2^log_s entries per (plan, divisor), and the closing pass ends with reduce_small instead of a multiplication.
// On success *fold_tw names the scaled table and *post3_dev is cleared; otherwise both stay as they were (the divisor remains a multiplication).
static int fold_divisor(NttPlan *p, uint32_t log_n, const fe_t *pre3_host, const fe_t *post3_host, const Tw29 **fold_tw, fe_t **post3_dev) {
  hipStream_t s = g.stream;
  if (post3_host && !pre3_host && g.ntt_fold_scale && p->levels >= 2 && memcmp(&post3_host[0], &post3_host[1], 32) == 0 && memcmp(&post3_host[0], &post3_host[2], 32) == 0) {
    const uint32_t l = p->levels - 2;
    uint32_t log_sl = log_n; for (uint32_t q = 0; q < l; q++) log_sl -= p->log_m[q];
    if ((p->split[l] == log_sl && !p->direct2[l]) || (p->direct2[l] && log_sl <= 22)) {   // that level reads ONE table of 2^log_sl entries (gathered by e = column k, or laid out [k][column]: scaling is element-wise either way; a big level's 2-D table is not duplicated per divisor)
      const std::string key((const char *)&post3_host[0], 32);
      auto it = p->scaled.find(key);
      if (it == p->scaled.end()) {
        const uint32_t cnt = 1u << log_sl; uint4 *lo, *hi; uint32_t *top;
        if (alloc_tw29(cnt, &lo, &hi, &top)) {
          p->owned.push_back(lo); p->owned.push_back(hi); p->owned.push_back(top);
          hipLaunchKernelGGL(k_scale_table29, dim3(ceil_div(cnt, 256)), dim3(256), 0, s, p->tw29_s_lo[l].lo, p->tw29_s_lo[l].hi, p->tw29_s_lo[l].top, lo, hi, top, post3_host[0], cnt);
          HIPCHK(hipGetLastError());
          Tw29 t; t.lo = lo; t.hi = hi; t.top = top;
          it = p->scaled.emplace(key, t).first;
        }
      }
      if (it != p->scaled.end()) { *fold_tw = &it->second; *post3_dev = nullptr; }   // allocation failed: the divisor stays a multiplication in the closing pass
    }
  }
  return MI355_OK;
}

// The coset shift a[i] *= f^i of coeff_to_extended_part folded into the FIRST strided pass (round 6): i = m 2^log_t + column there, so f^i = (f^(2^log_t))^m --
// a table of 2^log_m entries applied on load -- times f^column, which rides on the pass's inter-level twiddles (one [k][column] table per coset factor, the layout big levels
// already read).  One multiplication per element inside an ALU-bound pass instead of a pass of its own (2 multiplications + 64 B of HBM traffic per element).  Returns nullptr
// where the fold does not apply (single-pass plans, sizes above MI355_NTT_COSET_FOLD_MAX_LOG, the A/B kernels, no memory for the table, more than 16 factors on one plan): the
// caller then runs k_distribute_powers first, as before.  Same bits either way (exact field arithmetic, canonical output).
static const NttPlan::CosetTw *coset_fold_tables(NttPlan *p, uint32_t log_n, const void *omega, const void *factor) {
  if (p->levels < 2 || log_n > g.ntt_coset_fold_max_log) return nullptr;
  const std::string key((const char *)factor, 32);
  auto it = p->coset.find(key);
  if (it != p->coset.end()) return &it->second;
  if (p->coset.size() >= 16) return nullptr;
  const uint32_t lm = p->log_m[0], log_t = log_n - lm; const uint64_t cnt = 1ull << log_n;
  fe_t f, w; memcpy(&f, factor, 32); memcpy(&w, omega, 32);
  NttPlan::CosetTw T;
  uint4 *lo, *hi; uint32_t *top;
  // big tables (0.6 GB per factor at 2^24, 2.4 GB at 2^26) are built only into HBM that is really spare (hbm_spare_for_table): the first coset transform of a proof runs when most
  // of the proof's working set is already allocated, so what is free here is close to what stays free; below the margin the shift stays the separate pass
  if (!hbm_spare_for_table(cnt)) return nullptr;
  if (!alloc_tw29(cnt, &lo, &hi, &top)) return nullptr;
  p->owned.push_back(lo); p->owned.push_back(hi); p->owned.push_back(top);
  hipLaunchKernelGGL(k_pow_table29_2d, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, g.stream, lo, hi, top, w, log_t, cnt, f, 1);   // level 0: w_S = omega (S = N)
  if (hipGetLastError() != hipSuccess) return nullptr;
  T.s2d.lo = lo; T.s2d.hi = hi; T.s2d.top = top;
  if (pow_table29(*p, &T.in, Fr::pow_u64(f, 1ull << log_t), 1, 1u << lm) != MI355_OK) return nullptr;
  return &p->coset.emplace(key, T).first->second;
}

// dst[2^log_n] = NTT_omega( pre3-scaled, zero-padded src[src_len] ), then optional post3 scaling.  src may equal dst.
// coset: the tables of coset_fold_tables -- the transform is then of src[i] f^i (first pass; multi-pass plans only)
int ntt_dev_impl(const fe_t *src, uint64_t src_len, fe_t *dst, uint32_t log_n, const void *omega, const fe_t *pre3_host, const fe_t *post3_host, const NttPlan::CosetTw *coset = nullptr) {
  if (log_n > 28) return fail(MI355_EBADARG, "ntt: log_n > 28 (BN254 Fr two-adicity)");
  const uint64_t N = 1ull << log_n;
  hipStream_t s = g.stream;
  fe_t *pre3 = nullptr, *post3 = nullptr;
  if (pre3_host || post3_host) {
    fe_t *c; CHK(ws_get("ntt.consts", 6 * sizeof(fe_t), (void **)&c));
    if (pre3_host) { HIPCHK(hipMemcpyAsync(c, pre3_host, 3 * sizeof(fe_t), hipMemcpyHostToDevice, s)); pre3 = c; }
    if (post3_host) { HIPCHK(hipMemcpyAsync(c + 3, post3_host, 3 * sizeof(fe_t), hipMemcpyHostToDevice, s)); post3 = c + 3; }
    HIPCHK(hipStreamSynchronize(s));  // the host copies may be stack temporaries of the caller
  }
  if (log_n == 0) {
    if (src != dst || pre3 || post3 || src_len < 1) {
      // size-1 transform = identity (apart from scalings); handle through the generic final kernel
    }
  }
  NttPlan *p; CHK(get_plan(log_n, omega, &p));
  const Tw29 *fold_tw = nullptr;
  CHK(fold_divisor(p, log_n, pre3_host, post3_host, &fold_tw, &post3));
  if (coset && (p->levels < 2 || fold_tw || pre3)) return fail(MI355_EBADARG, "ntt: a folded coset shift needs a multi-pass plan and no other scaling");
  CallTrace tr("ntt_fr", N, 64.0);
  Scope total("ntt_total");
  if (p->levels == 1) {
    const uint32_t lm = p->log_m[0], tile = 1u << lm;
    Scope sc("ntt_pass");
    NTT29_LAUNCH(k_ntt29_final, 1u, tile, (size_t)36 * (tile + 1), src, dst, lm, 0u, 0u, 0u, p->tw29_m[0], src_len, pre3, post3);
  } else {
    fe_t *scratch; CHK(ws_get("ntt.scratch", N * sizeof(fe_t), (void **)&scratch));
    uint32_t log_s = log_n;
    const fe_t *cur = src; uint64_t cur_len = src_len; const fe_t *cur_pre = pre3;
    for (uint32_t l = 0; l + 1 < p->levels; l++) {
      Ntt29Level L9; L9.log_m = p->log_m[l]; L9.log_t = log_s - L9.log_m; L9.split = p->split[l]; L9.tw_m = p->tw29_m[l]; L9.tw_s_lo = p->tw29_s_lo[l]; L9.tw_s_hi = p->tw29_s_hi[l];
      L9.direct = p->direct2[l] ? 2u : (p->split[l] == log_s) ? 1u : 0u;
      if (fold_tw && l + 2 == p->levels) L9.tw_s_lo = *fold_tw;
      if (coset && l == 0) { L9.tw_in = coset->in; L9.has_in = 1; L9.tw_s_lo = coset->s2d; L9.tw_s_hi = coset->s2d; L9.direct = 2; }
      const uint32_t lc = std::min(cols_for(L9.log_m), L9.log_t), tile = 1u << (L9.log_m + lc);
      const uint64_t blocks = (N >> log_s) << (L9.log_t - lc);
      Scope sc("ntt_pass");
      NTT29_LAUNCH(k_ntt29_strided, (uint32_t)blocks, tile, (size_t)36 * tile, cur, scratch, L9, lc, cur_len, cur_pre);
      cur = scratch; cur_len = N; cur_pre = nullptr; log_s -= L9.log_m;
    }
    const uint32_t lm = p->log_m[p->levels - 1], log_a = p->log_m[0], log_b = p->levels == 3 ? p->log_m[1] : 0;
    const uint32_t lc = std::min(cols_for(lm), log_a), tile = 1u << (lm + lc);
    const uint64_t blocks = ((uint64_t)1 << log_b) << (log_a - lc);
    Scope sc("ntt_pass");
    NTT29_LAUNCH(k_ntt29_final, (uint32_t)blocks, tile, (size_t)36 * (((size_t)1 << lm) + 1) * ((size_t)1 << lc), cur, dst, lm, log_a, log_b, lc, p->tw29_m[p->levels - 1], N, (const fe_t *)nullptr, post3);
  }
  HIPCHK(hipGetLastError());
  total.close();
  tr.done();
  return MI355_OK;
}


// `data.size()` in-place transforms of 2^log_n elements (optionally times a divisor) as batched launches: blockIdx.y = vector.  Used by the batch entry
// points for small transforms; falls back to the loop of single transforms where the batched kernels do not apply.  Same results.
// srcs / coset: data[i] = transform of srcs[i][j] f^j (the folded coset shift; srcs[i] is only read, so it may be the coefficient vector itself)
int ntt_batch_inplace(const std::vector<fe_t *> &data, uint32_t log_n, const void *omega, const void *divisor, const std::vector<const fe_t *> *srcs = nullptr, const NttPlan::CosetTw *coset = nullptr) {
  const size_t cnt = data.size();
  auto single_loop = [&]() -> int {
    if (coset) { for (size_t i = 0; i < cnt; i++) CHK(ntt_dev_impl((*srcs)[i], 1ull << log_n, data[i], log_n, omega, nullptr, nullptr, coset)); return MI355_OK; }
    for (fe_t *d : data) {
      if (!divisor) CHK(ntt_dev_impl(d, 1ull << log_n, d, log_n, omega, nullptr, nullptr));
      else { fe_t post[3]; for (int i = 0; i < 3; i++) memcpy(&post[i], divisor, 32); CHK(ntt_dev_impl(d, 1ull << log_n, d, log_n, omega, nullptr, post)); }
    }
    return MI355_OK;
  };
  if (cnt < 2 || log_n > g.ntt_batch_max_log || log_n > 28) return single_loop();
  NttPlan *p; CHK(get_plan(log_n, omega, &p));
  if (p->levels < 2) return single_loop();
  const uint64_t N = 1ull << log_n;
  hipStream_t s = g.stream;
  fe_t post_host[3]; fe_t *post3 = nullptr; const Tw29 *fold_tw = nullptr;
  if (divisor) {
    for (int i = 0; i < 3; i++) memcpy(&post_host[i], divisor, 32);
    fe_t *c; CHK(ws_get("ntt.consts", 6 * sizeof(fe_t), (void **)&c));
    HIPCHK(hipMemcpyAsync(c + 3, post_host, 3 * sizeof(fe_t), hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s));
    post3 = c + 3;
    CHK(fold_divisor(p, log_n, nullptr, post_host, &fold_tw, &post3));
  }
  // in-place transforms of ONE buffer listed twice must run one after the other (the serial loop applied the transform twice; concurrent passes would race)
  { std::vector<const void *> seen(data.begin(), data.end()); std::sort(seen.begin(), seen.end()); if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return single_loop(); }
  const size_t chunk = std::min<size_t>(std::min<size_t>(cnt, 65535), std::max<size_t>(1, (size_t)((1ull << 30) / (N * sizeof(fe_t)))));   // at most 1 GiB of scratch, and gridDim.y <= 65535
  fe_t *scratch; CHK(ws_get("ntt.scratch.batch", chunk * N * sizeof(fe_t), (void **)&scratch));
  // pointer tables for the whole list: data[i] and the scratch slot of i (slots repeat chunk by chunk; the stream orders their reuse)
  std::vector<const fe_t *> tab(3 * cnt);
  for (size_t i = 0; i < cnt; i++) { tab[i] = data[i]; tab[cnt + i] = scratch + (i % chunk) * N; tab[2 * cnt + i] = coset ? (*srcs)[i] : data[i]; }   // [destination | scratch slot | first-pass source]
  const fe_t **dtab; CHK(ws_get("ntt.batch.ptrs", 3 * cnt * sizeof(void *), (void **)&dtab));
  HIPCHK(hipMemcpyAsync(dtab, tab.data(), 3 * cnt * sizeof(void *), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));   // `tab` is a stack-lifetime vector
  CallTrace tr("ntt_fr_batch", N * cnt, 64.0);
  for (size_t base = 0; base < cnt; base += chunk) {
    const uint32_t c = (uint32_t)std::min(chunk, cnt - base);
    const fe_t *const *d_data = dtab + base; fe_t *const *d_scr = (fe_t *const *)(dtab + cnt + base);
    uint32_t log_s = log_n;
    for (uint32_t l = 0; l + 1 < p->levels; l++) {
      Ntt29Level L9; L9.log_m = p->log_m[l]; L9.log_t = log_s - L9.log_m; L9.split = p->split[l]; L9.tw_m = p->tw29_m[l]; L9.tw_s_lo = p->tw29_s_lo[l]; L9.tw_s_hi = p->tw29_s_hi[l];
      L9.direct = p->direct2[l] ? 2u : (p->split[l] == log_s) ? 1u : 0u;
      if (fold_tw && l + 2 == p->levels) L9.tw_s_lo = *fold_tw;
      if (coset && l == 0) { L9.tw_in = coset->in; L9.has_in = 1; L9.tw_s_lo = coset->s2d; L9.tw_s_hi = coset->s2d; L9.direct = 2; }
      const uint32_t lc = std::min(cols_for(L9.log_m), L9.log_t), tile = 1u << (L9.log_m + lc);
      const uint64_t blocks = (N >> log_s) << (L9.log_t - lc);
      Scope sc("ntt_pass");
      const NttBatch B{l == 0 ? (const fe_t *const *)(dtab + 2 * cnt + base) : (const fe_t *const *)d_scr, d_scr};
      hipLaunchKernelGGL((k_ntt29_strided<2, 0>), dim3((uint32_t)blocks, c), dim3(std::max(64u, std::min(512u, tile / 4))), (size_t)36 * tile, s, (const fe_t *)nullptr, (fe_t *)nullptr, L9, lc, N, (const fe_t *)nullptr, Raw29{nullptr, nullptr, nullptr}, B);
      log_s -= L9.log_m;
    }
    const uint32_t lm = p->log_m[p->levels - 1], log_a = p->log_m[0], log_b = p->levels == 3 ? p->log_m[1] : 0;
    const uint32_t lc = std::min(cols_for(lm), log_a), tile = 1u << (lm + lc);
    const uint64_t blocks = ((uint64_t)1 << log_b) << (log_a - lc);
    Scope sc("ntt_pass");
    const NttBatch B{(const fe_t *const *)d_scr, (fe_t *const *)d_data};
    hipLaunchKernelGGL((k_ntt29_final<2, 0>), dim3((uint32_t)blocks, c), dim3(std::max(64u, std::min(512u, tile / 4))), (size_t)36 * (((size_t)1 << lm) + 1) * ((size_t)1 << lc), s, (const fe_t *)nullptr, (fe_t *)nullptr, lm, log_a, log_b, lc,
                       p->tw29_m[p->levels - 1], N, (const fe_t *)nullptr, (const fe_t *)post3, Raw29{nullptr, nullptr, nullptr}, B);
    HIPCHK(hipGetLastError());
  }
  tr.done();
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

namespace {
struct BatchItem { uint32_t index; int slot; };
// run(slot, index) for every item, one thread per device that has items; the first error wins
template <class F> int run_per_device_lists(const std::vector<BatchItem> &items, F run_list) {
  std::vector<std::vector<uint32_t>> by_slot(MAX_DEV);
  for (const auto &it : items) by_slot[it.slot].push_back(it.index);
  std::vector<int> slots; for (int s = 0; s < MAX_DEV; s++) if (!by_slot[s].empty()) slots.push_back(s);
  std::vector<int> rcs(slots.size(), MI355_OK); std::vector<std::string> errs(slots.size());
  auto work = [&](size_t k) {
    const int slot = slots[k];
    rcs[k] = guarded([&]() -> int {
      DevGuard lk(slot);
      CHK(need_init(slot));
      CHK(run_list(slot, by_slot[slot]));
      HIPCHK(hipStreamSynchronize(g.stream));
      resolve_spans();
      return MI355_OK;
    }, "batch worker (one per device)");
    if (rcs[k] != MI355_OK) errs[k] = g_err;
  };
  {
    struct Joiner { std::vector<std::thread> th; ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); } } workers;
    for (size_t k = 1; k < slots.size(); k++) workers.th.emplace_back(work, k);
    if (!slots.empty()) work(0);
  }
  for (size_t k = 0; k < slots.size(); k++) if (rcs[k] != MI355_OK) return fail(rcs[k], "device slot " + std::to_string(slots[k]) + ": " + errs[k]);
  return MI355_OK;
}
template <class F> int run_per_device(const std::vector<BatchItem> &items, F run) {
  return run_per_device_lists(items, [&](int slot, const std::vector<uint32_t> &idx) -> int { for (uint32_t i : idx) CHK(run(slot, i)); return MI355_OK; });
}
int transform_in_place(fe_t *data, uint32_t log_n, const void *omega, const void *divisor) {
  if (!divisor) return ntt_dev_impl(data, 1ull << log_n, data, log_n, omega, nullptr, nullptr);
  fe_t post[3]; for (int i = 0; i < 3; i++) memcpy(&post[i], divisor, 32);
  return ntt_dev_impl(data, 1ull << log_n, data, log_n, omega, nullptr, post);
}
// Host-pointer batch on ONE device (called with the device's lock held): upload of item i + 1 | transform of item i | download of item i - 1.
// PCIe is full duplex, but a copy from / to pageable memory blocks its calling thread for the whole transfer, so the downloads run on a helper
// thread with its own stream; two staging buffers alternate.  Results are those of the serial loop.
int ntt_host_pipeline(const std::vector<uint32_t> &idx, void *const *data_host, size_t bytes, uint32_t log_n, const void *omega, const void *divisor) {
  const size_t n = idx.size();
  void *dev[2] = {nullptr, nullptr};
  CHK(ws_get("io.ntt", bytes, &dev[0]));
  if (n < 2 || g.host_batch_overlap == 0) {
    for (uint32_t i : idx) {
      HIPCHK(hipMemcpyAsync(dev[0], data_host[i], bytes, hipMemcpyHostToDevice, g.stream));
      CHK(transform_in_place((fe_t *)dev[0], log_n, omega, divisor));
      HIPCHK(hipMemcpyAsync(data_host[i], dev[0], bytes, hipMemcpyDeviceToHost, g.stream));
      HIPCHK(hipStreamSynchronize(g.stream));   // the staging buffer is reused by the next item
    }
    return MI355_OK;
  }
  CHK(ws_get("io.ntt.b", bytes, &dev[1]));
  struct Events { hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    ~Events() { for (int i = 0; i < 2; i++) { if (up[i]) (void)hipEventDestroy(up[i]); if (done[i]) (void)hipEventDestroy(done[i]); } } } ev;
  for (int i = 0; i < 2; i++) { HIPCHK(hipEventCreateWithFlags(&ev.up[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev.done[i], hipEventDisableTiming)); }
  struct Shared { std::mutex mu; std::condition_variable cv; size_t computed = 0, downloaded = 0; bool abort = false; hipError_t err = hipSuccess; } sh;
  const int device = g.device; hipStream_t down_stream = g.aux_stream[0] ? g.aux_stream[0] : g.copy_stream;
  std::thread downloader([&]() {
    hipError_t e = hipSetDevice(device);
    for (size_t i = 0; i < n && e == hipSuccess; i++) {
      { std::unique_lock<std::mutex> lk(sh.mu); sh.cv.wait(lk, [&] { return sh.computed > i || sh.abort; }); if (sh.abort) return; }
      e = hipStreamWaitEvent(down_stream, ev.done[i & 1], 0);
      if (e == hipSuccess) e = hipMemcpyAsync(data_host[idx[i]], dev[i & 1], bytes, hipMemcpyDeviceToHost, down_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(down_stream);
      { std::lock_guard<std::mutex> lk(sh.mu); if (e == hipSuccess) sh.downloaded = i + 1; else { sh.err = e; sh.abort = true; } }
      sh.cv.notify_all();
    }
    if (e != hipSuccess) { { std::lock_guard<std::mutex> lk(sh.mu); sh.err = e; sh.abort = true; } sh.cv.notify_all(); }
  });
  // the helper is joined on every way out of this function (an exception from the loop below -- std::bad_alloc in the plan cache, say -- must not
  // reach a joinable std::thread's destructor)
  struct Joiner { std::thread &t; Shared &sh; ~Joiner() { if (!t.joinable()) return; { std::lock_guard<std::mutex> lk(sh.mu); sh.abort = true; } sh.cv.notify_all(); t.join(); } } joiner{downloader, sh};
  int rc = [&]() -> int {
    for (size_t i = 0; i < n; i++) {
      if (i >= 2) {   // staging buffer i & 1 still holds item i - 2 until its download has finished
        std::unique_lock<std::mutex> lk(sh.mu); sh.cv.wait(lk, [&] { return sh.downloaded + 1 >= i || sh.abort; });
        if (sh.abort) return MI355_EHIP;
      }
      HIPCHK(hipMemcpyAsync(dev[i & 1], data_host[idx[i]], bytes, hipMemcpyHostToDevice, g.copy_stream));
      HIPCHK(hipEventRecord(ev.up[i & 1], g.copy_stream));
      HIPCHK(hipStreamWaitEvent(g.stream, ev.up[i & 1], 0));
      CHK(transform_in_place((fe_t *)dev[i & 1], log_n, omega, divisor));
      HIPCHK(hipEventRecord(ev.done[i & 1], g.stream));
      { std::lock_guard<std::mutex> lk(sh.mu); sh.computed = i + 1; }
      sh.cv.notify_all();
    }
    return MI355_OK;
  }();
  if (rc != MI355_OK) { { std::lock_guard<std::mutex> lk(sh.mu); sh.abort = true; } sh.cv.notify_all(); }
  downloader.join();
  if (sh.err != hipSuccess) return fail(MI355_EHIP, std::string("ntt_batch download: ") + hipGetErrorString(sh.err));
  return rc;
}
}  // namespace

extern "C" {

// ---- NTT
int mi355_ntt_fr_dev(void *data_dev, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({data_dev}, &slot, "ntt")); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(data_dev, log_n, omega));
  CHK(ntt_dev_impl((const fe_t *)data_dev, 1ull << log_n, (fe_t *)data_dev, log_n, omega, nullptr, nullptr));
  return finish_async();
  });
}
int mi355_intt_fr_dev(void *data_dev, uint32_t log_n, const void *omega_inv, const void *divisor) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({data_dev}, &slot, "intt")); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(data_dev, log_n, omega_inv));
  if (!divisor) return fail(MI355_EBADARG, "intt: null divisor");
  fe_t post[3]; for (int i = 0; i < 3; i++) memcpy(&post[i], divisor, 32);
  CHK(ntt_dev_impl((const fe_t *)data_dev, 1ull << log_n, (fe_t *)data_dev, log_n, omega_inv, nullptr, post));
  return finish_async();
  });
}
int mi355_coeff_to_extended_dev(void *dst_dev, const void *coeffs_dev, uint32_t log_n, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({dst_dev, coeffs_dev}, &slot, "coeff_to_extended")); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(dst_dev, log_ext, extended_omega));
  if (!coeffs_dev || !g_coset || !g_coset_inv || log_n > log_ext) return fail(MI355_EBADARG, "coeff_to_extended: bad argument");
  fe_t pre[3]; pre[0] = Fr::one(); memcpy(&pre[1], g_coset, 32); memcpy(&pre[2], g_coset_inv, 32);
  CHK(ntt_dev_impl((const fe_t *)coeffs_dev, 1ull << log_n, (fe_t *)dst_dev, log_ext, extended_omega, pre, nullptr));
  return finish_async();
  });
}
static int extended_to_coeff_locked(void *data_dev, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega_inv, const void *extended_ifft_divisor);
int mi355_extended_to_coeff_dev(void *data_dev, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega_inv, const void *extended_ifft_divisor) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({data_dev}, &slot, "extended_to_coeff")); DevGuard lk(slot);
  CHK(need_init(slot));
  CHK(extended_to_coeff_locked(data_dev, log_ext, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor));
  return finish_async();
  });
}

int mi355_ntt_fr_host(void *data_host, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);   // host-pointer calls may run on any bound device (replicas): callers on different threads land on different devices
  CHK(need_init(slot)); CHK(check_ntt_args(data_host, log_n, omega));
  NttHostArgs a{log_n, omega, nullptr};
  const size_t bytes = sizeof(fe_t) << log_n;
  return with_host_io(data_host, bytes, bytes, bytes, "io.ntt", [](void *dev, void *ud) { auto *a = (NttHostArgs *)ud; return ntt_dev_impl((const fe_t *)dev, 1ull << a->log_n, (fe_t *)dev, a->log_n, a->omega, nullptr, nullptr); }, &a);
  });
}
int mi355_intt_fr_host(void *data_host, uint32_t log_n, const void *omega_inv, const void *divisor) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);   // host-pointer calls may run on any bound device (replicas): callers on different threads land on different devices
  CHK(need_init(slot)); CHK(check_ntt_args(data_host, log_n, omega_inv));
  if (!divisor) return fail(MI355_EBADARG, "intt: null divisor");
  NttHostArgs a{log_n, omega_inv, divisor};
  const size_t bytes = sizeof(fe_t) << log_n;
  return with_host_io(data_host, bytes, bytes, bytes, "io.ntt", [](void *dev, void *ud) {
    auto *a = (NttHostArgs *)ud; fe_t post[3]; for (int i = 0; i < 3; i++) memcpy(&post[i], a->divisor, 32);
    return ntt_dev_impl((const fe_t *)dev, 1ull << a->log_n, (fe_t *)dev, a->log_n, a->omega, nullptr, post); }, &a);
  });
}
int mi355_coeff_to_extended_host(void *dst_host, const void *coeffs_host, uint32_t log_n, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega) {
  return guarded([&]() -> int {
  {
    const int slot = pick_replica_slot(); DevGuard lk(slot);   // host-pointer calls may run on any bound device (replicas): callers on different threads land on different devices
    CHK(need_init(slot)); CHK(check_ntt_args(dst_host, log_ext, extended_omega));
    if (!coeffs_host || !g_coset || !g_coset_inv || log_n > log_ext) return fail(MI355_EBADARG, "coeff_to_extended: bad argument");
    void *src, *dst; CHK(ws_get("io.ntt_src", sizeof(fe_t) << log_n, &src)); CHK(ws_get("io.ntt", sizeof(fe_t) << log_ext, &dst));
    HIPCHK(hipMemcpyAsync(src, coeffs_host, sizeof(fe_t) << log_n, hipMemcpyHostToDevice, g.stream));
    fe_t pre[3]; pre[0] = Fr::one(); memcpy(&pre[1], g_coset, 32); memcpy(&pre[2], g_coset_inv, 32);
    CHK(ntt_dev_impl((const fe_t *)src, 1ull << log_n, (fe_t *)dst, log_ext, extended_omega, pre, nullptr));
    HIPCHK(hipMemcpyAsync(dst_host, dst, sizeof(fe_t) << log_ext, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream)); resolve_spans();
  }
  return MI355_OK;
  });
}
// body of mi355_extended_to_coeff_dev, to be called with the device lock held
static int extended_to_coeff_locked(void *data_dev, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega_inv, const void *extended_ifft_divisor) {
  CHK(check_ntt_args(data_dev, log_ext, extended_omega_inv));
  if (!g_coset || !g_coset_inv || !extended_ifft_divisor) return fail(MI355_EBADARG, "extended_to_coeff: null pointer");
  // post-scale table {d, d * g_coset_inv, d * g_coset}: three constant products formed on the host (setup, not data path)
  fe_t d, gc, gci, post[3]; memcpy(&d, extended_ifft_divisor, 32); memcpy(&gc, g_coset, 32); memcpy(&gci, g_coset_inv, 32);
  post[0] = d; post[1] = Fr::mul(d, gci); post[2] = Fr::mul(d, gc);
  return ntt_dev_impl((const fe_t *)data_dev, 1ull << log_ext, (fe_t *)data_dev, log_ext, extended_omega_inv, nullptr, post);
}
int mi355_extended_to_coeff_host(void *data_host, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega_inv, const void *extended_ifft_divisor) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);   // host-pointer calls may run on any bound device (replicas): callers on different threads land on different devices
  CHK(need_init(slot)); CHK(check_ntt_args(data_host, log_ext, extended_omega_inv));
  void *dev; CHK(ws_get("io.ntt", sizeof(fe_t) << log_ext, &dev));
  HIPCHK(hipMemcpyAsync(dev, data_host, sizeof(fe_t) << log_ext, hipMemcpyHostToDevice, g.stream));
  CHK(extended_to_coeff_locked(dev, log_ext, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor));
  HIPCHK(hipMemcpyAsync(data_host, dev, sizeof(fe_t) << log_ext, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream)); resolve_spans();
  return MI355_OK;
  });
}

// ---- distribute_powers / coset NTT
static DistFactors dist_factors(const void *factor) {   // f, f^256, f^(256 EVAL_RUN): a few dozen host multiplications per call
  DistFactors F; memcpy(&F.f, factor, 32);
  F.f256 = Fr::pow_u64(F.f, 256); F.fblock = Fr::pow_u64(F.f256, EVAL_RUN);
  return F;
}
static int distribute_powers_locked(void *data_dev, uint64_t n, const void *factor, const void *src_dev = nullptr) {
  if (!factor || (n && !data_dev)) return fail(MI355_EBADARG, "distribute_powers: null pointer");
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(k_distribute_powers, dim3(ceil_div(n, (uint64_t)EVAL_RUN * 256)), dim3(256), 0, g.stream, (const fe_t *)(src_dev ? src_dev : data_dev), (fe_t *)data_dev, n, dist_factors(factor));
  HIPCHK(hipGetLastError());
  return MI355_OK;
}
// the coset shift of `cnt` polynomials in one launch: dst[i][j] = src[i][j] * factor^j.  The pointer tables travel through a workspace buffer; the
// caller synchronises the stream before `ptrs` (host) goes away (the batch entry points end with a stream synchronisation).
static int distribute_powers_batch_locked(const std::vector<const void *> &src, const std::vector<void *> &dst, uint64_t n, const void *factor) {
  const size_t cnt = src.size();
  if (cnt == 0 || n == 0) return MI355_OK;
  void **tab; CHK(ws_get("batch.ptrs", 2 * cnt * sizeof(void *), (void **)&tab));
  for (size_t base = 0; base < cnt; base += 65535) {   // grid.y limit
    const size_t c = std::min<size_t>(65535, cnt - base);
    HIPCHK(hipMemcpyAsync(tab, src.data() + base, c * sizeof(void *), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(tab + cnt, dst.data() + base, c * sizeof(void *), hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_distribute_powers_batch, dim3(ceil_div(n, (uint64_t)EVAL_RUN * 256), (uint32_t)c), dim3(256), 0, g.stream, (const fe_t *const *)tab, (fe_t *const *)(tab + cnt), n, dist_factors(factor));
    HIPCHK(hipGetLastError());
    if (base + c < cnt) HIPCHK(hipStreamSynchronize(g.stream));   // the table is reused by the next slice
  }
  return MI355_OK;
}
int mi355_distribute_powers_fr_dev(void *data_dev, uint64_t n, const void *factor) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({data_dev}, &slot, "distribute_powers")); DevGuard lk(slot);
  CHK(need_init(slot));
  return distribute_powers_locked(data_dev, n, factor);
  });
}
int mi355_coset_ntt_fr_dev(void *dst_dev, const void *coeffs_dev, uint32_t log_n, const void *coset_factor, const void *omega) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({dst_dev, coeffs_dev}, &slot, "coset_ntt")); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(dst_dev, log_n, omega));
  if (!coeffs_dev || !coset_factor) return fail(MI355_EBADARG, "coset_ntt: null pointer");
  NttPlan *p; CHK(get_plan(log_n, omega, &p));
  if (const NttPlan::CosetTw *ct = coset_fold_tables(p, log_n, omega, coset_factor)) {
    CHK(ntt_dev_impl((const fe_t *)coeffs_dev, 1ull << log_n, (fe_t *)dst_dev, log_n, omega, nullptr, nullptr, ct));   // the shift rides on the first pass: no k_distribute_powers
    return finish_async();
  }
  CHK(distribute_powers_locked(dst_dev, 1ull << log_n, coset_factor, coeffs_dev));   // dst = coeffs[i] * factor^i (one pass, no copy first)
  CHK(ntt_dev_impl((const fe_t *)dst_dev, 1ull << log_n, (fe_t *)dst_dev, log_n, omega, nullptr, nullptr));
  return finish_async();
  });
}

// ---- batches of independent transforms, spread over the bound devices (the NTT does not shard at k <= 26 -- "replicas only" --
// but the 8 iNTTs and 32 coset NTTs of one layer-4 proof are independent of each other).  Host pointers are dealt round-robin, one worker
// thread, one staging buffer and one PCIe link per device; device pointers run on the device that owns them (mi355_buf_alloc(.., slot)),
// concurrently across devices.  Each worker holds only its own device's lock.  Results are those of the serial loop.

// `batch` x best_fft (divisor == NULL) or EvaluationDomain::ifft (divisor = n^-1), in place, host memory
int mi355_ntt_fr_batch_host(void *const *data_host, uint32_t batch, uint32_t log_n, const void *omega, const void *divisor) {
  return guarded([&]() -> int {
  if (batch == 0) return MI355_OK;
  if (!data_host || !omega || log_n > 28) return fail(MI355_EBADARG, "ntt_batch: bad argument");
  for (uint32_t i = 0; i < batch; i++) if (!data_host[i]) return fail(MI355_EBADARG, "ntt_batch: null polynomial pointer");
  const int D = std::max(1, g_ndev);
  std::vector<BatchItem> items(batch);
  for (uint32_t i = 0; i < batch; i++) items[i] = {i, (int)(i % (uint32_t)D)};
  const size_t bytes = sizeof(fe_t) << log_n;
  return run_per_device_lists(items, [&](int, const std::vector<uint32_t> &idx) -> int { return ntt_host_pipeline(idx, data_host, bytes, log_n, omega, divisor); });
  });
}
// the same on resident polynomials: each transform runs on the device that owns its buffer, asynchronously within a device
int mi355_ntt_fr_batch_dev(void *const *data_dev, uint32_t batch, uint32_t log_n, const void *omega, const void *divisor) {
  return guarded([&]() -> int {
  if (batch == 0) return MI355_OK;
  if (!data_dev || !omega || log_n > 28) return fail(MI355_EBADARG, "ntt_batch: bad argument");
  std::vector<BatchItem> items(batch);
  for (uint32_t i = 0; i < batch; i++) { if (!data_dev[i]) return fail(MI355_EBADARG, "ntt_batch: null polynomial pointer"); items[i] = {i, slot_of(data_dev[i])}; }
  return run_per_device_lists(items, [&](int, const std::vector<uint32_t> &idx) -> int {
    std::vector<fe_t *> list; for (uint32_t i : idx) list.push_back((fe_t *)data_dev[i]);
    return ntt_batch_inplace(list, log_n, omega, divisor);   // small transforms: batched launches (blockIdx.y = polynomial); otherwise the loop of single transforms
  });
  });
}
// `batch` x coeff_to_extended_part: dst[i] = best_fft(coeffs[i][j] * coset_factor^j, omega); dst[i] and coeffs[i] must live on one device
int mi355_coset_ntt_fr_batch_dev(void *const *dst_dev, const void *const *coeffs_dev, uint32_t batch, uint32_t log_n, const void *coset_factor, const void *omega) {
  return guarded([&]() -> int {
  if (batch == 0) return MI355_OK;
  if (!dst_dev || !coeffs_dev || !coset_factor || !omega || log_n > 28) return fail(MI355_EBADARG, "coset_ntt_batch: bad argument");
  std::vector<BatchItem> items(batch);
  for (uint32_t i = 0; i < batch; i++) {
    if (!dst_dev[i] || !coeffs_dev[i]) return fail(MI355_EBADARG, "coset_ntt_batch: null polynomial pointer");
    int slot; CHK(common_slot({dst_dev[i], coeffs_dev[i]}, &slot, "coset_ntt_batch")); items[i] = {i, slot};
  }
  // per device: ONE launch scales every polynomial of the list by the powers of the coset factor (round 4: 946 separate 64-block launches of
  // ~110 us each per coset part of the k = 20 inner circuit were latency, not work), then the transforms run in place one after the other
  return run_per_device_lists(items, [&](int, const std::vector<uint32_t> &idx) -> int {
    std::vector<const void *> src; std::vector<void *> dst;
    for (uint32_t i : idx) { src.push_back(coeffs_dev[i]); dst.push_back(dst_dev[i]); }
    std::vector<fe_t *> list; for (void *d : dst) list.push_back((fe_t *)d);
    NttPlan *p; CHK(get_plan(log_n, omega, &p));
    if (const NttPlan::CosetTw *ct = coset_fold_tables(p, log_n, omega, coset_factor)) {   // round 6: the shift rides on the first pass of every transform of the list
      std::vector<const fe_t *> from; for (const void *q : src) from.push_back((const fe_t *)q);
      return ntt_batch_inplace(list, log_n, omega, nullptr, &from, ct);
    }
    CHK(distribute_powers_batch_locked(src, dst, 1ull << log_n, coset_factor));
    return ntt_batch_inplace(list, log_n, omega, nullptr);
  });
  });
}

// ---- element-wise vector operations on resident polynomials
int mi355_fr_vec_op_dev(int op, void *dst_dev, const void *a_dev, const void *b_dev, uint64_t n) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({dst_dev, a_dev, b_dev}, &slot, "fr_vec_op")); DevGuard lk(slot);
  CHK(need_init(slot));
  if (op < 0 || op > 2 || (n && (!dst_dev || !a_dev || !b_dev))) return fail(MI355_EBADARG, "fr_vec_op: bad argument");
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(k_fr_vec_op, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, op, (fe_t *)dst_dev, (const fe_t *)a_dev, (const fe_t *)b_dev, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
// dst[i] (+)= sum_j coeffs[j] * prod_k polys[factor_poly[..]][(i + factor_rot[..]) mod n]: the gate / permutation / lookup expressions of
// evaluate_h as ONE launch (k_fr_gate_eval).  Term j owns term_len[j] consecutive entries of factor_poly / factor_rot.
int mi355_fr_gate_eval_dev(void *dst_dev, const void *const *polys_dev, uint32_t n_polys, const void *coeffs, const uint32_t *term_len, uint32_t n_terms,
                           const uint32_t *factor_poly, const int32_t *factor_rot, uint64_t n, int accumulate) {
  return guarded([&]() -> int {
  if (!dst_dev || (n_polys && !polys_dev) || (n_terms && (!coeffs || !term_len))) return fail(MI355_EBADARG, "fr_gate_eval: null pointer");
  if (n == 0 || (n & (n - 1))) return fail(MI355_EBADARG, "fr_gate_eval: n must be a power of two (rotations wrap modulo the domain size)");
  if (n_polys > GATE_MAX_POLYS || n_terms > GATE_MAX_TERMS) return fail(MI355_EBADARG, "fr_gate_eval: at most 24 polynomials and 16 terms per launch (split the expression)");
  GatePlan G; memset(&G, 0, sizeof G);
  G.n_terms = n_terms; G.accumulate = accumulate ? 1u : 0u;
  uint32_t nf = 0;
  for (uint32_t j = 0; j < n_terms; j++) {
    if (term_len[j] > GATE_MAX_TERM_LEN || nf + term_len[j] > GATE_MAX_FACTORS) return fail(MI355_EBADARG, "fr_gate_eval: at most 16 factors per term and 48 per launch");
    G.term_len[j] = (uint8_t)term_len[j]; nf += term_len[j];
    memcpy(&G.coeff[j], (const char *)coeffs + 32 * (size_t)j, 32);
    G.coeff29[j] = Fr29::from_sat(G.coeff[j]);
    { const fe_t one = Fr::one(), minus_one = Fr::neg(Fr::one()); G.coeff_kind[j] = memcmp(&G.coeff[j], &one, 32) == 0 ? 1 : memcmp(&G.coeff[j], &minus_one, 32) == 0 ? 2 : 0; }
  }
  if (nf && (!factor_poly || !factor_rot)) return fail(MI355_EBADARG, "fr_gate_eval: null factor list");
  for (uint32_t q = 0; q < nf; q++) {
    if (factor_poly[q] >= n_polys) return fail(MI355_EBADARG, "fr_gate_eval: factor refers to a polynomial outside the list");
    G.factor_poly[q] = (uint8_t)factor_poly[q]; G.factor_rot[q] = factor_rot[q];
  }
  int slot = slot_of(dst_dev);
  for (uint32_t p = 0; p < n_polys; p++) {
    if (!polys_dev[p]) return fail(MI355_EBADARG, "fr_gate_eval: null polynomial pointer");
    int s; CHK(common_slot({dst_dev, polys_dev[p]}, &s, "fr_gate_eval"));
    G.poly[p] = (const fe_t *)polys_dev[p];
    // dst may BE one of the operands only where every thread reads exactly the element it writes: same base address, every factor of that
    // polynomial un-rotated.  Any other overlap is a cross-thread race (thread i would read what thread i - r is writing): rejected.
    const uintptr_t d0 = (uintptr_t)dst_dev, d1 = d0 + n * sizeof(fe_t), p0 = (uintptr_t)polys_dev[p], p1 = p0 + n * sizeof(fe_t);
    if (p0 < d1 && d0 < p1) {
      if (p0 != d0) return fail(MI355_EBADARG, "fr_gate_eval: dst overlaps an operand at an offset");
      for (uint32_t q = 0; q < nf; q++) if (factor_poly[q] == p && (((uint64_t)(int64_t)factor_rot[q]) & (n - 1)) != 0)
        return fail(MI355_EBADARG, "fr_gate_eval: dst aliases an operand that is read with a non-zero rotation");
    }
  }
  DevGuard lk(slot);
  CHK(need_init(slot));
  hipLaunchKernelGGL(k_fr_gate_eval, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, (fe_t *)dst_dev, G, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
// extended-domain vector from its Q coset parts: dst[i * Q + q] = parts[q][i] (dst: Q * n elements, must not overlap a part)
int mi355_fr_interleave_dev(void *dst_dev, const void *const *parts_dev, uint32_t q_parts, uint64_t n) {
  return guarded([&]() -> int {
  if (!dst_dev || !parts_dev || q_parts == 0 || q_parts > 8) return fail(MI355_EBADARG, "fr_interleave: 1..8 parts");
  InterleavePlan P; memset(&P, 0, sizeof P); P.q = q_parts;
  int slot = slot_of(dst_dev);
  for (uint32_t q = 0; q < q_parts; q++) {
    if (!parts_dev[q]) return fail(MI355_EBADARG, "fr_interleave: null part pointer");
    int s; CHK(common_slot({dst_dev, parts_dev[q]}, &s, "fr_interleave"));
    const uintptr_t d0 = (uintptr_t)dst_dev, d1 = d0 + (uint64_t)q_parts * n * sizeof(fe_t), p0 = (uintptr_t)parts_dev[q], p1 = p0 + n * sizeof(fe_t);
    if (p0 < d1 && d0 < p1) return fail(MI355_EBADARG, "fr_interleave: dst overlaps a part");
    P.part[q] = (const fe_t *)parts_dev[q];
  }
  DevGuard lk(slot);
  CHK(need_init(slot));
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(k_fr_interleave, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, (fe_t *)dst_dev, P, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_vec_axpy_dev(void *dst_dev, const void *a_dev, const void *b_dev, const void *scalar, uint64_t n) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({dst_dev, a_dev, b_dev}, &slot, "fr_vec_axpy")); DevGuard lk(slot);
  CHK(need_init(slot));
  if (!scalar || (n && (!dst_dev || !b_dev))) return fail(MI355_EBADARG, "fr_vec_axpy: null pointer");
  if (n == 0) return MI355_OK;
  fe_t s; memcpy(&s, scalar, 32);
  hipLaunchKernelGGL(k_fr_vec_axpy, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, (fe_t *)dst_dev, (const fe_t *)a_dev, (const fe_t *)b_dev, s, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_vec_mul_periodic_dev(void *data_dev, uint64_t n, const void *table_host, uint32_t period) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({data_dev}, &slot, "fr_vec_mul_periodic")); DevGuard lk(slot);
  CHK(need_init(slot));
  if (!table_host || period == 0 || (period & (period - 1)) || period > 4096 || (n && !data_dev)) return fail(MI355_EBADARG, "fr_vec_mul_periodic: period must be a power of two <= 4096");
  if (n == 0) return MI355_OK;
  fe_t *tab; CHK(ws_get("vec.table", (size_t)period * sizeof(fe_t), (void **)&tab));
  HIPCHK(hipMemcpyAsync(tab, table_host, (size_t)period * sizeof(fe_t), hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  hipLaunchKernelGGL(k_fr_vec_mul_periodic, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, (fe_t *)data_dev, n, tab, period - 1);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}

// ---- eval_polynomial
static int eval_polynomial_locked(const void *poly_dev, uint64_t n, const void *point, void *out_fr_host) {
  if (!out_fr_host || !point || (n && !poly_dev)) return fail(MI355_EBADARG, "eval_polynomial: null pointer");
  fe_t res = Fr::zero();
  if (n == 0) { memcpy(out_fr_host, &res, 32); return MI355_OK; }
  fe_t x; memcpy(&x, point, 32);
  const uint32_t blocks = ceil_div(n, (uint64_t)EVAL_RUN * 256);
  fe_t *partial; CHK(ws_get("eval.partial", ((size_t)blocks + 1) * sizeof(fe_t), (void **)&partial));
  {
    Scope sc("eval_poly");
    hipLaunchKernelGGL(k_eval_poly_partial, dim3(blocks), dim3(256), 0, g.stream, (const fe_t *)poly_dev, n, x, partial);
    hipLaunchKernelGGL(k_fr_sum, dim3(1), dim3(256), 0, g.stream, (const fe_t *)partial, (uint64_t)blocks, partial + blocks, (uint64_t)0);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&res, partial + blocks, sizeof res, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  resolve_spans();
  memcpy(out_fr_host, &res, 32);
  return MI355_OK;
}
int mi355_eval_polynomial_dev(const void *poly_dev, uint64_t n, const void *point, void *out_fr_host) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({poly_dev}, &slot, "eval_polynomial")); DevGuard lk(slot);
  CHK(need_init(slot));
  return eval_polynomial_locked(poly_dev, n, point, out_fr_host);
  });
}
// `batch` x eval_polynomial with ONE device synchronisation: step 9 of create_proof evaluates every queried (polynomial, rotation) pair -- dozens
// for a compression layer, thousands for the k = 20 inner circuit -- and a copy-back + stream synchronisation per value (~100 us) would cost more than
// the evaluations.  polys_dev[i] has n coefficients, points holds batch x 32 bytes, out_fr_host receives batch x 32 bytes.  All on one device.
int mi355_eval_polynomial_batch_dev(const void *const *polys_dev, uint32_t batch, uint64_t n, const void *points, void *out_fr_host) {
  return guarded([&]() -> int {
  if (batch == 0) return MI355_OK;
  if (!polys_dev || !points || !out_fr_host) return fail(MI355_EBADARG, "eval_polynomial_batch: null pointer");
  int slot = -1;
  for (uint32_t i = 0; i < batch; i++) {
    if (n && !polys_dev[i]) return fail(MI355_EBADARG, "eval_polynomial_batch: null polynomial pointer");
    const int s = slot_of(polys_dev[i]);
    if (slot < 0) slot = s; else if (s != slot && g_ctx[s].device != g_ctx[slot].device) return fail(MI355_EBADARG, "eval_polynomial_batch: the polynomials live on different devices");
  }
  DevGuard lk(slot);
  CHK(need_init(slot));
  if (n == 0) { memset(out_fr_host, 0, (size_t)batch * 32); return MI355_OK; }
  const uint32_t blocks = ceil_div(n, (uint64_t)EVAL_RUN * 256);
  const size_t stride = (size_t)blocks;
  // slices of at most 4096 evaluations (and at most 256 MiB of partial sums) share one scratch area, one pair of launches and one copy-back
  const uint32_t CH = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4096, (256ull << 20) / (stride * sizeof(fe_t))));
  const uint32_t chunk = std::min(batch, CH);
  fe_t *partial; CHK(ws_get("eval.partial", stride * chunk * sizeof(fe_t), (void **)&partial));
  // [pointer table | points | results]: the pointer table is padded to 32 bytes so that the fe_t arrays behind it keep the 16-byte alignment their dwordx4 accesses assume (an odd `chunk` used to leave them 8-byte aligned)
  const size_t ptab_bytes = ((size_t)chunk * sizeof(void *) + 31) & ~(size_t)31;
  char *tab; CHK(ws_get("eval.batch", ptab_bytes + (size_t)chunk * 2 * sizeof(fe_t), (void **)&tab));
  const fe_t **ptab = (const fe_t **)tab; fe_t *xtab = (fe_t *)(tab + ptab_bytes); fe_t *res = xtab + chunk;
  for (uint32_t base = 0; base < batch; base += CH) {
    const uint32_t cnt = std::min(CH, batch - base);
    Scope sc("eval_poly");
    HIPCHK(hipMemcpyAsync(ptab, polys_dev + base, (size_t)cnt * sizeof(void *), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(xtab, (const char *)points + 32 * (size_t)base, (size_t)cnt * sizeof(fe_t), hipMemcpyHostToDevice, g.stream));
    hipLaunchKernelGGL(k_eval_poly_partial_batch, dim3(blocks, cnt), dim3(256), 0, g.stream, (const fe_t *const *)ptab, n, (const fe_t *)xtab, partial, (uint64_t)stride);
    hipLaunchKernelGGL(k_fr_sum, dim3(cnt), dim3(256), 0, g.stream, (const fe_t *)partial, (uint64_t)blocks, res, (uint64_t)stride);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync((char *)out_fr_host + 32 * (size_t)base, res, (size_t)cnt * sizeof(fe_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  resolve_spans();
  return MI355_OK;
  });
}
int mi355_eval_polynomial_host(const void *poly_host, uint64_t n, const void *point, void *out_fr_host) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);   // host-pointer calls may run on any bound device (replicas): callers on different threads land on different devices
  CHK(need_init(slot));
  if (n && !poly_host) return fail(MI355_EBADARG, "eval_polynomial: null pointer");
  void *dev = nullptr;
  if (n) { CHK(ws_get("io.ntt", n * sizeof(fe_t), &dev)); HIPCHK(hipMemcpyAsync(dev, poly_host, n * sizeof(fe_t), hipMemcpyHostToDevice, g.stream)); }
  return eval_polynomial_locked(dev, n, point, out_fr_host);
  });
}

}  // extern "C"
