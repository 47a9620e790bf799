// lib_msm.hip -- libmi355zk.so, the MSM translation unit: launch orchestration of msm.hpp (digits -> two-level LDS counting sort ->
// segmented accumulation -> fix-up -> reduction tail), the host-pointer and sharded (multi-device) schedules, the C-ABI entry points
// mi355_msm_* / mi355_g1_sum_*, and the SRS entry points that launch kernels of msm.hpp (params-file loader with on-device validation,
// window-table build, synthetic SRS).  Host logic only; all arithmetic runs in the kernels.
// kernel headers first: lib_common.hpp defines the macro `g` (the calling thread's device context), a name the kernels use for locals
#define ZK_FRSCAN_DEVICE_ONLY 1   // msm.hpp needs the block scans of frscan.hpp, not its kernels (those belong to lib_aux.hip)
#include "msm.hpp"
#include "lib_common.hpp"

namespace mi355 {

static_assert(sizeof(fe_t) == 32 && sizeof(g1_affine_t) == 64 && sizeof(g1_jac_t) == 96, "ABI element sizes");
static_assert(sizeof(g1_xyzz_t) == 128 && sizeof(g1_xyzz29_t) == 144, "device record sizes (workspace layout, 16-byte vector accesses)");

int msm_tu_init_device() {
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l1_scatter_split<24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l1_scatter_split<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l1_scatter_split<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l2_scatter_split<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l2_scatter_split<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return MI355_OK;
}

// ------------------------------------------------------------------------------------------------ MSM
// measured on MI355X, in units of one bucket addition (~0.07 ns at 14 G adds/s): sort ~0.18 per entry, bucket reduction ~8.2 per bucket
// (fix-up, running sums, tree); without window tables the result also waits for the serial Horner tail over the windows (255 doublings
// in one lane, ~2 ms = 3e7 units).  The ordering this model gives for c was re-checked against tools/bench_window_choice.py at the end
// of the round (k = 18 ... 24).
double msm_cost(uint64_t n, int c, bool shared) { const double W = (MSM_SCALAR_BITS + c - 1) / c, nb = (double)(1ull << (c - 1)); return 1.18 * W * (double)n + 8.2 * nb * (shared ? 1.0 : W) + (shared ? 0.0 : 3.0e7); }

int choose_c(uint64_t n) {
  if (t_opts.force_c) return t_opts.force_c;
  double best = 1e300; int best_c = 8;
  for (int c = 4; c <= g_auto_max_c; c++) {
    const double W = (MSM_SCALAR_BITS + c - 1) / c, nb = (double)(1ull << (c - 1));
    if (W * nb * sizeof(g1_xyzz29_t) > 6.0e9) continue;
    const double cost = msm_cost(n, c, false);
    if (cost < best) { best = cost; best_c = c; }
  }
  return best_c;
}

struct PreTable { const g1_affine_t *table = nullptr; uint64_t row_stride = 0; int c = 0, w = 0; };   // table already offset to the slice start

int msm_batch_impl(const g1_affine_t *bases, const fe_t *const *polys_host, uint32_t M, uint64_t n, void *out_host, const PreTable *pre, void *out_dev_user = nullptr, bool accumulate_plan = false);

int msm_dev_impl(const g1_affine_t *bases, const fe_t *scalars, uint64_t n, void *out_host, const PreTable *pre = nullptr) {
  return msm_batch_impl(bases, &scalars, 1, n, out_host, pre);
}

// The shape of one MSM pass -- window bits, table use, sorter key split, entry / bucket counts -- derived in ONE place for the launch code
// (msm_enqueue) and for the batch splitter (msm_batch_impl), which must agree on what fits.
struct MsmShape { bool shared; uint32_t c, W, fb, cb_bits, regions; uint64_t emax, nbuckets; };
int msm_shape(uint64_t n, uint32_t M, const PreTable *pre, MsmShape &o, const MsmShape *forced = nullptr) {
  if (forced) { o.c = forced->c; o.shared = forced->shared; }
  else {
    o.c = (uint32_t)choose_c(n);
    // precomputed rows 2^(c w) P available and cheaper than the per-window schedule at this n -> all windows share one bucket set
    o.shared = pre && pre->table && !t_opts.force_c && !t_opts.no_tables && msm_cost(n, pre->c, true) <= msm_cost(n, (int)o.c, false) && ((uint64_t)pre->w << log2_ceil(n)) < (1ull << 31);
    if (o.shared) o.c = (uint32_t)pre->c;
  }
  o.W = (MSM_SCALAR_BITS + o.c - 1) / o.c;
  o.emax = (uint64_t)M * n * o.W;
  const uint32_t kb = o.c - 1; uint32_t fb = kb < g.sort_fb ? kb : g.sort_fb; if (kb - fb > 11) fb = kb - 11;
  o.fb = fb; o.cb_bits = kb - fb;
  const uint64_t sets = (uint64_t)M * (o.shared ? 1 : o.W);
  o.nbuckets = sets << kb;
  const uint64_t regions = sets << o.cb_bits; o.regions = (uint32_t)std::min<uint64_t>(regions, 0xffffffffu);
  if (o.emax >= (1ull << 32)) return fail(MI355_EBADARG, "msm: n * windows must be < 2^32");
  if (o.nbuckets >= (1ull << 31)) return fail(MI355_EBADARG, "msm: too many buckets (split the batch)");
  if (regions * 4 > 48 * 1024) return fail(MI355_EBADARG, "msm: batch too large for the coarse histogram (split the batch)");
  if (o.fb > 12 || (1u << o.cb_bits) > SORT_MAX_BINS) return fail(MI355_EBADARG, "msm: window bits out of range for the sorter");
  return MI355_OK;
}

// The reduction tail of one MSM pass: sum_b (b + 1) B[b] per bucket set by short chunked running sums, a multi-block tree per set, the
// Horner over windows (none with window tables) and the normalisation.  Runs entirely on the 29-bit field.
int msm_reduce_tail(const MsmShape &sh, uint32_t M, const g1_xyzz29_t *buckets, g1_jac_t *out_dev, bool normalise, hipStream_t s, const std::string &sfx) {
  auto role = [&](const char *r) { return std::string(r) + sfx; };
  const uint32_t nb = 1u << (sh.c - 1), red_wpp = sh.shared ? 1 : sh.W, red_windows = M * red_wpp;
  // running-sum chunk per reduce thread: every thread is one serial chain of 2*chunk additions plus a ~(c-1)-bit scalar multiple, so the
  // chain is kept short (the kernel is latency-bound) as long as there are enough buckets to give the GPU ~128k chains (measured at 2^21 buckets: 2.58 ms with 256k chains of 8 buckets, 2.11 ms with 128k of 16, 2.63 ms with 64k of 32)
  uint32_t chunk = 64; while (chunk > nb) chunk >>= 1;
  // (bucket sets of small MSMs: chains down to g.reduce_min_chunk buckets -- the chain is then mostly the (c - 2)-bit multiple of the
  // chunk sum, ~26 instead of ~37 addition times at 2^16 buckets)
  while (chunk > g.reduce_min_chunk && (uint64_t)(nb / chunk) * red_windows < g.reduce_chains) chunk >>= 1;
  const uint32_t chunks_per_window = nb / chunk, nchunks = chunks_per_window * red_windows;
  g1_xyzz29_t *chunk_out, *tree_a, *tree_b, *window_sums;
  CHK(ws_get(role("msm.chunk_out").c_str(), (size_t)nchunks * sizeof(g1_xyzz29_t), (void **)&chunk_out));
  { const size_t lvl = (size_t)ceil_div(chunks_per_window, 64 * TREE_PER_THREAD) * red_windows + 1;   // 64: the quad form's logical threads per block
    CHK(ws_get(role("msm.tree_a").c_str(), lvl * sizeof(g1_xyzz29_t), (void **)&tree_a)); CHK(ws_get(role("msm.tree_b").c_str(), lvl * sizeof(g1_xyzz29_t), (void **)&tree_b)); }
  CHK(ws_get(role("msm.window_sums").c_str(), (size_t)red_windows * sizeof(g1_xyzz29_t), (void **)&window_sums));
  MsmPlan PR; PR.n = 0; PR.c = sh.c; PR.windows = red_windows; PR.nb = nb; PR.seg = 0; PR.batch = M;
  // latency form (a quad per logical thread, g1_xyzz29_add_q4: ~2.6x shorter dependent chain per addition) where the kernel cannot fill the
  // GPU anyway; throughput form (one lane per thread) for big bucket sets.  MI355_TAIL_COOP_MAX = largest logical thread count that takes the quads.
  const bool coop_reduce = nchunks <= g.tail_coop_max && (g.tail_coop_mask & 2u);
  if (coop_reduce) hipLaunchKernelGGL(k_msm_bucket_reduce<4>, dim3(ceil_div((uint64_t)nchunks * 4, 128)), dim3(128), 0, s, buckets, chunk_out, PR, chunk);
  else hipLaunchKernelGGL(k_msm_bucket_reduce<1>, dim3(ceil_div(nchunks, 128)), dim3(128), 0, s, buckets, chunk_out, PR, chunk);
  {
    // tree-sum the chunk results per window, ping-ponging between two small buffers
    const g1_xyzz29_t *cur = chunk_out; uint32_t cnt = chunks_per_window; g1_xyzz29_t *bufs[2] = {tree_a, tree_b}; int which = 0;
    while (true) {
      const bool coop = (uint64_t)cnt * red_windows <= g.tail_coop_max && (g.tail_coop_mask & 4u);
      const uint32_t per_block = (coop ? 64u : 256u) * TREE_PER_THREAD;
      const uint32_t outn = ceil_div(cnt, per_block);
      g1_xyzz29_t *dst = outn == 1 ? window_sums : bufs[which];
      if (coop) hipLaunchKernelGGL(k_msm_tree_sum29<4>, dim3(outn, red_windows), dim3(256), 0, s, cur, cnt, dst, outn);
      else hipLaunchKernelGGL(k_msm_tree_sum29<1>, dim3(outn, red_windows), dim3(256), 0, s, cur, cnt, dst, outn);
      if (outn == 1) break;
      cur = dst; cnt = outn; which ^= 1;
    }
  }
  if (g.tail_coop_max && (g.tail_coop_mask & 8u)) hipLaunchKernelGGL(k_msm_final29<4>, dim3(M), dim3(64), 0, s, (const g1_xyzz29_t *)window_sums, red_wpp, sh.shared ? 0u : sh.c, out_dev, normalise ? 1 : 0);
  else hipLaunchKernelGGL(k_msm_final29<1>, dim3(M), dim3(64), 0, s, (const g1_xyzz29_t *)window_sums, red_wpp, sh.shared ? 0u : sh.c, out_dev, normalise ? 1 : 0);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}

// One MSM (or one chunk of a pipelined MSM) enqueued on three streams: st.a digits + sort (HBM-bound), st.b bucket accumulation
// (ALU-bound), st.c fix-up + bucket reduction (latency-bound).  With st.a == st.b == st.c this is the plain serial schedule.
// M commitments over the same basis slice in one pass: (polynomial m, window w) is window m * W + w of one big bucket problem, so the
// per-call fixed costs (launches, the latency-bound reduction tail) are paid once per batch.  out_dev: M x 96 B, device.
struct MsmStreams { hipStream_t a, b, c; };

// forced: take (c, shared) from a shape computed for another length (the slices of a host-chunked MSM must agree on the bucket layout);
// set_index / set_count: this pass fills bucket set `set_index` of `set_count` (each nbuckets records); skip_tail: stop after the fix-up
// (the caller folds the sets and runs msm_reduce_tail once).
int msm_enqueue(const g1_affine_t *bases, const PolyPtrs &inl, const fe_t *const *polys_dev, uint32_t M, uint64_t n, g1_jac_t *out_dev, const PreTable *pre, MsmSlot &slot,
                const MsmStreams &st, bool normalise, const MsmShape *forced = nullptr, uint32_t set_index = 0, uint32_t set_count = 1, bool skip_tail = false) {
  const bool piped = st.a != st.b;
  const std::string sfx = slot.id ? "#" + std::to_string(slot.id) : std::string();
  auto role = [&](const char *r) { return std::string(r) + sfx; };
#define WS(name, bytes, ptr) CHK(ws_get(role(name).c_str(), bytes, (void **)&ptr))
  MsmShape sh; CHK(msm_shape(n, M, pre, sh, forced));
  const bool shared = sh.shared;
  if (shared) bases = pre->table;
  MsmPlan P; P.n = (uint32_t)n; P.batch = M; P.c = sh.c; P.windows = sh.W; P.nb = 1u << (P.c - 1);
  const uint64_t emax = sh.emax;
  const uint64_t want_threads = (uint64_t)g.prop.multiProcessorCount * 256 * g.seg_factor;   // segments per lane slot (MI355_SEG_FACTOR)
  uint64_t seg = (emax + want_threads - 1) / want_threads; if (seg < 16) seg = 16; if (seg > 4096) seg = 4096;
  P.seg = (uint32_t)seg;
  // worst-case segment (<= 4096) | minimum << 13 | (fill percentage / 2) << 26: the kernels derive the segment from the actual entry count.  With the
  // segmented fix-up (two-partial buckets are cheap) the entries may spread over more threads than with the per-bucket kernels
  const bool segfix = g.fixup_mode == 1 || (g.fixup_mode == 2 && (uint32_t)sh.nbuckets >= (1u << 19) && (uint64_t)P.seg * sh.nbuckets >= emax);
  const uint32_t seg_arg = P.seg | (std::min(g.seg_min, P.seg) << 13) | (((segfix ? g.seg_fill_segfix : g.seg_fill) / 2) << 26);
  const uint32_t red_wpp = shared ? 1 : P.windows;        // bucket sets per polynomial
  const uint32_t red_windows = M * red_wpp;               // bucket sets to reduce
  const uint32_t nbuckets = (uint32_t)sh.nbuckets;
  const uint32_t acc_threads = ceil_div(emax, seg), acc_blocks = ceil_div(acc_threads, 256);
  const uint32_t tn = acc_blocks * 256;
  // sort plan: fine bits fb (<= 12, LDS histogram of 2^fb bins), coarse bits = the rest
  SortPlan S; S.n = P.n; S.windows = M * P.windows; S.wpp = P.windows; S.nb = P.nb;
  S.fb = sh.fb; S.cb_bits = sh.cb_bits;
  S.shared = shared ? 1 : 0; S.nshift = log2_ceil(n);
  S.regions = sh.regions;
  S.t1 = g.sort_t1;                           // level-1 tile: 1024 threads x 8 or 16 entries (64 / 128 KiB of LDS staging)
  // split records (payload u32 + fine key u16 as two streams, MI355_SORT_SPLIT): 6 bytes of staging per entry -> 24 576-entry tiles where the bin
  // bookkeeping leaves room (<= 1024 coarse bins)
  if (g.sort_split == 1 && S.t1 == 16384 && (1u << S.cb_bits) <= 1024 && emax >= (1ull << 24)) S.t1 = 24576;   // MI355_SORT_SPLIT=2: split records, 16 384-entry tiles
  // level-2 tile (6 B of LDS per entry next to the 3 x 2^fb words of bin bookkeeping): 16384 entries give twice the run length in
  // `sorted` (fewer partial-line store transactions, the limiter of this kernel) at one workgroup per CU; worth it for big sorts
  S.t2 = g.sort_t2 ? g.sort_t2 : (emax >= (1ull << 27) && S.fb <= 11 ? 16384 : 8192);
  const uint32_t tiles1 = ceil_div(n, S.t1), l2_tiles_max = ceil_div(emax, S.t2) + S.regions + 8;   // + 8: the XCD-aware tile order rounds the tile count up to a multiple of 8
  const uint32_t vwindows = M * P.windows;   // (polynomial, window) pairs

  uint32_t *enc, *hist, *offsets, *cursor, *sorted, *scan_sums, *coarse_hist, *coarse_off, *coarse_cursor, *tile_start;
  g1_xyzz29_t *buckets, *part; int32_t *part_id;
  // stage-A-only buffers are shared by all slots (the sort stages of successive chunks run one after the other on st.a)
  CHK(ws_get("msm.digits", emax * 4, (void **)&enc));
  uint32_t *pairs_lo = nullptr; uint16_t *pairs_hi = nullptr;
  CHK(ws_get("msm.pairs_lo", emax * 4 + 64, (void **)&pairs_lo)); CHK(ws_get("msm.pairs_hi", emax * 2 + 64, (void **)&pairs_hi));
  CHK(ws_get("msm.hist", ((size_t)nbuckets + 1) * 4, (void **)&hist));
  CHK(ws_get("msm.cursor", ((size_t)nbuckets + 1) * 4, (void **)&cursor));
  CHK(ws_get("msm.coarse_hist", ((size_t)S.regions + 1) * 4, (void **)&coarse_hist));
  CHK(ws_get("msm.coarse_off", ((size_t)S.regions + 1) * 4, (void **)&coarse_off));
  CHK(ws_get("msm.coarse_cursor", ((size_t)S.regions + 1) * 4, (void **)&coarse_cursor));
  CHK(ws_get("msm.tile_start", ((size_t)S.regions + 1) * 4, (void **)&tile_start));
  const uint32_t scan_n = nbuckets + 1, scan_blocks = ceil_div(scan_n, SCAN_BLOCK * SCAN_ITEMS);
  const uint32_t cscan_n = S.regions + 1, cscan_blocks = ceil_div(cscan_n, SCAN_BLOCK * SCAN_ITEMS);
  CHK(ws_get("msm.scan_sums", (size_t)(scan_blocks + cscan_blocks) * 4, (void **)&scan_sums));
  // per-slot: what the accumulation and the reduction of this chunk read while the next chunk is being sorted
  WS("msm.offsets", ((size_t)nbuckets + 1) * 4, offsets);
  WS("msm.sorted", emax * 4, sorted);
  WS("msm.buckets", (size_t)nbuckets * set_count * sizeof(g1_xyzz29_t), buckets);
  buckets += (size_t)nbuckets * set_index;
  WS("msm.part", (size_t)tn * 2 * sizeof(g1_xyzz29_t), part);
  WS("msm.part_id", (size_t)tn * 2 * 4, part_id);
  const uint32_t big_cap = tn / g.fixup_serial_max + 2;
  uint32_t *big_list; WS("msm.big_list", ((size_t)big_cap * 3 + 1) * 4, big_list);
  uint32_t *big_count = big_list + (size_t)big_cap * 3;
  // buckets that span >= FIXUP_HUGE_MIN accumulate threads (at most tn / FIXUP_HUGE_MIN of them) get FIXUP_SLICES workgroups each
  const uint32_t huge_cap = tn / FIXUP_HUGE_MIN + 2;
  uint32_t *huge_list, *huge_count; g1_xyzz29_t *huge_part;
  WS("msm.huge_list", ((size_t)huge_cap * 3 + 1) * 4, huge_list); huge_count = huge_list + (size_t)huge_cap * 3;
  WS("msm.huge_part", (size_t)huge_cap * FIXUP_SLICES * sizeof(g1_xyzz29_t), huge_part);
#undef WS

  const int grid_stream = g.prop.multiProcessorCount * 8;
  {
    hipStream_t s = st.a;
    // the accumulation and the fix-up of the chunk that used this slot before must be done with `sorted` / `offsets`
    if (piped && slot.used) { HIPCHK(hipStreamWaitEvent(s, slot.acc_done, 0)); HIPCHK(hipStreamWaitEvent(s, slot.red_done, 0)); }
    {
      Scope sc("msm_digits", s);
      HIPCHK(hipMemsetAsync(coarse_hist, 0, ((size_t)S.regions + 1) * 4, s));
      hipLaunchKernelGGL(k_msm_digits, dim3(grid_stream / M > 0 ? grid_stream / M : 1, M), dim3(256), (size_t)S.regions * 4, s, inl, polys_dev, enc, P, coarse_hist, S.fb, S.cb_bits, S.shared);
    }
    {
      Scope sc("msm_sort", s);
      HIPCHK(hipMemsetAsync(hist, 0, ((size_t)nbuckets + 1) * 4, s));
      hipLaunchKernelGGL(k_scan_partial, dim3(cscan_blocks), dim3(SCAN_BLOCK), 0, s, coarse_hist, scan_sums + scan_blocks, cscan_n);
      hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, scan_sums + scan_blocks, cscan_blocks);
      hipLaunchKernelGGL(k_scan_final, dim3(cscan_blocks), dim3(SCAN_BLOCK), 0, s, coarse_hist, scan_sums + scan_blocks, coarse_off, coarse_cursor, cscan_n);
      {
        Scope s1("sort_l1", s);
        const uint32_t CBp = ((1u << S.cb_bits) + 1) & ~1u;
        const size_t lds1 = (size_t)(3 * CBp + 32 + SORT_SPLIT_TMAX) * 4 + (size_t)S.t1 * 6;
        if (S.t1 == 24576) hipLaunchKernelGGL(k_sort_l1_scatter_split<24>, dim3(tiles1 * vwindows), dim3(1024), lds1, s, enc, coarse_cursor, pairs_lo, pairs_hi, S);
        else if (S.t1 == 16384) hipLaunchKernelGGL(k_sort_l1_scatter_split<16>, dim3(tiles1 * vwindows), dim3(1024), lds1, s, enc, coarse_cursor, pairs_lo, pairs_hi, S);
        else hipLaunchKernelGGL(k_sort_l1_scatter_split<8>, dim3(tiles1 * vwindows), dim3(1024), lds1, s, enc, coarse_cursor, pairs_lo, pairs_hi, S);
      }
      hipLaunchKernelGGL(k_sort_tile_prefix, dim3(1), dim3(SCAN_BLOCK), 0, s, coarse_off, tile_start, S);
      { Scope s2("sort_hist", s);
      hipLaunchKernelGGL(k_sort_l2_hist_split, dim3(l2_tiles_max), dim3(256), 0, s, (const uint16_t *)pairs_hi, coarse_off, tile_start, hist, S); }
      hipLaunchKernelGGL(k_scan_partial, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, s, hist, scan_sums, scan_n);
      hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, scan_sums, scan_blocks);
      hipLaunchKernelGGL(k_scan_final, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, s, hist, scan_sums, offsets, cursor, scan_n);
      {
        Scope s3("sort_l2", s);
        const size_t lds2 = (size_t)(3 * (1u << S.fb) + 32) * 4 + (size_t)S.t2 * 6;   // histogram / offsets / bases + staged indices (4 B) and their bins (2 B)
        if (S.t2 == 8192) hipLaunchKernelGGL(k_sort_l2_scatter_split<8>, dim3(l2_tiles_max), dim3(1024), lds2, s, (const uint32_t *)pairs_lo, (const uint16_t *)pairs_hi, coarse_off, tile_start, cursor, sorted, S);
        else hipLaunchKernelGGL(k_sort_l2_scatter_split<16>, dim3(l2_tiles_max), dim3(1024), lds2, s, (const uint32_t *)pairs_lo, (const uint16_t *)pairs_hi, coarse_off, tile_start, cursor, sorted, S);
      }
    }
    if (piped) { HIPCHK(hipEventRecord(slot.sorted, s)); HIPCHK(hipStreamWaitEvent(st.b, slot.sorted, 0)); }
  }
  {
    hipStream_t s = st.b;
    // the reduction of the chunk that used this slot before must be done with `buckets` / `part`
    if (piped && slot.used) HIPCHK(hipStreamWaitEvent(s, slot.red_done, 0));
    {
      Scope sc("msm_accumulate", s);
      HIPCHK(hipMemsetAsync(buckets, 0, (size_t)nbuckets * sizeof(g1_xyzz29_t), s));
#define ACC_LAUNCH(V) hipLaunchKernelGGL(k_msm_accumulate<V>, dim3(acc_blocks), dim3(256), 0, s, bases, sorted, offsets, nbuckets, buckets, part, part_id, seg_arg, S.nshift, shared ? pre->row_stride : (uint64_t)0, g.debug_gather_mask)
      ACC_LAUNCH(4);   // <4>: limb products as explicitly chained v_mad (fp29.hpp mac_*).  The C++-multiplier variant <0> (59.5 vs 57.1 ms at 2^26, round 3) and the older A/B variants are no longer instantiated (round 6)
#undef ACC_LAUNCH
    }
    if (piped) { HIPCHK(hipEventRecord(slot.acc_done, s)); HIPCHK(hipStreamWaitEvent(st.c, slot.acc_done, 0)); }
  }
  {
    hipStream_t s = st.c;
    Scope sc("msm_reduce", s);
    HIPCHK(hipMemsetAsync(big_count, 0, 4, s));
    // the whole tail runs on the 29-bit field (g1_xyzz29_add / _dbl): records are never converted to the saturated form on the way
    HIPCHK(hipMemsetAsync(huge_count, 0, 4, s));
    // four lanes per bucket where buckets straddle many short segments (small and mid-size MSMs); with 2^19 buckets and more the extra
    // threads cost more than the shorter chains save (measured: +0.15 ms at 2^18, +0.5 ms at 2^21 buckets)
    // auto (2): big bucket sets whose buckets are no longer than a segment straddle two accumulate threads as a rule -- the case the
    // segmented reduction is measured faster in (2^24 .. 2^26 with c = 22); everything else takes the per-bucket kernels
    if (segfix) {
      // one segmented reduction over the 2 * tn partial slots, level by level (64 slots -> 2 per level) until one wavefront holds the rest
      uint32_t N = 2 * tn; const uint32_t waves1 = ceil_div(N, 64);
      int32_t *lv_ids; g1_xyzz29_t *lv_recs;
      { const std::string r1 = role("msm.segfix_ids"), r2 = role("msm.segfix_recs"); const size_t cap = (size_t)2 * waves1 + (size_t)waves1 / 8 + 512;
        CHK(ws_get(r1.c_str(), cap * 4, (void **)&lv_ids)); CHK(ws_get(r2.c_str(), cap * sizeof(g1_xyzz29_t), (void **)&lv_recs)); }
      const int32_t *cur_ids = part_id; const g1_xyzz29_t *cur_recs = part; size_t used = 0;
      while (N > 64) {
        const uint32_t waves = ceil_div(N, 64);
        hipLaunchKernelGGL(k_msm_segfix, dim3(ceil_div(N, 256)), dim3(256), 0, s, cur_ids, cur_recs, N, buckets, lv_ids + used, lv_recs + used, 0);
        cur_ids = lv_ids + used; cur_recs = lv_recs + used; used += (size_t)2 * waves; N = 2 * waves;
      }
      hipLaunchKernelGGL(k_msm_segfix, dim3(1), dim3(64), 0, s, cur_ids, cur_recs, N, buckets, (int32_t *)nullptr, (g1_xyzz29_t *)nullptr, 1);
    } else {
    if ((uint64_t)nbuckets * 4 <= g.tail_coop_max && (g.tail_coop_mask & 1u)) hipLaunchKernelGGL((k_msm_fixup<4, 4>), dim3(ceil_div((uint64_t)nbuckets * 16, 256)), dim3(256), 0, s, offsets, nbuckets, buckets, part, part_id, seg_arg, tn, big_list, big_count, big_cap, huge_list, huge_count, huge_cap, g.fixup_serial_max, g.fixup_huge_min);
    else if (nbuckets <= (1u << g.fixup_lanes_max_log)) hipLaunchKernelGGL((k_msm_fixup<4, 1>), dim3(ceil_div((uint64_t)nbuckets * 4, 256)), dim3(256), 0, s, offsets, nbuckets, buckets, part, part_id, seg_arg, tn, big_list, big_count, big_cap, huge_list, huge_count, huge_cap, g.fixup_serial_max, g.fixup_huge_min);
    else hipLaunchKernelGGL((k_msm_fixup<1, 1>), dim3(ceil_div(nbuckets, 256)), dim3(256), 0, s, offsets, nbuckets, buckets, part, part_id, seg_arg, tn, big_list, big_count, big_cap, huge_list, huge_count, huge_cap, g.fixup_serial_max, g.fixup_huge_min);
    hipLaunchKernelGGL(k_msm_fixup_big, dim3(big_cap), dim3(256), 0, s, buckets, part, part_id, big_list, big_count);
    hipLaunchKernelGGL(k_msm_fixup_huge, dim3(huge_cap * FIXUP_SLICES), dim3(256), 0, s, part, part_id, huge_list, huge_count, huge_part, huge_cap);
    hipLaunchKernelGGL(k_msm_fixup_huge_fold, dim3(huge_cap), dim3(64), 0, s, buckets, huge_list, huge_count, (const g1_xyzz29_t *)huge_part);
    }
    if (!skip_tail) CHK(msm_reduce_tail(sh, M, buckets, out_dev, normalise, s, sfx));
  }
  if (piped) HIPCHK(hipEventRecord(slot.red_done, st.c));
  slot.used = true;
  HIPCHK(hipGetLastError());
  g.last_c = (int)P.c; g.last_w = (int)P.windows; g.last_entries += emax; g.last_shared = shared;
  return MI355_OK;
}

// Chunks of a pipelined MSM: the point range is cut into K slices, every slice is a complete MSM over its part of the basis (and of the
// window tables), and the memory-bound sort of slice k + 1 runs under the ALU-bound accumulation of slice k; the K partial results are
// added at the end.  OFF by default (MI355_MSM_CHUNKS / mi355_msm_set_pipeline): on MI355X the accumulation holds every wave slot for its
// whole run, so the sort of the next slice barely progresses next to it -- measured 74.9 ms (1 chunk), 75.6 (2), 80.2 (4), 91.1 (8) at 2^26.
uint32_t msm_chunks_for(uint32_t M, uint64_t n) {
  if (g.msm_chunks <= 1 || M != 1 || n < (1ull << g.msm_chunk_min_log)) return 1;
  uint32_t k = g.msm_chunks;
  while (k > 1 && n / k < ((1ull << g.msm_chunk_min_log) >> 2)) k--;
  return k;
}

// out_dev_user: the M results stay in device memory and the call returns without waiting for the stream.  accumulate_plan: second half of
// a split batch (mi355_msm_last_plan reports the entries of the whole batch).
int msm_batch_impl(const g1_affine_t *bases, const fe_t *const *polys_host, uint32_t M, uint64_t n, void *out_host, const PreTable *pre, void *out_dev_user, bool accumulate_plan) {
  if (M == 0) return MI355_OK;
  if (n == 0) {
    if (out_dev_user) { HIPCHK(hipMemsetAsync(out_dev_user, 0, (size_t)M * sizeof(g1_jac_t), g.stream)); return MI355_OK; }
    memset(out_host, 0, (size_t)M * sizeof(g1_jac_t)); return MI355_OK;
  }
  if (n >= (1ull << 31)) return fail(MI355_EBADARG, "msm: n must be < 2^31");
  if (M > 1) {
    // a batch that would overflow the 32-bit entry index, the coarse histogram's LDS or 8 GiB of bucket records is processed as two
    // half batches (the same shape computation as the launch code: msm_shape)
    MsmShape sh; const int rc0 = msm_shape(n, M, pre, sh);
    if (rc0 != MI355_OK || sh.emax > (1ull << 29) || sh.nbuckets * sizeof(g1_xyzz29_t) > (8ull << 30)) {
      const uint32_t h = M / 2;
      int rc = msm_batch_impl(bases, polys_host, h, n, out_host, pre, out_dev_user, accumulate_plan);
      if (rc != MI355_OK) return rc;
      return msm_batch_impl(bases, polys_host + h, M - h, n, out_host ? (char *)out_host + (size_t)h * sizeof(g1_jac_t) : nullptr, pre,
                            out_dev_user ? (char *)out_dev_user + (size_t)h * sizeof(g1_jac_t) : nullptr, true);
    }
  }
  const uint32_t K = msm_chunks_for(M, n);
  if (!accumulate_plan) { g.last_entries = 0; g.last_host_slices = 1; }
  CallTrace tr("msm_g1", (uint64_t)M * n, 96.0);
  g1_jac_t *out_dev; const fe_t **polys_dev = nullptr;
  CHK(ws_get("msm.out", (size_t)(K + 1) * M * sizeof(g1_jac_t), (void **)&out_dev));
  hipStream_t s = g.stream;
  Scope total("msm_total", s);
  PolyPtrs inl; for (int i = 0; i < 8; i++) inl.p[i] = nullptr;
  if (K == 1) {
    if (M <= 8) for (uint32_t m = 0; m < M; m++) inl.p[m] = polys_host[m];
    else {
      // the pointer array is staged through a buffer the library owns (the caller's array may be a temporary)
      CHK(ws_get("msm.polys", (size_t)M * sizeof(void *), (void **)&polys_dev));
      g.polys_stage.assign(polys_host, polys_host + M);
      HIPCHK(hipMemcpyAsync(polys_dev, g.polys_stage.data(), (size_t)M * sizeof(void *), hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
    }
    MsmStreams st{s, s, s};
    CHK(msm_enqueue(bases, inl, polys_dev, M, n, out_dev, pre, g.msm_slot[0], st, t_opts.normalise));
  } else {
    std::vector<uint64_t> lo(K + 1);
    for (uint32_t k = 0; k <= K; k++) lo[k] = n * k / K;
    // the side streams start after everything already queued on the caller's stream (the scalars may still be in flight there)
    HIPCHK(hipEventRecord(g.ev_fork, s));
    HIPCHK(hipStreamWaitEvent(g.aux_stream[0], g.ev_fork, 0)); HIPCHK(hipStreamWaitEvent(g.aux_stream[1], g.ev_fork, 0));
    MsmStreams st{g.aux_stream[0], s, g.aux_stream[1]};
    for (uint32_t k = 0; k < K; k++) {
      PreTable pk; const PreTable *pp = nullptr;
      if (pre) { pk = *pre; if (pk.table) pk.table += lo[k]; pp = &pk; }
      inl.p[0] = polys_host[0] + lo[k];   // pipelined MSMs are single-polynomial (msm_chunks_for)
      CHK(msm_enqueue(bases + lo[k], inl, nullptr, M, lo[k + 1] - lo[k], out_dev + (size_t)(k + 1) * M, pp, g.msm_slot[k & 1], st, false));
    }
    HIPCHK(hipStreamWaitEvent(s, g.msm_slot[0].red_done, 0)); HIPCHK(hipStreamWaitEvent(s, g.msm_slot[1].red_done, 0));
    HIPCHK(hipEventRecord(g.ev_fork, st.a)); HIPCHK(hipStreamWaitEvent(s, g.ev_fork, 0));   // join the sort stream too
    hipLaunchKernelGGL(k_g1_sum_strided, dim3(M), dim3(64), 0, s, out_dev + M, K, M, out_dev, t_opts.normalise ? 1 : 0);
  }
  HIPCHK(hipGetLastError());
  total.close();
  g.msm_slot[0].used = g.msm_slot[1].used = false;
  g.last_chunks = (int)K;
  if (out_dev_user) {   // asynchronous: the caller's stream order protects the result
    HIPCHK(hipMemcpyAsync(out_dev_user, out_dev, (size_t)M * sizeof(g1_jac_t), hipMemcpyDeviceToDevice, s));
    if (g.profiling) resolve_spans();
    tr.done(" (device result)");
    return MI355_OK;
  }
  HIPCHK(hipMemcpyAsync(out_host, out_dev, (size_t)M * sizeof(g1_jac_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  resolve_spans();
  { char buf[64]; snprintf(buf, sizeof buf, " batch=%u c=%d W=%d", M, g.last_c, g.last_w); tr.done(buf); }
  return MI355_OK;
}

// mi355_msm_g1_host on one device.  Big single MSMs are cut into K point-range slices: slice k + 1 crosses PCIe on the copy stream while
// slice k is sorted and accumulated; every slice fills its OWN bucket set with the common layout (c, W, table rows), the K sets are
// added bucket by bucket and the latency-bound reduction tail runs once.  (The pipelined device-resident schedule above pays a tail
// per slice; here the slices only differ from the one-pass MSM by K - 1 extra bucket records per bucket.)  Everything else -- batches,
// small sizes -- is one copy followed by the device-resident path.
int msm_host_single(const g1_affine_t *bases, const fe_t *const *polys_host, uint32_t M, uint64_t n, void *out_host, const PreTable *pre, void *out_dev_user = nullptr) {
  fe_t *sc; CHK(ws_get("io.scalars", (size_t)M * n * sizeof(fe_t), (void **)&sc));
  hipStream_t s = g.stream;
  // Slice plan: the copy of slice k + 1 must fit under the compute of slice k.  PCIe moves a pair's 32 bytes ~1.7x faster than the GPU
  // consumes them, so slices may GROW by that factor: a small first slice (its copy is the only exposed one), then x1.7 each -- 5 slices
  // for 2^26 pairs instead of 8 equal ones.  Every extra slice costs bucket crossings in the accumulation (each bucket is visited once
  // per slice) and a fix-up pass, which is why fewer, growing slices win (measured: 8 equal slices 86.7 ms).
  std::vector<uint64_t> cut;   // slice k = [cut[k], cut[k + 1])
  if (M == 1 && n >= (1ull << g.host_slice_min_log) && g.host_chunks > 1) {
    double w = 1.0, tot = 0; std::vector<double> ws;
    const double first = 1.0 / 16.0;
    for (double rem = 1.0; rem > 1e-9 && ws.size() + 1 < g.host_chunks;) { const double take = std::min(rem, first * w); ws.push_back(take); rem -= take; w *= 1.7; tot += take; }
    if (tot < 1.0 - 1e-9) ws.push_back(1.0 - tot);
    cut.push_back(0); double acc = 0;
    const uint64_t align = n >= (1ull << 20) ? ~1023ull : ~0ull;
    for (size_t i = 0; i + 1 < ws.size(); i++) { acc += ws[i]; const uint64_t c = std::min<uint64_t>(n, (uint64_t)(acc * (double)n) & align); if (c > cut.back() && c < n) cut.push_back(c); }
    cut.push_back(n);
  }
  uint32_t K = cut.empty() ? 1 : (uint32_t)cut.size() - 1;
  uint64_t biggest = 0; for (uint32_t k = 0; k < K && !cut.empty(); k++) biggest = std::max(biggest, cut[k + 1] - cut[k]);
  MsmShape shape;
  if (K > 1 && (msm_shape(biggest, 1, pre, shape) != MI355_OK || shape.nbuckets * K * sizeof(g1_xyzz29_t) > (8ull << 30))) K = 1;
  if (K <= 1) {
    std::vector<const fe_t *> ptrs(M);
    for (uint32_t m = 0; m < M; m++) { ptrs[m] = sc + (size_t)m * n; HIPCHK(hipMemcpyAsync(sc + (size_t)m * n, polys_host[m], n * sizeof(fe_t), hipMemcpyHostToDevice, s)); }
    return msm_batch_impl(bases, ptrs.data(), M, n, out_host, pre, out_dev_user);
  }
  g.last_entries = 0;
  CallTrace tr("msm_g1_host", n, 96.0);
  g1_jac_t *out_dev; CHK(ws_get("msm.out", 2 * sizeof(g1_jac_t), (void **)&out_dev));
  Scope total("msm_total", s);
  // the staging buffer may still be read by work queued earlier on the compute stream
  HIPCHK(hipEventRecord(g.ev_fork, s)); HIPCHK(hipStreamWaitEvent(g.copy_stream, g.ev_fork, 0));
  PolyPtrs inl; for (int i = 0; i < 8; i++) inl.p[i] = nullptr;
  MsmStreams st{s, s, s};
  for (uint32_t k = 0; k < K; k++) {
    const uint64_t lo = cut[k], hi = cut[k + 1];
    // pageable source: the call blocks this thread while the DMA runs, which is exactly when the GPU works on the slices queued before
    HIPCHK(hipMemcpyAsync(sc + lo, polys_host[0] + lo, (hi - lo) * sizeof(fe_t), hipMemcpyHostToDevice, g.copy_stream));
    HIPCHK(hipEventRecord(g.ev_copy[k & 3], g.copy_stream));
    HIPCHK(hipStreamWaitEvent(s, g.ev_copy[k & 3], 0));
    PreTable pk; const PreTable *pp = nullptr;
    if (pre) { pk = *pre; if (pk.table) pk.table += lo; pp = &pk; }
    inl.p[0] = sc + lo;
    CHK(msm_enqueue(bases + lo, inl, nullptr, 1, hi - lo, out_dev, pp, g.msm_slot[0], st, t_opts.normalise, &shape, k, K, true));
  }
  g.msm_slot[0].used = false;
  g1_xyzz29_t *buckets; CHK(ws_get("msm.buckets", (size_t)shape.nbuckets * K * sizeof(g1_xyzz29_t), (void **)&buckets));
  {
    Scope sc2("msm_reduce", s);
    hipLaunchKernelGGL(k_msm_bucket_fold, dim3(ceil_div(shape.nbuckets, 256)), dim3(256), 0, s, buckets, (uint32_t)shape.nbuckets, K);
    CHK(msm_reduce_tail(shape, 1, buckets, out_dev, t_opts.normalise, s, std::string()));
  }
  total.close();
  g.last_chunks = (int)K; g.last_host_slices = (int)K;
  if (out_dev_user) {   // sharded MSM: the partial stays on the device, in stream order
    HIPCHK(hipMemcpyAsync(out_dev_user, out_dev, sizeof(g1_jac_t), hipMemcpyDeviceToDevice, s));
    if (g.profiling) resolve_spans();
    return MI355_OK;
  }
  HIPCHK(hipMemcpyAsync(out_host, out_dev, sizeof(g1_jac_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  resolve_spans();
  { char buf[64]; snprintf(buf, sizeof buf, " host slices=%u c=%d W=%d", K, g.last_c, g.last_w); tr.done(buf); }
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

// Prover::load_params for one degree: stream a RawBytes params file into device memory (two pinned staging buffers: the read of chunk
// i + 1 overlaps the DMA of chunk i), optionally validate every point on the device, register both bases as library-owned handles.
static int stream_file_to_device(FILE *f, void *dev, size_t bytes, void *pinned[2], size_t chunk) {
  hipEvent_t ev[2] = {nullptr, nullptr};
  for (int i = 0; i < 2; i++) { HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); HIPCHK(hipEventRecord(ev[i], g.stream)); }
  size_t done = 0; int which = 0; int rc = MI355_OK;
  while (done < bytes && rc == MI355_OK) {
    const size_t len = std::min(chunk, bytes - done);
    if (hipEventSynchronize(ev[which]) != hipSuccess) { rc = fail(MI355_EHIP, "srs_load_params_file: event wait failed"); break; }   // the previous copy out of this staging buffer has finished
    if (fread(pinned[which], 1, len, f) != len) { rc = fail(MI355_EBADARG, "srs_load_params_file: short read"); break; }
    if (hipMemcpyAsync((char *)dev + done, pinned[which], len, hipMemcpyHostToDevice, g.stream) != hipSuccess || hipEventRecord(ev[which], g.stream) != hipSuccess) { (void)hipGetLastError(); rc = fail(MI355_EHIP, "srs_load_params_file: copy failed"); break; }
    done += len; which ^= 1;
  }
  (void)hipStreamSynchronize(g.stream);
  for (int i = 0; i < 2; i++) (void)hipEventDestroy(ev[i]);
  return rc;
}
int mi355_srs_load_params_file(const char *path, uint32_t flags, uint32_t *k_out, uint64_t *g_handle_out, uint64_t *g_lagrange_handle_out, void *g2_out, void *s_g2_out) {
  return guarded([&]() -> int {
  AllGuard lk;
  CHK(need_init());
  if (!path || !k_out || !g_handle_out || !g_lagrange_handle_out) return fail(MI355_EBADARG, "srs_load_params_file: null pointer");
  FILE *f = fopen(path, "rb");
  if (!f) return fail(MI355_EBADARG, std::string("srs_load_params_file: cannot open ") + path);
  uint8_t hdr[4];
  int rc = MI355_OK; Srs sg, sl; void *pinned[2] = {nullptr, nullptr};
  do {
    if (fread(hdr, 1, 4, f) != 4) { rc = fail(MI355_EBADARG, "srs_load_params_file: empty file"); break; }
    const uint32_t k = (uint32_t)hdr[0] | ((uint32_t)hdr[1] << 8) | ((uint32_t)hdr[2] << 16) | ((uint32_t)hdr[3] << 24);
    if (k == 0 || k > 28) { rc = fail(MI355_EBADARG, "srs_load_params_file: k out of range (not a RawBytes params file)"); break; }
    const uint64_t n = 1ull << k, want = 4 + 2 * n * sizeof(g1_affine_t) + 256;
    if (fseek(f, 0, SEEK_END) != 0 || (uint64_t)ftell(f) != want) { rc = fail(MI355_EBADARG, "srs_load_params_file: file length does not match 4 + 2 * 2^k * 64 + 256 (load_params rejects it too)"); break; }
    fseek(f, 4, SEEK_SET);
    const size_t chunk = std::min<uint64_t>(64ull << 20, n * sizeof(g1_affine_t));
    sg.n = sl.n = n; sg.mem = std::make_shared<SrsMem>(); sl.mem = std::make_shared<SrsMem>();
    if (srs_alloc(*sg.mem, n) != MI355_OK || srs_alloc(*sl.mem, n) != MI355_OK) { (void)hipGetLastError(); rc = fail(MI355_EOOM, "srs_load_params_file: device allocation failed"); break; }
    bool okp = true;
    for (int i = 0; i < 2; i++) okp = okp && hipHostMalloc(&pinned[i], chunk, hipHostMallocPortable) == hipSuccess;
    if (!okp) { rc = fail(MI355_EOOM, "srs_load_params_file: pinned staging allocation failed"); break; }
    for (Srs *b : {&sg, &sl}) for (auto &sh : b->mem->sh) {   // the file holds g then g_lagrange, each in point order = shard order
      if (rc != MI355_OK) break;
      if ((rc = bind_ctx(sh.slot)) != MI355_OK) break;
      rc = stream_file_to_device(f, sh.dev, sh.n * sizeof(g1_affine_t), pinned, chunk);
    }
    if (rc != MI355_OK) break;
    uint8_t tail[256];
    if (fread(tail, 1, 256, f) != 256) { rc = fail(MI355_EBADARG, "srs_load_params_file: short read (g2 / s_g2)"); break; }
    if (g2_out) memcpy(g2_out, tail, 128);
    if (s_g2_out) memcpy(s_g2_out, tail + 128, 128);
    if (flags & 1u) {
      uint64_t nbad_total = 0;
      for (Srs *b : {&sg, &sl}) for (auto &sh : b->mem->sh) {
        if ((rc = bind_ctx(sh.slot)) != MI355_OK) break;
        uint32_t *bad = nullptr;
        if (ws_get("io.validate", 4, (void **)&bad) != MI355_OK) { rc = MI355_EHIP; break; }
        (void)hipMemsetAsync(bad, 0, 4, g.stream);
        hipLaunchKernelGGL(k_g1_validate, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, sh.dev, sh.n, bad);
        uint32_t nbad = 0;
        if (hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, g.stream) != hipSuccess || hipStreamSynchronize(g.stream) != hipSuccess) { rc = fail(MI355_EHIP, "srs_load_params_file: validation failed to run"); break; }
        nbad_total += nbad;
      }
      if (rc != MI355_OK) break;
      if (nbad_total) { rc = fail(MI355_EBADARG, "srs_load_params_file: " + std::to_string(nbad_total) + " point(s) are not on the curve"); break; }
    }
    *k_out = k;
    *g_handle_out = srs_insert(sg);
    *g_lagrange_handle_out = srs_insert(sl);
  } while (false);
  fclose(f);
  for (int i = 0; i < 2; i++) if (pinned[i]) (void)hipHostFree(pinned[i]);
  const std::string keep = g_err;
  (void)bind_ctx(0);
  if (rc != MI355_OK) g_err = keep;
  return rc;   // on failure the shared_ptrs in sg / sl free whatever was allocated
  });
}
int mi355_srs_precompute(uint64_t handle, uint64_t n_hint, int c) {
  return guarded([&]() -> int {
  AllGuard lk;
  CHK(need_init());
  Srs *sp; CHK(srs_find(handle, &sp, "srs_precompute"));
  SrsMem &mem = *sp->mem;
  if (n_hint == 0 || n_hint > sp->n) n_hint = sp->n;
  size_t live = 0; for (const auto &sh : mem.sh) if (sh.lo < sp->n) live++;
  const uint64_t per_shard = std::max<uint64_t>(1, n_hint / std::max<size_t>(1, live));
  const bool automatic = c == 0;
  if (automatic) {   // best shared-bucket window for MSMs of n_hint points (per shard), within the sorter's key range
    double best = 1e300;
    for (int cc = 4; cc <= g_auto_max_c; cc++) { const double co = msm_cost(per_shard, cc, true); if (co < best) { best = co; c = cc; } }
  }
  if (c < 2 || c > MSM_MAX_C) return fail(MI355_EBADARG, "srs_precompute: window bits out of range");
  if (sp->tab) {
    if (sp->tab->c == c) return MI355_OK;
    // tables inherited from the parent basis (or built earlier with another width) stay when they are within 10 % of the best schedule
    if (automatic && msm_cost(per_shard, sp->tab->c, true) <= 1.10 * msm_cost(per_shard, c, true)) return MI355_OK;
  }
  const int W = (MSM_SCALAR_BITS + c - 1) / c;
  auto tab = std::make_shared<SrsTables>();
  tab->c = c; tab->w = W;
  tab->pre.assign(mem.sh.size(), nullptr); tab->stride.assign(mem.sh.size(), 0); tab->slot.assign(mem.sh.size(), 0);
  for (size_t i = 0; i < mem.sh.size(); i++) {
    const Shard &sh = mem.sh[i];
    tab->slot[i] = sh.slot;
    if (sh.lo >= sp->n) continue;
    const uint64_t cnt = std::min(sh.n, sp->n - sh.lo);   // a prefix view tabulates only its own points
    CHK(bind_ctx(sh.slot));
    CHK(dev_malloc((void **)&tab->pre[i], (size_t)W * cnt * sizeof(g1_affine_t), "srs_precompute (window table)"));
    tab->stride[i] = cnt;
    hipLaunchKernelGGL(k_srs_precompute, dim3(ceil_div(cnt, 256)), dim3(256), 0, g.stream, sh.dev, tab->pre[i], cnt, (uint32_t)W, (uint32_t)c);
    HIPCHK(hipGetLastError());
  }
  for (const auto &sh : mem.sh) { CHK(bind_ctx(sh.slot)); HIPCHK(hipStreamSynchronize(g.stream)); }
  sp->tab = tab;   // the previous tables (if any) are freed here unless another handle still shares them
  return bind_ctx(0);
  });
}

// ---- MSM
// the pieces of [off, off + n) of a registered basis, one per shard that intersects it
struct Piece { int slot; const g1_affine_t *bases; PreTable pre; uint64_t a /* first scalar of the call */, cnt; };
static int srs_pieces(uint64_t handle, uint64_t off, uint64_t n, std::vector<Piece> &out) {
  Srs *sp; CHK(srs_find(handle, &sp, "msm"));
  if (off > sp->n || n > sp->n - off) return fail(MI355_EBADARG, "msm: base_offset + n exceeds the registered basis (best_multiexp panics on length mismatch)");
  const SrsMem &mem = *sp->mem;
  for (size_t i = 0; i < mem.sh.size(); i++) {
    const Shard &sh = mem.sh[i];
    const uint64_t lo = std::max(off, sh.lo), hi = std::min(off + n, sh.lo + sh.n);
    if (hi <= lo) continue;
    Piece p; p.slot = sh.slot; p.bases = sh.dev + (lo - sh.lo); p.a = lo - off; p.cnt = hi - lo;
    if (sp->tab && sp->tab->pre[i] && hi - sh.lo <= sp->tab->stride[i]) { p.pre.table = sp->tab->pre[i] + (lo - sh.lo); p.pre.row_stride = sp->tab->stride[i]; p.pre.c = sp->tab->c; p.pre.w = sp->tab->w; }
    out.push_back(p);
  }
  if (out.empty()) { Piece p; p.slot = 0; p.bases = nullptr; p.a = 0; p.cnt = 0; out.push_back(p); }   // n == 0
  return MI355_OK;
}

enum ScalarLoc { SCALARS_HOST = 0, SCALARS_DEV = 1 };
// Sharded MSM: every device that owns part of the point range computes the partial sum over its shard on its own stream, driven by
// its own host thread (copies from pageable memory block the issuing thread, so one thread per device keeps the PCIe links busy in parallel);
// the M x 96-byte partials are exchanged with ONE ncclAllGather per device (grouped) and folded + normalised on the primary device.
// polys: M pointers to n scalars each, host memory or device memory of any bound device.
static int msm_multi(const std::vector<Piece> &pieces, const fe_t *const *polys, ScalarLoc loc, uint32_t M, uint64_t n, void *out_host) {
  (void)n;
  const int D = g_ndev;
  const MsmOpts opts = t_opts;
  std::vector<int> rcs(D, MI355_OK); std::vector<std::string> errs(D);
  std::vector<const Piece *> by_slot(D, nullptr);
  for (const auto &p : pieces) by_slot[p.slot] = &p;
  // device-resident scalars may still be in flight on their owner's stream: the other devices' copies must wait for it
  std::vector<int> owner(M, 0);
  if (loc == SCALARS_DEV) {
    bool seen[MAX_DEV] = {false};
    for (uint32_t m = 0; m < M; m++) { owner[m] = slot_of(polys[m]); seen[owner[m]] = true; }
    if (D > 1) for (int s = 0; s < D; s++) if (seen[s]) { CHK(bind_ctx(s)); HIPCHK(hipStreamSynchronize(g.stream)); }
  }
  auto work = [&](int slot) {
    t_opts = opts; t_opts.normalise = false;            // partials are folded (and normalised once) after the exchange
    auto body = [&]() -> int {
      CHK(bind_ctx(slot));
      g1_jac_t *send; CHK(ws_get("xchg.send", (size_t)M * sizeof(g1_jac_t), (void **)&send));
      const Piece *p = by_slot[slot];
      if (!p || p->cnt == 0) { HIPCHK(hipMemsetAsync(send, 0, (size_t)M * sizeof(g1_jac_t), g.stream)); return MI355_OK; }
      std::vector<const fe_t *> ptrs(M);
      if (loc == SCALARS_HOST) {   // this device's slice of every polynomial crosses its own PCIe link, chunk-overlapped with the compute
        for (uint32_t m = 0; m < M; m++) ptrs[m] = polys[m] + p->a;
        return msm_host_single(p->bases, ptrs.data(), M, p->cnt, nullptr, &p->pre, send);
      }
      // device-resident scalars: a polynomial that lives on this (physical) device is read in place; one that lives on another bound device --
      // the primary as a rule, any device with mi355_buf_alloc(.., slot) -- crosses xGMI into a staging buffer first
      fe_t *sc = nullptr;
      for (uint32_t m = 0; m < M; m++) {
        const int o = owner[m];
        if (g_ctx[o].device == g.device) { ptrs[m] = polys[m] + p->a; continue; }
        if (!sc) CHK(ws_get("io.scalars", (size_t)M * p->cnt * sizeof(fe_t), (void **)&sc));
        ptrs[m] = sc + (size_t)m * p->cnt;
        HIPCHK(hipMemcpyPeerAsync(sc + (size_t)m * p->cnt, g.device, polys[m] + p->a, g_ctx[o].device, p->cnt * sizeof(fe_t), g.stream));
      }
      return msm_batch_impl(p->bases, ptrs.data(), M, p->cnt, nullptr, &p->pre, send);
    };
    rcs[slot] = guarded(body, "msm shard worker");
    if (rcs[slot] != MI355_OK) errs[slot] = g_err;
  };
  {
    // every started worker is joined on every path (a std::thread that is destroyed while joinable terminates the process)
    struct Joiner { std::vector<std::thread> th; ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); } } workers;
    workers.th.reserve(D);
    for (int s = 1; s < D; s++) workers.th.emplace_back(work, s);
    work(0);
  }
  t_opts = opts;
  CHK(bind_ctx(0));
  for (int s = 0; s < D; s++) if (rcs[s] != MI355_OK) return fail(rcs[s], "device slot " + std::to_string(s) + ": " + errs[s]);
  // ---- exchange: D x M x 96 bytes
  const size_t part_bytes = (size_t)M * sizeof(g1_jac_t);
  std::vector<g1_jac_t *> send(D), recv(D);
  for (int s = 0; s < D; s++) { CHK(bind_ctx(s)); CHK(ws_get("xchg.send", part_bytes, (void **)&send[s])); CHK(ws_get("xchg.recv", part_bytes * D, (void **)&recv[s])); }
  if (g_ctx[0].comm) {
    int r = g_rccl.GroupStart(); if (r != 0) return rccl_fail("ncclGroupStart", r);
    for (int s = 0; s < D; s++) {
      CHK(bind_ctx(s));
      r = g_rccl.AllGather(send[s], recv[s], part_bytes, /* ncclUint8 */ 1, g_ctx[s].comm, g_ctx[s].stream);
      if (r != 0) { (void)g_rccl.GroupEnd(); return rccl_fail("ncclAllGather", r); }
    }
    r = g_rccl.GroupEnd(); if (r != 0) return rccl_fail("ncclGroupEnd", r);
    g_last_exchange = "rccl_allgather";
  } else {
    // several slots on one physical device (test mode): no communicator can exist; the partials are copied device-to-device
    for (int s = 0; s < D; s++) {
      CHK(bind_ctx(s)); HIPCHK(hipEventRecord(g.ev_xchg, g.stream));
      CHK(bind_ctx(0)); HIPCHK(hipStreamWaitEvent(g.stream, g_ctx[s].ev_xchg, 0));
      HIPCHK(hipMemcpyAsync((char *)recv[0] + part_bytes * s, send[s], part_bytes, hipMemcpyDeviceToDevice, g.stream));
    }
    g_last_exchange = "device_copy";
  }
  CHK(bind_ctx(0));
  g1_jac_t *res; CHK(ws_get("xchg.result", part_bytes, (void **)&res));
  hipLaunchKernelGGL(k_g1_sum_strided, dim3(M), dim3(64), 0, g.stream, (const g1_jac_t *)recv[0], (uint32_t)D, M, res, opts.normalise ? 1 : 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_host, res, part_bytes, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  for (int s = D - 1; s >= 0; s--) { CHK(bind_ctx(s)); if (s) HIPCHK(hipStreamSynchronize(g.stream)); resolve_spans(); }
  g_last_devices = D;
  return MI355_OK;
}
static bool single_device_call(const std::vector<Piece> &pieces) { return pieces.size() == 1 && pieces[0].slot == 0 && !g_force_exchange; }

// host scalars, one device: mi355_msm_g1_host.  The copy is cut into chunks on a second stream; the digit extraction of chunk j runs while
// chunk j + 1 crosses PCIe (msm_host_single in the MSM section above).
static int msm_host_dispatch(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_host, uint32_t M, uint64_t n, void *out_g1_host) {
  std::vector<Piece> pieces; CHK(srs_pieces(srs_handle, base_offset, n, pieces));
  if (M == 0 || n == 0) { if (M) memset(out_g1_host, 0, (size_t)M * sizeof(g1_jac_t)); return MI355_OK; }
  if (!single_device_call(pieces)) return msm_multi(pieces, (const fe_t *const *)scalars_host, SCALARS_HOST, M, n, out_g1_host);
  const Piece &p = pieces[0];
  g_last_devices = 1; g_last_exchange = "none";
  // staged in groups of at most 4 GiB of scalars
  const uint32_t group_max = (uint32_t)std::max<uint64_t>(1, (4ull << 30) / (n * sizeof(fe_t)));
  for (uint32_t m0 = 0; m0 < M; m0 += group_max) {
    const uint32_t mg = std::min(group_max, M - m0);
    CHK(msm_host_single(p.bases, (const fe_t *const *)scalars_host + m0, mg, n, (char *)out_g1_host + (size_t)m0 * sizeof(g1_jac_t), &p.pre));
  }
  return MI355_OK;
}
static int msm_dev_dispatch(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_dev, uint32_t M, uint64_t n, void *out_g1_host) {
  std::vector<Piece> pieces; CHK(srs_pieces(srs_handle, base_offset, n, pieces));
  if (M == 0 || n == 0) { if (M) memset(out_g1_host, 0, (size_t)M * sizeof(g1_jac_t)); return MI355_OK; }
  if (!single_device_call(pieces)) return msm_multi(pieces, (const fe_t *const *)scalars_dev, SCALARS_DEV, M, n, out_g1_host);
  g_last_devices = 1; g_last_exchange = "none";
  // Every scalar block is marked as used by queued work (slot_of's `touch`), with one device as with several: an upload into a block whose only
  // use since mi355_buf_alloc was as an MSM input must wait for the compute stream, not take the "fresh block" path.
  // Several devices bound, basis on the primary only: scalars that live on another device are read through peer access, after their producer;
  // when peer access to that device could not be enabled they are staged on the primary first (hipMemcpyPeerAsync needs no peer mapping).
  std::vector<const fe_t *> ptrs(M);
  fe_t *stage = nullptr; size_t staged = 0;
  for (uint32_t m = 0; m < M; m++) {
    const int o = slot_of(scalars_dev[m], true);
    ptrs[m] = (const fe_t *)scalars_dev[m];
    if (g_ndev > 1 && o != 0) {
      CHK(bind_ctx(o)); HIPCHK(hipStreamSynchronize(g.stream));
      if (!g_peer_ok[0][o]) {
        CHK(bind_ctx(0));
        if (!stage) CHK(ws_get("io.scalars", (size_t)M * n * sizeof(fe_t), (void **)&stage));
        HIPCHK(hipMemcpyPeerAsync(stage + staged, g.device, scalars_dev[m], g_ctx[o].device, n * sizeof(fe_t), g.stream));
        ptrs[m] = stage + staged; staged += n;
      }
    }
  }
  CHK(bind_ctx(0));
  return msm_batch_impl(pieces[0].bases, ptrs.data(), M, n, out_g1_host, &pieces[0].pre);
}

int mi355_msm_g1_dev(uint64_t srs_handle, uint64_t base_offset, const void *scalars_dev, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  MsmGuard lk;
  CHK(need_init());
  if (!out_g1_host || (n && !scalars_dev)) return fail(MI355_EBADARG, "msm: null pointer");
  return msm_dev_dispatch(srs_handle, base_offset, &scalars_dev, 1, n, out_g1_host);
  });
}
int mi355_msm_g1_dev_async(uint64_t srs_handle, uint64_t base_offset, const void *scalars_dev, uint64_t n, void *out_g1_dev) {
  return guarded([&]() -> int {
  MsmGuard lk;
  CHK(need_init());
  if (!out_g1_dev || (n && !scalars_dev)) return fail(MI355_EBADARG, "msm: null pointer");
  std::vector<Piece> pieces; CHK(srs_pieces(srs_handle, base_offset, n, pieces));
  if (pieces.size() != 1 || pieces[0].slot != 0) return fail(MI355_EBADARG, "msm_g1_dev_async: the range must lie on the primary device (one process per GPU drives its own shard)");
  const fe_t *sc = (const fe_t *)scalars_dev;
  (void)slot_of(scalars_dev, true); (void)slot_of(out_g1_dev, true);   // both blocks now carry queued work: a later upload into either waits for the compute stream
  return msm_batch_impl(pieces[0].bases, &sc, 1, n, nullptr, &pieces[0].pre, out_g1_dev);
  });
}
int mi355_g1_sum_dev(const void *pts_dev, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  const int slot = slot_of(pts_dev);
  DevGuard lk(slot);
  CHK(need_init(slot));
  if (!out_g1_host || (n && !pts_dev) || n > (1u << 20)) return fail(MI355_EBADARG, "g1_sum: bad argument");
  g1_jac_t *dev; CHK(ws_get("io.g1sum", sizeof(g1_jac_t), (void **)&dev));
  hipLaunchKernelGGL(k_g1_sum, dim3(1), dim3(64), 0, g.stream, (const g1_jac_t *)pts_dev, (uint32_t)n, dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_g1_host, dev, sizeof(g1_jac_t), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return MI355_OK;
  });
}
int mi355_msm_g1_host(uint64_t srs_handle, uint64_t base_offset, const void *scalars_host, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  MsmGuard lk;
  CHK(need_init());
  if (!out_g1_host || (n && !scalars_host)) return fail(MI355_EBADARG, "msm: null pointer");
  return msm_host_dispatch(srs_handle, base_offset, &scalars_host, 1, n, out_g1_host);
  });
}
int mi355_msm_g1_batch_dev(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_dev, uint32_t batch, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  MsmGuard lk;
  CHK(need_init());
  if (!out_g1_host || (batch && !scalars_dev)) return fail(MI355_EBADARG, "msm_batch: null pointer");
  for (uint32_t m = 0; m < batch; m++) if (n && !scalars_dev[m]) return fail(MI355_EBADARG, "msm_batch: null polynomial pointer");
  return msm_dev_dispatch(srs_handle, base_offset, scalars_dev, batch, n, out_g1_host);
  });
}
int mi355_msm_g1_batch_host(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_host, uint32_t batch, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  MsmGuard lk;
  CHK(need_init());
  if (!out_g1_host || (batch && !scalars_host)) return fail(MI355_EBADARG, "msm_batch: null pointer");
  for (uint32_t m = 0; m < batch; m++) if (n && !scalars_host[m]) return fail(MI355_EBADARG, "msm_batch: null polynomial pointer");
  return msm_host_dispatch(srs_handle, base_offset, scalars_host, batch, n, out_g1_host);
  });
}
int mi355_msm_set_pipeline(uint32_t chunks, uint32_t min_log_n) {
  return guarded([&]() -> int {
  AllGuard lk;
  if (chunks > 16 || min_log_n > 31) return fail(MI355_EBADARG, "msm_set_pipeline: chunks <= 16, min_log_n <= 31");
  for (int s = 0; s < std::max(1, g_ndev); s++) { g_ctx[s].msm_chunks = chunks ? chunks : 1; g_ctx[s].msm_chunk_min_log = chunks ? min_log_n : 23; }
  return MI355_OK;
  });
}
int mi355_msm_g1_adhoc_host(const void *bases_host, const void *scalars_host, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  MsmGuard lk;
  CHK(need_init());
  if (!out_g1_host || (n && (!scalars_host || !bases_host))) return fail(MI355_EBADARG, "msm: null pointer");
  fe_t *sc = nullptr; g1_affine_t *bs = nullptr;
  if (n) {
    CHK(ws_get("io.scalars", n * sizeof(fe_t), (void **)&sc)); CHK(ws_get("io.bases", n * sizeof(g1_affine_t), (void **)&bs));
    HIPCHK(hipMemcpyAsync(sc, scalars_host, n * sizeof(fe_t), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(bs, bases_host, n * sizeof(g1_affine_t), hipMemcpyHostToDevice, g.stream));
  }
  g_last_devices = 1; g_last_exchange = "none";
  return msm_dev_impl(bs, sc, n, out_g1_host);
  });
}
int mi355_g1_sum_host(const void *pts_host, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  CHK(need_init());
  if (!out_g1_host || (n && !pts_host) || n > (1u << 20)) return fail(MI355_EBADARG, "g1_sum: bad argument");
  g1_jac_t *dev; CHK(ws_get("io.g1sum", (n + 1) * sizeof(g1_jac_t), (void **)&dev));
  if (n) HIPCHK(hipMemcpyAsync(dev + 1, pts_host, n * sizeof(g1_jac_t), hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_g1_sum, dim3(1), dim3(64), 0, g.stream, dev + 1, (uint32_t)n, dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_g1_host, dev, sizeof(g1_jac_t), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return MI355_OK;
  });
}
// The two setters below act on the CALLING THREAD only (thread-local options): concurrent callers never see each other's settings.
int mi355_msm_set_window_bits(int c) {
  return guarded([&]() -> int {
  if (c == -1) { t_opts.force_c = 0; t_opts.no_tables = true; return MI355_OK; }   // automatic window, window tables ignored
  if (c != 0 && (c < 2 || c > MSM_MAX_C)) return fail(MI355_EBADARG, "window bits must be 0 (auto), -1 (auto, no window tables) or in [2, 24]");
  t_opts.force_c = c; t_opts.no_tables = false; return MI355_OK;
  });
}
int mi355_msm_set_normalise(int on) { t_opts.normalise = on != 0; return MI355_OK; }
int mi355_msm_last_plan(int *c_out, int *windows_out, uint64_t *entries_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  const Ctx &p = g_ctx[0];
  if (c_out) *c_out = p.last_c; if (windows_out) *windows_out = p.last_w; if (entries_out) *entries_out = p.last_entries; return MI355_OK;
  });
}
// how the last MSM ran: device slots that took part, the exchange ("none" | "rccl_allgather" | "device_copy"), whether the window tables
// (one shared bucket set) were used, and the number of point-range slices of the host-pointer path
int mi355_msm_last_run(int *devices_out, const char **exchange_out, int *shared_tables_out, int *host_slices_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  if (devices_out) *devices_out = g_last_devices; if (exchange_out) *exchange_out = g_last_exchange;
  if (shared_tables_out) *shared_tables_out = g_ctx[0].last_shared ? 1 : 0; if (host_slices_out) *host_slices_out = g_ctx[0].last_host_slices;
  return MI355_OK;
  });
}

// ---- synthetic SRS
static int ensure_fixed_base_table() {
  if (g.fixed_base_table) return MI355_OK;
  CHK(dev_malloc((void **)&g.fixed_base_table, 32 * 256 * sizeof(g1_affine_t), "fixed-base table"));
  hipLaunchKernelGGL(k_fixed_base_table, dim3(1), dim3(256), 0, g.stream, g.fixed_base_table);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}
int mi355_g1_fixed_base_mul_dev(void *points_dev, const void *scalars_dev, uint64_t n) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({points_dev, scalars_dev}, &slot, "fixed_base_mul"));
  DevGuard lk(slot);
  CHK(need_init(slot));
  if (!points_dev || !scalars_dev) return fail(MI355_EBADARG, "fixed_base_mul: null pointer");
  CHK(ensure_fixed_base_table());
  hipLaunchKernelGGL(k_fixed_base_mul, dim3(ceil_div(n, 256)), dim3(256), 0, g.stream, g.fixed_base_table, (const fe_t *)scalars_dev, (g1_affine_t *)points_dev, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_srs_setup_dev(void *g_dev, void *g_lagrange_dev, uint32_t k, const void *tau, const void *omega) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({g_dev, g_lagrange_dev}, &slot, "srs_setup"));
  DevGuard lk(slot);
  CHK(need_init(slot));
  if (!g_dev || !g_lagrange_dev || !tau || !omega || k > 28) return fail(MI355_EBADARG, "srs_setup: bad argument");
  CHK(ensure_fixed_base_table());
  const uint64_t n = 1ull << k;
  fe_t *sc; CHK(ws_get("srs.scalars", 2 * n * sizeof(fe_t), (void **)&sc));
  fe_t t, w; memcpy(&t, tau, 32); memcpy(&w, omega, 32);
  // (tau^n - 1) / n: one-off constant, formed on the host with the same limb code
  fe_t nn = Fr::zero(); nn.l[0] = (uint32_t)n; nn.l[1] = (uint32_t)(n >> 32);
  const fe_t tn1_over_n = Fr::mul(Fr::sub(Fr::pow_u64(t, n), Fr::one()), Fr::inv(Fr::from_canonical(nn)));
  hipLaunchKernelGGL(k_srs_scalars, dim3(ceil_div(n, 256)), dim3(256), 0, g.stream, sc, sc + n, t, w, tn1_over_n, n);
  hipLaunchKernelGGL(k_fixed_base_mul, dim3(ceil_div(n, 256)), dim3(256), 0, g.stream, g.fixed_base_table, sc, (g1_affine_t *)g_dev, n);
  hipLaunchKernelGGL(k_fixed_base_mul, dim3(ceil_div(n, 256)), dim3(256), 0, g.stream, g.fixed_base_table, sc + n, (g1_affine_t *)g_lagrange_dev, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(g.stream));
  return MI355_OK;
  });
}

}  // extern "C"
