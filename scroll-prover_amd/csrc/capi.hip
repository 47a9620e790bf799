// capi.hip -- the C-ABI of libmi355zk.so (include/mi355zk.h): context, SRS handles, workspace arena,
// MSM / NTT launch orchestration and HIP-event profiling.  Host logic only; all arithmetic runs in the
// kernels of msm.cuh / ntt.cuh.  There is deliberately no CPU fallback: without a gfx950 device every
// compute entry point returns MI355_ENODEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mi355zk.h"
#include "msm.cuh"
#include "ntt.cuh"
#include "ntt29.cuh"
#include "g1fft.cuh"
#include "frscan.cuh"
#include "g2.cuh"

using namespace zk;

// the ABI is plain bytes: these sizes are what the Rust / C++ / Python bindings assume (halo2curves Fr 32 B, G1Affine 64 B, G1 96 B)
static_assert(sizeof(fe_t) == 32 && sizeof(g1_affine_t) == 64 && sizeof(g1_jac_t) == 96, "ABI element sizes");
static_assert(sizeof(g2_affine_t) == 128, "G2Affine is 128 bytes (x.c0, x.c1, y.c0, y.c1)");
static_assert(sizeof(g1_xyzz_t) == 128 && sizeof(g1_xyzz29_t) == 144, "device record sizes (workspace layout, 16-byte vector accesses)");

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                                     \
  do {                                                                                                                   \
    hipError_t _e = (expr);                                                                                              \
    if (_e != hipSuccess) {                                                                                              \
      char _b[512]; snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      (void)hipGetLastError();                                                                                           \
      return fail(_e == hipErrorOutOfMemory ? MI355_EOOM : MI355_EHIP, _b);                                              \
    }                                                                                                                    \
  } while (0)
#define CHK(expr) do { int _r = (expr); if (_r != MI355_OK) return _r; } while (0)

// A registered basis.  One device: ONE shard holding all n points.  mi355_init_multi with D devices: shard d = points [lo, lo + n) on device slot d
// (fixed point-range shards); window tables are per shard (row stride = shard length).  The device memory (SrsMem) is shared
// between a handle and its prefix views (mi355_srs_register_prefix) and freed with the last of them.
struct Shard { int slot = 0; uint64_t lo = 0, n = 0; g1_affine_t *dev = nullptr; bool owned = false; };
void free_shards(std::vector<Shard> &sh);
struct SrsMem { std::vector<Shard> sh; ~SrsMem() { free_shards(sh); } };
// window tables T[w][i] = 2^(c w) P_i, one allocation per shard (pre[i] belongs to shard i of the basis, rows `stride[i]` points apart).
// A prefix view starts out sharing its parent's tables and gets private ones only when a much smaller n calls for another window width.
struct SrsTables { std::vector<g1_affine_t *> pre; std::vector<uint64_t> stride; std::vector<int> slot; int c = 0, w = 0; ~SrsTables(); };
struct Srs { uint64_t n = 0; std::shared_ptr<SrsMem> mem; std::shared_ptr<SrsTables> tab; };
struct Buf { void *p = nullptr; size_t cap = 0; };
struct NttPlan {
  uint32_t log_n = 0, levels = 0, log_m[3] = {0, 0, 0};
  fe_t *tw_m[3] = {nullptr, nullptr, nullptr};
  fe_t *tw_s_lo[2] = {nullptr, nullptr}, *tw_s_hi[2] = {nullptr, nullptr};
  uint32_t split[2] = {0, 0};
  uint32_t direct2[2] = {0, 0};
  std::map<std::string, Tw29> scaled;   // the last strided level's direct twiddle table times a constant (key: the 32 bytes of the constant)   // tw29_s_lo[l] is the 2^log_s-entry table [k][column] of a big level (ntt29.cuh)
  // the same tables as w * 2^261 mod r in 29-bit limbs (SoA) for the unsaturated kernels (ntt29.cuh)
  Tw29 tw29_m[3] = {}, tw29_s_lo[2] = {}, tw29_s_hi[2] = {};
  std::vector<void *> owned;
};
struct Prof { double ms = 0; uint64_t launches = 0; };

struct MsmSlot { int id = 0; hipEvent_t sorted = nullptr, acc_done = nullptr, red_done = nullptr; bool used = false; };   // per-chunk buffers + events of the pipelined MSM

struct Span { std::string name; hipEvent_t a, b; };
struct Ctx {
  bool inited = false;
  int slot = 0;                 // index in g_ctx (0 = primary)
  std::vector<Span> spans;      // profiling: (name, start, stop) event pairs resolved after the stream is idle
  g1_jac_t *xchg_send = nullptr, *xchg_recv = nullptr;   // multi-device exchange: this device's 96-byte partial / the gathered D x 96 bytes
  void *comm = nullptr;         // ncclComm_t of this device (mi355_init_multi with distinct devices)
  hipEvent_t ev_xchg = nullptr;
  // host-pointer MSM: chunked copy on its own stream, overlapped with the digit extraction (msm_host_single)
  hipStream_t copy_stream = nullptr; hipEvent_t ev_copy[4] = {nullptr, nullptr, nullptr, nullptr};
  void *pin_ring = nullptr; size_t pin_ring_bytes = 0;
  uint32_t host_slice_min_log = 22;   // MI355_HOST_SLICE_MIN_LOG: smallest log2(n) the host-pointer MSM cuts into slices (tests lower it)
  uint32_t host_chunks = 8;     // MI355_HOST_CHUNKS: upper bound on the point-range slices of the host-pointer MSM (1 = one copy, then compute)
  int device = -1;
  hipDeviceProp_t prop;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipStream_t aux_stream[2] = {nullptr, nullptr};   // side streams of the pipelined MSM (sort | reduction); the accumulation stays on `stream`
  hipEvent_t ev_fork = nullptr;
  MsmSlot msm_slot[2];
  std::vector<const fe_t *> polys_stage;
  uint32_t msm_chunks = 1;       // MI355_MSM_CHUNKS / mi355_msm_set_pipeline (off by default: measured slower)
  uint32_t msm_chunk_min_log = 23;
  int last_chunks = 1;
  std::map<std::string, Buf> ws;           // grow-only workspace arena, keyed by role
  std::map<std::string, NttPlan> ntt_plans;  // key = log_n | omega bytes
  g1_affine_t *fixed_base_table = nullptr;
  uint32_t sort_t2 = 0;         // MI355_SORT_T2 = 8192 | 16384 (0: by size)
  uint32_t debug_gather_mask = 0x7fffffffu;   // MI355_DEBUG_GATHER_MASK (timing experiments only: results become wrong)
  uint32_t acc_variant = 4;     // MI355_ACC_VARIANT: 4 = limb products of k_msm_accumulate as column blocks of chained v_mad (fp29_asm_gen.inc): 57.1 vs 59.5 ms at 2^26, bit-identical; 0 = the plain C++ multiplier
  uint32_t seg_factor = 16;
  uint32_t sort_fb = 11;        // MI355_SORT_FB: fine (level-2) key bits of the sorter, 9..12
  uint32_t reduce_chains = 131072;   // MI355_REDUCE_CHAINS: target number of running-sum chains of the bucket reduction
  uint32_t seg_fill = 40, seg_fill_segfix = 40;   // MI355_SEG_FILL / MI355_SEG_FILL_SEGFIX (even, 2..100 %: above 100 the segments would no longer cover the entries): share of the launched accumulate threads the actual entries are spread over
  uint32_t seg_min = 16;             // MI355_SEG_MIN: shortest accumulate segment (entries per thread) when few digits are non-zero
  uint32_t fixup_mode = 2;           // MI355_FIXUP_MODE: 2 = by shape (see msm_enqueue), 0 = per-bucket kernels (four lanes / workgroup / several workgroups per bucket), 1 = one segmented reduction over the partial sums (k_msm_segfix: measured better for two-partial buckets, worse for spans of 15-30, profiles/r02b_segfix_ab.log)
  uint32_t fixup_huge_min = 2048;    // MI355_FIXUP_HUGE_MIN (>= 2048): bucket spans from this many accumulate threads on are summed by several workgroups
  uint32_t fixup_serial_max = 32;    // MI355_FIXUP_SERIAL_MAX: bucket spans (in accumulate threads) above this go to the workgroup-per-bucket fix-up
  uint32_t fixup_lanes_max_log = 17;  // MI355_FIXUP_LANES_MAX_LOG: bucket sets up to 2^this records take the four-lanes-per-bucket fix-up
  uint32_t reduce_min_chunk = 4;     // MI355_REDUCE_MIN_CHUNK: shortest running-sum chain (buckets per reduce thread) small bucket sets are cut into
  uint32_t sort_t1 = 16384;     // MI355_SORT_T1=8192 selects the smaller level-1 tile (2 workgroups per CU)
  uint32_t ntt_direct2_max_log = 25;   // MI355_NTT_DIRECT2_MAX_LOG: levels up to 2^this elements read their inter-level twiddles from a full table (36 B per element of the level: 2^24 transform 2.41 -> 2.28 ms for 0.6 GB; at 2^26 the 2.4 GB table only buys 1.7 %, so the default stops at 2^25); 0 disables
  uint32_t ntt_fold_scale = 1;  // MI355_NTT_FOLD_SCALE=0: the inverse transform's divisor stays a multiplication in the closing pass
  uint32_t ntt_radix_log = 2;   // MI355_NTT_RADIX_LOG
  uint32_t ntt_tile_log = 11;   // log2 of the LDS tile in elements (MI355_NTT_TILE_LOG)
  bool ntt29 = true;   // unsaturated 29-bit NTT kernels (MI355_NTT_SAT=1 selects the saturated 8x32 ones for A/B runs)
  bool profiling = false;
  bool trace = false;           // MI355_TRACE=1: one stderr line per MSM / NTT call (host wall time; device transforms are synchronised for it)
  std::map<std::string, Prof> prof;
  int last_c = 0, last_w = 0; uint64_t last_entries = 0; bool last_shared = false; int last_host_slices = 1;
};

// One context per bound device; slot 0 is the primary (every single-device entry point runs there).  All entry points are serialised by
// g_mu.  `g` names the context the CURRENT THREAD works on: the API thread after need_init() (primary), or one of the per-device worker
// threads of a sharded MSM (msm_multi), which is why the pointer is thread-local.
constexpr int MAX_DEV = 16;
std::mutex g_mu;
Ctx g_ctx[MAX_DEV];
int g_ndev = 0;
bool g_dup_devices = false;     // test mode: the same physical device bound to several slots (exchange by device copies instead of RCCL)
uint32_t g_shard_min_log = 14;  // MI355_SHARD_MIN_LOG: a basis with fewer than 2^this points per device stays on the primary device (tests lower it)
bool g_force_exchange = false;  // MI355_MULTI_FORCE=1: take the sharded path (partials + exchange + fold) even with one device
thread_local Ctx *g_cur = &g_ctx[0];
#define g (*g_cur)
std::unordered_map<uint64_t, Srs> g_srs;
uint64_t g_next_handle = 1;
int g_last_devices = 1; const char *g_last_exchange = "none";

// per-THREAD MSM options (mi355_msm_set_normalise / _set_window_bits): a rayon worker that asks for un-normalised partial sums must not
// change what another worker's commit returns (SURVEY 8b "Threading")
struct MsmOpts { bool normalise = true; int force_c = 0; bool no_tables = false; };
thread_local MsmOpts t_opts;

void use_ctx(int slot) { g_cur = &g_ctx[slot]; }
void free_shards(std::vector<Shard> &sh) {
  for (auto &x : sh) {
    if (x.slot < g_ndev && g_ctx[x.slot].inited) (void)hipSetDevice(g_ctx[x.slot].device);
    if (x.owned && x.dev) (void)hipFree(x.dev);
    x.dev = nullptr;
  }
  sh.clear();
}
SrsTables::~SrsTables() {
  for (size_t i = 0; i < pre.size(); i++) if (pre[i]) {
    if (slot[i] < g_ndev && g_ctx[slot[i]].inited) { (void)hipSetDevice(g_ctx[slot[i]].device); (void)hipStreamSynchronize(g_ctx[slot[i]].stream); }
    (void)hipFree(pre[i]);
  }
}
// signed-digit recoding of min(k, r - k) < 2^253: W = ceil(254 / c) windows always absorb the carry of the top digit (253 bits of
// magnitude + 1); c <= 24 is the sorter's key range (23 bucket bits = 12 fine + 11 coarse)
constexpr int MSM_SCALAR_BITS = 255, MSM_MAX_C = 24;   // hard limit of the sorter (23 key bits); the automatic choices stop at g_auto_max_c
int g_auto_max_c = 22;          // MI355_MSM_AUTO_MAX_C (22..24): widest window the automatic choices may take.  c = 24 (W = 11) trades 8 % fewer
                                // bucket additions for a 4x larger bucket reduction; explicit requests (mi355_srs_precompute(c), _set_window_bits) may always use 23 / 24

// Every compute entry point starts here.  The HIP current device is per host thread and calls arrive from whichever thread runs
// create_proof (rayon workers included, SURVEY 8b "Threading"), so the bound device is re-selected on the calling thread each time.
int bind_ctx(int slot) {
  use_ctx(slot);
  if (hipSetDevice(g.device) != hipSuccess) { (void)hipGetLastError(); return fail(MI355_EHIP, "hipSetDevice failed on the calling thread"); }
  return MI355_OK;
}
int need_init() {
  use_ctx(0);
  if (g_ndev == 0 || !g.inited) return fail(MI355_ENODEVICE, "mi355_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)");
  return bind_ctx(0);
}

// MI355_TRACE: per-call counters for the integrator (metrics / logging)
struct CallTrace {
  const char *what; uint64_t n; double bytes_per_unit; std::chrono::steady_clock::time_point t0;
  CallTrace(const char *w, uint64_t n_, double bpu) : what(w), n(n_), bytes_per_unit(bpu), t0(std::chrono::steady_clock::now()) {}
  void done(const char *extra = "") {
    if (!g.trace) return;
    (void)hipStreamSynchronize(g.stream);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[mi355zk] %s n=%llu %.3f ms %.2f M units/s %.1f GB/s algorithmic%s\n", what, (unsigned long long)n, ms, n / ms / 1e3, n * bytes_per_unit / ms / 1e6, extra);
  }
};

int ws_get(const char *role, size_t bytes, void **out) {
  Buf &b = g.ws[role];
  if (b.cap < bytes) {
    if (b.p) { HIPCHK(hipStreamSynchronize(g.stream)); for (int i = 0; i < 2; i++) if (g.aux_stream[i]) HIPCHK(hipStreamSynchronize(g.aux_stream[i])); HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t cap = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&b.p, cap)); b.cap = cap;
  }
  *out = b.p; return MI355_OK;
}

// ---- profiling: a list of (name, start, stop) event pairs per context, resolved after the stream is idle
struct Scope {
  bool on; hipEvent_t a = nullptr, b = nullptr; std::string name; hipStream_t st;
  Scope(const char *n, hipStream_t s = nullptr) : on(g.profiling), name(n), st(s ? s : g.stream) { if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); } }
  void close() { if (on) { (void)hipEventRecord(b, st); g.spans.push_back({name, a, b}); on = false; } }
  ~Scope() { close(); }
};
void resolve_spans() {
  if (g.spans.empty()) return;
  (void)hipStreamSynchronize(g.stream);
  for (int i = 0; i < 2; i++) if (g.aux_stream[i]) (void)hipStreamSynchronize(g.aux_stream[i]);
  for (auto &s : g.spans) { float ms = 0; (void)hipEventElapsedTime(&ms, s.a, s.b); Prof &p = g.prof[s.name]; p.ms += ms; p.launches++; (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  g.spans.clear();
}

// ------------------------------------------------------------------------------------------------ MSM
uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
uint32_t log2_ceil(uint64_t n) { uint32_t l = 0; while ((1ull << l) < n) l++; return l; }

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
  { const size_t lvl = (size_t)ceil_div(chunks_per_window, 256 * TREE_PER_THREAD) * red_windows + 1;
    CHK(ws_get(role("msm.tree_a").c_str(), lvl * sizeof(g1_xyzz29_t), (void **)&tree_a)); CHK(ws_get(role("msm.tree_b").c_str(), lvl * sizeof(g1_xyzz29_t), (void **)&tree_b)); }
  CHK(ws_get(role("msm.window_sums").c_str(), (size_t)red_windows * sizeof(g1_xyzz29_t), (void **)&window_sums));
  MsmPlan PR; PR.n = 0; PR.c = sh.c; PR.windows = red_windows; PR.nb = nb; PR.seg = 0; PR.batch = M;
  hipLaunchKernelGGL(k_msm_bucket_reduce, dim3(ceil_div(nchunks, 128)), dim3(128), 0, s, buckets, chunk_out, PR, chunk);
  {
    // tree-sum the chunk results per window, ping-ponging between two small buffers
    const g1_xyzz29_t *cur = chunk_out; uint32_t cnt = chunks_per_window; g1_xyzz29_t *bufs[2] = {tree_a, tree_b}; int which = 0;
    while (true) {
      const uint32_t outn = ceil_div(cnt, 256 * TREE_PER_THREAD);
      g1_xyzz29_t *dst = outn == 1 ? window_sums : bufs[which];
      hipLaunchKernelGGL(k_msm_tree_sum29, dim3(outn, red_windows), dim3(256), 0, s, cur, cnt, dst, outn);
      if (outn == 1) break;
      cur = dst; cnt = outn; which ^= 1;
    }
  }
  hipLaunchKernelGGL(k_msm_final29, dim3(M), dim3(64), 0, s, (const g1_xyzz29_t *)window_sums, red_wpp, sh.shared ? 0u : sh.c, out_dev, normalise ? 1 : 0);
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
  // level-2 tile (6 B of LDS per entry next to the 3 x 2^fb words of bin bookkeeping): 16384 entries give twice the run length in
  // `sorted` (fewer partial-line store transactions, the limiter of this kernel) at one workgroup per CU; worth it for big sorts
  S.t2 = g.sort_t2 ? g.sort_t2 : (emax >= (1ull << 27) && S.fb <= 11 ? 16384 : 8192);
  const uint32_t tiles1 = ceil_div(n, S.t1), l2_tiles_max = ceil_div(emax, S.t2) + S.regions + 8;   // + 8: the XCD-aware tile order rounds the tile count up to a multiple of 8
  const uint32_t vwindows = M * P.windows;   // (polynomial, window) pairs

  uint32_t *enc, *hist, *offsets, *cursor, *sorted, *scan_sums, *coarse_hist, *coarse_off, *coarse_cursor, *tile_start; uint64_t *pairs;
  g1_xyzz29_t *buckets, *part; int32_t *part_id;
  // stage-A-only buffers are shared by all slots (the sort stages of successive chunks run one after the other on st.a)
  CHK(ws_get("msm.digits", emax * 4, (void **)&enc));
  CHK(ws_get("msm.pairs", emax * 8, (void **)&pairs));
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
        const uint32_t CBp = ((1u << S.cb_bits) + 1) & ~1u;
        if (S.t1 == 16384) hipLaunchKernelGGL(k_sort_l1_scatter<16>, dim3(tiles1 * vwindows), dim3(1024), (size_t)(3 * CBp + 32) * 4 + (size_t)S.t1 * 8, s, enc, coarse_cursor, pairs, S);
        else hipLaunchKernelGGL(k_sort_l1_scatter<8>, dim3(tiles1 * vwindows), dim3(1024), (size_t)(3 * CBp + 32) * 4 + (size_t)S.t1 * 8, s, enc, coarse_cursor, pairs, S);
      }
      hipLaunchKernelGGL(k_sort_tile_prefix, dim3(1), dim3(SCAN_BLOCK), 0, s, coarse_off, tile_start, S);
      hipLaunchKernelGGL(k_sort_l2_hist, dim3(l2_tiles_max), dim3(256), 0, s, pairs, coarse_off, tile_start, hist, S);
      hipLaunchKernelGGL(k_scan_partial, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, s, hist, scan_sums, scan_n);
      hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, scan_sums, scan_blocks);
      hipLaunchKernelGGL(k_scan_final, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, s, hist, scan_sums, offsets, cursor, scan_n);
      {
        const size_t lds2 = (size_t)(3 * (1u << S.fb) + 32) * 4 + (size_t)S.t2 * 6;   // histogram / offsets / bases + staged indices (4 B) and their bins (2 B)
        if (S.t2 == 8192) hipLaunchKernelGGL(k_sort_l2_scatter<8>, dim3(l2_tiles_max), dim3(1024), lds2, s, pairs, coarse_off, tile_start, cursor, sorted, S);
        else hipLaunchKernelGGL(k_sort_l2_scatter<16>, dim3(l2_tiles_max), dim3(1024), lds2, s, pairs, coarse_off, tile_start, cursor, sorted, S);
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
      if (g.acc_variant & 4) ACC_LAUNCH(4); else ACC_LAUNCH(0);   // 4: limb products as explicitly chained v_mad (fp29.cuh mac_*); the older A/B variants (index stream further ahead, prefetched bucket ends) did not help and are no longer instantiated
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
    if (nbuckets <= (1u << g.fixup_lanes_max_log)) hipLaunchKernelGGL(k_msm_fixup<4>, dim3(ceil_div((uint64_t)nbuckets * 4, 256)), dim3(256), 0, s, offsets, nbuckets, buckets, part, part_id, seg_arg, tn, big_list, big_count, big_cap, huge_list, huge_count, huge_cap, g.fixup_serial_max, g.fixup_huge_min);
    else hipLaunchKernelGGL(k_msm_fixup<1>, dim3(ceil_div(nbuckets, 256)), dim3(256), 0, s, offsets, nbuckets, buckets, part, part_id, seg_arg, tn, big_list, big_count, big_cap, huge_list, huge_count, huge_cap, g.fixup_serial_max, g.fixup_huge_min);
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
  if (!g.copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventCreateWithFlags(&g.ev_copy[i], hipEventDisableTiming));
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

// ------------------------------------------------------------------------------------------------ NTT
constexpr uint32_t NTT_DIRECT_TW_MAX_LOG = 20;   // 2^20 x 36 B = 38 MB per table at most
std::string plan_key(uint32_t log_n, const void *omega) { std::string k((const char *)omega, 32); k.push_back((char)log_n); return k; }

int pow_table(fe_t **out, const fe_t &base, uint64_t step, uint32_t count) {
  HIPCHK(hipMalloc((void **)out, (size_t)count * sizeof(fe_t)));
  hipLaunchKernelGGL(k_pow_table, dim3(ceil_div(count, 256)), dim3(256), 0, g.stream, *out, base, step, count);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}

int pow_table29(NttPlan &p, Tw29 *out, const fe_t &base, uint64_t step, uint32_t count) {
  uint4 *lo, *hi; uint32_t *top;
  HIPCHK(hipMalloc((void **)&lo, (size_t)count * 16)); p.owned.push_back(lo);
  HIPCHK(hipMalloc((void **)&hi, (size_t)count * 16)); p.owned.push_back(hi);
  HIPCHK(hipMalloc((void **)&top, (size_t)count * 4)); p.owned.push_back(top);
  hipLaunchKernelGGL(k_pow_table29, dim3(ceil_div(count, 256)), dim3(256), 0, g.stream, lo, hi, top, base, step, count);
  HIPCHK(hipGetLastError());
  out->lo = lo; out->hi = hi; out->top = top;
  return MI355_OK;
}

int get_plan(uint32_t log_n, const void *omega, NttPlan **out) {
  const std::string key = plan_key(log_n, omega);
  auto it = g.ntt_plans.find(key);
  if (it != g.ntt_plans.end()) { *out = &it->second; return MI355_OK; }
  NttPlan p; p.log_n = log_n;
  if (log_n <= 8) { p.levels = 1; p.log_m[0] = log_n; }
  else if (log_n <= 18) { p.levels = 2; p.log_m[0] = (log_n + 1) / 2; p.log_m[1] = log_n / 2; }
  else { p.levels = 3; p.log_m[0] = (log_n + 2) / 3; p.log_m[1] = (log_n + 1) / 3; p.log_m[2] = log_n / 3; }
  fe_t w; memcpy(&w, omega, 32);
  const uint64_t N = 1ull << log_n;
  uint32_t log_s = log_n;
  for (uint32_t l = 0; l < p.levels; l++) {
    const uint32_t lm = p.log_m[l];
    if (lm >= 1) CHK(pow_table(&p.tw_m[l], w, N >> lm, std::max(1u, 1u << (lm - 1))));
    CHK(pow_table29(p, &p.tw29_m[l], w, N >> lm, std::max(1u, (1u << lm) >> 1)));
    if (l + 1 < p.levels) {
      // inter-level twiddles w_S^e, e < 2^log_s: ONE table when it is small enough to live in L2 (no lo x hi product per element),
      // otherwise the usual two half-size tables
      p.split[l] = log_s <= NTT_DIRECT_TW_MAX_LOG ? log_s : (log_s + 1) / 2;
      CHK(pow_table(&p.tw_s_lo[l], w, N >> log_s, 1u << p.split[l]));
      CHK(pow_table(&p.tw_s_hi[l], w, (N >> log_s) << p.split[l], 1u << (log_s - p.split[l])));
      if (g.ntt29 && log_s > NTT_DIRECT_TW_MAX_LOG && log_s <= g.ntt_direct2_max_log) {
        // big level: every twiddle w_S^(column k) once, in the order the pass reads them (36 B x 2^log_s: 2.4 GB at 2^26, read coalesced
        // next to the data by a pass that is ALU-bound); saves the lo x hi product per element
        const uint64_t cnt = 1ull << log_s; uint4 *lo, *hi; uint32_t *top;
        HIPCHK(hipMalloc((void **)&lo, cnt * 16)); p.owned.push_back(lo);
        HIPCHK(hipMalloc((void **)&hi, cnt * 16)); p.owned.push_back(hi);
        HIPCHK(hipMalloc((void **)&top, cnt * 4)); p.owned.push_back(top);
        hipLaunchKernelGGL(k_pow_table29_2d, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, g.stream, lo, hi, top, Fr::pow_u64(w, N >> log_s), log_s - lm, cnt);
        HIPCHK(hipGetLastError());
        p.tw29_s_lo[l].lo = lo; p.tw29_s_lo[l].hi = hi; p.tw29_s_lo[l].top = top; p.tw29_s_hi[l] = p.tw29_s_lo[l]; p.direct2[l] = 1;
      } else {
      CHK(pow_table29(p, &p.tw29_s_lo[l], w, N >> log_s, 1u << p.split[l]));
      CHK(pow_table29(p, &p.tw29_s_hi[l], w, (N >> log_s) << p.split[l], 1u << (log_s - p.split[l])));
      }
    }
    log_s -= lm;
  }
  g.ntt_plans[key] = p; *out = &g.ntt_plans[key];
  return MI355_OK;
}

// radix-2^R register rounds: R = g.ntt_radix_log (1..3); one work item per 2^R elements
#define NTT29_LAUNCH(KERN, BLOCKS, TILE, LDS, ...)                                                                                          \
  do {                                                                                                                                     \
    if (g.ntt_radix_log == 3) hipLaunchKernelGGL(KERN<3>, dim3(BLOCKS), dim3(std::max(64u, std::min(512u, (TILE) / 8))), LDS, s, __VA_ARGS__);       \
    else if (g.ntt_radix_log == 2) hipLaunchKernelGGL(KERN<2>, dim3(BLOCKS), dim3(std::max(64u, std::min(512u, (TILE) / 4))), LDS, s, __VA_ARGS__);  \
    else hipLaunchKernelGGL(KERN<1>, dim3(BLOCKS), dim3(std::max(64u, std::min(1024u, (TILE) / 2))), LDS, s, __VA_ARGS__);                            \
  } while (0)

uint32_t cols_for(uint32_t log_m) { uint32_t lc = 3; while (lc > 0 && log_m + lc > g.ntt_tile_log) lc--; return lc; }

// dst[2^log_n] = NTT_omega( pre3-scaled, zero-padded src[src_len] ), then optional post3 scaling.  src may equal dst.
int ntt_dev_impl(const fe_t *src, uint64_t src_len, fe_t *dst, uint32_t log_n, const void *omega, const fe_t *pre3_host, const fe_t *post3_host) {
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
  // an inverse transform's divisor (the three post-scaling constants equal) is folded into the inter-level twiddles of the last strided
  // pass: one table of 2^log_s entries per (plan, divisor), and the closing pass ends with reduce_small instead of a multiplication
  const Tw29 *fold_tw = nullptr;
  if (post3_host && !pre3_host && g.ntt29 && g.ntt_fold_scale && p->levels >= 2 && memcmp(&post3_host[0], &post3_host[1], 32) == 0 && memcmp(&post3_host[0], &post3_host[2], 32) == 0) {
    const uint32_t l = p->levels - 2;
    uint32_t log_sl = log_n; for (uint32_t q = 0; q < l; q++) log_sl -= p->log_m[q];
    if (p->split[l] == log_sl && !p->direct2[l]) {   // that level reads ONE direct table
      const std::string key((const char *)&post3_host[0], 32);
      auto it = p->scaled.find(key);
      if (it == p->scaled.end()) {
        const uint32_t cnt = 1u << log_sl; uint4 *lo, *hi; uint32_t *top;
        HIPCHK(hipMalloc((void **)&lo, (size_t)cnt * 16)); p->owned.push_back(lo);
        HIPCHK(hipMalloc((void **)&hi, (size_t)cnt * 16)); p->owned.push_back(hi);
        HIPCHK(hipMalloc((void **)&top, (size_t)cnt * 4)); p->owned.push_back(top);
        hipLaunchKernelGGL(k_scale_table29, dim3(ceil_div(cnt, 256)), dim3(256), 0, s, p->tw29_s_lo[l].lo, p->tw29_s_lo[l].hi, p->tw29_s_lo[l].top, lo, hi, top, post3_host[0], cnt);
        HIPCHK(hipGetLastError());
        Tw29 t; t.lo = lo; t.hi = hi; t.top = top;
        it = p->scaled.emplace(key, t).first;
      }
      fold_tw = &it->second; post3 = nullptr;
    }
  }
  CallTrace tr("ntt_fr", N, 64.0);
  Scope total("ntt_total");
  if (p->levels == 1) {
    const uint32_t lm = p->log_m[0], tile = 1u << lm;
    const uint32_t threads = std::max(64u, std::min(1024u, tile / 2));
    const size_t lds = (size_t)2 * 16 * (tile + 1);
    Scope sc("ntt_pass");
    if (g.ntt29) NTT29_LAUNCH(k_ntt29_final, 1u, tile, (size_t)36 * (tile + 1), src, dst, lm, 0u, 0u, 0u, p->tw29_m[0], src_len, pre3, post3);
    else hipLaunchKernelGGL(k_ntt_final, dim3(1), dim3(threads), lds, s, src, dst, lm, 0u, 0u, 0u, p->tw_m[0], src_len, pre3, post3);
  } else {
    fe_t *scratch; CHK(ws_get("ntt.scratch", N * sizeof(fe_t), (void **)&scratch));
    uint32_t log_s = log_n;
    const fe_t *cur = src; uint64_t cur_len = src_len; const fe_t *cur_pre = pre3;
    for (uint32_t l = 0; l + 1 < p->levels; l++) {
      NttLevel L; L.log_m = p->log_m[l]; L.log_t = log_s - L.log_m; L.tw_m = p->tw_m[l]; L.tw_s_lo = p->tw_s_lo[l]; L.tw_s_hi = p->tw_s_hi[l]; L.split = p->split[l];
      const uint32_t lc = std::min(cols_for(L.log_m), L.log_t), tile = 1u << (L.log_m + lc);
      const uint32_t threads = std::max(64u, std::min(1024u, tile / 2));
      const size_t lds = (size_t)2 * 16 * tile;
      const uint64_t blocks = (N >> log_s) << (L.log_t - lc);
      Scope sc("ntt_pass");
      if (g.ntt29) {
        Ntt29Level L9; L9.log_m = L.log_m; L9.log_t = L.log_t; L9.split = L.split; L9.tw_m = p->tw29_m[l]; L9.tw_s_lo = p->tw29_s_lo[l]; L9.tw_s_hi = p->tw29_s_hi[l]; L9.direct = p->direct2[l] ? 2u : (p->split[l] == log_s) ? 1u : 0u;
        if (fold_tw && l + 2 == p->levels) L9.tw_s_lo = *fold_tw;
        NTT29_LAUNCH(k_ntt29_strided, (uint32_t)blocks, tile, (size_t)36 * tile, cur, scratch, L9, lc, cur_len, cur_pre);
      } else
      hipLaunchKernelGGL(k_ntt_strided, dim3((uint32_t)blocks), dim3(threads), lds, s, cur, scratch, L, lc, cur_len, cur_pre);
      cur = scratch; cur_len = N; cur_pre = nullptr; log_s -= L.log_m;
    }
    const uint32_t lm = p->log_m[p->levels - 1], log_a = p->log_m[0], log_b = p->levels == 3 ? p->log_m[1] : 0;
    const uint32_t lc = std::min(cols_for(lm), log_a), tile = 1u << (lm + lc);
    const uint32_t threads = std::max(64u, std::min(1024u, tile / 2));
    const size_t lds = (size_t)2 * 16 * (((size_t)1 << lm) + 1) * ((size_t)1 << lc);
    const uint64_t blocks = ((uint64_t)1 << log_b) << (log_a - lc);
    Scope sc("ntt_pass");
    if (g.ntt29) NTT29_LAUNCH(k_ntt29_final, (uint32_t)blocks, tile, (size_t)36 * (((size_t)1 << lm) + 1) * ((size_t)1 << lc), cur, dst, lm, log_a, log_b, lc, p->tw29_m[p->levels - 1], N, (const fe_t *)nullptr, post3);
    else hipLaunchKernelGGL(k_ntt_final, dim3((uint32_t)blocks), dim3(threads), lds, s, cur, dst, lm, log_a, log_b, lc, p->tw_m[p->levels - 1], N, (const fe_t *)nullptr, post3);
  }
  HIPCHK(hipGetLastError());
  total.close();
  tr.done();
  return MI355_OK;
}

int finish_async() { if (g.profiling) resolve_spans(); return MI355_OK; }

// DFT over G1 points (g1fft.cuh).  in: n x (96 B Jacobian | 64 B affine), out likewise (may alias in); scale: optional Fr (Montgomery).
int g1fft_impl(const void *in, int in_jac, void *out, int out_jac, uint32_t log_n, const void *omega, const void *scale_host) {
  const uint32_t n = 1u << log_n, half = std::max(1u, n / 2);
  g1_xyzz_t *work; fe_t *tw; fe_t scale = Fr::zero(); if (scale_host) memcpy(&scale, scale_host, 32);
  CHK(ws_get("g1fft.work", (size_t)n * sizeof(g1_xyzz_t), (void **)&work));
  CHK(ws_get("g1fft.tw", (size_t)half * sizeof(fe_t), (void **)&tw));
  hipStream_t s = g.stream;
  fe_t w; memcpy(&w, omega, 32);
  Scope total("g1_fft");
  hipLaunchKernelGGL(k_pow_table, dim3(ceil_div(half, 256)), dim3(256), 0, s, tw, w, (uint64_t)1, half);
  if (in_jac) hipLaunchKernelGGL(k_g1fft_load<1>, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, work, log_n);
  else hipLaunchKernelGGL(k_g1fft_load<0>, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, work, log_n);
  for (uint32_t st = 0; st < log_n; st++) hipLaunchKernelGGL(k_g1fft_stage, dim3(ceil_div(n / 2, 256)), dim3(256), 0, s, work, tw, log_n, st);
  if (out_jac) hipLaunchKernelGGL(k_g1fft_store<1>, dim3(ceil_div(n, 256)), dim3(256), 0, s, work, out, log_n, scale, scale_host ? 1 : 0);
  else hipLaunchKernelGGL(k_g1fft_store<0>, dim3(ceil_div(n, 256)), dim3(256), 0, s, work, out, log_n, scale, scale_host ? 1 : 0);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}

// data[i] <- data[i]^-1 (zeros kept): tile products -> (recursively) their inverses -> per-tile completion
int batch_invert_impl(fe_t *data, uint64_t n, int level) {
  const uint32_t tiles = (uint32_t)ceil_div(n, FRSCAN_TILE);
  hipStream_t s = g.stream;
  if (tiles <= 32) { hipLaunchKernelGGL(k_fr_batch_invert<0>, dim3(tiles), dim3(FRSCAN_THREADS), 0, s, data, n, (fe_t *)nullptr); return MI355_OK; }
  fe_t *tile_prod; const std::string role = "frscan.inv_tiles" + std::to_string(level);
  CHK(ws_get(role.c_str(), (size_t)tiles * sizeof(fe_t), (void **)&tile_prod));
  hipLaunchKernelGGL(k_fr_batch_invert<1>, dim3(tiles), dim3(FRSCAN_THREADS), 0, s, data, n, tile_prod);
  CHK(batch_invert_impl(tile_prod, tiles, level + 1));
  hipLaunchKernelGGL(k_fr_batch_invert<2>, dim3(tiles), dim3(FRSCAN_THREADS), 0, s, data, n, tile_prod);
  return MI355_OK;
}

// P_j = src_j + m * P_(j-1) over n elements (dst may alias src); reverse: index j lives at memory position n - 1 - j
int linrec_impl(const fe_t *src, fe_t *dst, uint64_t n, const fe_t &m, bool reverse, int level) {
  const uint32_t tiles = (uint32_t)ceil_div(n, FRSCAN_TILE);
  const size_t lds = (size_t)(FRSCAN_THREADS * 65 + 2 * FRSCAN_THREADS * 9) * 4;
  hipStream_t s = g.stream;
  if (tiles <= 1) { hipLaunchKernelGGL(k_fr_linrec<1>, dim3(1), dim3(FRSCAN_THREADS), lds, s, src, dst, n, m, reverse ? 1 : 0, (fe_t *)nullptr); return MI355_OK; }
  fe_t *tile_tot; const std::string role = "frscan.linrec" + std::to_string(level);
  CHK(ws_get(role.c_str(), (size_t)tiles * sizeof(fe_t), (void **)&tile_tot));
  hipLaunchKernelGGL(k_fr_linrec<0>, dim3(tiles), dim3(FRSCAN_THREADS), lds, s, src, dst, n, m, reverse ? 1 : 0, tile_tot);
  CHK(linrec_impl(tile_tot, tile_tot, tiles, Fr::pow_u64(m, FRSCAN_TILE), false, level + 1));   // the value of the recurrence at every tile end
  hipLaunchKernelGGL(k_fr_linrec<1>, dim3(tiles), dim3(FRSCAN_THREADS), lds, s, src, dst, n, m, reverse ? 1 : 0, tile_tot);
  return MI355_OK;
}

int with_host_io(void *data_host, size_t in_bytes, size_t out_bytes, size_t dev_bytes, const char *role, int (*body)(void *dev, void *ud), void *ud) {
  void *dev; CHK(ws_get(role, dev_bytes, &dev));
  HIPCHK(hipMemcpyAsync(dev, data_host, in_bytes, hipMemcpyHostToDevice, g.stream));
  CHK(body(dev, ud));
  HIPCHK(hipMemcpyAsync(data_host, dev, out_bytes, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  resolve_spans();
  return MI355_OK;
}

}  // namespace

// No exception may cross the C boundary (the Rust callers are `extern "C"` and would abort): every multi-line entry point runs inside
// this guard, which turns allocation failures and anything else the host-side C++ might throw into error codes.
template <class F> int guarded(F body) {
  try { return body(); }
  catch (const std::bad_alloc &) { return fail(MI355_EOOM, "host allocation failed"); }
  catch (const std::exception &e) { return fail(MI355_EHIP, std::string("unexpected host exception: ") + e.what()); }
  catch (...) { return fail(MI355_EHIP, "unexpected host exception"); }
}

// ================================================================================================ C ABI
extern "C" {

const char *mi355_last_error(void) { return g_err.c_str(); }
const char *mi355_version(void) { return "mi355zk 0.1.0 (gfx950; BN254 G1 MSM + Fr NTT)"; }

// ---- RCCL, bound lazily: librccl.so.1 is dlopen()ed by mi355_init_multi only when it needs a communicator, so single-device users (and the
// CPU-only symbol checks) carry no RCCL dependency; a process that imported torch first gets torch's copy (same SONAME), as with libamdhip64.
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
} g_rccl;
static int rccl_fail(const char *what, int rc) { return fail(MI355_ERCCL, std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?")); }
static int rccl_load() {
  if (g_rccl.lib) return MI355_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(MI355_ERCCL, std::string("cannot load librccl.so.1: ") + dlerror());
  g_rccl.CommInitAll = (int (*)(void **, int, const int *))dlsym(h, "ncclCommInitAll");
  g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
  g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
  g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
  g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GroupStart || !g_rccl.GroupEnd) { dlclose(h); return fail(MI355_ERCCL, "librccl.so.1 lacks the expected nccl* symbols"); }
  g_rccl.lib = h;
  return MI355_OK;
}

// everything mi355_init does for ONE device slot (the calling thread ends up bound to that device)
static int init_ctx(int slot, int device_id) {
  use_ctx(slot);
  g = Ctx();
  g.slot = slot;
  HIPCHK(hipSetDevice(device_id));
  g.device = device_id;                     // from here on destroy_ctx() releases whatever the steps below created
  HIPCHK(hipGetDeviceProperties(&g.prop, device_id));
  if (strncmp(g.prop.gcnArchName, "gfx950", 6) != 0) return fail(MI355_ENODEVICE, std::string("device is ") + g.prop.gcnArchName + ", this library is built for gfx950 only");
  HIPCHK(hipStreamCreateWithFlags(&g.own_stream, hipStreamNonBlocking));
  g.stream = g.own_stream;
  for (int i = 0; i < 2; i++) {
    { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); const char *e = getenv("MI355_AUX_PRIO"); const bool high = !(e && e[0] == '0');
      HIPCHK(hipStreamCreateWithPriority(&g.aux_stream[i], hipStreamNonBlocking, high ? hi : lo)); }   // the side streams outrank the accumulation
    g.msm_slot[i].id = i; g.msm_slot[i].used = false;
    HIPCHK(hipEventCreateWithFlags(&g.msm_slot[i].sorted, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&g.msm_slot[i].acc_done, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&g.msm_slot[i].red_done, hipEventDisableTiming));
  }
  HIPCHK(hipEventCreateWithFlags(&g.ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g.ev_xchg, hipEventDisableTiming));
  { const char *e = getenv("MI355_MSM_CHUNKS"); if (e) { int v = atoi(e); if (v >= 1 && v <= 16) g.msm_chunks = (uint32_t)v; } }
  // dynamic-LDS limits are per device
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt_strided, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt_final, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l1_scatter<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l1_scatter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l2_scatter<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_sort_l2_scatter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_strided<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_final<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_strided<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_final<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_strided<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_ntt29_final<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_prefix_product<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_prefix_product<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_linrec<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_linrec<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  { const char *e = getenv("MI355_TRACE"); g.trace = e && e[0] == '1'; }
  { const char *e = getenv("MI355_NTT_SAT"); g.ntt29 = !(e && e[0] == '1'); }
#ifdef MI355_DEBUG_KNOBS
  { const char *e = getenv("MI355_DEBUG_GATHER_MASK"); if (e) g.debug_gather_mask = (uint32_t)strtoul(e, nullptr, 0); }
#endif
  { const char *e = getenv("MI355_ACC_VARIANT"); if (e) g.acc_variant = (uint32_t)atoi(e) & 7; }
  { const char *e = getenv("MI355_REDUCE_CHAINS"); if (e) { int v = atoi(e); if (v >= 1024) g.reduce_chains = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_FILL"); if (e) { int v = atoi(e); if (v >= 2 && v <= 100) g.seg_fill = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_FILL_SEGFIX"); if (e) { int v = atoi(e); if (v >= 2 && v <= 100) g.seg_fill_segfix = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_MIN"); if (e) { int v = atoi(e); if (v >= 1 && v <= 4096) g.seg_min = (uint32_t)v; } }
  { const char *e = getenv("MI355_FIXUP_MODE"); if (e && e[0] >= '0' && e[0] <= '2') g.fixup_mode = (uint32_t)(e[0] - '0'); }
  { const char *e = getenv("MI355_FIXUP_HUGE_MIN"); if (e) { long v = atol(e); if (v >= 2048 && v <= 0x7fffffffL) g.fixup_huge_min = (uint32_t)v; } }
  { const char *e = getenv("MI355_FIXUP_SERIAL_MAX"); if (e) { int v = atoi(e); if (v >= 1 && v <= 1024) g.fixup_serial_max = (uint32_t)v; } }
  { const char *e = getenv("MI355_FIXUP_LANES_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 31) g.fixup_lanes_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_REDUCE_MIN_CHUNK"); if (e) { int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) g.reduce_min_chunk = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_FB"); if (e) { int v = atoi(e); if (v >= 9 && v <= 12) g.sort_fb = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_FACTOR"); if (e) { int v = atoi(e); if (v >= 1 && v <= 256) g.seg_factor = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_T2"); if (e) { int v = atoi(e); if (v == 8192 || v == 16384) g.sort_t2 = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_T1"); if (e && atoi(e) == 8192) g.sort_t1 = 8192; }
  { const char *e = getenv("MI355_NTT_DIRECT2_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 28) g.ntt_direct2_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_FOLD_SCALE"); if (e) g.ntt_fold_scale = e[0] == '0' ? 0u : 1u; }
  { const char *e = getenv("MI355_NTT_RADIX_LOG"); if (e) { int v = atoi(e); if (v >= 1 && v <= 3) g.ntt_radix_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_TILE_LOG"); if (e) { int v = atoi(e); if (v >= 8 && v <= 12) g.ntt_tile_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_HOST_CHUNKS"); if (e) { int v = atoi(e); if (v >= 1 && v <= 64) g.host_chunks = (uint32_t)v; } }
  { const char *e = getenv("MI355_HOST_SLICE_MIN_LOG"); if (e) { int v = atoi(e); if (v >= 4 && v <= 31) g.host_slice_min_log = (uint32_t)v; } }
  g.inited = true;
  return MI355_OK;
}
static void destroy_ctx(int slot) {
  use_ctx(slot);
  if (g.device < 0) return;                 // never reached hipSetDevice: nothing was created
  (void)hipSetDevice(g.device);
  if (g.stream) (void)hipStreamSynchronize(g.stream);
  for (auto &kv : g.ws) if (kv.second.p) (void)hipFree(kv.second.p);
  g.ws.clear();
  for (auto &kv : g.ntt_plans) { for (void *q : kv.second.owned) (void)hipFree(q); for (int i = 0; i < 3; i++) if (kv.second.tw_m[i]) (void)hipFree(kv.second.tw_m[i]); for (int i = 0; i < 2; i++) { if (kv.second.tw_s_lo[i]) (void)hipFree(kv.second.tw_s_lo[i]); if (kv.second.tw_s_hi[i]) (void)hipFree(kv.second.tw_s_hi[i]); } }
  g.ntt_plans.clear();
  if (g.fixed_base_table) { (void)hipFree(g.fixed_base_table); g.fixed_base_table = nullptr; }
  if (g.pin_ring) { (void)hipHostFree(g.pin_ring); g.pin_ring = nullptr; g.pin_ring_bytes = 0; }
  if (g.comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(g.comm); g.comm = nullptr; }
  if (g.own_stream) (void)hipStreamDestroy(g.own_stream);
  for (int i = 0; i < 2; i++) {
    if (g.aux_stream[i]) { (void)hipStreamDestroy(g.aux_stream[i]); g.aux_stream[i] = nullptr; }
    if (g.msm_slot[i].sorted) { (void)hipEventDestroy(g.msm_slot[i].sorted); (void)hipEventDestroy(g.msm_slot[i].acc_done); (void)hipEventDestroy(g.msm_slot[i].red_done); g.msm_slot[i] = MsmSlot(); }
  }
  if (g.ev_fork) { (void)hipEventDestroy(g.ev_fork); g.ev_fork = nullptr; }
  if (g.ev_xchg) { (void)hipEventDestroy(g.ev_xchg); g.ev_xchg = nullptr; }
  for (int i = 0; i < 4; i++) if (g.ev_copy[i]) { (void)hipEventDestroy(g.ev_copy[i]); g.ev_copy[i] = nullptr; }
  if (g.copy_stream) { (void)hipStreamDestroy(g.copy_stream); g.copy_stream = nullptr; }
  g.own_stream = g.stream = nullptr; g.inited = false; g.device = -1;
}
static void shutdown_all() {
  for (auto &kv : g_srs) { kv.second.tab.reset(); kv.second.mem.reset(); }   // the destructors bind each shard's device and free
  g_srs.clear();
  for (int s = g_ndev - 1; s >= 0; s--) destroy_ctx(s);
  g_ndev = 0; g_dup_devices = false; g_force_exchange = false;
  use_ctx(0);
}

int mi355_init_multi(const int *device_ids, int n_devices) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!device_ids || n_devices < 1 || n_devices > MAX_DEV) return fail(MI355_EBADARG, "init_multi: need 1..16 device ids");
  if (g_ndev) {
    bool same = g_ndev == n_devices;
    for (int i = 0; same && i < n_devices; i++) same = g_ctx[i].device == device_ids[i];
    use_ctx(0);
    return same ? MI355_OK : fail(MI355_EBADARG, "already bound to a different device list (mi355_shutdown first)");
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); return fail(MI355_ENODEVICE, "no HIP device visible"); }
  bool dup = false;
  for (int i = 0; i < n_devices; i++) {
    if (device_ids[i] < 0 || device_ids[i] >= count) return fail(MI355_EBADARG, "device_id out of range");
    for (int j = 0; j < i; j++) if (device_ids[j] == device_ids[i]) dup = true;
  }
  if (dup) { const char *e = getenv("MI355_ALLOW_DUP_DEVICES"); if (!(e && e[0] == '1')) return fail(MI355_EBADARG, "init_multi: the same device listed twice (test mode needs MI355_ALLOW_DUP_DEVICES=1)"); }
  int rc = MI355_OK;
  for (int s = 0; s < n_devices && rc == MI355_OK; s++) { rc = init_ctx(s, device_ids[s]); if (rc == MI355_OK) g_ndev = s + 1; else { g_ndev = s + 1; } }
  if (rc != MI355_OK) { const std::string keep = g_err; shutdown_all(); g_err = keep; return rc; }
  g_dup_devices = dup;
  { const char *e = getenv("MI355_MULTI_FORCE"); g_force_exchange = e && e[0] == '1'; }
  { const char *e = getenv("MI355_MSM_AUTO_MAX_C"); g_auto_max_c = 22; if (e) { int v = atoi(e); if (v >= 16 && v <= MSM_MAX_C) g_auto_max_c = v; } }
  { const char *e = getenv("MI355_SHARD_MIN_LOG"); g_shard_min_log = 14; if (e) { int v = atoi(e); if (v >= 0 && v <= 30) g_shard_min_log = (uint32_t)v; } }
  if (n_devices > 1 || g_force_exchange) {
    // one communicator per process over the bound devices: ncclCommInitAll.  Duplicate devices (test mode) cannot form a
    // communicator; their exchange is a device-to-device copy.
    if (!dup) {
      rc = rccl_load();
      if (rc == MI355_OK) {
        void *comms[MAX_DEV] = {nullptr};
        const int r = g_rccl.CommInitAll(comms, n_devices, device_ids);
        if (r != 0) rc = rccl_fail("ncclCommInitAll", r);
        else for (int s = 0; s < n_devices; s++) g_ctx[s].comm = comms[s];
      }
    }
    for (int s = 0; s < n_devices && rc == MI355_OK; s++) {
      use_ctx(s);
      if (hipSetDevice(g.device) != hipSuccess) { rc = fail(MI355_EHIP, "hipSetDevice failed"); break; }
      if (!dup) for (int t = 0; t < n_devices; t++) if (t != s) { int can = 0; if (hipDeviceCanAccessPeer(&can, g.device, g_ctx[t].device) == hipSuccess && can) { if (hipDeviceEnablePeerAccess(g_ctx[t].device, 0) != hipSuccess) (void)hipGetLastError(); } }
    }
    if (rc != MI355_OK) { const std::string keep = g_err; shutdown_all(); g_err = keep; return rc; }
  }
  use_ctx(0);
  (void)hipSetDevice(g.device);
  return MI355_OK;
  });
}
int mi355_init(int device_id) { return mi355_init_multi(&device_id, 1); }
int mi355_device_count(int *n_out) { std::lock_guard<std::mutex> lk(g_mu); if (!n_out) return fail(MI355_EBADARG, "device_count: null pointer"); *n_out = g_ndev; return MI355_OK; }

int mi355_shutdown(void) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ndev) return MI355_OK;
  shutdown_all();
  return MI355_OK;
  });
}

int mi355_set_stream(void *hip_stream) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  HIPCHK(hipStreamSynchronize(g.stream));
  g.stream = (hipStream_t)hip_stream;   // NULL = the HIP null (legacy default) stream, which is torch's default stream
  return MI355_OK;
  });
}
int mi355_reset_stream(void) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  HIPCHK(hipStreamSynchronize(g.stream));
  g.stream = g.own_stream;
  return MI355_OK;
  });
}
int mi355_synchronize(void) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  for (int s = g_ndev - 1; s >= 0; s--) { CHK(bind_ctx(s)); HIPCHK(hipStreamSynchronize(g.stream)); resolve_spans(); }
  return MI355_OK;
  });
}

// ---- SRS
// shard plan of a basis of n points: D equal point ranges, or everything on the primary device when the basis is too small to be worth
// spreading (a 2^14-point shard is already latency-bound)
static std::vector<Shard> plan_shards(uint64_t n) {
  std::vector<Shard> v;
  int D = g_ndev;
  if (D > 1 && n / (uint64_t)D < (1ull << g_shard_min_log)) D = 1;
  for (int d = 0; d < D; d++) { Shard s; s.slot = d; s.lo = n * d / D; s.n = n * (d + 1) / D - s.lo; if (s.n) v.push_back(s); }
  return v;
}
static int srs_find(uint64_t handle, Srs **out, const char *who) {
  auto it = g_srs.find(handle);
  if (it == g_srs.end() || !it->second.mem) return fail(MI355_EBADARG, std::string(who) + ": unknown SRS handle");
  *out = &it->second; return MI355_OK;
}
static uint64_t srs_insert(const Srs &s) { const uint64_t h = g_next_handle++; g_srs[h] = s; return h; }
// allocate the shards of `mem` on their devices (leaves the primary bound)
static int srs_alloc(SrsMem &mem, uint64_t n) {
  mem.sh = plan_shards(n);
  for (auto &sh : mem.sh) {
    CHK(bind_ctx(sh.slot));
    HIPCHK(hipMalloc((void **)&sh.dev, sh.n * sizeof(g1_affine_t))); sh.owned = true;
  }
  return bind_ctx(0);
}
int mi355_srs_register_host(const void *bases, uint64_t n, uint64_t *handle_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!bases || !handle_out || n == 0) return fail(MI355_EBADARG, "srs_register: null pointer or n == 0");
  Srs s; s.n = n; s.mem = std::make_shared<SrsMem>();
  CHK(srs_alloc(*s.mem, n));
  for (auto &sh : s.mem->sh) {
    CHK(bind_ctx(sh.slot));
    HIPCHK(hipMemcpy(sh.dev, (const g1_affine_t *)bases + sh.lo, sh.n * sizeof(g1_affine_t), hipMemcpyHostToDevice));
  }
  CHK(bind_ctx(0));
  *handle_out = srs_insert(s); return MI355_OK;
  });
}
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
  std::lock_guard<std::mutex> lk(g_mu);
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
// copy a basis that sits contiguously on the primary device into freshly allocated shards
static int srs_scatter_from_primary(SrsMem &mem, const g1_affine_t *src_dev, uint64_t n, bool alias_shard0) {
  mem.sh = plan_shards(n);
  for (auto &sh : mem.sh) {
    if (sh.slot == 0 && alias_shard0) { sh.dev = const_cast<g1_affine_t *>(src_dev) + sh.lo; sh.owned = false; continue; }
    CHK(bind_ctx(sh.slot));
    HIPCHK(hipMalloc((void **)&sh.dev, sh.n * sizeof(g1_affine_t))); sh.owned = true;
    if (sh.slot == 0 || g_ctx[sh.slot].device == g_ctx[0].device) HIPCHK(hipMemcpyAsync(sh.dev, src_dev + sh.lo, sh.n * sizeof(g1_affine_t), hipMemcpyDeviceToDevice, g.stream));
    else HIPCHK(hipMemcpyPeerAsync(sh.dev, g.device, src_dev + sh.lo, g_ctx[0].device, sh.n * sizeof(g1_affine_t), g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  return bind_ctx(0);
}
int mi355_srs_register_dev(const void *bases_dev, uint64_t n, int copy, uint64_t *handle_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!bases_dev || !handle_out || n == 0) return fail(MI355_EBADARG, "srs_register_dev: null pointer or n == 0");
  HIPCHK(hipStreamSynchronize(g.stream));   // the producer of bases_dev may have run on the library stream
  Srs s; s.n = n; s.mem = std::make_shared<SrsMem>();
  CHK(srs_scatter_from_primary(*s.mem, (const g1_affine_t *)bases_dev, n, copy == 0));
  *handle_out = srs_insert(s); return MI355_OK;
  });
}
// ParamsKZG::downsize / `&params.g[..n]` as a handle of its own: the first n points of a registered basis, sharing its device memory AND
// its window tables (the clone + downsize of load_params_map [REF integration/tests/integration.rs:17-22] must not double-allocate 48 GiB
// tables).  The memory is freed when the last handle that shares it is released, in any order.
int mi355_srs_register_prefix(uint64_t parent_handle, uint64_t n, uint64_t *handle_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  Srs *p; CHK(srs_find(parent_handle, &p, "srs_register_prefix"));
  if (!handle_out || n == 0 || n > p->n) return fail(MI355_EBADARG, "srs_register_prefix: n must be in [1, len(parent)]");
  Srs s; s.n = n; s.mem = p->mem; s.tab = p->tab;
  *handle_out = srs_insert(s); return MI355_OK;
  });
}
int mi355_srs_release(uint64_t handle) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_srs.find(handle);
  if (it == g_srs.end()) return fail(MI355_EBADARG, "srs_release: unknown handle");
  for (int s = 0; s < g_ndev; s++) if (g_ctx[s].inited) { (void)hipSetDevice(g_ctx[s].device); (void)hipStreamSynchronize(g_ctx[s].stream); }
  g_srs.erase(it);   // the last handle sharing the memory frees it (SrsMem::~SrsMem)
  if (g_ndev) (void)hipSetDevice(g_ctx[0].device);
  return MI355_OK;
  });
}
int mi355_srs_precompute(uint64_t handle, uint64_t n_hint, int c) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
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
    HIPCHK(hipMalloc((void **)&tab->pre[i], (size_t)W * cnt * sizeof(g1_affine_t)));
    tab->stride[i] = cnt;
    hipLaunchKernelGGL(k_srs_precompute, dim3(ceil_div(cnt, 256)), dim3(256), 0, g.stream, sh.dev, tab->pre[i], cnt, (uint32_t)W, (uint32_t)c);
    HIPCHK(hipGetLastError());
  }
  for (const auto &sh : mem.sh) { CHK(bind_ctx(sh.slot)); HIPCHK(hipStreamSynchronize(g.stream)); }
  sp->tab = tab;   // the previous tables (if any) are freed here unless another handle still shares them
  return bind_ctx(0);
  });
}
int mi355_srs_pre_dev_ptr(uint64_t handle, void **dev_ptr_out, int *c_out, int *windows_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  Srs *sp; CHK(srs_find(handle, &sp, "srs_pre_dev_ptr"));
  if (!dev_ptr_out) return fail(MI355_EBADARG, "srs_pre_dev_ptr: null pointer");
  *dev_ptr_out = sp->tab && !sp->tab->pre.empty() ? sp->tab->pre[0] : nullptr; if (c_out) *c_out = sp->tab ? sp->tab->c : 0; if (windows_out) *windows_out = sp->tab ? sp->tab->w : 0; return MI355_OK;
  });
}
int mi355_srs_len(uint64_t handle, uint64_t *n_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  Srs *sp; CHK(srs_find(handle, &sp, "srs_len"));
  if (!n_out) return fail(MI355_EBADARG, "srs_len: null pointer");
  *n_out = sp->n; return MI355_OK;
  });
}
int mi355_srs_dev_ptr(uint64_t handle, void **dev_ptr_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  Srs *sp; CHK(srs_find(handle, &sp, "srs_dev_ptr"));
  if (!dev_ptr_out) return fail(MI355_EBADARG, "srs_dev_ptr: null pointer");
  *dev_ptr_out = sp->mem->sh.empty() ? nullptr : sp->mem->sh[0].dev; return MI355_OK;   // primary shard (all points with one device)
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
// polys: M pointers to n scalars each, host memory or primary-device memory.
static int msm_multi(const std::vector<Piece> &pieces, const fe_t *const *polys, ScalarLoc loc, uint32_t M, uint64_t n, void *out_host) {
  (void)n;
  const int D = g_ndev;
  const MsmOpts opts = t_opts;
  std::vector<int> rcs(D, MI355_OK); std::vector<std::string> errs(D);
  std::vector<const Piece *> by_slot(D, nullptr);
  for (const auto &p : pieces) by_slot[p.slot] = &p;
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
      if (slot == 0 || g.device == g_ctx[0].device) for (uint32_t m = 0; m < M; m++) ptrs[m] = polys[m] + p->a;
      else {
        fe_t *sc; CHK(ws_get("io.scalars", (size_t)M * p->cnt * sizeof(fe_t), (void **)&sc));
        for (uint32_t m = 0; m < M; m++) {
          ptrs[m] = sc + (size_t)m * p->cnt;
          HIPCHK(hipMemcpyPeerAsync(sc + (size_t)m * p->cnt, g.device, polys[m] + p->a, g_ctx[0].device, p->cnt * sizeof(fe_t), g.stream));   // over xGMI
        }
      }
      return msm_batch_impl(p->bases, ptrs.data(), M, p->cnt, nullptr, &p->pre, send);
    };
    rcs[slot] = guarded(body);
    if (rcs[slot] != MI355_OK) errs[slot] = g_err;
  };
  // scalars on the primary device may still be in flight on the primary stream: the peers' copies must wait for it
  if (loc == SCALARS_DEV && D > 1) { CHK(bind_ctx(0)); HIPCHK(hipStreamSynchronize(g.stream)); }
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
  return msm_batch_impl(pieces[0].bases, (const fe_t *const *)scalars_dev, M, n, out_g1_host, &pieces[0].pre);
}

int mi355_msm_g1_dev(uint64_t srs_handle, uint64_t base_offset, const void *scalars_dev, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!out_g1_host || (n && !scalars_dev)) return fail(MI355_EBADARG, "msm: null pointer");
  return msm_dev_dispatch(srs_handle, base_offset, &scalars_dev, 1, n, out_g1_host);
  });
}
int mi355_msm_g1_dev_async(uint64_t srs_handle, uint64_t base_offset, const void *scalars_dev, uint64_t n, void *out_g1_dev) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!out_g1_dev || (n && !scalars_dev)) return fail(MI355_EBADARG, "msm: null pointer");
  std::vector<Piece> pieces; CHK(srs_pieces(srs_handle, base_offset, n, pieces));
  if (pieces.size() != 1 || pieces[0].slot != 0) return fail(MI355_EBADARG, "msm_g1_dev_async: the range must lie on the primary device (one process per GPU drives its own shard)");
  const fe_t *sc = (const fe_t *)scalars_dev;
  return msm_batch_impl(pieces[0].bases, &sc, 1, n, nullptr, &pieces[0].pre, out_g1_dev);
  });
}
int mi355_g1_sum_dev(const void *pts_dev, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
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
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!out_g1_host || (n && !scalars_host)) return fail(MI355_EBADARG, "msm: null pointer");
  return msm_host_dispatch(srs_handle, base_offset, &scalars_host, 1, n, out_g1_host);
  });
}
int mi355_msm_g1_batch_dev(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_dev, uint32_t batch, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!out_g1_host || (batch && !scalars_dev)) return fail(MI355_EBADARG, "msm_batch: null pointer");
  for (uint32_t m = 0; m < batch; m++) if (n && !scalars_dev[m]) return fail(MI355_EBADARG, "msm_batch: null polynomial pointer");
  return msm_dev_dispatch(srs_handle, base_offset, scalars_dev, batch, n, out_g1_host);
  });
}
int mi355_msm_g1_batch_host(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_host, uint32_t batch, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!out_g1_host || (batch && !scalars_host)) return fail(MI355_EBADARG, "msm_batch: null pointer");
  for (uint32_t m = 0; m < batch; m++) if (n && !scalars_host[m]) return fail(MI355_EBADARG, "msm_batch: null polynomial pointer");
  return msm_host_dispatch(srs_handle, base_offset, scalars_host, batch, n, out_g1_host);
  });
}
int mi355_msm_set_pipeline(uint32_t chunks, uint32_t min_log_n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  if (chunks > 16 || min_log_n > 31) return fail(MI355_EBADARG, "msm_set_pipeline: chunks <= 16, min_log_n <= 31");
  for (int s = 0; s < std::max(1, g_ndev); s++) { g_ctx[s].msm_chunks = chunks ? chunks : 1; g_ctx[s].msm_chunk_min_log = chunks ? min_log_n : 23; }
  return MI355_OK;
  });
}
int mi355_msm_g1_adhoc_host(const void *bases_host, const void *scalars_host, uint64_t n, void *out_g1_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
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
  std::lock_guard<std::mutex> lk(g_mu);
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
// Curve::batch_normalize: n Jacobian points (96 B, any representative) -> n affine points (64 B); k_g1_batch_normalize (frscan.cuh)
int mi355_g1_batch_normalize_dev(const void *jac_dev, void *affine_dev, uint64_t n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (n && (!jac_dev || !affine_dev)) return fail(MI355_EBADARG, "g1_batch_normalize: null pointer");
  if (n >= (1ull << 31)) return fail(MI355_EBADARG, "g1_batch_normalize: n must be < 2^31");
  if (n) {
    const char *a = (const char *)jac_dev, *b = (const char *)affine_dev;
    if (a < b + n * sizeof(g1_affine_t) && b < a + n * sizeof(g1_jac_t)) return fail(MI355_EBADARG, "g1_batch_normalize: input and output must not overlap");
    hipLaunchKernelGGL(k_g1_batch_normalize, dim3(ceil_div(n, FRSCAN_THREADS)), dim3(FRSCAN_THREADS), 0, g.stream, (const g1_jac_t *)jac_dev, (g1_affine_t *)affine_dev, n);
    HIPCHK(hipGetLastError());
  }
  return finish_async();
  });
}
int mi355_g1_batch_normalize_host(const void *jac_host, void *affine_host, uint64_t n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (n && (!jac_host || !affine_host)) return fail(MI355_EBADARG, "g1_batch_normalize: null pointer");
  if (n >= (1ull << 31)) return fail(MI355_EBADARG, "g1_batch_normalize: n must be < 2^31");
  if (!n) return MI355_OK;
  char *dev; CHK(ws_get("io.g1norm", n * (sizeof(g1_jac_t) + sizeof(g1_affine_t)), (void **)&dev));
  g1_jac_t *in = (g1_jac_t *)dev; g1_affine_t *out = (g1_affine_t *)(dev + n * sizeof(g1_jac_t));
  HIPCHK(hipMemcpyAsync(in, jac_host, n * sizeof(g1_jac_t), hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_g1_batch_normalize, dim3(ceil_div(n, FRSCAN_THREADS)), dim3(FRSCAN_THREADS), 0, g.stream, (const g1_jac_t *)in, out, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(affine_host, out, n * sizeof(g1_affine_t), hipMemcpyDeviceToHost, g.stream));
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
  std::lock_guard<std::mutex> lk(g_mu);
  const Ctx &p = g_ctx[0];
  if (c_out) *c_out = p.last_c; if (windows_out) *windows_out = p.last_w; if (entries_out) *entries_out = p.last_entries; return MI355_OK;
  });
}
// how the last MSM ran: device slots that took part, the exchange ("none" | "rccl_allgather" | "device_copy"), whether the window tables
// (one shared bucket set) were used, and the number of point-range slices of the host-pointer path
int mi355_msm_last_run(int *devices_out, const char **exchange_out, int *shared_tables_out, int *host_slices_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (devices_out) *devices_out = g_last_devices; if (exchange_out) *exchange_out = g_last_exchange;
  if (shared_tables_out) *shared_tables_out = g_ctx[0].last_shared ? 1 : 0; if (host_slices_out) *host_slices_out = g_ctx[0].last_host_slices;
  return MI355_OK;
}

// ---- NTT
static int check_ntt_args(const void *data, uint32_t log_n, const void *omega) {
  if (!data || !omega) return fail(MI355_EBADARG, "ntt: null pointer");
  if (log_n > 28) return fail(MI355_EBADARG, "ntt: log_n > 28");
  return MI355_OK;
}
int mi355_ntt_fr_dev(void *data_dev, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(data_dev, log_n, omega));
  CHK(ntt_dev_impl((const fe_t *)data_dev, 1ull << log_n, (fe_t *)data_dev, log_n, omega, nullptr, nullptr));
  return finish_async();
  });
}
int mi355_intt_fr_dev(void *data_dev, uint32_t log_n, const void *omega_inv, const void *divisor) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(data_dev, log_n, omega_inv));
  if (!divisor) return fail(MI355_EBADARG, "intt: null divisor");
  fe_t post[3]; for (int i = 0; i < 3; i++) memcpy(&post[i], divisor, 32);
  CHK(ntt_dev_impl((const fe_t *)data_dev, 1ull << log_n, (fe_t *)data_dev, log_n, omega_inv, nullptr, post));
  return finish_async();
  });
}
int mi355_coeff_to_extended_dev(void *dst_dev, const void *coeffs_dev, uint32_t log_n, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(dst_dev, log_ext, extended_omega));
  if (!coeffs_dev || !g_coset || !g_coset_inv || log_n > log_ext) return fail(MI355_EBADARG, "coeff_to_extended: bad argument");
  fe_t pre[3]; pre[0] = Fr::one(); memcpy(&pre[1], g_coset, 32); memcpy(&pre[2], g_coset_inv, 32);
  CHK(ntt_dev_impl((const fe_t *)coeffs_dev, 1ull << log_n, (fe_t *)dst_dev, log_ext, extended_omega, pre, nullptr));
  return finish_async();
  });
}
static int extended_to_coeff_locked(void *data_dev, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega_inv, const void *extended_ifft_divisor);
int mi355_extended_to_coeff_dev(void *data_dev, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega_inv, const void *extended_ifft_divisor) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  CHK(extended_to_coeff_locked(data_dev, log_ext, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor));
  return finish_async();
  });
}

struct NttHostArgs { uint32_t log_n; const void *omega; const void *divisor; };
int mi355_ntt_fr_host(void *data_host, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(data_host, log_n, omega));
  NttHostArgs a{log_n, omega, nullptr};
  const size_t bytes = sizeof(fe_t) << log_n;
  return with_host_io(data_host, bytes, bytes, bytes, "io.ntt", [](void *dev, void *ud) { auto *a = (NttHostArgs *)ud; return ntt_dev_impl((const fe_t *)dev, 1ull << a->log_n, (fe_t *)dev, a->log_n, a->omega, nullptr, nullptr); }, &a);
  });
}
int mi355_intt_fr_host(void *data_host, uint32_t log_n, const void *omega_inv, const void *divisor) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(data_host, log_n, omega_inv));
  if (!divisor) return fail(MI355_EBADARG, "intt: null divisor");
  NttHostArgs a{log_n, omega_inv, divisor};
  const size_t bytes = sizeof(fe_t) << log_n;
  return with_host_io(data_host, bytes, bytes, bytes, "io.ntt", [](void *dev, void *ud) {
    auto *a = (NttHostArgs *)ud; fe_t post[3]; for (int i = 0; i < 3; i++) memcpy(&post[i], a->divisor, 32);
    return ntt_dev_impl((const fe_t *)dev, 1ull << a->log_n, (fe_t *)dev, a->log_n, a->omega, nullptr, post); }, &a);
  });
}
// ---- DFT over G1 points (best_fft::<Fr, G1>, g_to_lagrange)
int mi355_g1_fft_dev(void *points_jac_dev, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(points_jac_dev, log_n, omega));
  CHK(g1fft_impl(points_jac_dev, 1, points_jac_dev, 1, log_n, omega, nullptr));
  return finish_async();
  });
}
int mi355_g1_fft_host(void *points_jac_host, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(points_jac_host, log_n, omega));
  NttHostArgs a{log_n, omega, nullptr};
  const size_t bytes = sizeof(g1_jac_t) << log_n;
  return with_host_io(points_jac_host, bytes, bytes, bytes, "io.g1fft", [](void *dev, void *ud) { auto *a = (NttHostArgs *)ud; return g1fft_impl(dev, 1, dev, 1, a->log_n, a->omega, nullptr); }, &a);
  });
}
int mi355_g_to_lagrange_dev(const void *g_affine_dev, void *g_lagrange_affine_dev, uint32_t log_n, const void *omega_inv, const void *n_inv) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init()); CHK(check_ntt_args(g_affine_dev, log_n, omega_inv));
  if (!g_lagrange_affine_dev || !n_inv) return fail(MI355_EBADARG, "g_to_lagrange: null pointer");
  CHK(g1fft_impl(g_affine_dev, 0, g_lagrange_affine_dev, 0, log_n, omega_inv, n_inv));
  return finish_async();
  });
}
// the first n points of a registered basis as ONE contiguous array on the primary device: the primary shard itself when it covers them,
// otherwise a temporary assembled from the shards (xGMI peer copies)
static int srs_gather_to_primary(const Srs &sr, uint64_t n, const g1_affine_t **out) {
  const SrsMem &mem = *sr.mem;
  if (!mem.sh.empty() && mem.sh[0].slot == 0 && mem.sh[0].lo == 0 && mem.sh[0].n >= n) { *out = mem.sh[0].dev; return MI355_OK; }
  CHK(bind_ctx(0));
  g1_affine_t *tmp; CHK(ws_get("srs.gather", n * sizeof(g1_affine_t), (void **)&tmp));
  for (const auto &sh : mem.sh) {
    if (sh.lo >= n) continue;
    const uint64_t cnt = std::min(sh.n, n - sh.lo);
    if (g_ctx[sh.slot].device == g.device) HIPCHK(hipMemcpyAsync(tmp + sh.lo, sh.dev, cnt * sizeof(g1_affine_t), hipMemcpyDeviceToDevice, g.stream));
    else HIPCHK(hipMemcpyPeerAsync(tmp + sh.lo, g.device, sh.dev, g_ctx[sh.slot].device, cnt * sizeof(g1_affine_t), g.stream));
  }
  *out = tmp; return MI355_OK;
}
int mi355_srs_downsize(uint64_t g_handle, uint32_t k, const void *omega_inv, const void *n_inv, uint64_t *g_lagrange_handle_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  Srs *sp; CHK(srs_find(g_handle, &sp, "srs_downsize"));
  if (!omega_inv || !n_inv || !g_lagrange_handle_out || k > 28 || (1ull << k) > sp->n) return fail(MI355_EBADARG, "srs_downsize: bad argument (2^k must not exceed the registered basis)");
  const uint64_t n = 1ull << k;
  for (int sl = 1; sl < g_ndev; sl++) { CHK(bind_ctx(sl)); HIPCHK(hipStreamSynchronize(g.stream)); }
  CHK(bind_ctx(0));
  const g1_affine_t *src; CHK(srs_gather_to_primary(*sp, n, &src));
  g1_affine_t *res; HIPCHK(hipMalloc((void **)&res, n * sizeof(g1_affine_t)));
  int rc = g1fft_impl(src, 0, res, 0, k, omega_inv, n_inv);
  if (rc == MI355_OK) rc = finish_async();
  if (rc == MI355_OK && hipStreamSynchronize(g.stream) != hipSuccess) rc = fail(MI355_EHIP, "srs_downsize: stream synchronize failed");
  Srs s; s.n = n; s.mem = std::make_shared<SrsMem>();
  if (rc == MI355_OK) {
    if (plan_shards(n).size() == 1) { Shard one; one.slot = 0; one.lo = 0; one.n = n; one.dev = res; one.owned = true; s.mem->sh.push_back(one); res = nullptr; }
    else rc = srs_scatter_from_primary(*s.mem, res, n, false);
  }
  if (res) { (void)bind_ctx(0); (void)hipFree(res); }
  if (rc != MI355_OK) return rc;
  *g_lagrange_handle_out = srs_insert(s); return MI355_OK;
  });
}
int mi355_srs_read_host(uint64_t handle, uint64_t offset, uint64_t n, void *out_affine_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  Srs *sp; CHK(srs_find(handle, &sp, "srs_read_host"));
  if (n && !out_affine_host) return fail(MI355_EBADARG, "srs_read_host: null pointer");
  if (offset > sp->n || n > sp->n - offset) return fail(MI355_EBADARG, "srs_read_host: range exceeds the registered basis");
  for (const auto &sh : sp->mem->sh) {
    const uint64_t lo = std::max(offset, sh.lo), hi = std::min(offset + n, sh.lo + sh.n);
    if (hi <= lo) continue;
    CHK(bind_ctx(sh.slot));
    HIPCHK(hipMemcpyAsync((g1_affine_t *)out_affine_host + (lo - offset), sh.dev + (lo - sh.lo), (hi - lo) * sizeof(g1_affine_t), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  return bind_ctx(0);
  });
}
int mi355_coeff_to_extended_host(void *dst_host, const void *coeffs_host, uint32_t log_n, uint32_t log_ext, const void *g_coset, const void *g_coset_inv, const void *extended_omega) {
  return guarded([&]() -> int {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    CHK(need_init()); CHK(check_ntt_args(dst_host, log_ext, extended_omega));
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
// body of mi355_extended_to_coeff_dev, to be called with g_mu held
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
  std::lock_guard<std::mutex> lk(g_mu);   // held across stage, compute and copy-back: the staging buffer is shared between the host entry points
  CHK(need_init()); CHK(check_ntt_args(data_host, log_ext, extended_omega_inv));
  void *dev; CHK(ws_get("io.ntt", sizeof(fe_t) << log_ext, &dev));
  HIPCHK(hipMemcpyAsync(dev, data_host, sizeof(fe_t) << log_ext, hipMemcpyHostToDevice, g.stream));
  CHK(extended_to_coeff_locked(dev, log_ext, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor));
  HIPCHK(hipMemcpyAsync(data_host, dev, sizeof(fe_t) << log_ext, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream)); resolve_spans();
  return MI355_OK;
  });
}

// ---- distribute_powers / coset NTT
static int distribute_powers_locked(void *data_dev, uint64_t n, const void *factor, const void *src_dev = nullptr) {
  if (!factor || (n && !data_dev)) return fail(MI355_EBADARG, "distribute_powers: null pointer");
  if (n == 0) return MI355_OK;
  fe_t f; memcpy(&f, factor, 32);
  hipLaunchKernelGGL(k_distribute_powers, dim3(ceil_div(n, (uint64_t)EVAL_RUN * 256)), dim3(256), 0, g.stream, (const fe_t *)(src_dev ? src_dev : data_dev), (fe_t *)data_dev, n, f);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}
int mi355_distribute_powers_fr_dev(void *data_dev, uint64_t n, const void *factor) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  return distribute_powers_locked(data_dev, n, factor);
  });
}
int mi355_coset_ntt_fr_dev(void *dst_dev, const void *coeffs_dev, uint32_t log_n, const void *coset_factor, const void *omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);   // one critical section: copy, coset scaling and transform are queued back to back
  CHK(need_init()); CHK(check_ntt_args(dst_dev, log_n, omega));
  if (!coeffs_dev || !coset_factor) return fail(MI355_EBADARG, "coset_ntt: null pointer");
  CHK(distribute_powers_locked(dst_dev, 1ull << log_n, coset_factor, coeffs_dev));   // dst = coeffs[i] * factor^i (one pass, no copy first)
  CHK(ntt_dev_impl((const fe_t *)dst_dev, 1ull << log_n, (fe_t *)dst_dev, log_n, omega, nullptr, nullptr));
  return finish_async();
  });
}

// ---- element-wise vector operations on resident polynomials
int mi355_fr_vec_op_dev(int op, void *dst_dev, const void *a_dev, const void *b_dev, uint64_t n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (op < 0 || op > 2 || (n && (!dst_dev || !a_dev || !b_dev))) return fail(MI355_EBADARG, "fr_vec_op: bad argument");
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(k_fr_vec_op, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, op, (fe_t *)dst_dev, (const fe_t *)a_dev, (const fe_t *)b_dev, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_vec_axpy_dev(void *dst_dev, const void *a_dev, const void *b_dev, const void *scalar, uint64_t n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!scalar || (n && (!dst_dev || !b_dev))) return fail(MI355_EBADARG, "fr_vec_axpy: null pointer");
  if (n == 0) return MI355_OK;
  fe_t s; memcpy(&s, scalar, 32);
  hipLaunchKernelGGL(k_fr_vec_axpy, dim3(g.prop.multiProcessorCount * 8), dim3(256), 0, g.stream, (fe_t *)dst_dev, (const fe_t *)a_dev, (const fe_t *)b_dev, s, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_kate_division_dev(void *dst_dev, const void *poly_dev, uint64_t n, const void *z) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!z || n == 0 || !poly_dev || (n > 1 && !dst_dev)) return fail(MI355_EBADARG, "fr_kate_division: null pointer or empty polynomial");
  if (n >= (1ull << 40)) return fail(MI355_EBADARG, "fr_kate_division: n too large");
  if (n == 1) return MI355_OK;   // a constant: the quotient is empty
  fe_t m; memcpy(&m, z, 32);
  CHK(linrec_impl((const fe_t *)poly_dev + 1, (fe_t *)dst_dev, n - 1, m, true, 0));
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_batch_invert_dev(void *data_dev, uint64_t n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (n && !data_dev) return fail(MI355_EBADARG, "fr_batch_invert: null pointer");
  if (n >= (1ull << 40)) return fail(MI355_EBADARG, "fr_batch_invert: n too large");
  if (n == 0) return MI355_OK;
  CHK(batch_invert_impl((fe_t *)data_dev, n, 0));
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_prefix_product_dev(void *dst_dev, const void *src_dev, uint64_t n, void *total_out_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (n && (!dst_dev || !src_dev)) return fail(MI355_EBADARG, "fr_prefix_product: null pointer");
  if (n >= (1ull << 40)) return fail(MI355_EBADARG, "fr_prefix_product: n too large");
  const uint32_t tiles = (uint32_t)ceil_div(n, FRSCAN_TILE);
  fe_t *tile_prod, *tile_prefix, *total;
  CHK(ws_get("frscan.tile_prod", ((size_t)tiles + 1) * sizeof(fe_t), (void **)&tile_prod));
  CHK(ws_get("frscan.tile_prefix", ((size_t)tiles + 1) * sizeof(fe_t), (void **)&tile_prefix));
  CHK(ws_get("frscan.total", sizeof(fe_t), (void **)&total));
  const size_t lds = (size_t)(FRSCAN_THREADS * 65 + 2 * FRSCAN_THREADS * 9) * 4;
  hipStream_t s = g.stream;
  if (tiles) hipLaunchKernelGGL(k_fr_prefix_product<0>, dim3(tiles), dim3(FRSCAN_THREADS), lds, s, (const fe_t *)src_dev, (fe_t *)dst_dev, n, tile_prod, (const fe_t *)tile_prefix);
  hipLaunchKernelGGL(k_fr_scan_tile_products, dim3(1), dim3(FRSCAN_THREADS), 0, s, (const fe_t *)tile_prod, tile_prefix, tiles, total);
  if (tiles) hipLaunchKernelGGL(k_fr_prefix_product<1>, dim3(tiles), dim3(FRSCAN_THREADS), lds, s, (const fe_t *)src_dev, (fe_t *)dst_dev, n, tile_prod, (const fe_t *)tile_prefix);
  HIPCHK(hipGetLastError());
  if (total_out_host) { HIPCHK(hipMemcpyAsync(total_out_host, total, sizeof(fe_t), hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); }
  return MI355_OK;
  });
}
int mi355_fr_vec_mul_periodic_dev(void *data_dev, uint64_t n, const void *table_host, uint32_t period) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
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
    hipLaunchKernelGGL(k_fr_sum, dim3(1), dim3(256), 0, g.stream, partial, (uint64_t)blocks, partial + blocks);
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
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  return eval_polynomial_locked(poly_dev, n, point, out_fr_host);
  });
}
int mi355_eval_polynomial_host(const void *poly_host, uint64_t n, const void *point, void *out_fr_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);   // held across staging and evaluation (the staging buffer is shared between the host entry points)
  CHK(need_init());
  if (n && !poly_host) return fail(MI355_EBADARG, "eval_polynomial: null pointer");
  void *dev = nullptr;
  if (n) { CHK(ws_get("io.ntt", n * sizeof(fe_t), &dev)); HIPCHK(hipMemcpyAsync(dev, poly_host, n * sizeof(fe_t), hipMemcpyHostToDevice, g.stream)); }
  return eval_polynomial_locked(dev, n, point, out_fr_host);
  });
}

// ---- synthetic SRS
static int ensure_fixed_base_table() {
  if (g.fixed_base_table) return MI355_OK;
  HIPCHK(hipMalloc((void **)&g.fixed_base_table, 32 * 256 * sizeof(g1_affine_t)));
  hipLaunchKernelGGL(k_fixed_base_table, dim3(1), dim3(256), 0, g.stream, g.fixed_base_table);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}
int mi355_g1_fixed_base_mul_dev(void *points_dev, const void *scalars_dev, uint64_t n) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!points_dev || !scalars_dev) return fail(MI355_EBADARG, "fixed_base_mul: null pointer");
  CHK(ensure_fixed_base_table());
  hipLaunchKernelGGL(k_fixed_base_mul, dim3(ceil_div(n, 256)), dim3(256), 0, g.stream, g.fixed_base_table, (const fe_t *)scalars_dev, (g1_affine_t *)points_dev, n);
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_srs_setup_dev(void *g_dev, void *g_lagrange_dev, uint32_t k, const void *tau, const void *omega) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
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

// ---- G2: s_g2 = tau * G2 of ParamsKZG::setup (the only G2 arithmetic on the path)
int mi355_g2_mul_host(const void *p_affine_host, const void *scalar, void *out_affine_host) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  if (!p_affine_host || !scalar || !out_affine_host) return fail(MI355_EBADARG, "g2_mul: null pointer");
  g2_affine_t *dev; CHK(ws_get("g2.io", 2 * sizeof(g2_affine_t) + 16, (void **)&dev));
  uint32_t *flag = reinterpret_cast<uint32_t *>(dev + 2);
  fe_t k; memcpy(&k, scalar, 32);
  HIPCHK(hipMemcpyAsync(dev, p_affine_host, sizeof(g2_affine_t), hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_g2_mul, dim3(1), dim3(64), 0, g.stream, (const g2_affine_t *)dev, k, dev + 1, flag);
  HIPCHK(hipGetLastError());
  uint32_t ok = 0; g2_affine_t res;
  HIPCHK(hipMemcpyAsync(&res, dev + 1, sizeof res, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipMemcpyAsync(&ok, flag, 4, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  if (!ok) return fail(MI355_EBADARG, "g2_mul: the point is not on the twist y^2 = x^3 + 3 / (9 + u)");
  memcpy(out_affine_host, &res, sizeof res);
  return MI355_OK;
  });
}

// ---- test hook: read back a workspace buffer ("msm.sorted", "msm.offsets", ...) after a call
int mi355_debug_ws_read(const char *role, uint64_t offset, void *dst_host, uint64_t bytes) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  CHK(need_init());
  auto it = g.ws.find(role ? role : "");
  if (it == g.ws.end() || !dst_host || offset + bytes > it->second.cap) return fail(MI355_EBADARG, "debug_ws_read: unknown role or range");
  HIPCHK(hipStreamSynchronize(g.stream));
  HIPCHK(hipMemcpy(dst_host, (const char *)it->second.p + offset, bytes, hipMemcpyDeviceToHost));
  return MI355_OK;
  });
}

// ---- profiling
int mi355_profile_enable(int on) { std::lock_guard<std::mutex> lk(g_mu); for (int s = 0; s < MAX_DEV; s++) g_ctx[s].profiling = on != 0; return MI355_OK; }
int mi355_profile_reset(void) { std::lock_guard<std::mutex> lk(g_mu); for (int s = g_ndev - 1; s >= 0; s--) { if (bind_ctx(s) == MI355_OK) resolve_spans(); g.prof.clear(); } use_ctx(0); return MI355_OK; }
int mi355_profile_get(const char *name, double *ms_out, uint64_t *launches_out) {
  return guarded([&]() -> int {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!name) return fail(MI355_EBADARG, "profile_get: null name");
  use_ctx(0);   // the primary device's record (a sharded MSM runs the same kernels on every device)
  if (g_ndev && g.inited && bind_ctx(0) == MI355_OK) resolve_spans();
  auto it = g.prof.find(name);
  if (ms_out) *ms_out = it == g.prof.end() ? 0.0 : it->second.ms;
  if (launches_out) *launches_out = it == g.prof.end() ? 0 : it->second.launches;
  return MI355_OK;
  });
}

}  // extern "C"

