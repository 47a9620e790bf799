// lib_common.hpp -- host-side state shared by the translation units of libmi355zk.so (lib_core / lib_msm / lib_ntt / lib_aux .hip):
// device contexts, the per-device locks, SRS handles, the workspace arena, the resident-buffer registry, HIP-event profiling and the
// error plumbing of the C-ABI (include/mi355zk.h).  Host logic only; no kernel is declared here -- every lib_*.hip includes the
// kernel headers it launches and nothing else, so a change to one kernel family rebuilds one translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <condition_variable>
#include <unordered_map>
#include <vector>

#include "../../include/mi355zk.h"
#include "fp.hpp"
#include "g1.hpp"
#include "g1_29.hpp"
#include "ntt_types.hpp"

namespace mi355 {
using namespace zk;

extern thread_local std::string g_err;
int fail(int code, const std::string &msg);

#define HIPCHK(expr)                                                                                                     \
  do {                                                                                                                   \
    hipError_t _e = (expr);                                                                                              \
    if (_e != hipSuccess) {                                                                                              \
      char _b[512]; snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      (void)hipGetLastError();                                                                                           \
      return ::mi355::fail(_e == hipErrorOutOfMemory ? MI355_EOOM : MI355_EHIP, _b);                                     \
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
  uint32_t split[2] = {0, 0};
  uint32_t direct2[2] = {0, 0};
  std::map<std::string, Tw29> scaled;   // the last strided level's direct twiddle table times a constant (key: the 32 bytes of the constant)
  // the tables of a coset shift folded into the first pass, per coset factor f (key: its 32 bytes): in[m] = (f^(2^log_t))^m, s2d[k][column] = w_S^(column k) f^column
  struct CosetTw { Tw29 in, s2d; };
  std::map<std::string, CosetTw> coset;
  // twiddles as w * 2^261 mod r in 29-bit limbs (SoA) for the kernels of ntt29.hpp; tw29_s_lo[l] of a big level is the 2^log_s-entry table [k][column]
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
  void *comm = nullptr;         // ncclComm_t of this device (mi355_init_multi with distinct devices)
  hipEvent_t ev_xchg = nullptr;
  hipEvent_t ev_xchg2 = nullptr;   // recorded behind a cross-slot mi355_buf_copy on the destination's stream; the source slot's stream waits for it
  // host-pointer MSM: chunked copy on its own stream, overlapped with the digit extraction (msm_host_single); also the stream the
  // resident-buffer uploads run on (mi355_buf_upload)
  hipStream_t copy_stream = nullptr; hipEvent_t ev_copy[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_up_fork = nullptr;   // mi355_buf_upload into a block in use: recorded on the compute stream, awaited by the copy stream (upload mutex)
  hipEvent_t ev_up[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; uint32_t up_next = 0;   // uploads on copy_stream (ring: several threads may upload at once); the compute stream waits for each
  uint32_t host_slice_min_log = 22;   // MI355_HOST_SLICE_MIN_LOG: smallest log2(n) the host-pointer MSM cuts into slices (tests lower it)
  uint32_t host_batch_overlap = 1;   // MI355_HOST_BATCH_OVERLAP=0: the host-pointer batch transforms copy and compute one item at a time (no helper thread)
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
  uint32_t seg_factor = 16;
  uint32_t sort_fb = 11;        // MI355_SORT_FB: fine (level-2) key bits of the sorter, 9..12
  uint32_t sort_split = 1;      // MI355_SORT_SPLIT: the level-1 output is two streams (payload u32 + fine key u16); 1 = 24 576-entry tiles where the bin bookkeeping leaves room, 2 = 16 384-entry tiles (the single-stream u64 records were an A/B path of round 3 and left the library in round 6)
  uint32_t reduce_chains = 131072;   // MI355_REDUCE_CHAINS: target number of running-sum chains of the bucket reduction
  uint32_t seg_fill = 40, seg_fill_segfix = 40;   // MI355_SEG_FILL / MI355_SEG_FILL_SEGFIX (even, 2..100 %: above 100 the segments would no longer cover the entries): share of the launched accumulate threads the actual entries are spread over
  uint32_t seg_min = 16;             // MI355_SEG_MIN: shortest accumulate segment (entries per thread) when few digits are non-zero
  uint32_t fixup_mode = 2;           // MI355_FIXUP_MODE: 2 = by shape (see msm_enqueue), 0 = per-bucket kernels (four lanes / workgroup / several workgroups per bucket), 1 = one segmented reduction over the partial sums (k_msm_segfix: measured better for two-partial buckets, worse for spans of 15-30, profiles/r02b_segfix_ab.log)
  uint32_t fixup_huge_min = 2048;    // MI355_FIXUP_HUGE_MIN (>= 2048): bucket spans from this many accumulate threads on are summed by several workgroups
  uint32_t fixup_serial_max = 32;    // MI355_FIXUP_SERIAL_MAX: bucket spans (in accumulate threads) above this go to the workgroup-per-bucket fix-up
  uint32_t fixup_lanes_max_log = 17;  // MI355_FIXUP_LANES_MAX_LOG: bucket sets up to 2^this records take the four-lanes-per-bucket fix-up
  uint32_t tail_coop_mask = 15;      // MI355_TAIL_COOP_MASK: which tail kernels may take the quad form (1 fix-up, 2 running sums, 4 trees, 8 final Horner)
  uint32_t tail_coop_max = 65536;    // MI355_TAIL_COOP_MAX: reduction-tail kernels with at most this many logical threads run the quad-cooperative (latency) form of the point addition; 0 disables
  uint32_t reduce_min_chunk = 4;     // MI355_REDUCE_MIN_CHUNK: shortest running-sum chain (buckets per reduce thread) small bucket sets are cut into
  uint32_t sort_t1 = 16384;     // MI355_SORT_T1=8192 selects the smaller level-1 tile (2 workgroups per CU)
  uint32_t ntt_direct2_max_log = 25;   // MI355_NTT_DIRECT2_MAX_LOG: levels up to 2^this elements read their inter-level twiddles from a full [k][column] table (36 B per element of the level: 0.6 GB at 2^24; tables of 256 MB and more only while HBM has the table + max(16 GiB, 1/12 of the device) to spare, lib_ntt.hip hbm_spare_for_table; otherwise, and above the cap, the level multiplies two half-size table entries per element).  At 2^26 the 2.4 GB table buys 0.5 % (8.53 vs 8.58 ms, profiles/r06_direct2_k26_ab.json; round 3 had measured 1.7 %), so the default stops at 2^25; 0 disables
  uint32_t ntt_coset_fold_max_log = 26;  // MI355_NTT_COSET_FOLD_MAX_LOG: coset transforms up to 2^this fold distribute_powers into their first pass (one 36 B x 2^log_n table per coset factor: 38 MB at 2^20, 0.6 GB at 2^24, 2.4 GB at 2^26; tables of 256 MB and more are only built while HBM has the table + max(16 GiB, 1/12 of the device) to spare -- otherwise, and above the cap, the separate pass runs); 0: always the separate k_distribute_powers pass
  uint32_t ntt_direct2_min_log = 0;   // MI355_NTT_DIRECT2_MIN_LOG: levels of at least 2^this elements read their inter-level twiddles from the [k][column] table (coalesced, next to the data); smaller ones gather from ONE 1-D table w_S^e by e = column k.  Round 6: 0 (every level) -- the gather cost more than it looked: 2^20 transform 111.5 -> 103 us, 2^18 26.1 -> 25.3, 2^24 1 965 -> 1 932, 2^26 8.89 -> 8.73 ms (profiles/r06_direct2_small_levels_ab.json; 21 restores round 5)
  uint32_t ntt_fold_scale = 1;  // MI355_NTT_FOLD_SCALE=0: the inverse transform's divisor stays a multiplication in the closing pass
  uint32_t ntt_batch_max_log = 22;       // MI355_NTT_BATCH_MAX_LOG: the batch entry points run transforms up to 2^this as batched launches (blockIdx.y = polynomial); 0: always the loop of single transforms
  uint32_t ntt_two_level_max_log = 18;   // MI355_NTT_TWO_LEVEL_MAX_LOG (18..20): transforms up to 2^this run as two passes instead of three
  uint32_t ntt_tile_log = 11;   // log2 of the LDS tile in elements (MI355_NTT_TILE_LOG)
  bool profiling = false;
  bool trace = false;           // MI355_TRACE=1: one stderr line per MSM / NTT call (host wall time; device transforms are synchronised for it)
  std::map<std::string, Prof> prof;
  int last_c = 0, last_w = 0; uint64_t last_entries = 0; bool last_shared = false; int last_host_slices = 1;
};

// One context per bound device; slot 0 is the primary.  LOCKING (round 3): one mutex PER DEVICE SLOT instead of one for the library.
//   * an entry point that works on one device (every transform, scan, evaluation, element-wise operation, buffer copy) holds that slot's
//     lock only, so callers on different devices -- rayon workers of one prover -- run concurrently;
//   * MSM entry points hold slot 0 and, with several devices bound, every other slot as well (the point range is sharded over all of them);
//   * lifecycle and SRS management (init, shutdown, register, release, precompute, downsize, profile control) hold every slot.
// Locks are always taken in ascending slot order.  The SRS handle table is read under slot 0's lock and written under all of them.
// `g` names the context the CURRENT THREAD works on (bind_ctx), which is why the pointer is thread-local: API threads and the per-device
// worker threads of sharded MSMs / batched transforms each have their own.
constexpr int MAX_DEV = 16;
extern std::mutex g_ctx_mu[MAX_DEV];
extern std::mutex g_upload_mu[MAX_DEV];
extern Ctx g_ctx[MAX_DEV];
extern int g_ndev;
extern bool g_dup_devices;     // test mode: the same physical device bound to several slots (exchange by device copies instead of RCCL)
extern uint32_t g_shard_min_log;  // MI355_SHARD_MIN_LOG: a basis with fewer than 2^this points per device stays on the primary device (tests lower it)
extern bool g_peer_ok[MAX_DEV][MAX_DEV];   // [s][t]: kernels on slot s may read memory of slot t directly
extern bool g_force_exchange;  // MI355_MULTI_FORCE=1: take the sharded path (partials + exchange + fold) even with one device
extern thread_local Ctx *g_cur;
#define g (*::mi355::g_cur)
extern std::unordered_map<uint64_t, Srs> &g_srs;   // heap-allocated, never destroyed (no HIP calls from static destruction)
extern uint64_t g_next_handle;
extern int g_last_devices; extern const char *g_last_exchange;
extern int g_auto_max_c;          // MI355_MSM_AUTO_MAX_C (22..24): widest window the automatic choices may take

struct DevGuard {   // one device slot
  std::unique_lock<std::mutex> lk;
  explicit DevGuard(int slot) : lk(g_ctx_mu[slot]) {}
};
struct AllGuard {   // every slot, ascending
  std::unique_lock<std::mutex> lk[MAX_DEV];
  AllGuard() { for (int i = 0; i < MAX_DEV; i++) lk[i] = std::unique_lock<std::mutex>(g_ctx_mu[i]); }
};
struct MsmGuard {   // slot 0, plus every other bound slot when the process drives several devices
  std::unique_lock<std::mutex> lk[MAX_DEV];
  MsmGuard() { lk[0] = std::unique_lock<std::mutex>(g_ctx_mu[0]); for (int i = 1; i < g_ndev && i < MAX_DEV; i++) lk[i] = std::unique_lock<std::mutex>(g_ctx_mu[i]); }
};

// per-THREAD MSM options (mi355_msm_set_normalise / _set_window_bits): a rayon worker that asks for un-normalised partial sums must not
// change what another worker's commit returns (SURVEY 8b "Threading")
struct MsmOpts { bool normalise = true; int force_c = 0; bool no_tables = false; };
extern thread_local MsmOpts t_opts;

inline void use_ctx(int slot) { g_cur = &g_ctx[slot]; }
// Every compute entry point starts here.  The HIP current device is per host thread and calls arrive from whichever thread runs
// create_proof (rayon workers included, SURVEY 8b "Threading"), so the bound device is re-selected on the calling thread each time.
int bind_ctx(int slot);
int need_init(int slot = 0);

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

int ws_get(const char *role, size_t bytes, void **out);
// hipMalloc on the current context's device; on out-of-memory the device's pooled (free) mi355_buf blocks go back to HIP and the call is retried once
int dev_malloc(void **out, size_t bytes, const char *what);

// ---- profiling: a list of (name, start, stop) event pairs per context, resolved after the stream is idle
struct Scope {
  bool on; hipEvent_t a = nullptr, b = nullptr; std::string name; hipStream_t st;
  Scope(const char *n, hipStream_t s = nullptr) : on(g.profiling), name(n), st(s ? s : g.stream) { if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); } }
  void close() { if (on) { (void)hipEventRecord(b, st); g.spans.push_back({name, a, b}); on = false; } }
  ~Scope() { close(); }
};
void resolve_spans();
inline int finish_async() { if (g.profiling) resolve_spans(); return MI355_OK; }

inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
inline uint32_t log2_ceil(uint64_t n) { uint32_t l = 0; while ((1ull << l) < n) l++; return l; }

// No exception may cross the C boundary (the Rust callers are `extern "C"` and would abort): every multi-line entry point runs inside
// this guard, which turns allocation failures and anything else the host-side C++ might throw into error codes.
// MI355_TRACE=2: every entry point runs inside a roctx range named after it ("wrap every FFI call in a roctx range"), so a
// rocprofv3 --marker-trace timeline of a real proof shows which create_proof call each kernel belongs to.  The marker library
// (librocprofiler-sdk-roctx.so.1, else libroctx64.so.4) is dlopen()ed on first use; without it, or with MI355_TRACE != 2, this costs one load.
struct Roctx { int state = 0; int (*push)(const char *) = nullptr; int (*pop)() = nullptr; };
extern Roctx g_roctx;
void roctx_bind();
struct ApiRange {
  bool on;
  explicit ApiRange(const char *name) : on(false) { if (g_roctx.state == 0) roctx_bind(); if (g_roctx.state == 2) { g_roctx.push(name); on = true; } }
  ~ApiRange() { if (on) g_roctx.pop(); }
};
template <class F> int guarded(F body, const char *who = __builtin_FUNCTION()) {
  ApiRange range(who);
  try { return body(); }
  catch (const std::bad_alloc &) { return fail(MI355_EOOM, "host allocation failed"); }
  catch (const std::exception &e) { return fail(MI355_EHIP, std::string("unexpected host exception: ") + e.what()); }
  catch (...) { return fail(MI355_EHIP, "unexpected host exception"); }
}

// ---- RCCL, bound lazily (lib_core.hip)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
extern Rccl g_rccl;
int rccl_fail(const char *what, int rc);

// ---- SRS bookkeeping (lib_core.hip; no kernels)
std::vector<Shard> plan_shards(uint64_t n);
int srs_find(uint64_t handle, Srs **out, const char *who);
uint64_t srs_insert(const Srs &s);
int srs_alloc(SrsMem &mem, uint64_t n);
int srs_scatter_from_primary(SrsMem &mem, const g1_affine_t *src_dev, uint64_t n, bool alias_shard0);
int srs_gather_to_primary(const Srs &sr, uint64_t n, const g1_affine_t **out);

// ---- resident buffers (mi355_buf_*, lib_core.hip): which device slot owns a device pointer.  Pointers handed out by mi355_buf_alloc are
// looked up in the registry; anything else (a torch tensor, an SRS pointer) is asked of the HIP runtime when several devices are bound.
// `touch` marks a registered block as used by queued work (an upload into it must then wait for the compute stream).
int slot_of(const void *dev_ptr, bool touch = true);
int buf_check_range(const void *dev_ptr, uint64_t bytes, const char *who);   // MI355_EBADARG when [dev_ptr, dev_ptr + bytes) leaves the library block that holds dev_ptr
// the slot that owns ALL of the given device pointers (null pointers ignored); MI355_EBADARG when they live on different devices
int common_slot(std::initializer_list<const void *> ptrs, int *slot_out, const char *who);
// round-robin choice of a device for a host-pointer call (replicas: any bound device can run it); prefers a device whose lock is free
int pick_replica_slot();

// ---- per-translation-unit hooks called by init_ctx (dynamic-LDS limits of the kernels that TU launches)
int msm_tu_init_device();
int ntt_tu_init_device();
int aux_tu_init_device();
// twiddle table out[i] = (base^step)^i on the current context's stream (kernel in ntt.hpp; used by the G1 DFT as well)
int launch_pow_table(fe_t *out, const fe_t &base, uint64_t step, uint32_t count);
// window-cost model shared by the MSM launch code and mi355_srs_precompute
constexpr int MSM_SCALAR_BITS = 255, MSM_MAX_C = 24;   // hard limit of the sorter (23 key bits); the automatic choices stop at g_auto_max_c
double msm_cost(uint64_t n, int c, bool shared);

// ---- small helpers shared by the transform entry points (lib_ntt.hip, lib_aux.hip)
inline int check_ntt_args(const void *data, uint32_t log_n, const void *omega) {
  if (!data || !omega) return fail(MI355_EBADARG, "ntt: null pointer");
  if (log_n > 28) return fail(MI355_EBADARG, "ntt: log_n > 28");
  return MI355_OK;
}
struct NttHostArgs { uint32_t log_n; const void *omega; const void *divisor; };
// host buffer -> this context's staging buffer `role` -> body(dev) -> host buffer, synchronous (the device lock is held by the caller)
inline int with_host_io(void *data_host, size_t in_bytes, size_t out_bytes, size_t dev_bytes, const char *role, int (*body)(void *dev, void *ud), void *ud) {
  void *dev; CHK(ws_get(role, dev_bytes, &dev));
  HIPCHK(hipMemcpyAsync(dev, data_host, in_bytes, hipMemcpyHostToDevice, g.stream));
  CHK(body(dev, ud));
  HIPCHK(hipMemcpyAsync(data_host, dev, out_bytes, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  resolve_spans();
  return MI355_OK;
}

}  // namespace mi355
