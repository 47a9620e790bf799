// lib_core.hip -- libmi355zk.so, host state and the part of the C-ABI (include/mi355zk.h) that launches no kernel: lifecycle
// (mi355_init / _init_multi / _shutdown, streams), the per-device locks, SRS handle bookkeeping (register / prefix / release / read-back),
// the resident-buffer API (mi355_buf_*), the RCCL binding, HIP-event profiling.  There is deliberately no CPU fallback: without a gfx950
// device every compute entry point returns MI355_ENODEVICE.
#include <dlfcn.h>

#include "lib_common.hpp"
#include "slab_ranges.hpp"

namespace mi355 {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

std::mutex g_ctx_mu[MAX_DEV];
std::mutex g_upload_mu[MAX_DEV];   // mi355_buf_upload's bookkeeping (event ring, fork event): uploads never take the device lock
Ctx g_ctx[MAX_DEV];
int g_ndev = 0;
bool g_dup_devices = false;
uint32_t g_shard_min_log = 14;
bool g_force_exchange = false;
bool g_peer_ok[MAX_DEV][MAX_DEV] = {};   // [s][t]: kernels on slot s may dereference memory of slot t (same device, or hipDeviceEnablePeerAccess succeeded)
thread_local Ctx *g_cur = &g_ctx[0];
// The handle table is heap-allocated and never destroyed: the destructors of its entries (SrsMem / SrsTables) call hipFree, and a process
// that exits with bases still registered -- the Rust shim's static params_map never calls mi355_shutdown -- must not run HIP calls from
// static destruction, after the HIP runtime's own exit handlers.  Device memory is only ever freed by mi355_srs_release / mi355_shutdown.
std::unordered_map<uint64_t, Srs> &g_srs = *new std::unordered_map<uint64_t, Srs>();
uint64_t g_next_handle = 1;
int g_last_devices = 1; const char *g_last_exchange = "none";
int g_auto_max_c = 22;
thread_local MsmOpts t_opts;
Rccl g_rccl;

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

int bind_ctx(int slot) {
  use_ctx(slot);
  if (hipSetDevice(g.device) != hipSuccess) { (void)hipGetLastError(); return fail(MI355_EHIP, "hipSetDevice failed on the calling thread"); }
  return MI355_OK;
}
int need_init(int slot) {
  if (slot < 0 || slot >= MAX_DEV) return fail(MI355_EBADARG, "device slot out of range");
  use_ctx(slot);
  if (g_ndev == 0 || slot >= g_ndev || !g.inited) { use_ctx(0); return fail(MI355_ENODEVICE, "mi355_init() has not succeeded: no gfx950 device bound (there is no CPU fallback)"); }
  return bind_ctx(slot);
}

static size_t pool_release_slot(int slot);
// hipMalloc on the current context's device.  HBM may be full of blocks that only sit in the library's own buffer pool (mi355_buf_free keeps
// them for reuse): before reporting MI355_EOOM the pooled blocks of THIS device go back to HIP and the allocation is tried once more
// (ws_get, the twiddle tables and mi355_srs_precompute used to fail while gigabytes of free pooled blocks sat idle).
int dev_malloc(void **out, size_t bytes, const char *what) {
  *out = nullptr;
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    if (pool_release_slot(g.slot) > 0) e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError(); *out = nullptr;
    char b[256]; snprintf(b, sizeof b, "%s: hipMalloc of %zu bytes failed: %s", what, bytes, hipGetErrorString(e));
    return fail(e == hipErrorOutOfMemory ? MI355_EOOM : MI355_EHIP, b);
  }
  return MI355_OK;
}

int ws_get(const char *role, size_t bytes, void **out) {
  Buf &b = g.ws[role];
  if (b.cap < bytes) {
    if (b.p) {   // every stream that may still touch the old block: compute, the two auxiliary streams, and the copy stream (host-pointer entry points stage through workspace blocks)
      HIPCHK(hipStreamSynchronize(g.stream)); for (int i = 0; i < 2; i++) if (g.aux_stream[i]) HIPCHK(hipStreamSynchronize(g.aux_stream[i]));
      if (g.copy_stream) HIPCHK(hipStreamSynchronize(g.copy_stream));
      HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0;
    }
    size_t cap = bytes + bytes / 8 + 256;
    // the head-room is a convenience (fewer regrowths), not a requirement: near the HBM limit the exact size is tried as well
    if (dev_malloc(&b.p, cap, role) != MI355_OK) { cap = bytes; CHK(dev_malloc(&b.p, cap, role)); }
    b.cap = cap;
  }
  *out = b.p; return MI355_OK;
}

void resolve_spans() {
  if (g.spans.empty()) return;
  (void)hipStreamSynchronize(g.stream);
  for (int i = 0; i < 2; i++) if (g.aux_stream[i]) (void)hipStreamSynchronize(g.aux_stream[i]);
  for (auto &s : g.spans) { float ms = 0; (void)hipEventElapsedTime(&ms, s.a, s.b); Prof &p = g.prof[s.name]; p.ms += ms; p.launches++; (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  g.spans.clear();
}

Roctx g_roctx;
void roctx_bind() {   // state 1: off (default, or the library is absent), 2: ranges on
  static std::mutex mu; std::lock_guard<std::mutex> lk(mu);
  if (g_roctx.state != 0) return;
  const char *e = getenv("MI355_TRACE");
  if (!(e && e[0] == '2')) { g_roctx.state = 1; return; }
  void *h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
  if (h) { g_roctx.push = (int (*)(const char *))dlsym(h, "roctxRangePushA"); g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop"); }
  if (!h || !g_roctx.push || !g_roctx.pop) { fprintf(stderr, "[mi355zk] MI355_TRACE=2: no roctx library found, ranges stay off\n"); g_roctx.state = 1; return; }
  g_roctx.state = 2;
}
int rccl_fail(const char *what, int rc) { return fail(MI355_ERCCL, std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?")); }
// librccl.so.1 is dlopen()ed by mi355_init_multi only when it needs a communicator, so single-device users (and the CPU-only symbol checks)
// carry no RCCL dependency; a process that imported torch first gets torch's copy (same SONAME), as with libamdhip64.
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

// ------------------------------------------------------------------------------------------------ resident buffers
// A block handed out by mi355_buf_alloc.  free_ev: recorded on the owner's compute stream when the block was last returned to the pool --
// the only work an upload into the recycled block has to wait for.  used: the block has been passed to a library call since it was
// allocated (an upload into it must then wait for the whole compute stream).
struct BufBlock { void *p = nullptr; size_t bytes = 0; int slot = 0; hipEvent_t free_ev = nullptr; bool used = false; bool arena = false; };
static std::mutex g_buf_mu;
static std::map<uintptr_t, BufBlock> &g_bufs = *new std::map<uintptr_t, BufBlock>();                         // live blocks by base address
static std::multimap<std::pair<int, size_t>, BufBlock> &g_pool = *new std::multimap<std::pair<int, size_t>, BufBlock>();   // free blocks by (slot, size)
static std::atomic<uint32_t> g_rr{0};
// Slabs (round 5).  The pool above keeps freed blocks by EXACT size: right for one layer's proofs in a row (the second proof allocates nothing), wrong for a prover process that
// holds several layers -- a chunk prover runs 2^20-, 2^24- and 2^25-row proofs back to back [REF integration/src/prove.rs:30-43], every layer's blocks have another size, the working
// sets do not fit HBM together, and each proof used to give the previous layer's blocks back to HIP and hipMalloc its own (measured: 8.4 s per chunk round against 4.3 s for the three
// layers alone, profiles/r05_prover_process_*.json).  Now blocks are CARVED out of slabs (hipMalloc'd once, >= MI355_BUF_SLAB_MB each, default 1 GiB; a larger request gets a slab of its
// own size).  Order on a pool miss: carve from the free ranges; else move this device's pooled blocks into the free ranges (adjacent ranges of one slab coalesce) and carve; else a new
// slab.  A range only enters the free list after the last use of the block it came from has COMPLETED (its free event is synchronised first -- the events of a previous layer's proof are
// long done), so a carved block is fresh: an upload into it waits for nothing, exactly like a block straight from hipMalloc.  Slabs go back to HIP when they are entirely free and
// somebody needs the memory (mi355_buf_trim, an out-of-memory retry, shutdown).  MI355_BUF_ARENA=0 restores one hipMalloc per block.
static mi355zk::SlabRanges *g_arena = new mi355zk::SlabRanges[MAX_DEV];   // per device slot: its slabs and their quiescent free ranges (csrc/slab_ranges.hpp, model-checked on the host)
static bool arena_on() { static const bool on = [] { const char *e = getenv("MI355_BUF_ARENA"); return !(e && e[0] == '0'); }(); return on; }
static size_t slab_min_bytes() { static const size_t v = [] { const char *e = getenv("MI355_BUF_SLAB_MB"); const long mb = e ? atol(e) : 1024; return (size_t)std::max<long>(1, mb) << 20; }(); return v; }
// this device's pooled blocks -> free ranges (the calling thread is bound to the device: events are synchronised outside the registry mutex).  The copy stream is drained as well: a
// block's free event only covers the compute stream, and an upload that was queued into a block and never consumed before its mi355_buf_free must not land in whatever is carved there next
static void arena_recycle_pool(int slot) {
  std::vector<BufBlock> take;
  { std::lock_guard<std::mutex> bl(g_buf_mu); for (auto it = g_pool.begin(); it != g_pool.end();) { if (it->first.first == slot && it->second.arena) { take.push_back(it->second); it = g_pool.erase(it); } else ++it; } }
  if (take.empty()) return;
  for (auto &b : take) if (b.free_ev) { (void)hipEventSynchronize(b.free_ev); (void)hipEventDestroy(b.free_ev); }
  if (g_ctx[slot].copy_stream) (void)hipStreamSynchronize(g_ctx[slot].copy_stream);
  std::lock_guard<std::mutex> bl(g_buf_mu);
  for (auto &b : take) g_arena[slot].insert((uintptr_t)b.p, b.bytes);
}
// slabs of this device that are entirely free go back to HIP; returns the bytes freed
static size_t arena_release_free_slabs(int slot) {
  std::vector<uintptr_t> drop; size_t freed = 0;
  { std::lock_guard<std::mutex> bl(g_buf_mu); freed = g_arena[slot].take_whole_slabs(drop); }
  for (uintptr_t q : drop) (void)hipFree((void *)q);
  return freed;
}

static BufBlock *buf_find_locked(const void *p) {
  auto it = g_bufs.upper_bound((uintptr_t)p);
  if (it == g_bufs.begin()) return nullptr;
  --it;
  if ((uintptr_t)p < it->first + it->second.bytes) return &it->second;
  return nullptr;
}
// a write of `bytes` starting at dev_ptr must stay inside the library block that holds dev_ptr (blocks are carved next to each other inside slabs: an overrun lands
// in a neighbouring LIVE polynomial, not in a fault).  Pointers the library did not hand out (torch tensors, caller allocations) cannot be checked and pass.
int buf_check_range(const void *dev_ptr, uint64_t bytes, const char *who) {
  std::lock_guard<std::mutex> lk(g_buf_mu);
  if (BufBlock *b = buf_find_locked(dev_ptr)) {
    const uint64_t off = (uint64_t)((uintptr_t)dev_ptr - (uintptr_t)b->p);
    if (bytes > b->bytes - off) return fail(MI355_EBADARG, std::string(who) + ": range exceeds the block");
  }
  return MI355_OK;
}
int slot_of(const void *dev_ptr, bool touch) {
  if (!dev_ptr) return 0;
  {
    std::lock_guard<std::mutex> lk(g_buf_mu);
    if (BufBlock *b = buf_find_locked(dev_ptr)) { if (touch) b->used = true; return b->slot; }
  }
  if (g_ndev <= 1) return 0;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, dev_ptr) == hipSuccess) { for (int s = 0; s < g_ndev; s++) if (g_ctx[s].device == a.device) return s; }
  else (void)hipGetLastError();
  return 0;
}
int common_slot(std::initializer_list<const void *> ptrs, int *slot_out, const char *who) {
  int slot = -1;
  for (const void *p : ptrs) {
    if (!p) continue;
    const int s = slot_of(p);
    if (slot < 0) slot = s;
    else if (s != slot && g_ctx[s].device != g_ctx[slot].device) return fail(MI355_EBADARG, std::string(who) + ": the operands live on different devices (copy one with mi355_buf_copy first)");
  }
  *slot_out = slot < 0 ? 0 : slot;
  return MI355_OK;
}
int pick_replica_slot() {
  const int D = g_ndev;
  if (D <= 1) return 0;
  const int start = (int)(g_rr.fetch_add(1) % (uint32_t)D);
  for (int i = 0; i < D; i++) { const int s = (start + i) % D; if (g_ctx_mu[s].try_lock()) { g_ctx_mu[s].unlock(); return s; } }
  return start;
}
// give the pooled (free) blocks of one device slot back to HIP.  Returns bytes freed.  Callers: dev_malloc on an out-of-memory retry -- reached WITH the slot's lock from the
// compute entry points and WITHOUT it from mi355_buf_alloc -- and mi355_buf_trim.  The invariant that makes both safe: this function touches only state under g_buf_mu
// (the pool map) plus thread-safe HIP calls (stream synchronisation, hipFree) on the calling thread's bound device; it must never read or write Ctx members that the
// slot's lock protects (g.ws, the event rings, the plans).
static size_t pool_release_slot(int slot) {
  std::vector<BufBlock> drop; bool any_arena = false;
  { std::lock_guard<std::mutex> bl(g_buf_mu); for (auto it = g_pool.begin(); it != g_pool.end();) { if (it->first.first == slot && !it->second.arena) { drop.push_back(it->second); it = g_pool.erase(it); } else { any_arena = any_arena || (it->first.first == slot); ++it; } }
    any_arena = any_arena || !g_arena[slot].free_ranges.empty(); }
  if (drop.empty() && !any_arena) return 0;
  (void)hipStreamSynchronize(g_ctx[slot].stream);   // work queued on a block before its mi355_buf_free
  if (g_ctx[slot].copy_stream) (void)hipStreamSynchronize(g_ctx[slot].copy_stream);
  size_t freed = 0;
  for (auto &b : drop) { if (b.free_ev) (void)hipEventDestroy(b.free_ev); (void)hipFree(b.p); freed += b.bytes; }
  arena_recycle_pool(slot);                         // carved blocks return to their slabs; slabs that are whole again go back to HIP
  freed += arena_release_free_slabs(slot);
  return freed;
}
static void buf_release_all_locked() {   // shutdown: every slot's lock is held, the devices are still bound
  auto drop = [](BufBlock &b) {
    if (b.slot < g_ndev && g_ctx[b.slot].inited) (void)hipSetDevice(g_ctx[b.slot].device);
    if (b.free_ev) (void)hipEventDestroy(b.free_ev);
    if (b.p && !b.arena) (void)hipFree(b.p);        // carved blocks are freed with their slabs below
  };
  for (auto &kv : g_bufs) drop(kv.second);
  for (auto &kv : g_pool) drop(kv.second);
  for (int d = 0; d < MAX_DEV; d++) {
    if (!g_arena[d].slabs.empty() && d < g_ndev && g_ctx[d].inited) (void)hipSetDevice(g_ctx[d].device);
    for (auto &sl : g_arena[d].slabs) (void)hipFree((void *)sl.base);
    g_arena[d].clear();
  }
  g_bufs.clear(); g_pool.clear();
}

// ------------------------------------------------------------------------------------------------ contexts
// everything mi355_init does for ONE device slot (the calling thread ends up bound to that device)
static int init_ctx(int slot, int device_id) {
  use_ctx(slot);
  { const bool keep_profiling = g.profiling; g = Ctx(); g.profiling = keep_profiling; }   // mi355_profile_enable before mi355_init stays in force
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
  HIPCHK(hipEventCreateWithFlags(&g.ev_xchg2, hipEventDisableTiming));
  HIPCHK(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreateWithFlags(&g.ev_copy[i], hipEventDisableTiming));
  for (int i = 0; i < 8; i++) HIPCHK(hipEventCreateWithFlags(&g.ev_up[i], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g.ev_up_fork, hipEventDisableTiming));
  { const char *e = getenv("MI355_MSM_CHUNKS"); if (e) { int v = atoi(e); if (v >= 1 && v <= 16) g.msm_chunks = (uint32_t)v; } }
  // dynamic-LDS limits are per device and per kernel: each translation unit sets the ones of the kernels it launches
  CHK(msm_tu_init_device());
  CHK(ntt_tu_init_device());
  CHK(aux_tu_init_device());
  { const char *e = getenv("MI355_TRACE"); g.trace = e && e[0] == '1'; }
#ifdef MI355_DEBUG_KNOBS
  { const char *e = getenv("MI355_DEBUG_GATHER_MASK"); if (e) g.debug_gather_mask = (uint32_t)strtoul(e, nullptr, 0); }
#endif
  { const char *e = getenv("MI355_REDUCE_CHAINS"); if (e) { int v = atoi(e); if (v >= 1024) g.reduce_chains = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_FILL"); if (e) { int v = atoi(e); if (v >= 2 && v <= 100) g.seg_fill = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_FILL_SEGFIX"); if (e) { int v = atoi(e); if (v >= 2 && v <= 100) g.seg_fill_segfix = (uint32_t)v; } }
  { const char *e = getenv("MI355_SEG_MIN"); if (e) { int v = atoi(e); if (v >= 1 && v <= 4096) g.seg_min = (uint32_t)v; } }
  { const char *e = getenv("MI355_FIXUP_MODE"); if (e && e[0] >= '0' && e[0] <= '2') g.fixup_mode = (uint32_t)(e[0] - '0'); }
  { const char *e = getenv("MI355_FIXUP_HUGE_MIN"); if (e) { long v = atol(e); if (v >= 2048 && v <= 0x7fffffffL) g.fixup_huge_min = (uint32_t)v; } }
  { const char *e = getenv("MI355_FIXUP_SERIAL_MAX"); if (e) { int v = atoi(e); if (v >= 1 && v <= 1024) g.fixup_serial_max = (uint32_t)v; } }
  { const char *e = getenv("MI355_FIXUP_LANES_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 31) g.fixup_lanes_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_TAIL_COOP_MASK"); if (e) g.tail_coop_mask = (uint32_t)atoi(e) & 15u; }
  { const char *e = getenv("MI355_TAIL_COOP_MAX"); if (e) { long v = atol(e); if (v >= 0 && v <= (1l << 24)) g.tail_coop_max = (uint32_t)v; } }
  { const char *e = getenv("MI355_REDUCE_MIN_CHUNK"); if (e) { int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) g.reduce_min_chunk = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_FB"); if (e) { int v = atoi(e); if (v >= 9 && v <= 12) g.sort_fb = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_SPLIT"); if (e && e[0] >= '1' && e[0] <= '2') g.sort_split = (uint32_t)(e[0] - '0'); }
  { const char *e = getenv("MI355_SEG_FACTOR"); if (e) { int v = atoi(e); if (v >= 1 && v <= 256) g.seg_factor = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_T2"); if (e) { int v = atoi(e); if (v == 8192 || v == 16384) g.sort_t2 = (uint32_t)v; } }
  { const char *e = getenv("MI355_SORT_T1"); if (e && atoi(e) == 8192) g.sort_t1 = 8192; }
  { const char *e = getenv("MI355_NTT_DIRECT2_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 28) g.ntt_direct2_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_DIRECT2_MIN_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 28) g.ntt_direct2_min_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_FOLD_SCALE"); if (e) g.ntt_fold_scale = e[0] == '0' ? 0u : 1u; }
  { const char *e = getenv("MI355_NTT_COSET_FOLD_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 28) g.ntt_coset_fold_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_BATCH_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 0 && v <= 28) g.ntt_batch_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_TWO_LEVEL_MAX_LOG"); if (e) { int v = atoi(e); if (v >= 9 && v <= 20) g.ntt_two_level_max_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_NTT_TILE_LOG"); if (e) { int v = atoi(e); if (v >= 8 && v <= 12) g.ntt_tile_log = (uint32_t)v; } }
  { const char *e = getenv("MI355_HOST_BATCH_OVERLAP"); if (e) g.host_batch_overlap = atoi(e) != 0; }
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
  if (g.copy_stream) (void)hipStreamSynchronize(g.copy_stream);
  for (auto &kv : g.ws) if (kv.second.p) (void)hipFree(kv.second.p);
  g.ws.clear();
  for (auto &kv : g.ntt_plans) for (void *q : kv.second.owned) (void)hipFree(q);
  g.ntt_plans.clear();
  if (g.fixed_base_table) { (void)hipFree(g.fixed_base_table); g.fixed_base_table = nullptr; }
  if (g.comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(g.comm); g.comm = nullptr; }
  if (g.own_stream) (void)hipStreamDestroy(g.own_stream);
  for (int i = 0; i < 2; i++) {
    if (g.aux_stream[i]) { (void)hipStreamDestroy(g.aux_stream[i]); g.aux_stream[i] = nullptr; }
    if (g.msm_slot[i].sorted) { (void)hipEventDestroy(g.msm_slot[i].sorted); (void)hipEventDestroy(g.msm_slot[i].acc_done); (void)hipEventDestroy(g.msm_slot[i].red_done); g.msm_slot[i] = MsmSlot(); }
  }
  if (g.ev_fork) { (void)hipEventDestroy(g.ev_fork); g.ev_fork = nullptr; }
  if (g.ev_xchg) { (void)hipEventDestroy(g.ev_xchg); g.ev_xchg = nullptr; }
  if (g.ev_xchg2) { (void)hipEventDestroy(g.ev_xchg2); g.ev_xchg2 = nullptr; }
  for (int i = 0; i < 8; i++) if (g.ev_up[i]) { (void)hipEventDestroy(g.ev_up[i]); g.ev_up[i] = nullptr; }
  if (g.ev_up_fork) { (void)hipEventDestroy(g.ev_up_fork); g.ev_up_fork = nullptr; }
  for (int i = 0; i < 4; i++) if (g.ev_copy[i]) { (void)hipEventDestroy(g.ev_copy[i]); g.ev_copy[i] = nullptr; }
  if (g.copy_stream) { (void)hipStreamDestroy(g.copy_stream); g.copy_stream = nullptr; }
  g.own_stream = g.stream = nullptr; g.inited = false; g.device = -1;
}
static void shutdown_all() {
  for (int s = 0; s < g_ndev; s++) if (g_ctx[s].inited) { (void)hipSetDevice(g_ctx[s].device); (void)hipStreamSynchronize(g_ctx[s].stream); }
  { std::lock_guard<std::mutex> lk(g_buf_mu); buf_release_all_locked(); }
  for (auto &kv : g_srs) { kv.second.tab.reset(); kv.second.mem.reset(); }   // the destructors bind each shard's device and free
  g_srs.clear();
  for (int s = g_ndev - 1; s >= 0; s--) destroy_ctx(s);
  g_ndev = 0; g_dup_devices = false; g_force_exchange = false;
  use_ctx(0);
}

// ------------------------------------------------------------------------------------------------ SRS bookkeeping
// shard plan of a basis of n points: D equal point ranges, or everything on the primary device when the basis is too small to be worth
// spreading (a 2^14-point shard is already latency-bound)
std::vector<Shard> plan_shards(uint64_t n) {
  std::vector<Shard> v;
  int D = g_ndev;
  if (D > 1 && n / (uint64_t)D < (1ull << g_shard_min_log)) D = 1;
  for (int d = 0; d < D; d++) { Shard s; s.slot = d; s.lo = n * d / D; s.n = n * (d + 1) / D - s.lo; if (s.n) v.push_back(s); }
  return v;
}
int srs_find(uint64_t handle, Srs **out, const char *who) {
  auto it = g_srs.find(handle);
  if (it == g_srs.end() || !it->second.mem) return fail(MI355_EBADARG, std::string(who) + ": unknown SRS handle");
  *out = &it->second; return MI355_OK;
}
uint64_t srs_insert(const Srs &s) { const uint64_t h = g_next_handle++; g_srs[h] = s; return h; }
// allocate the shards of `mem` on their devices (leaves the primary bound)
int srs_alloc(SrsMem &mem, uint64_t n) {
  mem.sh = plan_shards(n);
  for (auto &sh : mem.sh) {
    CHK(bind_ctx(sh.slot));
    CHK(dev_malloc((void **)&sh.dev, sh.n * sizeof(g1_affine_t), "srs shard")); sh.owned = true;
  }
  return bind_ctx(0);
}
// copy a basis that sits contiguously on the primary device into freshly allocated shards
int srs_scatter_from_primary(SrsMem &mem, const g1_affine_t *src_dev, uint64_t n, bool alias_shard0) {
  mem.sh = plan_shards(n);
  for (auto &sh : mem.sh) {
    if (sh.slot == 0 && alias_shard0) { sh.dev = const_cast<g1_affine_t *>(src_dev) + sh.lo; sh.owned = false; continue; }
    CHK(bind_ctx(sh.slot));
    CHK(dev_malloc((void **)&sh.dev, sh.n * sizeof(g1_affine_t), "srs shard")); sh.owned = true;
    if (sh.slot == 0 || g_ctx[sh.slot].device == g_ctx[0].device) HIPCHK(hipMemcpyAsync(sh.dev, src_dev + sh.lo, sh.n * sizeof(g1_affine_t), hipMemcpyDeviceToDevice, g.stream));
    else HIPCHK(hipMemcpyPeerAsync(sh.dev, g.device, src_dev + sh.lo, g_ctx[0].device, sh.n * sizeof(g1_affine_t), g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  return bind_ctx(0);
}
// the first n points of a registered basis as ONE contiguous array on the primary device: the primary shard itself when it covers them,
// otherwise a temporary assembled from the shards (xGMI peer copies)
int srs_gather_to_primary(const Srs &sr, uint64_t n, const g1_affine_t **out) {
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

}  // namespace mi355

using namespace mi355;

// ================================================================================================ C ABI
extern "C" {

const char *mi355_last_error(void) { return g_err.c_str(); }
const char *mi355_version(void) { return "mi355zk 0.3.0 (gfx950; BN254 G1 MSM + Fr NTT)"; }

int mi355_init_multi(const int *device_ids, int n_devices) {
  return guarded([&]() -> int {
  AllGuard lk;
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
  for (int s = 0; s < n_devices && rc == MI355_OK; s++) { rc = init_ctx(s, device_ids[s]); g_ndev = s + 1; }
  if (rc != MI355_OK) { const std::string keep = g_err; shutdown_all(); g_err = keep; return rc; }
  g_dup_devices = dup;
  { const char *e = getenv("MI355_MULTI_FORCE"); g_force_exchange = e && e[0] == '1'; }
  { const char *e = getenv("MI355_MSM_AUTO_MAX_C"); g_auto_max_c = 22; if (e) { int v = atoi(e); if (v >= 16 && v <= MSM_MAX_C) g_auto_max_c = v; } }
  { const char *e = getenv("MI355_SHARD_MIN_LOG"); g_shard_min_log = 14; if (e) { int v = atoi(e); if (v >= 0 && v <= 30) g_shard_min_log = (uint32_t)v; } }
  for (int s = 0; s < MAX_DEV; s++) for (int t = 0; t < MAX_DEV; t++) g_peer_ok[s][t] = s == t;
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
      for (int t = 0; t < n_devices; t++) {
        g_peer_ok[s][t] = dup || t == s;
        if (dup || t == s) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g.device, g_ctx[t].device) == hipSuccess && can) {
          const hipError_t pe = hipDeviceEnablePeerAccess(g_ctx[t].device, 0);
          if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) g_peer_ok[s][t] = true;
          if (pe != hipSuccess) (void)hipGetLastError();
        }
      }
    }
    if (rc != MI355_OK) { const std::string keep = g_err; shutdown_all(); g_err = keep; return rc; }
  }
  use_ctx(0);
  (void)hipSetDevice(g.device);
  return MI355_OK;
  });
}
int mi355_init(int device_id) { return mi355_init_multi(&device_id, 1); }
int mi355_device_count(int *n_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  if (!n_out) return fail(MI355_EBADARG, "device_count: null pointer");
  *n_out = g_ndev; return MI355_OK;
  });
}

int mi355_shutdown(void) {
  return guarded([&]() -> int {
  AllGuard lk;
  if (!g_ndev) return MI355_OK;
  shutdown_all();
  return MI355_OK;
  });
}

int mi355_set_stream(void *hip_stream) {
  return guarded([&]() -> int {
  AllGuard lk;
  CHK(need_init());
  HIPCHK(hipStreamSynchronize(g.stream));
  g.stream = (hipStream_t)hip_stream;   // NULL = the HIP null (legacy default) stream, which is torch's default stream
  return MI355_OK;
  });
}
int mi355_reset_stream(void) {
  return guarded([&]() -> int {
  AllGuard lk;
  CHK(need_init());
  HIPCHK(hipStreamSynchronize(g.stream));
  g.stream = g.own_stream;
  return MI355_OK;
  });
}
int mi355_synchronize(void) {
  return guarded([&]() -> int {
  AllGuard lk;
  CHK(need_init());
  for (int s = g_ndev - 1; s >= 0; s--) { CHK(bind_ctx(s)); HIPCHK(hipStreamSynchronize(g.copy_stream)); HIPCHK(hipStreamSynchronize(g.stream)); resolve_spans(); }
  return MI355_OK;
  });
}

// ---- SRS handles (the entry points that launch kernels -- load_params_file, precompute, downsize -- live in lib_msm.hip / lib_aux.hip)
int mi355_srs_register_host(const void *bases, uint64_t n, uint64_t *handle_out) {
  return guarded([&]() -> int {
  AllGuard lk;
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
int mi355_srs_register_dev(const void *bases_dev, uint64_t n, int copy, uint64_t *handle_out) {
  return guarded([&]() -> int {
  AllGuard lk;
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
  AllGuard lk;
  Srs *p; CHK(srs_find(parent_handle, &p, "srs_register_prefix"));
  if (!handle_out || n == 0 || n > p->n) return fail(MI355_EBADARG, "srs_register_prefix: n must be in [1, len(parent)]");
  Srs s; s.n = n; s.mem = p->mem; s.tab = p->tab;
  *handle_out = srs_insert(s); return MI355_OK;
  });
}
int mi355_srs_release(uint64_t handle) {
  return guarded([&]() -> int {
  AllGuard lk;
  auto it = g_srs.find(handle);
  if (it == g_srs.end()) return fail(MI355_EBADARG, "srs_release: unknown handle");
  for (int s = 0; s < g_ndev; s++) if (g_ctx[s].inited) { (void)hipSetDevice(g_ctx[s].device); (void)hipStreamSynchronize(g_ctx[s].stream); }
  g_srs.erase(it);   // the last handle sharing the memory frees it (SrsMem::~SrsMem)
  if (g_ndev) (void)hipSetDevice(g_ctx[0].device);
  return MI355_OK;
  });
}
int mi355_srs_pre_dev_ptr(uint64_t handle, void **dev_ptr_out, int *c_out, int *windows_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  Srs *sp; CHK(srs_find(handle, &sp, "srs_pre_dev_ptr"));
  if (!dev_ptr_out) return fail(MI355_EBADARG, "srs_pre_dev_ptr: null pointer");
  *dev_ptr_out = sp->tab && !sp->tab->pre.empty() ? sp->tab->pre[0] : nullptr; if (c_out) *c_out = sp->tab ? sp->tab->c : 0; if (windows_out) *windows_out = sp->tab ? sp->tab->w : 0; return MI355_OK;
  });
}
int mi355_srs_len(uint64_t handle, uint64_t *n_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  Srs *sp; CHK(srs_find(handle, &sp, "srs_len"));
  if (!n_out) return fail(MI355_EBADARG, "srs_len: null pointer");
  *n_out = sp->n; return MI355_OK;
  });
}
int mi355_srs_dev_ptr(uint64_t handle, void **dev_ptr_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  Srs *sp; CHK(srs_find(handle, &sp, "srs_dev_ptr"));
  if (!dev_ptr_out) return fail(MI355_EBADARG, "srs_dev_ptr: null pointer");
  *dev_ptr_out = sp->mem->sh.empty() ? nullptr : sp->mem->sh[0].dev; return MI355_OK;   // primary shard (all points with one device)
  });
}
int mi355_srs_read_host(uint64_t handle, uint64_t offset, uint64_t n, void *out_affine_host) {
  return guarded([&]() -> int {
  AllGuard lk;
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

// ---- resident buffers: the memory a proof's polynomials live in between the calls of create_proof.  Every `*_dev` entry
// point accepts pointers into these blocks (and any other HIP device pointer of a bound device) and runs on the device that owns them.
int mi355_buf_alloc(uint64_t bytes, int device_slot, void **dev_ptr_out) {
  return guarded([&]() -> int {
  if (!dev_ptr_out || bytes == 0) return fail(MI355_EBADARG, "buf_alloc: null pointer or zero size");
  if (device_slot < 0 || device_slot >= MAX_DEV) return fail(MI355_EBADARG, "buf_alloc: device slot out of range");
  // No device lock (round 4): a pool hit is bookkeeping under the registry mutex, a miss is hipMalloc on the calling thread (thread-safe in HIP).
  // With the lock, a witness uploader's next allocation waited for whatever call held the device -- a 12 ms batched MSM, say -- and the many-column
  // layers uploaded at 34 GB/s instead of the link's 53 (kernel timeline in profiles/r04_kernel_vs_wall_L0_L3.md: 18.6 ms of idle device per batch).
  CHK(need_init(device_slot));
  const size_t want = ((size_t)bytes + 255) & ~(size_t)255;
  {
    std::lock_guard<std::mutex> bl(g_buf_mu);
    auto it = g_pool.find({device_slot, want});
    if (it != g_pool.end()) { BufBlock b = it->second; g_pool.erase(it); b.used = false; g_bufs[(uintptr_t)b.p] = b; *dev_ptr_out = b.p; return MI355_OK; }
  }
  void *p = nullptr;
  BufBlock b; b.bytes = want; b.slot = device_slot;
  if (arena_on()) {
    { std::lock_guard<std::mutex> bl(g_buf_mu); p = (void *)g_arena[device_slot].carve(want); }
    if (!p) { arena_recycle_pool(device_slot); std::lock_guard<std::mutex> bl(g_buf_mu); p = (void *)g_arena[device_slot].carve(want); }
    if (!p) {
      size_t sbytes = std::max(want, slab_min_bytes()); void *base = nullptr;
      // a new slab; near the HBM limit the head-room of a shared slab is given up and the request gets exactly its size (dev_malloc frees whole slabs and retries before failing)
      if (sbytes > want && dev_malloc(&base, sbytes, "buf_alloc (slab)") != MI355_OK) { base = nullptr; sbytes = want; }
      if (!base) CHK(dev_malloc(&base, sbytes, "buf_alloc"));
      std::lock_guard<std::mutex> bl(g_buf_mu);
      // a slab made for this one (large) request is dedicated: small blocks are never carved out of it, so it is whole again as soon as its polynomial comes back (slab_ranges.hpp)
      g_arena[device_slot].small_limit = slab_min_bytes() / 64;   // 16 MiB with the default 1 GiB slabs: staging blocks, pointer tables, short vectors
      g_arena[device_slot].add_slab((uintptr_t)base, sbytes, want >= slab_min_bytes());
      p = (void *)g_arena[device_slot].carve(want);
      if (!p) return fail(MI355_EHIP, "buf_alloc: slab bookkeeping");   // another thread took the new range: cannot happen with best fit on a range >= want, kept as a guard
    }
    b.arena = true;
  } else {
    CHK(dev_malloc(&p, want, "buf_alloc"));   // the pool of this device is given back to the allocator before giving up
  }
  b.p = p;
  { std::lock_guard<std::mutex> bl(g_buf_mu); g_bufs[(uintptr_t)p] = b; }
  *dev_ptr_out = p; return MI355_OK;
  });
}
// Returns the block to the library's pool (a later mi355_buf_alloc of the same size on the same device reuses it without a hipMalloc /
// hipFree, which would synchronise the device).  Work already queued on the block stays valid: reuse waits for it.
int mi355_buf_free(void *dev_ptr) {
  return guarded([&]() -> int {
  if (!dev_ptr) return MI355_OK;
  int slot;
  { std::lock_guard<std::mutex> bl(g_buf_mu); auto it = g_bufs.find((uintptr_t)dev_ptr); if (it == g_bufs.end()) return fail(MI355_EBADARG, "buf_free: not the base pointer of a live mi355_buf_alloc block"); slot = it->second.slot; }
  // no device lock either: the event below marks "everything queued on the owner's compute stream so far", which includes every use the freeing
  // thread issued before this call (HIP streams take work from several threads); nobody may use the block after its free
  CHK(need_init(slot));
  BufBlock b;
  { std::lock_guard<std::mutex> bl(g_buf_mu); auto it = g_bufs.find((uintptr_t)dev_ptr); if (it == g_bufs.end()) return fail(MI355_EBADARG, "buf_free: block freed twice"); b = it->second; g_bufs.erase(it); }
  if (!b.free_ev) HIPCHK(hipEventCreateWithFlags(&b.free_ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(b.free_ev, g_ctx[slot].stream));
  { std::lock_guard<std::mutex> bl(g_buf_mu); g_pool.insert({{b.slot, b.bytes}, b}); }
  return MI355_OK;
  });
}
// give every pooled (free) block back to HIP
int mi355_buf_trim(void) {
  return guarded([&]() -> int {
  AllGuard lk;
  if (!g_ndev) return MI355_OK;
  std::vector<BufBlock> drop;
  { std::lock_guard<std::mutex> bl(g_buf_mu); for (auto it = g_pool.begin(); it != g_pool.end();) { if (!it->second.arena) { drop.push_back(it->second); it = g_pool.erase(it); } else ++it; } }
  for (int s = 0; s < g_ndev; s++) { CHK(bind_ctx(s)); HIPCHK(hipStreamSynchronize(g.stream)); }
  for (auto &b : drop) { CHK(bind_ctx(b.slot)); if (b.free_ev) (void)hipEventDestroy(b.free_ev); (void)hipFree(b.p); }
  for (int s = 0; s < g_ndev; s++) { CHK(bind_ctx(s)); arena_recycle_pool(s); (void)arena_release_free_slabs(s); }   // carved blocks back to their slabs; whole slabs back to HIP
  return bind_ctx(0);
  });
}
int mi355_buf_slot(const void *dev_ptr, int *slot_out) {
  return guarded([&]() -> int {
  if (!dev_ptr || !slot_out) return fail(MI355_EBADARG, "buf_slot: null pointer");
  *slot_out = slot_of(dev_ptr, false); return MI355_OK;
  });
}
// Host -> device.  The copy runs on the owner device's COPY stream, so it overlaps whatever the compute stream is doing: an upload into a
// block that no library call has used since mi355_buf_alloc only waits for the work that was queued on the block before its last
// mi355_buf_free; an upload into a block in use waits for the compute stream.  Later library calls on the device see the data (the compute
// stream waits for the copy).  On return the host buffer may be reused.  The device lock is NOT held while the DMA runs (a pageable source
// blocks the calling thread for the length of the copy): another thread's commitments and transforms proceed on the device meanwhile --
// the witness of column i + 1 crosses PCIe while column i is being committed.
int mi355_buf_upload(void *dst_dev, const void *src_host, uint64_t bytes) {
  return guarded([&]() -> int {
  if (bytes == 0) return MI355_OK;
  if (!dst_dev || !src_host) return fail(MI355_EBADARG, "buf_upload: null pointer");
  const int slot = slot_of(dst_dev, false);
  // Round 4: the upload takes the device's UPLOAD mutex only, never the device lock.  An MSM over a batch of columns holds the device lock for tens
  // of milliseconds; with the lock also needed here, column i + 1 could not start (or finish) crossing PCIe until that MSM had returned, and the
  // many-column layers ran upload and commitment back to back (layer 0: 0.85 s for steps 2-3 = 0.55 s of DMA + 0.30 s of MSM).  What the lock
  // protected is replaced: the event ring and the fork event are this path's own (upload mutex), HIP streams accept work from several threads,
  // and the ordering is by events -- the copy waits for the work queued on the block, the compute stream waits for the copy.
  CHK(need_init(slot));
  Ctx &c = g_ctx[slot];
  hipEvent_t done = nullptr;
  {
    std::lock_guard<std::mutex> ul(g_upload_mu[slot]);
    bool fresh = false; hipEvent_t free_ev = nullptr;
    { std::lock_guard<std::mutex> bl(g_buf_mu); if (BufBlock *b = buf_find_locked(dst_dev)) { if ((uintptr_t)dst_dev + bytes > (uintptr_t)b->p + b->bytes) return fail(MI355_EBADARG, "buf_upload: range exceeds the block"); fresh = !b->used; free_ev = b->free_ev; b->used = true; } }
    if (fresh) { if (free_ev) HIPCHK(hipStreamWaitEvent(c.copy_stream, free_ev, 0)); }
    else { HIPCHK(hipEventRecord(c.ev_up_fork, c.stream)); HIPCHK(hipStreamWaitEvent(c.copy_stream, c.ev_up_fork, 0)); }   // everything queued on the compute stream so far
    done = c.ev_up[c.up_next++ & 7];
    HIPCHK(hipEventSynchronize(done));   // the ring slot's previous use (eight uploads ago) has long completed; an unrecorded event returns at once
  }
  HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c.copy_stream));   // pageable source: blocks this thread for the transfer
  {
    std::lock_guard<std::mutex> ul(g_upload_mu[slot]);
    HIPCHK(hipEventRecord(done, c.copy_stream));
  }
  // The call returns when the copy HAS COMPLETED, so whatever the caller queues on this block afterwards needs no ordering against it.  Until round 4 the
  // compute stream was made to wait for every upload's event: with columns streaming in on another thread, each kernel of the commitment in flight then
  // queued behind the uploads of columns it never reads -- upload and compute ran back to back instead of side by side (layer 0, page-locked witness:
  // 19 ms of DMA + 17 ms of kernels per 32 columns = 0.88 s for steps 2-3 instead of 0.5 s).
  HIPCHK(hipEventSynchronize(done));   // page-locked sources return from hipMemcpyAsync at once
  return MI355_OK;
  });
}
// Device -> host, ordered after everything queued on the owner device; synchronous.
int mi355_buf_download(void *dst_host, const void *src_dev, uint64_t bytes) {
  return guarded([&]() -> int {
  if (bytes == 0) return MI355_OK;
  if (!dst_host || !src_dev) return fail(MI355_EBADARG, "buf_download: null pointer");
  const int slot = slot_of(src_dev);
  DevGuard lk(slot);
  CHK(need_init(slot));
  HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return MI355_OK;
  });
}
// Device -> device, within a device or between two bound devices (xGMI peer copy); asynchronous on the destination's compute stream.
int mi355_buf_copy(void *dst_dev, const void *src_dev, uint64_t bytes) {
  return guarded([&]() -> int {
  if (bytes == 0) return MI355_OK;
  if (!dst_dev || !src_dev) return fail(MI355_EBADARG, "buf_copy: null pointer");
  const int sd = slot_of(dst_dev), ss = slot_of(src_dev);
  const int lo = std::min(sd, ss), hi = std::max(sd, ss);
  DevGuard l1(lo); std::unique_ptr<DevGuard> l2; if (hi != lo) l2.reset(new DevGuard(hi));
  if (ss != sd) { CHK(need_init(ss)); HIPCHK(hipStreamSynchronize(g.stream)); }   // the producer of src ran on another stream
  CHK(need_init(sd));
  if (g_ctx[sd].device == g_ctx[ss].device) HIPCHK(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, g.stream));
  else HIPCHK(hipMemcpyPeerAsync(dst_dev, g.device, src_dev, g_ctx[ss].device, bytes, g.stream));
  if (ss != sd) {
    // The copy is queued on the DESTINATION slot's stream only.  Whatever the source slot queues next on src -- a kernel that rewrites it, or the
    // free_ev of mi355_buf_free after which the pool may hand the block to a fresh upload -- must come after the copy has read it: the source
    // slot's compute stream waits for an event recorded behind the copy (both slots' locks are held here).  ADVICE r3 (medium).
    hipEvent_t ev = g.ev_xchg2;
    HIPCHK(hipEventRecord(ev, g.stream));
    HIPCHK(hipStreamWaitEvent(g_ctx[ss].stream, ev, 0));
  }
  return MI355_OK;
  });
}
int mi355_buf_zero(void *dst_dev, uint64_t bytes) {
  return guarded([&]() -> int {
  if (bytes == 0) return MI355_OK;
  if (!dst_dev) return fail(MI355_EBADARG, "buf_zero: null pointer");
  const int slot = slot_of(dst_dev);
  DevGuard lk(slot);
  CHK(need_init(slot));
  HIPCHK(hipMemsetAsync(dst_dev, 0, bytes, g.stream));
  return MI355_OK;
  });
}

int mi355_host_alloc(uint64_t bytes, void **host_ptr_out) {
  return guarded([&]() -> int {
  if (!host_ptr_out || bytes == 0) return fail(MI355_EBADARG, "host_alloc: null pointer or zero size");
  CHK(need_init(0));   // page-locking goes through the bound device's runtime; no device lock: nothing of the library's state is touched
  void *p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(e == hipErrorOutOfMemory ? MI355_EOOM : MI355_EHIP, std::string("host_alloc: hipHostMalloc failed: ") + hipGetErrorString(e)); }
  *host_ptr_out = p; return MI355_OK;
  });
}
int mi355_host_free(void *host_ptr) {
  return guarded([&]() -> int {
  if (!host_ptr) return MI355_OK;
  CHK(need_init(0));
  HIPCHK(hipHostFree(host_ptr));
  return MI355_OK;
  });
}
int mi355_mem_info(int device_slot, uint64_t *free_bytes, uint64_t *total_bytes, uint64_t *live_buf_bytes, uint64_t *pooled_bytes, uint64_t *workspace_bytes) {
  return guarded([&]() -> int {
  if (device_slot < 0 || device_slot >= MAX_DEV) return fail(MI355_EBADARG, "mem_info: device slot out of range");
  DevGuard lk(device_slot);
  CHK(need_init(device_slot));
  size_t fr = 0, tot = 0;
  HIPCHK(hipMemGetInfo(&fr, &tot));
  uint64_t live = 0, pooled = 0, ws = 0;
  { std::lock_guard<std::mutex> bl(g_buf_mu);
    for (const auto &kv : g_bufs) if (kv.second.slot == device_slot) live += kv.second.bytes;
    for (const auto &kv : g_pool) if (kv.first.first == device_slot) pooled += kv.second.bytes;
    pooled += g_arena[device_slot].free_bytes(); }   // free ranges of the slabs: held by the library, reusable, not live
  for (const auto &kv : g.ws) ws += kv.second.cap;
  if (free_bytes) *free_bytes = fr; if (total_bytes) *total_bytes = tot; if (live_buf_bytes) *live_buf_bytes = live;
  if (pooled_bytes) *pooled_bytes = pooled; if (workspace_bytes) *workspace_bytes = ws;
  return MI355_OK;
  });
}

// ---- test hook: read back a workspace buffer ("msm.sorted", "msm.offsets", ...) after a call
int mi355_debug_ws_read(const char *role, uint64_t offset, void *dst_host, uint64_t bytes) {
  return guarded([&]() -> int {
  DevGuard lk(0);
  CHK(need_init());
  auto it = g.ws.find(role ? role : "");
  if (it == g.ws.end() || !dst_host || offset + bytes > it->second.cap) return fail(MI355_EBADARG, "debug_ws_read: unknown role or range");
  HIPCHK(hipStreamSynchronize(g.stream));
  HIPCHK(hipMemcpy(dst_host, (const char *)it->second.p + offset, bytes, hipMemcpyDeviceToHost));
  return MI355_OK;
  });
}

// ---- profiling
int mi355_profile_enable(int on) { return guarded([&]() -> int { AllGuard lk; for (int s = 0; s < MAX_DEV; s++) g_ctx[s].profiling = on != 0; return MI355_OK; }); }
int mi355_profile_reset(void) { return guarded([&]() -> int { AllGuard lk; for (int s = g_ndev - 1; s >= 0; s--) { if (bind_ctx(s) == MI355_OK) resolve_spans(); g.prof.clear(); } use_ctx(0); return MI355_OK; }); }
int mi355_profile_get(const char *name, double *ms_out, uint64_t *launches_out) {
  return guarded([&]() -> int {
  DevGuard lk(0);
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
