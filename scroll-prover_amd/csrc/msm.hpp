// msm.hpp -- BN254 G1 multi-scalar multiplication (Pippenger / bucket method) for gfx950.
// Stands in for halo2_proofs::arithmetic::best_multiexp and the ParamsKZG::commit / commit_lagrange wrappers
// (SURVEY.md §8a a1/a2).  The result is a group element, so any correct schedule is bit-exact with the CPU
// path after normalisation; what differs from the CPU reference is the schedule:
//
//   0 k_srs_precompute  (optional, once per basis) T[w][i] = 2^(c w) P_i: all windows then share ONE bucket set
//   1 k_msm_digits      scalars (Montgomery) -> canonical -> signed c-bit digits d in [-2^(c-1), 2^(c-1)], digit plane [w][i]
//   2 k_sort_l1_* / k_sort_l2_*   two-level counting sort of (window, bucket) keys: LDS histograms + LDS-staged coalesced
//                       stores, one global atomic per (tile, non-empty bin); exclusive scans give the bucket offsets
//   3 k_msm_accumulate  SEGMENTED bucket accumulation: thread t owns entries [t*L, (t+1)*L) of the sorted list whatever
//                       bucket boundaries fall inside, keeps a 9x29-bit XYZZ accumulator in registers, gathers affine
//                       bases (64 B each) and does mixed additions.  Perfectly load-balanced for any scalar
//                       distribution (witness columns are mostly 0/1/small values).
//   4 k_msm_fixup(_big) buckets that straddle thread boundaries: sum their partials (one lane, or a workgroup for giant ones)
//   5 k_msm_bucket_reduce / k_msm_tree_sum29   sum_b (b+1) * B[b] by short chunked running sums, then multi-block
//                       wavefront-shuffle + LDS trees
//   6 k_msm_final29     Horner over windows (none with window tables), normalise to (x, y, 1) with a one-lane Euclidean inverse
//
// A batch of M polynomials over one basis (mi355_msm_g1_batch_*) runs the same kernels once with (polynomial m, window w) as
// window m * W + w: grid.y = m in the digits kernel, a bucket set per polynomial, M workgroups in k_msm_final29.  Steps 4-6 run on the 29-bit field as well (g1_xyzz29_add / _dbl).
//
// Algorithmic HBM bytes: 96 B per (scalar, point) pair.  The accumulation is VALU-integer bound
// (10 field multiplications = ~1650 v_mad_u64_u32 + ~700 other instructions per mixed addition), see DESIGN.md section 4.
#pragma once
#include "fp_asm.hpp"
#include "g1_29.hpp"
#include "frscan.hpp"

namespace zk {

#if defined(__HIPCC__)

struct MsmPlan {
  uint32_t n;        // pairs
  uint32_t c;        // window bits
  uint32_t windows;  // W = ceil(255 / c)
  uint32_t nb;       // buckets per window = 2^(c-1)
  uint32_t seg;      // entries per accumulate thread
  uint32_t batch;    // M polynomials committed in one pass (same basis); (polynomial m, window w) acts as window index m * W + w
};

__device__ __forceinline__ g1_affine_t load_affine(const g1_affine_t *p) {
  g1_affine_t r; r.x = g_load(&p->x); r.y = g_load(&p->y); return r;
}
// streaming variant for the bucket gathers: every 64-byte base of the (48 GiB) table is used once per MSM, so it should not
// displace the index / offset streams from the caches
__device__ __forceinline__ g1_affine_t load_affine_nt(const g1_affine_t *p) {
  const uint32_t *q = reinterpret_cast<const uint32_t *>(p); g1_affine_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) { r.x.l[i] = __builtin_nontemporal_load(q + i); r.y.l[i] = __builtin_nontemporal_load(q + 8 + i); }
  return r;
}
__device__ __forceinline__ g1_xyzz_t load_xyzz(const g1_xyzz_t *p) {
  g1_xyzz_t r; r.x = g_load(&p->x); r.y = g_load(&p->y); r.zz = g_load(&p->zz); r.zzz = g_load(&p->zzz); return r;
}
__device__ __forceinline__ void store_xyzz(g1_xyzz_t *p, const g1_xyzz_t &v) {
  g_store(&p->x, v.x); g_store(&p->y, v.y); g_store(&p->zz, v.zz); g_store(&p->zzz, v.zzz);
}

// raw 29-bit accumulator records (144 B = 9 x 16 B): what k_msm_accumulate flushes without any arithmetic
__device__ __forceinline__ void store_xyzz29(g1_xyzz29_t *p, const g1_xyzz29_t &v) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(&v); uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ g1_xyzz29_t load_xyzz29(const g1_xyzz29_t *p) {
  g1_xyzz29_t v; uint32_t *w = reinterpret_cast<uint32_t *>(&v); const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 9; i++) { const uint4 a = q[i]; w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w; }
  return v;
}

// ---- quad-cooperative point addition / doubling for the latency-bound reduction tail (fix-up of small bucket sets, running sums, trees,
// the final Horner): ONE addition spread over the 4 lanes of a quad.  All four lanes enter with the same operands and leave with the same
// result; in each of four rounds every lane multiplies a different pair (selected by its position in the quad) and the four products are
// broadcast inside the quad with DPP quad_perm moves (full-rate VALU, no LDS).  The dependent chain of an addition drops from 14 field
// multiplications to 4 (a doubling: 9 -> 3): ~1 250 instructions per wavefront instead of ~2 900, i.e. the latency of every step of the
// ~60-step serial chain of a small MSM's tail.  It costs 4x the lanes and ~1.7x the total instructions, so the launch code only uses it where
// the kernels are latency-bound (few logical threads).  Same formulas, bounds and exceptional cases as g1_xyzz29_add / _dbl (g1_29.hpp); the
// only difference is Y3 = A - B + 4p as two products instead of one fused reduction (value < 5.6 p, inside the accumulator invariant).
template <int J> __device__ __forceinline__ fe29_t quad_bcast(const fe29_t &v) {
  fe29_t r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int x = __builtin_amdgcn_mov_dpp((int)v.l[i], J * 0x55, 0xf, 0xf, true);   // quad_perm(J, J, J, J)
    // the result is pinned in a register: left to itself the compiler folds the DPP move into the consuming instruction, and for the
    // pattern (bcast<0>(m) + c) - bcast<1>(m) of Y3 it dropped the lane permutation of the subtrahend (every lane then subtracted its OWN
    // product; found on the device with a staged comparison against the plain addition, tools/_scratch in round 3)
    asm volatile("" : "+v"(x));
    r.l[i] = (uint32_t)x;
  }   // quad_perm(J, J, J, J)
  return r;
}
// operand of this lane's product: a_q.  Written as limb-wise selects on VALUES passed by value: with references and a nested ternary the
// compiler selects between the four ADDRESSES instead and keeps the operands in scratch memory (ScratchSize 508, checked with
// -Rpass-analysis=kernel-resource-usage; this form: 0)
__device__ __forceinline__ fe29_t quad_sel(uint32_t q, const fe29_t a0, const fe29_t a1, const fe29_t a2, const fe29_t a3) {
  fe29_t r;
#pragma unroll
  for (int i = 0; i < 9; i++) { const uint32_t x0 = a0.l[i], x1 = a1.l[i], x2 = a2.l[i], x3 = a3.l[i]; const uint32_t lo = q & 1u ? x1 : x0, hi = q & 1u ? x3 : x2; r.l[i] = q & 2u ? hi : lo; }
  return r;
}
__device__ __forceinline__ g1_xyzz29_t g1_xyzz29_dbl_q4(const g1_xyzz29_t &a) {
  if (g1_xyzz29_is_identity(a)) return a;
  const uint32_t q = threadIdx.x & 3u;
  const fe29_t xt = Fq29::reduce_small(Fq29::normalise(a.x));   // tight, < 2 p
  const fe29_t yc = Fq29::carry(a.y);                           // limbs <= 2^29 + 2, value < 6.1 p
  const fe29_t U = Fq29::dbl(yc);                               // limbs <= 2^30 + 4, < 12.2 p
  fe29_t m = FQ29_MUL(quad_sel(q, U, xt, U, xt), quad_sel(q, U, xt, U, xt));
  const fe29_t V = quad_bcast<0>(m), xx = quad_bcast<1>(m);
  const fe29_t M = Fq29::carry(Fq29::add(Fq29::dbl(xx), xx));   // 3 x^2, limbs <= 2^29 + 8, < 3.3 p
  m = FQ29_MUL(quad_sel(q, U, xt, M, V), quad_sel(q, V, V, M, a.zz));
  const fe29_t W = quad_bcast<0>(m), S = quad_bcast<1>(m), MM = quad_bcast<2>(m), ZZ3 = quad_bcast<3>(m);
  g1_xyzz29_t r;
  r.x = Fq29::sub8(MM, Fq29::dbl(S));                           // < 1.1 p + 8 p
  const fe29_t t = Fq29::sub16(S, r.x);                          // < 17.1 p
  m = FQ29_MUL(quad_sel(q, M, W, W, W), quad_sel(q, t, yc, a.zzz, a.zzz));
  r.y = Fq29::sub4(quad_bcast<0>(m), quad_bcast<1>(m));         // < 1.4 p + 4 p
  r.zz = ZZ3; r.zzz = quad_bcast<2>(m);
  return r;
}
__device__ __forceinline__ void g1_xyzz29_add_q4(g1_xyzz29_t &acc, const g1_xyzz29_t &o) {
  if (g1_xyzz29_is_identity(o)) return;
  if (g1_xyzz29_is_identity(acc)) { acc = o; return; }
  const uint32_t q = threadIdx.x & 3u;
  fe29_t m = FQ29_MUL(quad_sel(q, acc.x, acc.y, o.x, o.y), quad_sel(q, o.zz, o.zzz, acc.zz, acc.zzz));
  const fe29_t U1 = quad_bcast<0>(m), S1 = quad_bcast<1>(m), U2 = quad_bcast<2>(m), S2 = quad_bcast<3>(m);   // tight, < 1.1 p
  const fe29_t Pd = Fq29::sub4(U2, U1), Rd = Fq29::sub4(S2, S1);                                              // < 5.1 p
  m = FQ29_MUL(quad_sel(q, Pd, Rd, acc.zz, acc.zzz), quad_sel(q, Pd, Rd, o.zz, o.zzz));
  const fe29_t PP = quad_bcast<0>(m), RR = quad_bcast<1>(m), ZZ12 = quad_bcast<2>(m), ZZZ12 = quad_bcast<3>(m);
  const fe29_t one = Fq29::one();
  m = FQ29_MUL(quad_sel(q, Pd, U1, ZZ12, Rd), quad_sel(q, PP, PP, PP, one));
  const fe29_t PPP = quad_bcast<0>(m), Q = quad_bcast<1>(m), ZZ3 = quad_bcast<2>(m), Rone = quad_bcast<3>(m);
  if (Fq29::is_zero_tight(ZZ3)) {   // Pd == 0 (both zz != 0): o == +-acc, doubling or annihilation
    if (Fq29::is_zero_tight(Rone)) acc = g1_xyzz29_dbl_q4(acc);
    else acc = g1_xyzz29_identity();
    return;
  }
  const fe29_t X3 = Fq29::sub4_8(RR, PPP, Fq29::dbl(Q));                                                      // 13.2 p, one carry
  const fe29_t t = Fq29::sub16(Q, X3);
  m = FQ29_MUL(quad_sel(q, Rd, S1, ZZZ12, ZZZ12), quad_sel(q, t, PPP, PPP, PPP));
  acc.x = X3; acc.y = Fq29::sub4(quad_bcast<0>(m), quad_bcast<1>(m));                                         // < 1.6 p + 4 p
  acc.zz = ZZ3; acc.zzz = quad_bcast<2>(m);
}
// Q = 1: one lane per logical thread (the throughput form); Q = 4: a quad per logical thread (the latency form)
template <int Q> __device__ __forceinline__ void g1_vadd(g1_xyzz29_t &acc, const g1_xyzz29_t &o) { if (Q == 4) g1_xyzz29_add_q4(acc, o); else g1_xyzz29_add(acc, o); }
template <int Q> __device__ __forceinline__ g1_xyzz29_t g1_vdbl(const g1_xyzz29_t &a) { return Q == 4 ? g1_xyzz29_dbl_q4(a) : g1_xyzz29_dbl(a); }

// ---- 1. digits.  Plane layout enc[w * n + i]: 0 for a zero digit, else |d| (1 .. 2^(c-1)) with bit 31 = sign.
// up to 8 polynomial pointers travel as a kernel argument (copied at launch: no staging copy, nothing for an asynchronous caller to keep
// alive); larger batches pass a device array
struct PolyPtrs { const fe_t *p[8]; };
__global__ void __launch_bounds__(256) k_msm_digits(PolyPtrs inl, const fe_t *const *__restrict__ polys, uint32_t *__restrict__ enc, MsmPlan P,
                                                     uint32_t *__restrict__ coarse_hist, uint32_t fb, uint32_t cb_bits, uint32_t shared) {
  // the level-1 (coarse) histogram of the sorter is taken here, while the digits are in registers: LDS counters per block,
  // one global atomic per non-empty bin at the end (dynamic LDS = regions * 4 bytes)
  extern __shared__ uint32_t hist_lds[];
  const uint32_t CB = 1u << cb_bits, per_poly = (shared ? 1u : P.windows) * CB, regions = P.batch * per_poly;
  for (uint32_t b = threadIdx.x; b < regions; b += blockDim.x) hist_lds[b] = 0;
  __syncthreads();
  const uint32_t stride = gridDim.x * blockDim.x, m = blockIdx.y;   // grid.y = batch
  const fe_t *__restrict__ poly = polys ? polys[m] : inl.p[m];
  const uint32_t half = 1u << (P.c - 1), mask = (1u << P.c) - 1;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += stride) {
    const fe_t k = Fr::redc(g_load(&poly[i]));   // Montgomery -> canonical (= to_repr()): the reduction alone, 64 multiply-adds instead of a product by one
    // the scalar is consumed c bits at a time by shifting the whole 256-bit value right (8 funnel shifts per window, static
    // register indices); indexing k.l[bit >> 5] with a runtime window position would put the limbs in scratch memory
    // k and r - k name the same term up to the sign of the point: take the smaller one.  Uniform scalars gain nothing, but the small
    // NEGATIVE values of real witness columns (-1, -2, ...) become one-window scalars instead of 254-bit ones (an all "-1" column:
    // 18.2 -> 3.3 ms at 2^24).
    fe_t t; { uint32_t borrow = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) { const uint64_t d = (uint64_t)FrP::mod(q) - k.l[q] - borrow; t.l[q] = (uint32_t)d; borrow = (uint32_t)(d >> 63); } }
    bool flip = false;
#pragma unroll
    for (int q = 7; q >= 0; q--) { if (t.l[q] != k.l[q]) { flip = t.l[q] < k.l[q]; break; } }
    const uint32_t sign_flip = flip ? 0x80000000u : 0u;
    uint32_t l0 = flip ? t.l[0] : k.l[0], l1 = flip ? t.l[1] : k.l[1], l2 = flip ? t.l[2] : k.l[2], l3 = flip ? t.l[3] : k.l[3];
    uint32_t l4 = flip ? t.l[4] : k.l[4], l5 = flip ? t.l[5] : k.l[5], l6 = flip ? t.l[6] : k.l[6], l7 = flip ? t.l[7] : k.l[7];
    uint32_t carry = 0;
    for (uint32_t w = 0; w < P.windows; w++) {
      const uint32_t raw = (l0 & mask) + carry;
      l0 = __funnelshift_r(l0, l1, P.c); l1 = __funnelshift_r(l1, l2, P.c); l2 = __funnelshift_r(l2, l3, P.c); l3 = __funnelshift_r(l3, l4, P.c);
      l4 = __funnelshift_r(l4, l5, P.c); l5 = __funnelshift_r(l5, l6, P.c); l6 = __funnelshift_r(l6, l7, P.c); l7 >>= P.c;
      uint32_t e;
      if (raw > half) { e = (1u << P.c) - raw; if (e) e |= 0x80000000u; carry = 1; } else { e = raw; carry = 0; }   // raw == 2^c: digit 0, carry 1
      if (e) e ^= sign_flip;
      enc[((uint64_t)m * P.windows + w) * P.n + i] = e;
      if (e) atomicAdd(&hist_lds[m * per_poly + (shared ? 0 : w * CB) + (((e & 0x7fffffffu) - 1) >> fb)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < regions; b += blockDim.x) if (hist_lds[b]) atomicAdd(&coarse_hist[b], hist_lds[b]);
}

// ---- 2. exclusive scan (three small kernels; the array has W * 2^(c-1) + 1 entries)
constexpr uint32_t SCAN_BLOCK = 1024, SCAN_ITEMS = 4;  // 4096 per workgroup
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total) {
  // wave-level inclusive scan by shuffles, then across the 16 waves through LDS
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
  for (uint32_t o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  if (wave == 0) { uint32_t s = lane < (blockDim.x >> 6) ? lds[lane] : 0; for (uint32_t o = 1; o < 16; o <<= 1) { uint32_t y = __shfl_up(s, o); if (lane >= o) s += y; } if (lane < 16) lds[16 + lane] = s; }
  __syncthreads();
  const uint32_t wave_off = wave ? lds[16 + wave - 1] : 0;
  total = lds[16 + (blockDim.x >> 6) - 1];
  __syncthreads();
  return wave_off + x - v;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_partial(const uint32_t *__restrict__ in, uint32_t *__restrict__ block_sums, uint32_t n) {
  __shared__ uint32_t lds[32];
  const uint32_t base = blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
  uint32_t total; block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_block_sums(uint32_t *block_sums, uint32_t nblocks) {
  __shared__ uint32_t lds[32];
  __shared__ uint32_t running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblocks; base += SCAN_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? block_sums[i] : 0;
    uint32_t total; const uint32_t ex = block_exclusive_scan(v, lds, total);
    const uint32_t r = running;
    if (i < nblocks) block_sums[i] = r + ex;
    __syncthreads();
    if (threadIdx.x == 0) running = r + total;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_final(const uint32_t *__restrict__ in, const uint32_t *__restrict__ block_sums, uint32_t *__restrict__ out_a, uint32_t *__restrict__ out_b, uint32_t n) {
  __shared__ uint32_t lds[32];
  const uint32_t base = blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
  uint32_t total; uint32_t ex = block_exclusive_scan(s, lds, total) + block_sums[blockIdx.x];
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) if (base + k < n) { out_a[base + k] = ex; out_b[base + k] = ex; ex += v[k]; }
}

// ---- 3. two-level counting sort of the (window, bucket) keys, built for 19-22 bit keys and ~10^9 entries.
// A global atomic per entry (histogram + cursor) costs ~120 ms at 2^26 x 13 windows; here every workgroup first
// aggregates a tile in an LDS histogram and issues ONE global atomic per (tile, non-empty bin).
//   level 1: tiles over the digit plane, bin = window * CB + (bucket >> fb)          (CB = 2^cb_bits coarse bins per window;
//            its histogram is taken by k_msm_digits while the digits are in registers)
//            -> pairs[] = (fine key << 32 | point index | sign), grouped by coarse region
//   level 2: tiles inside each coarse region, bin = fine key (2^fb bins)             -> exact bucket offsets + sorted[]
// Order inside a bucket is arbitrary (group addition commutes), so nothing needs to be stable.
constexpr uint32_t SORT_MAX_BINS = 4096;
constexpr uint32_t SORT_SPLIT_TMAX = 128;   // entries of the block-start table of k_sort_l1_scatter_split (coarse bins >> spare key bits, at most 2048 >> 4)
struct SortPlan { uint32_t n, windows /* batch * W */, wpp /* W */, nb, fb, cb_bits, t1, t2, regions, shared, nshift /* ceil(log2 n): table row of an entry = payload >> nshift */; };   // shared = 1: all windows feed ONE bucket set (precomputed 2^(cw) P tables)

// Per-tile bin bookkeeping shared by both scatter kernels (1024 threads): lstart[] = exclusive scan of the tile histogram,
// gb[] = start of this tile's run inside each global bin (ONE returning global atomic per non-empty bin).  The atomics' results stay in
// registers: the caller stores them to LDS (tile_bin_publish) only after the staging pass, so their ~2 us round trip to the memory-side
// atomic unit overlaps with the LDS work instead of stalling the whole workgroup at a barrier.
__device__ __forceinline__ uint32_t tile_bin_offsets(const uint32_t *h, uint32_t *lstart, uint32_t (&gb)[4], uint32_t nbins, uint32_t *global_cursor, uint32_t *scratch32) {
  const uint32_t bpt = (nbins + 1023) >> 10, b0 = threadIdx.x * bpt;   // bpt <= 4
  uint32_t local[4], sum = 0;
  for (uint32_t k = 0; k < 4; k++) { local[k] = (k < bpt && b0 + k < nbins) ? h[b0 + k] : 0; sum += local[k]; }
  uint32_t total; uint32_t ex = block_exclusive_scan(sum, scratch32, total);
  for (uint32_t k = 0; k < 4; k++) { gb[k] = 0; if (k < bpt && b0 + k < nbins) { lstart[b0 + k] = ex; if (local[k]) gb[k] = atomicAdd(&global_cursor[b0 + k], local[k]); ex += local[k]; } }
  return total;
}
// gbase[b] = (start of this tile's run in global bin b) - lstart[b]: the write-out then places staged entry sidx at gbase[bin] + sidx with ONE
// LDS lookup per entry (the difference wraps modulo 2^32 and un-wraps in the sum)
__device__ __forceinline__ void tile_bin_publish(uint32_t *gbase, const uint32_t *lstart, const uint32_t (&gb)[4], uint32_t nbins) {
  const uint32_t bpt = (nbins + 1023) >> 10, b0 = threadIdx.x * bpt;
  for (uint32_t k = 0; k < 4; k++) if (k < bpt && b0 + k < nbins) gbase[b0 + k] = gb[k] - lstart[b0 + k];
}
// Level-1 scatter, LDS-staged: the tile is counting-sorted inside LDS first so that the global stores are coalesced runs
// (the direct version wrote 8-byte records at random: 3.4x write amplification measured with WRITE_SIZE).
// ---- split records (round 3, MI355_SORT_SPLIT): the level-1 output as TWO streams, payload (u32) and fine key (u16), instead of one u64 per
// entry of which 43 bits carry information.  The histogram pass then reads 2 bytes per entry instead of 8, the level-2 scatter 6 instead of 8, and
// with 6-byte staging a level-1 tile holds 24 576 entries (147 KB of LDS) -- runs of 24 entries per (tile, coarse bin) instead of 16.  The
// staged record no longer names its coarse bin; the write-out recovers it from the key's spare bits and a 32-entry table of block starts.
// (The single-stream u64 kernels k_sort_l1_scatter / k_sort_l2_hist / k_sort_l2_scatter were the A/B baseline of round 3 and left the library in round 6.)
template <int EPT> __global__ void __launch_bounds__(1024) k_sort_l1_scatter_split(const uint32_t *__restrict__ enc, uint32_t *__restrict__ coarse_cursor, uint32_t *__restrict__ pairs_lo,
                                                                                    uint16_t *__restrict__ pairs_hi, SortPlan S) {
  extern __shared__ uint32_t sm[];
  const uint32_t CB = 1u << S.cb_bits, CBp = (CB + 1) & ~1u, fmask = (1u << S.fb) - 1;
  // the 16 - fb spare bits of a staged key carry the low bits of the entry's coarse bin; T[k] = lstart[k << spare] locates the rest (see the write-out)
  const uint32_t spare = 16u - S.fb, rmask = (1u << spare) - 1, tcount = (CB + rmask) >> spare;
  uint32_t *h = sm, *lstart = sm + CBp, *gbase = sm + 2 * CBp, *scratch32 = sm + 3 * CBp, *T = sm + 3 * CBp + 32;
  uint32_t *stage_lo = T + SORT_SPLIT_TMAX;
  uint16_t *stage_hi = reinterpret_cast<uint16_t *>(stage_lo + 1024 * EPT);
  const uint32_t tiles1 = (S.n + S.t1 - 1) / S.t1;
  const uint32_t w = blockIdx.x / tiles1, j = blockIdx.x - w * tiles1;
  for (uint32_t b = threadIdx.x; b < CB; b += 1024) h[b] = 0;
  __syncthreads();
  const uint32_t i0 = j * S.t1, i1 = min(S.n, i0 + S.t1);
  const uint32_t *plane = enc + (uint64_t)w * S.n;
  uint32_t e[EPT], rank[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) {
    const uint32_t i = i0 + k * 1024 + threadIdx.x;
    e[k] = i < i1 ? plane[i] : 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every load in flight before the first returning LDS atomic (see k_sort_l1_scatter)
#pragma unroll
  for (int k = 0; k < EPT; k++) if (e[k]) rank[k] = atomicAdd(&h[((e[k] & 0x7fffffffu) - 1) >> S.fb], 1u);
  __syncthreads();
  const uint32_t m_poly = w / S.wpp, w_in = w - m_poly * S.wpp;
  uint32_t gb[4];
  const uint32_t total = tile_bin_offsets(h, lstart, gb, CB, coarse_cursor + (S.shared ? m_poly * CB : w * CB), scratch32);
  const uint32_t idx_base = S.shared ? w_in << S.nshift : 0;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < tcount; k += 1024) T[k] = lstart[k << spare];
#pragma unroll
  for (int k = 0; k < EPT; k++) if (e[k]) {
    const uint32_t bucket = (e[k] & 0x7fffffffu) - 1, i = i0 + k * 1024 + threadIdx.x, bin = bucket >> S.fb, pos = lstart[bin] + rank[k];
    stage_lo[pos] = (idx_base + i) | (e[k] & 0x80000000u);
    stage_hi[pos] = (uint16_t)((bucket & fmask) | ((bin & rmask) << S.fb));
  }
  tile_bin_publish(gbase, lstart, gb, CB);
  __syncthreads();
  // dense write-out: consecutive lanes take consecutive staged entries (coalesced runs).  The entry's coarse bin = (k << spare) | r with r from
  // the key's spare bits and k = the last block of 2^spare bins that starts at or before the entry: lstart is monotone, so the walk over T is
  // exact whatever the distribution (empty bins and blocks included); neighbouring lanes probe the same words (LDS broadcasts)
  uint32_t k = 0;   // a thread's entries are 1024 apart (~1.3 blocks of a uniform tile): the block index advances by a probe or two per step
  for (uint32_t sidx = threadIdx.x; sidx < total; sidx += 1024) {
    const uint32_t key = stage_hi[sidx];
    while (k + 1 < tcount && T[k + 1] <= sidx) k++;
    const uint32_t bin = (k << spare) | (key >> S.fb), pos = gbase[bin] + sidx;
    pairs_lo[pos] = stage_lo[sidx]; pairs_hi[pos] = (uint16_t)(key & fmask);
  }
}
// tile_start[r] = sum_{r' < r} ceil(size_r' / t2); one workgroup of SCAN_BLOCK threads, regions <= 8192
__global__ void __launch_bounds__(SCAN_BLOCK) k_sort_tile_prefix(const uint32_t *__restrict__ coarse_off, uint32_t *__restrict__ tile_start, SortPlan S) {
  __shared__ uint32_t lds[32];
  __shared__ uint32_t running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < S.regions; base += SCAN_BLOCK) {
    const uint32_t r = base + threadIdx.x;
    const uint32_t v = r < S.regions ? (coarse_off[r + 1] - coarse_off[r] + S.t2 - 1) / S.t2 : 0;
    uint32_t total; const uint32_t ex = block_exclusive_scan(v, lds, total);
    const uint32_t run = running;
    if (r < S.regions) tile_start[r] = run + ex;
    __syncthreads();
    if (threadIdx.x == 0) running = run + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_start[S.regions] = running;
}
__device__ __forceinline__ bool sort_l2_tile(const uint32_t *__restrict__ coarse_off, const uint32_t *__restrict__ tile_start, const SortPlan &S, uint32_t &region, uint32_t &s, uint32_t &e) {
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so workgroup b = 8 q + x takes the q-th tile of XCD x's contiguous
  // eighth of the tile list -- consecutive tiles (same coarse region, same few MB of `sorted`) then meet in ONE L2, which merges their
  // short interleaved runs before they reach HBM
  const uint32_t total = tile_start[S.regions], per_xcd = (total + 7) >> 3, q = blockIdx.x >> 3;
  const uint32_t tile = (blockIdx.x & 7) * per_xcd + q;
  if (q >= per_xcd || tile >= total) return false;
  uint32_t lo = 0, hi = S.regions;  // tile_start[lo] <= tile < tile_start[hi]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tile_start[mid] <= tile) lo = mid; else hi = mid; }
  region = lo;
  s = coarse_off[lo] + (tile - tile_start[lo]) * S.t2;
  e = min(coarse_off[lo + 1], s + S.t2);
  return true;
}
// Level-2 scatter, LDS-staged (same idea as level 1): tile <= 1024 * EPT entries of one coarse region, 2^fb fine bins; the bin of every staged entry travels in a 16-bit
// side array (fine < 4096), so the write-out needs no search over lstart.
// level-2 histogram and scatter over the split records
__global__ void __launch_bounds__(256) k_sort_l2_hist_split(const uint16_t *__restrict__ pairs_hi, const uint32_t *__restrict__ coarse_off, const uint32_t *__restrict__ tile_start,
                                                            uint32_t *__restrict__ hist, SortPlan S) {
  __shared__ uint32_t h[SORT_MAX_BINS];
  uint32_t region, s, e;
  if (!sort_l2_tile(coarse_off, tile_start, S, region, s, e)) return;
  const uint32_t FB = 1u << S.fb;
  for (uint32_t b = threadIdx.x; b < FB; b += blockDim.x) h[b] = 0;
  __syncthreads();
  // 8 keys (one 16-byte load) per lane and step over the aligned body, single keys at the ragged ends
  const uint32_t a0 = min(e, (s + 7u) & ~7u), a1 = a0 + ((e - a0) & ~7u);
  for (uint32_t p = s + threadIdx.x; p < a0; p += blockDim.x) atomicAdd(&h[pairs_hi[p]], 1u);
  for (uint32_t q = a0 + 8u * threadIdx.x; q < a1; q += 8u * blockDim.x) {
    const uint4 v = *reinterpret_cast<const uint4 *>(pairs_hi + q);
    atomicAdd(&h[v.x & 0xffffu], 1u); atomicAdd(&h[v.x >> 16], 1u); atomicAdd(&h[v.y & 0xffffu], 1u); atomicAdd(&h[v.y >> 16], 1u);
    atomicAdd(&h[v.z & 0xffffu], 1u); atomicAdd(&h[v.z >> 16], 1u); atomicAdd(&h[v.w & 0xffffu], 1u); atomicAdd(&h[v.w >> 16], 1u);
  }
  for (uint32_t p = a1 + threadIdx.x; p < e; p += blockDim.x) atomicAdd(&h[pairs_hi[p]], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < FB; b += blockDim.x) if (h[b]) atomicAdd(&hist[(region << S.fb) + b], h[b]);
}
template <int EPT> __global__ void __launch_bounds__(1024) k_sort_l2_scatter_split(const uint32_t *__restrict__ pairs_lo, const uint16_t *__restrict__ pairs_hi, const uint32_t *__restrict__ coarse_off,
                                                                                    const uint32_t *__restrict__ tile_start, uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted, SortPlan S) {
  extern __shared__ uint32_t sm[];
  const uint32_t FB = 1u << S.fb;
  uint32_t *h = sm, *lstart = sm + FB, *gbase = sm + 2 * FB, *scratch32 = sm + 3 * FB, *stage = sm + 3 * FB + 32;
  uint32_t region, s, e;
  if (!sort_l2_tile(coarse_off, tile_start, S, region, s, e)) return;
  for (uint32_t b = threadIdx.x; b < FB; b += 1024) h[b] = 0;
  __syncthreads();
  uint32_t idx[EPT], fine[EPT], rank[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) {
    const uint32_t p = s + k * 1024 + threadIdx.x;
    fine[k] = 0xffffffffu;
    if (p < e) { idx[k] = pairs_lo[p]; fine[k] = pairs_hi[p]; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < EPT; k++) if (fine[k] != 0xffffffffu) rank[k] = atomicAdd(&h[fine[k]], 1u);
  __syncthreads();
  uint32_t gb[4];
  const uint32_t total = tile_bin_offsets(h, lstart, gb, FB, cursor + (region << S.fb), scratch32);
  __syncthreads();
  uint16_t *stage_bin = reinterpret_cast<uint16_t *>(stage + 1024 * EPT);
#pragma unroll
  for (int k = 0; k < EPT; k++) if (fine[k] != 0xffffffffu) { const uint32_t pos = lstart[fine[k]] + rank[k]; stage[pos] = idx[k]; stage_bin[pos] = (uint16_t)fine[k]; }
  tile_bin_publish(gbase, lstart, gb, FB);
  __syncthreads();
  for (uint32_t sidx = threadIdx.x; sidx < total; sidx += 1024) {
    const uint32_t b = stage_bin[sidx];
    sorted[gbase[b] + sidx] = stage[sidx];
  }
}

// ---- 4. segmented accumulation
// offsets[b] .. offsets[b+1] = entries of global bucket b (b = w * nb + bucket); offsets has nbuckets+1 entries.
// bucket_sums must be zero-filled (all-zero XYZZ = identity) before launch.
// Segment length: the launch is sized for the worst case (every digit non-zero: n * W entries, `seg_max` per thread); the actual count is
// only known on the device (offsets[nbuckets]).  Witness-like columns (mostly zero / one-window scalars) leave 14 % of the entries, and
// with the worst-case segment 86 % of the launched threads had nothing to do while the rest ran at ~2 wavefronts per SIMD (11.6 ms for
// 1.1e8 entries at 2^26).  Every kernel that maps entries to threads therefore derives the segment from the actual total.
// The segment is chosen so that the entries fill ~40 % of the launched threads (about two rounds of resident wavefronts: enough to keep
// every SIMD at its three waves, while every further thread only adds partial sums for the fix-up to merge -- calibrated on witness-like,
// byte-valued and all-ones columns at 2^26, profiles/r02b_segment_calibration.log).
__device__ __forceinline__ uint32_t msm_seg_eff(uint32_t total, uint32_t threads, uint32_t seg_max, uint32_t seg_min, uint32_t fill_pct) {
  const uint64_t target = ((uint64_t)threads * (fill_pct > 100u ? 100u : fill_pct) + 99) / 100;   // never more than the launched threads: threads * segment must cover the entries
  uint32_t sg = (uint32_t)(((uint64_t)total + target - 1) / (target ? target : 1));
  sg = sg < seg_min ? seg_min : sg;
  return sg > seg_max ? seg_max : sg;
}
template <int VARIANT> __global__ void __launch_bounds__(256) k_msm_accumulate(const g1_affine_t *__restrict__ bases, const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ offsets,
                                                        uint32_t nbuckets, g1_xyzz29_t *__restrict__ bucket_sums, g1_xyzz29_t *__restrict__ part, int32_t *__restrict__ part_id, uint32_t seg_max,
                                                        uint32_t nshift, uint64_t row_stride, uint32_t gather_mask) {
  const uint32_t total = offsets[nbuckets];
  const uint32_t seg = msm_seg_eff(total, gridDim.x * blockDim.x, seg_max & 0x1fffu, (seg_max >> 13) & 0x1fffu, (seg_max >> 26) * 2);   // worst-case segment | minimum << 13 | (fill % / 2) << 26
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t start64 = (uint64_t)t * seg;
  if (start64 >= total) { part_id[2 * t] = -1; part_id[2 * t + 1] = -1; return; }
  const uint32_t start = (uint32_t)start64, end = (uint32_t)min((uint64_t)total, start64 + seg);
  // largest b with offsets[b] <= start  (then offsets[b+1] > start: the bucket that contains `start`)
  uint32_t lo = 0, hi = nbuckets;  // invariant offsets[lo] <= start < offsets[hi]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (offsets[mid] <= start) lo = mid; else hi = mid; }
  uint32_t b = lo, b_start = offsets[b], b_end = offsets[b + 1];
  uint32_t next_end = (b + 2 <= nbuckets) ? offsets[b + 2] : b_end;   // end of bucket b+1, fetched one bucket ahead so that a crossing does not stall on a load
  int32_t id_first = -1, id_last = -1;
  // the accumulator lives in the 9 x 29-bit unsaturated field (g1_29.hpp): one v_mad_u64_u32 per limb product, no carry chain;
  // it is flushed as a raw 144-byte record
  g1_xyzz29_t acc = g1_xyzz29_identity();
  // row_stride != 0: entry payload (w << nshift) | i names row w of the precomputed table T[w][.] = 2^(c w) P (rows row_stride points apart)
  auto base_of = [&](uint32_t e) -> const g1_affine_t * {
    const uint32_t gi = e & gather_mask;   // gather_mask = 0x7fffffff (index bits); smaller only in timing experiments
    if (row_stride == 0) return &bases[gi];
    const uint32_t w = gi >> nshift;       // entry payload = (table row << nshift) | point
    return &bases[(uint64_t)w * row_stride + (gi & ((1u << nshift) - 1))];
  };
  auto gather = [&](uint32_t e) -> g1_affine_t { return load_affine(base_of(e)); };
  // software pipeline: the gather of entry pos+1 (index, then 64-byte base) is issued before the ~10 field multiplications of entry pos.
  // VARIANT & 1 (A/B knob): the index stream runs one entry further ahead than the bases, so that the gather address is already in a
  // register when the iteration starts.
  uint32_t ent = sorted[start];
  uint32_t ent_next = start + 1 < end ? sorted[start + 1] : 0;
  g1_affine_t p = gather(ent);
  for (uint32_t pos = start; pos < end; pos++) {
    g1_affine_t p_next = p;
    uint32_t ent_next2 = 0;
    if (VARIANT & 1) {
      if (pos + 1 < end) p_next = gather(ent_next);
      if (pos + 2 < end) ent_next2 = sorted[pos + 2];
    } else {
      if (pos + 1 < end) { ent_next = sorted[pos + 1]; p_next = gather(ent_next); }
    }
    if (pos >= b_end) {
      // leave bucket b: it ends inside this thread's range
      if (b_start >= start) store_xyzz29(&bucket_sums[b], acc);                           // began here too: sole owner
      else { store_xyzz29(&part[2 * (uint64_t)t], acc); id_first = (int32_t)b; }          // began in an earlier thread
      acc = g1_xyzz29_identity();
      b++; b_start = b_end; b_end = (VARIANT & 2) ? next_end : offsets[b + 1];
      if (pos >= b_end) {
        // a run of empty buckets follows (low-entropy scalars: an all-equal column leaves ~2^21 / W empty buckets between two giant
        // ones, and walking them one dependent load at a time kept a handful of lanes busy for 35 ms): binary search for the bucket
        // that contains `pos` -- offsets[lo] <= pos < offsets[hi]
        uint32_t l2 = b, h2 = nbuckets;
        while (h2 - l2 > 1) { const uint32_t mid = (l2 + h2) >> 1; if (offsets[mid] <= pos) l2 = mid; else h2 = mid; }
        b = l2; b_start = offsets[b]; b_end = offsets[b + 1];
      }
      if (VARIANT & 2) next_end = (b + 2 <= nbuckets) ? offsets[b + 2] : b_end;
    }
    g1_xyzz29_madd<true, (VARIANT & 4) != 0>(acc, p, (ent >> 31) != 0);
    ent = ent_next; if (VARIANT & 1) ent_next = ent_next2; p = p_next;
  }
  // bucket b is still open at `end`
  if (b_start >= start && b_end <= end) store_xyzz29(&bucket_sums[b], acc);
  else if (b_start < start) { store_xyzz29(&part[2 * (uint64_t)t], acc); id_first = (int32_t)b; }   // spans the whole segment or just its head
  else { store_xyzz29(&part[2 * (uint64_t)t + 1], acc); id_last = (int32_t)b; }                        // began here, continues in the next thread
  part_id[2 * t] = id_first; part_id[2 * t + 1] = id_last;
}

// ---- 5. fix-up of buckets that straddle thread boundaries.  Small spans are summed by one lane; a bucket that spans more than
//         FIXUP_SERIAL_MAX accumulate-threads (skewed scalars: zeros/ones/small values, or the short top window) is queued and
//         reduced by a whole workgroup (wavefront-shuffle tree + LDS) in k_msm_fixup_big.
constexpr uint32_t FIXUP_SERIAL_MAX = 32, FIXUP_HUGE_MIN = 2048, FIXUP_SLICES = 64;   // SLICES <= 64: one wavefront folds the slice sums
// The fix-up and the whole reduction tail stay in the 29-bit field (g1_xyzz29_add / _dbl): no conversion of the 144-byte records to the
// saturated form (4 multiplications each) and the faster multiplier; records always hold valid accumulators (g1_29.hpp invariants).
template <int Q = 1> __device__ __forceinline__ void fixup_take29(g1_xyzz29_t &acc, const g1_xyzz29_t *__restrict__ part, const int32_t *__restrict__ part_id, uint32_t t, uint32_t b) {
  if (part_id[2 * t] == (int32_t)b) g1_vadd<Q>(acc, load_xyzz29(&part[2 * (uint64_t)t]));
  else if (part_id[2 * t + 1] == (int32_t)b) g1_vadd<Q>(acc, load_xyzz29(&part[2 * (uint64_t)t + 1]));
}
__device__ __forceinline__ g1_xyzz29_t shfl_down_xyzz29(const g1_xyzz29_t &v, uint32_t o) {
  g1_xyzz29_t r; const uint32_t *s = reinterpret_cast<const uint32_t *>(&v); uint32_t *d = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
  for (int i = 0; i < 36; i++) d[i] = __shfl_down(s[i], o);
  return r;
}
__device__ __forceinline__ g1_xyzz29_t shfl_xor_xyzz29(const g1_xyzz29_t &v, uint32_t o) {
  g1_xyzz29_t r; const uint32_t *s = reinterpret_cast<const uint32_t *>(&v); uint32_t *d = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
  for (int i = 0; i < 36; i++) d[i] = __shfl_xor(s[i], o);
  return r;
}
// FIXUP_LANES lanes per bucket: a bucket of a mid-size MSM straddles ~15 accumulate threads (2^20 pairs: 240 entries per bucket, 16 per
// thread), and one lane summing them serially left the kernel at one wavefront per SIMD on a 15-addition chain (184 us of a 2 ms MSM).
// Each lane of the group sums every FIXUP_LANES-th partial and two shuffle steps combine them; launch = nbuckets * FIXUP_LANES threads
// (0.60 -> 0.45 ms of tail at 2^14 pairs, 0.73 -> 0.67 ms at 2^20).
// FIXUP_LANES = 1 (big bucket sets: 2^21 buckets of which few straddle more than two threads) is the plain one-lane-per-bucket kernel.
template <uint32_t FIXUP_LANES, int Q = 1> __global__ void __launch_bounds__(256) k_msm_fixup(const uint32_t *__restrict__ offsets, uint32_t nbuckets, g1_xyzz29_t *__restrict__ bucket_sums,
                                                   const g1_xyzz29_t *__restrict__ part, const int32_t *__restrict__ part_id, uint32_t seg_max, uint32_t acc_threads,
                                                   uint32_t *__restrict__ big_list, uint32_t *__restrict__ big_count, uint32_t big_cap,
                                                   uint32_t *__restrict__ huge_list, uint32_t *__restrict__ huge_count, uint32_t huge_cap, uint32_t serial_max, uint32_t huge_min) {
  const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / Q, b = gid / FIXUP_LANES, sub = gid % FIXUP_LANES;   // Q lanes per logical thread
  const bool lead = (threadIdx.x & (Q - 1)) == 0;
  const uint32_t seg = msm_seg_eff(offsets[nbuckets], acc_threads, seg_max & 0x1fffu, (seg_max >> 13) & 0x1fffu, (seg_max >> 26) * 2);   // the segment length k_msm_accumulate used
  g1_xyzz29_t acc = g1_xyzz29_identity();
  bool mine = false;   // this group sums a straddling bucket of moderate span
  if (b < nbuckets) {
    const uint32_t s = offsets[b], e = offsets[b + 1];
    if (e != s) {
      const uint32_t t0 = s / seg, t1 = (e - 1) / seg;
      if (t0 != t1) {   // t0 == t1: the sole owner wrote it
        if (t1 - t0 > serial_max) {
          if (sub == 0 && lead) {
            if (huge_cap && t1 - t0 >= huge_min) {   // thousands of partials: several workgroups (k_msm_fixup_huge)
              const uint32_t idx = atomicAdd(huge_count, 1u);
              if (idx < huge_cap) { huge_list[3 * idx] = b; huge_list[3 * idx + 1] = t0; huge_list[3 * idx + 2] = t1; }
            } else {
              const uint32_t idx = atomicAdd(big_count, 1u);
              if (idx < big_cap) { big_list[3 * idx] = b; big_list[3 * idx + 1] = t0; big_list[3 * idx + 2] = t1; }
            }
          }
        } else {
          mine = true;
          for (uint32_t t = t0 + sub; t <= t1; t += FIXUP_LANES) fixup_take29<Q>(acc, part, part_id, t, b);
        }
      }
    }
  }
  // every lane of the wavefront reaches the shuffles; groups with nothing to do carry identities (the addition returns at once)
  if (FIXUP_LANES > 1) { const g1_xyzz29_t other = shfl_xor_xyzz29(acc, 1 * Q); g1_vadd<Q>(acc, other); }
  if (FIXUP_LANES > 2) { const g1_xyzz29_t other = shfl_xor_xyzz29(acc, 2 * Q); g1_vadd<Q>(acc, other); }
  static_assert(FIXUP_LANES == 1 || FIXUP_LANES == 2 || FIXUP_LANES == 4, "two shuffle steps");
  if (mine && sub == 0 && lead) store_xyzz29(&bucket_sums[b], acc);
}
__global__ void __launch_bounds__(256) k_msm_fixup_big(g1_xyzz29_t *__restrict__ bucket_sums, const g1_xyzz29_t *__restrict__ part, const int32_t *__restrict__ part_id,
                                                       const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ big_count) {
  __shared__ g1_xyzz29_t lds29[4];
  if (blockIdx.x >= *big_count) return;
  const uint32_t b = big_list[3 * blockIdx.x], t0 = big_list[3 * blockIdx.x + 1], t1 = big_list[3 * blockIdx.x + 2];
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (uint32_t t = t0 + threadIdx.x; t <= t1; t += blockDim.x) fixup_take29(acc, part, part_id, t, b);
  for (uint32_t o = 32; o >= 1; o >>= 1) { const g1_xyzz29_t other = shfl_down_xyzz29(acc, o); g1_xyzz29_add(acc, other); }
  if ((threadIdx.x & 63) == 0) lds29[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { for (uint32_t k = 1; k < (blockDim.x >> 6); k++) g1_xyzz29_add(acc, lds29[k]); store_xyzz29(&bucket_sums[b], acc); }
}
// A giant bucket (an all-ones selector column puts every point into ONE bucket) spans up to a million accumulate threads: its partials
// are summed by FIXUP_SLICES workgroups (one slice of the span each) into huge_part and folded by one wavefront.
// grid = huge_cap * FIXUP_SLICES resp. huge_cap; at most (accumulate threads) / FIXUP_HUGE_MIN buckets can qualify.
// slices of one giant bucket: a workgroup per ~2048 partial sums (8 per thread), at most FIXUP_SLICES -- a tree per slice costs ~10 additions
// whatever it sums, so a bucket of a few thousand partials gets 2-3 workgroups and the all-ones column (2^18 .. 2^20 partials) all 64
__device__ __forceinline__ uint32_t fixup_huge_slices(uint32_t span) { const uint32_t s = span >> 11; return s < 1u ? 1u : (s > FIXUP_SLICES ? FIXUP_SLICES : s); }
__global__ void __launch_bounds__(256) k_msm_fixup_huge(const g1_xyzz29_t *__restrict__ part, const int32_t *__restrict__ part_id, const uint32_t *__restrict__ huge_list,
                                                        const uint32_t *__restrict__ huge_count, g1_xyzz29_t *__restrict__ huge_part, uint32_t huge_cap) {
  __shared__ g1_xyzz29_t lds29[4];
  // block = slice * count + bucket (count = number of giant buckets, known on the device only): the blocks that have work are CONSECUTIVE
  // whether the column has one giant bucket or hundreds.  With bucket-major order over the allocated capacity the working blocks of a
  // column with ~200 giant buckets of 2-3 slices each sat at block indices 64 k + {0, 1, 2}, which the dispatcher maps to a handful of
  // CUs: 2.7 ms for work the one-workgroup-per-bucket kernel does in 0.15 ms (profiles/r02b_giant_bucket_fixup.md)
  const uint32_t count = min(*huge_count, huge_cap);
  if (count == 0) return;
  const uint32_t sl = blockIdx.x / count, idx = blockIdx.x - sl * count;
  if (sl >= FIXUP_SLICES) return;
  const uint32_t b = huge_list[3 * idx], t0 = huge_list[3 * idx + 1], t1 = huge_list[3 * idx + 2], span = t1 - t0 + 1;
  const uint32_t nsl = fixup_huge_slices(span);
  if (sl >= nsl) return;
  const uint32_t lo = t0 + (uint32_t)((uint64_t)span * sl / nsl), hi = t0 + (uint32_t)((uint64_t)span * (sl + 1) / nsl);   // [lo, hi)
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) fixup_take29(acc, part, part_id, t, b);
  for (uint32_t o = 32; o >= 1; o >>= 1) { const g1_xyzz29_t other = shfl_down_xyzz29(acc, o); g1_xyzz29_add(acc, other); }
  if ((threadIdx.x & 63) == 0) lds29[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { for (uint32_t k = 1; k < (blockDim.x >> 6); k++) g1_xyzz29_add(acc, lds29[k]); store_xyzz29(&huge_part[(uint64_t)idx * FIXUP_SLICES + sl], acc); }
}
__global__ void __launch_bounds__(64) k_msm_fixup_huge_fold(g1_xyzz29_t *__restrict__ bucket_sums, const uint32_t *__restrict__ huge_list, const uint32_t *__restrict__ huge_count,
                                                            const g1_xyzz29_t *__restrict__ huge_part) {
  const uint32_t idx = blockIdx.x;
  if (idx >= *huge_count) return;
  const uint32_t nsl = fixup_huge_slices(huge_list[3 * idx + 2] - huge_list[3 * idx + 1] + 1);
  g1_xyzz29_t acc = threadIdx.x < nsl ? load_xyzz29(&huge_part[(uint64_t)idx * FIXUP_SLICES + threadIdx.x]) : g1_xyzz29_identity();
  for (uint32_t o = FIXUP_SLICES / 2; o >= 1; o >>= 1) { const g1_xyzz29_t other = shfl_down_xyzz29(acc, o); g1_xyzz29_add(acc, other); }
  if (threadIdx.x == 0) store_xyzz29(&bucket_sums[huge_list[3 * idx]], acc);
}

// ---- 5b. the same fix-up as ONE segmented reduction over the partial sums (MI355_FIXUP_MODE=1).
// The partial records of the accumulate threads, read in thread order (slot 2t: the bucket that began in an earlier thread, slot 2t + 1: the
// bucket that continues in the next one, -1: none), are sorted by bucket, so the sums of straddling buckets are a reduction by key over a
// sorted sequence: every wavefront takes 64 consecutive slots, propagates keys over the empty slots (fill-forward), forms the suffix sums of
// equal keys by doubling (an addition is only issued for a step in which some lane has a partner of its own key: one step for the common
// two-partial bucket, six for a run that fills the wavefront) and stores the runs that lie strictly inside it; its first and last run may
// continue in the neighbouring wavefronts and go, two slots per wavefront, to the next level (1/32 of the size), which is the same kernel.
// Work and depth no longer depend on how the scalars are distributed: O(partials) additions, log-many levels, no lists, no thresholds.
__global__ void __launch_bounds__(256) k_msm_segfix(const int32_t *__restrict__ ids, const g1_xyzz29_t *__restrict__ recs, uint32_t n, g1_xyzz29_t *__restrict__ bucket_sums,
                                                    int32_t *__restrict__ ids_out, g1_xyzz29_t *__restrict__ recs_out, int last_level) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = j >> 6;
  if (wave >= ((n + 63) >> 6)) return;   // whole wavefronts past the end (the grid is rounded up to workgroups of four)
  const int32_t key0 = j < n ? ids[j] : -1;
  g1_xyzz29_t val = key0 >= 0 ? load_xyzz29(&recs[j]) : g1_xyzz29_identity();
  int32_t key = key0;
  for (uint32_t o = 1; o < 64; o <<= 1) { const int32_t kk = __shfl_up(key, o); if (lane >= o && key < 0) key = kk; }   // fill-forward over empty slots (they hold identities)
  for (uint32_t o = 1; o < 64; o <<= 1) {
    const int32_t k2 = __shfl_down(key, o);
    const bool take = lane + o < 64 && key >= 0 && k2 == key;
    if (__ballot(take) == 0) continue;                 // wave-uniform: no run is longer than o here
    const g1_xyzz29_t other = shfl_down_xyzz29(val, o);
    if (take) g1_xyzz29_add(val, other);
  }
  const int32_t kprev = __shfl_up(key, 1);
  const bool head = key >= 0 && (lane == 0 || kprev != key);
  const uint64_t valid = __ballot(key >= 0);
  const uint32_t first_lane = valid ? (uint32_t)__ffsll((unsigned long long)valid) - 1 : 64u;   // after the fill every lane from first_lane on is valid
  const int32_t key_last = __shfl(key, 63), key_first = __shfl(key, first_lane < 64u ? (int)first_lane : 0);
  const bool is_first = head && lane == first_lane, is_last = head && key == key_last;
  if (last_level) { if (head) store_xyzz29(&bucket_sums[key], val); return; }
  if (head && !is_first && !is_last) store_xyzz29(&bucket_sums[key], val);
  if (is_first) { ids_out[2 * wave] = key; store_xyzz29(&recs_out[2 * (uint64_t)wave], val); }
  if (is_last && !is_first) { ids_out[2 * wave + 1] = key; store_xyzz29(&recs_out[2 * (uint64_t)wave + 1], val); }
  if (lane == 0) {   // unused output slots are marked empty (the slot writers above and these never touch the same slot)
    if (first_lane == 64u) { ids_out[2 * wave] = -1; ids_out[2 * wave + 1] = -1; }
    else if (key_first == key_last) ids_out[2 * wave + 1] = -1;
  }
}

// host-chunked MSM: every slice of the point range fills its own bucket set (same layout); buckets[b] += sum_k buckets[k * nbuckets + b]
__global__ void __launch_bounds__(256) k_msm_bucket_fold(g1_xyzz29_t *__restrict__ buckets, uint32_t nbuckets, uint32_t sets) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  g1_xyzz29_t acc = load_xyzz29(&buckets[b]);
  for (uint32_t k = 1; k < sets; k++) g1_xyzz29_add(acc, load_xyzz29(&buckets[(uint64_t)k * nbuckets + b]));
  store_xyzz29(&buckets[b], acc);
}

// ---- 6a. chunked running sums: thread j of window w covers buckets [j*K, (j+1)*K) and emits
//          T + (j*K) * S  where S = sum B_i, T = sum (i_local + 1) B_i
template <int Q> __global__ void __launch_bounds__(128) k_msm_bucket_reduce(const g1_xyzz29_t *__restrict__ bucket_sums, g1_xyzz29_t *__restrict__ chunk_out, MsmPlan P, uint32_t chunk) {
  const uint32_t chunks_per_window = P.nb / chunk;
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / Q;   // logical thread (Q lanes each)
  if (g >= chunks_per_window * P.windows) return;
  const uint32_t w = g / chunks_per_window, j = g - w * chunks_per_window;
  const g1_xyzz29_t *B = bucket_sums + (uint64_t)w * P.nb + (uint64_t)j * chunk;
  g1_xyzz29_t run = g1_xyzz29_identity(), T = g1_xyzz29_identity();
  for (uint32_t i = chunk; i-- > 0;) { g1_vadd<Q>(run, load_xyzz29(&B[i])); g1_vadd<Q>(T, run); }
  if (j != 0) {
    const uint32_t k = j * chunk;
    g1_xyzz29_t kS = g1_xyzz29_identity();
    for (int bit = 31 - __clz(k); bit >= 0; bit--) { kS = g1_vdbl<Q>(kS); if ((k >> bit) & 1) g1_vadd<Q>(kS, run); }
    g1_vadd<Q>(T, kS);
  }
  if ((threadIdx.x & (Q - 1)) == 0) store_xyzz29(&chunk_out[g], T);
}
// ---- 6b. per window: tree-sum of the chunk results.  grid = (blocks, windows); a block folds up to 256 * TREE_PER_THREAD inputs
//          (wavefront shuffles, then LDS across the 4 waves) into one output; launched repeatedly until one value per window is left.
constexpr uint32_t TREE_PER_THREAD = 4;
// a block of 256 lanes = 256 / Q logical threads folds up to (256 / Q) * TREE_PER_THREAD inputs
template <int Q> __global__ void __launch_bounds__(256) k_msm_tree_sum29(const g1_xyzz29_t *__restrict__ in, uint32_t in_per_window, g1_xyzz29_t *__restrict__ out, uint32_t out_per_window) {
  __shared__ g1_xyzz29_t lds[4];
  constexpr uint32_t VT = 256 / Q;                               // logical threads per block, 64 / Q per wavefront
  const uint32_t w = blockIdx.y, first = blockIdx.x * VT * TREE_PER_THREAD, vt = threadIdx.x / Q;
  const g1_xyzz29_t *src = in + (uint64_t)w * in_per_window;
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (uint32_t k = 0; k < TREE_PER_THREAD; k++) { const uint32_t i = first + k * VT + vt; if (i < in_per_window) g1_vadd<Q>(acc, load_xyzz29(&src[i])); }
  for (uint32_t o = 32 / Q; o >= 1; o >>= 1) { const g1_xyzz29_t other = shfl_down_xyzz29(acc, o * Q); g1_vadd<Q>(acc, other); }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x < Q) { for (uint32_t k = 1; k < 4; k++) g1_vadd<Q>(acc, lds[k]); if (threadIdx.x == 0) store_xyzz29(&out[(uint64_t)w * out_per_window + blockIdx.x], acc); }
}
__device__ __forceinline__ void msm_emit_result(const g1_xyzz_t &acc, g1_jac_t *out, int normalise) {
  if (normalise) { *out = g1_xyzz_to_jac_normalised(acc); return; }
  // un-normalised Jacobian representative (X ZZ^2, Y ZZZ^2, ZZZ): skips the inversion; used for the per-GPU partial sums, which are
  // folded (and normalised once) by k_g1_sum
  g1_jac_t r;
  if (g1_xyzz_is_identity(acc)) { r.x = Fq::zero(); r.y = Fq::zero(); r.z = Fq::zero(); }
  else { r.x = fq_mul_ps(acc.x, fq_sqr_ps(acc.zz)); r.y = fq_mul_ps(acc.y, fq_sqr_ps(acc.zzz)); r.z = acc.zzz; }
  *out = r;
}
// ---- 7. Horner over windows + normalisation.  One lane; 255 doublings are inherently serial (none with window tables).
template <int Q> __global__ void __launch_bounds__(64) k_msm_final29(const g1_xyzz29_t *__restrict__ window_sums, uint32_t windows, uint32_t c, g1_jac_t *__restrict__ out, int normalise) {
  if (threadIdx.x >= Q) return;
  window_sums += (uint64_t)blockIdx.x * windows; out += blockIdx.x;
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (uint32_t w = windows; w-- > 0;) {
    for (uint32_t k = 0; k < c; k++) acc = g1_vdbl<Q>(acc);
    g1_vadd<Q>(acc, load_xyzz29(&window_sums[w]));
  }
  if (threadIdx.x == 0) msm_emit_result(g1_xyzz29_to_sat(acc), out, normalise);
}
// sum of n Jacobian points (fold of per-GPU partial results), normalised
__global__ void k_g1_sum(const g1_jac_t *__restrict__ pts, uint32_t n, g1_jac_t *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  g1_xyzz_t acc = g1_xyzz_identity();
  for (uint32_t i = 0; i < n; i++) { g1_jac_t p = pts[i]; g1_xyzz_add_ps(acc, g1_jac_to_xyzz(p)); }
  *out = g1_xyzz_to_jac_normalised(acc);
}

// out[m] = sum_k pts[k * stride + m], k < count: the fold of the per-chunk partial results of a pipelined MSM (one workgroup per polynomial)
__global__ void k_g1_sum_strided(const g1_jac_t *__restrict__ pts, uint32_t count, uint32_t stride, g1_jac_t *__restrict__ out, int normalise) {
  if (threadIdx.x != 0) return;
  g1_xyzz_t acc = g1_xyzz_identity();
  for (uint32_t k = 0; k < count; k++) { g1_jac_t p = pts[(size_t)k * stride + blockIdx.x]; g1_xyzz_add_ps(acc, g1_jac_to_xyzz(p)); }
  if (normalise) { out[blockIdx.x] = g1_xyzz_to_jac_normalised(acc); return; }
  g1_jac_t r;
  if (g1_xyzz_is_identity(acc)) { r.x = Fq::zero(); r.y = Fq::zero(); r.z = Fq::zero(); }
  else { r.x = fq_mul_ps(acc.x, fq_sqr_ps(acc.zz)); r.y = fq_mul_ps(acc.y, fq_sqr_ps(acc.zzz)); r.z = acc.zzz; }
  out[blockIdx.x] = r;
}

// every point is the identity (0, 0) or satisfies y^2 = x^3 + 3 with both coordinates reduced: what SerdeFormat::RawBytes checks point by
// point on the CPU while reading a params file [EXT-recalled halo2curves]; here one streaming pass over the basis in HBM
__global__ void __launch_bounds__(256) k_g1_validate(const g1_affine_t *__restrict__ pts, uint64_t n, uint32_t *__restrict__ bad) {
  uint32_t local = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const g1_affine_t p = load_affine(&pts[i]);
    if (g1_affine_is_identity(p)) continue;
    bool ok = true;
    for (int c = 0; c < 2 && ok; c++) {   // coordinate < p (canonical Montgomery residue)
      const fe_t &v = c ? p.y : p.x; bool lt = false;
      for (int k = 7; k >= 0; k--) { if (v.l[k] != FqP::mod(k)) { lt = v.l[k] < FqP::mod(k); break; } }
      ok = lt;
    }
    if (ok) {
      fe_t three = Fq::zero(); three.l[0] = 3; three = Fq::from_canonical(three);
      ok = Fq::eq(fq_sqr_ps(p.y), Fq::add(fq_mul_ps(fq_sqr_ps(p.x), p.x), three));
    }
    if (!ok) local++;
  }
  if (local) atomicAdd(bad, local);
}

// ---- window precomputation for a registered basis: T[w][i] = 2^(c w) * P_i, affine, w < W (row 0 = the basis itself).
// One thread per point and c doublings per row; the conversion back to affine coordinates shares ONE inversion per row among the 256
// points of a workgroup (Montgomery's trick through the LDS scans of frscan.hpp, the tile product inverted by lane 0's Euclidean
// inverse): ~210 field multiplications per point and row instead of ~550 with a Fermat ladder per point (3.05 s per 2^26 basis before).
// One-off cost at registration.
__global__ void __launch_bounds__(FRSCAN_THREADS) k_srs_precompute(const g1_affine_t *__restrict__ base, g1_affine_t *__restrict__ table, uint64_t n, uint32_t W, uint32_t c) {
  __shared__ uint32_t buf[2 * FRSCAN_THREADS * 9];
  __shared__ uint32_t inv_total[8];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  g1_affine_t P; P.x = Fq::zero(); P.y = Fq::zero();
  if (live) { P = load_affine(&base[i]); g_store(&table[i].x, P.x); g_store(&table[i].y, P.y); }
  const bool ident = !live || g1_affine_is_identity(P);   // the identity stays the identity in every row
  bool dead = false;                                       // a degenerate input whose multiple hit Z = 0 (see below)
  for (uint32_t w = 1; w < W; w++) {
    g1_xyzz_t acc = g1_xyzz_identity();
    fe_t z = Fq::one();
    if (!ident && !dead) {
      acc = g1_xyzz_dbl_affine_ps(P);
      for (uint32_t k = 1; k < c; k++) acc = g1_xyzz_dbl_ps(acc);
      // BN254 G1 has prime order and no 2-torsion: doubling a non-identity CURVE point never gives the identity, so ZZ ZZZ != 0.  Bases are
      // not validated at registration (mi355_srs_register_host / _dev), and an input with y = 0 or off the curve can reach ZZ ZZZ = 0: such a
      // point becomes the identity in this row and every later one, and must not zero the shared product of its 255 neighbours
      z = fq_mul_ps(acc.zz, acc.zzz);
      if (Fq::is_zero(z)) { dead = true; z = Fq::one(); }
    }
    fe_t total, total_r;
    const fe_t left = block_exclusive_mul_scan<false, true>(z, buf, total);
    const fe_t right = block_exclusive_mul_scan<true, true>(z, buf, total_r);
    if (threadIdx.x == 0) lds_put(inv_total, Fq::inv_sgcd(total));
    __syncthreads();
    if (dead) { P.x = Fq::zero(); P.y = Fq::zero(); }
    else if (!ident) {
      const fe_t inv = fq_mul_ps(fq_mul_ps(lds_get(inv_total), left), right);   // 1 / (ZZ ZZZ) of this thread's point
      P.x = fq_mul_ps(acc.x, fq_mul_ps(inv, acc.zzz)); P.y = fq_mul_ps(acc.y, fq_mul_ps(inv, acc.zz));
    }
    if (live) { g_store(&table[w * n + i].x, P.x); g_store(&table[w * n + i].y, P.y); }
  }
}

// ---- synthetic SRS helpers (ParamsKZG::setup restated): fixed-base multiplication by 8-bit windows.
// table[j][d] = d * 2^(8j) * G  for j < 32, d < 256 (affine; entry d = 0 is the identity (0,0)).
__global__ void k_fixed_base_table(g1_affine_t *table) {
  // one workgroup of 256 threads; thread d computes d * 2^(8j) G for every j by repeated doubling of its own point
  const uint32_t d = threadIdx.x;
  g1_affine_t G; G.x = Fq::one(); G.y = Fq::dbl(Fq::one());
  g1_xyzz_t P = g1_xyzz_mul_small(g1_xyzz_from_affine(G), d);
  for (uint32_t j = 0; j < 32; j++) {
    table[j * 256 + d] = g1_xyzz_to_affine(P);
    for (int k = 0; k < 8; k++) P = g1_xyzz_dbl_ps(P);
  }
}
// points[i] = scalars[i] * G, affine.  One thread per point: 32 mixed additions + one inversion.
__global__ void __launch_bounds__(256) k_fixed_base_mul(const g1_affine_t *__restrict__ table, const fe_t *__restrict__ scalars, g1_affine_t *__restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_t one_c = Fr::zero(); one_c.l[0] = 1;
  const fe_t k = fr_mul_ps(g_load(&scalars[i]), one_c);
  g1_xyzz_t acc = g1_xyzz_identity();
  for (uint32_t j = 0; j < 32; j++) {
    const uint32_t d = (k.l[j >> 2] >> ((j & 3) * 8)) & 0xff;
    if (d) g1_xyzz_madd_ps(acc, load_affine(&table[j * 256 + d]));
  }
  g1_affine_t r = g1_xyzz_to_affine(acc);
  g_store(&out[i].x, r.x); g_store(&out[i].y, r.y);
}
// scalars for setup: g_scal[i] = tau^i ; gl_scal[i] = omega^i (tau^n - 1) / (n (tau - omega^i))
__global__ void __launch_bounds__(256) k_srs_scalars(fe_t *__restrict__ g_scal, fe_t *__restrict__ gl_scal, fe_t tau, fe_t omega, fe_t tn1_over_n, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g_store(&g_scal[i], Fr::pow_u64(tau, i));
  const fe_t wi = Fr::pow_u64(omega, i);
  const fe_t d = Fr::inv_sgcd(Fr::sub(tau, wi));   // division steps: the same instruction stream in every lane
  g_store(&gl_scal[i], fr_mul_ps(fr_mul_ps(wi, tn1_over_n), d));
}
#endif  // __HIPCC__

}  // namespace zk
