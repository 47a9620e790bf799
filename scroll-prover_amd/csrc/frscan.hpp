// frscan.hpp -- multiplicative scans over device-resident Fr vectors: the grand products and the batch inversion of the permutation and
// lookup arguments of create_proof [EXT-recalled halo2_proofs src/plonk/permutation/prover.rs,
// src/plonk/lookup/prover.rs: `modified_values.batch_invert()`, then z[0] = 1, z[i + 1] = z[i] * modified_values[i]].
//
//   k_fr_batch_invert     data[i] = data[i]^-1, zeros stay zero (ff::BatchInvert semantics).  A thread multiplies its 8 (strided,
//                         coalesced) elements, the 256 thread products are scanned from both ends through LDS; the products of the
//                         2048-element tiles are themselves batch-inverted (recursion on the host side), so the whole vector costs a
//                         handful of one-lane inversions (division steps, fp.hpp inv_sgcd).
//   k_fr_prefix_product   dst[i] = prod_{j < i} src[j]  (dst[0] = 1): tile products -> scan of the tile products -> tile-local rescan
//                         with the carried-in prefix.  Order matters here, so tiles go through LDS to turn coalesced 16-byte-per-lane
//                         global accesses into 8 consecutive elements per thread (chunk stride 65 dwords: conflict-free).
// Both are streaming kernels (one read + one write of the vector, plus one re-read for the prefix product) with ~5 multiplications
// per element; Montgomery form in, Montgomery form out, fully reduced.
#pragma once
#include "fp_asm.hpp"

namespace zk {
#ifdef __HIPCC__

constexpr uint32_t FRSCAN_THREADS = 256, FRSCAN_EPT = 8, FRSCAN_TILE = FRSCAN_THREADS * FRSCAN_EPT;

__device__ __forceinline__ void lds_put(uint32_t *p, const fe_t &v) {
#pragma unroll
  for (int k = 0; k < 8; k++) p[k] = v.l[k];
}
__device__ __forceinline__ fe_t lds_get(const uint32_t *p) {
  fe_t v;
#pragma unroll
  for (int k = 0; k < 8; k++) v.l[k] = p[k];
  return v;
}

// exclusive multiplicative scan of one value per thread across the workgroup (Hillis-Steele through LDS, 9 dwords per slot).
// buf: 2 * 256 * 9 dwords.  Returns prod_{t' < t} x_t'; total = product of all 256 values.  REVERSE scans from the other end.
// FQ: the same scan in the base field (k_g1_batch_normalize multiplies Z coordinates).
// ADD: the additive scan of Fr (the grand SUM of the log-derivative lookup argument) with the same structure: sum_{t' < t} x_t', identity zero.
template <bool REVERSE, bool FQ = false, bool ADD = false> __device__ fe_t block_exclusive_mul_scan(const fe_t &x, uint32_t *buf, fe_t &total) {
  const uint32_t t = REVERSE ? FRSCAN_THREADS - 1 - threadIdx.x : threadIdx.x;
  uint32_t *cur = buf, *nxt = buf + FRSCAN_THREADS * 9;
  lds_put(cur + t * 9, x);
  __syncthreads();
  fe_t v = x;
  for (uint32_t o = 1; o < FRSCAN_THREADS; o <<= 1) {
    if (t >= o) v = ADD ? Fr::add(lds_get(cur + (t - o) * 9), v) : FQ ? fq_mul_ps(lds_get(cur + (t - o) * 9), v) : fr_mul_ps(lds_get(cur + (t - o) * 9), v);
    lds_put(nxt + t * 9, v);
    __syncthreads();
    uint32_t *tmp = cur; cur = nxt; nxt = tmp;
  }
  total = lds_get(cur + (FRSCAN_THREADS - 1) * 9);
  fe_t ex = t ? lds_get(cur + (t - 1) * 9) : (ADD ? Fr::zero() : FQ ? Fq::one() : Fr::one());
  __syncthreads();
  return ex;
}
template <bool ADD> __device__ __forceinline__ fe_t fr_scan_op(const fe_t &a, const fe_t &b) { return ADD ? Fr::add(a, b) : fr_mul_ps(a, b); }

// MODE 0: self-contained (the tile product is inverted by lane 0) -- for short vectors and the last level;
// MODE 1: tile_prod[b] = product of the tile's non-zero elements;  MODE 2: invert with tile_prod[b] already holding the INVERSE of the
// tile product (the host recursion inverts the tile products with the same three steps, so a 2^26 vector costs 17 one-lane inversions).
template <int MODE> __global__ void __launch_bounds__(FRSCAN_THREADS) k_fr_batch_invert(fe_t *__restrict__ data, uint64_t n, fe_t *__restrict__ tile_prod) {
  __shared__ uint32_t buf[2 * FRSCAN_THREADS * 9];
  __shared__ uint32_t inv_total[8];
  const uint64_t base = (uint64_t)blockIdx.x * FRSCAN_TILE;
  fe_t a[FRSCAN_EPT], pre[FRSCAN_EPT];   // pre[j] = a[0] * ... * a[j] with zeros replaced by one
  uint32_t zero_mask = 0;
  fe_t run = Fr::one();
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) {
    const uint64_t i = base + j * FRSCAN_THREADS + threadIdx.x;
    a[j] = i < n ? g_load(&data[i]) : Fr::one();
    if (Fr::is_zero(a[j])) { zero_mask |= 1u << j; a[j] = Fr::one(); }
    run = j ? fr_mul_ps(run, a[j]) : a[j];
    pre[j] = run;
  }
  fe_t total, total_r;
  const fe_t left = block_exclusive_mul_scan<false>(run, buf, total);
  if (MODE == 1) { if (threadIdx.x == 0) g_store(&tile_prod[blockIdx.x], total); return; }
  const fe_t right = block_exclusive_mul_scan<true>(run, buf, total_r);
  fe_t inv_t;
  if (MODE == 0) {
    if (threadIdx.x == 0) lds_put(inv_total, Fr::inv_sgcd(total));   // division-step inverse (fp.hpp): a fraction of the Fermat ladder's latency
    __syncthreads();
    inv_t = lds_get(inv_total);
  } else inv_t = g_load(&tile_prod[blockIdx.x]);
  fe_t inv = fr_mul_ps(fr_mul_ps(inv_t, left), right);   // (product of this thread's elements)^-1
#pragma unroll
  for (int j = FRSCAN_EPT - 1; j >= 0; j--) {
    const fe_t out = j ? fr_mul_ps(inv, pre[j - 1]) : inv;
    inv = fr_mul_ps(inv, a[j]);
    const uint64_t i = base + (uint64_t)j * FRSCAN_THREADS + threadIdx.x;
    if (i < n) g_store(&data[i], (zero_mask >> j) & 1u ? Fr::zero() : out);
  }
}

// tile <-> registers: thread t gets elements [t * 8, t * 8 + 8) of the tile; global accesses are coalesced (lane = consecutive element)
__device__ __forceinline__ void tile_load(const fe_t *__restrict__ src, uint64_t base, uint64_t n, uint32_t *tile, fe_t (&a)[FRSCAN_EPT], bool pad_zero = false) {
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) {
    const uint32_t e = j * FRSCAN_THREADS + threadIdx.x;
    const fe_t v = base + e < n ? g_load(&src[base + e]) : (pad_zero ? Fr::zero() : Fr::one());
    lds_put(tile + (e >> 3) * 65 + (e & 7) * 8, v);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) a[j] = lds_get(tile + threadIdx.x * 65 + j * 8);
  __syncthreads();
}

// PHASE 0: tile_prod[b] = product of tile b.  PHASE 1: dst[i] = tile_prefix[b] * (product of the tile's elements before i).
// ADD: sums instead of products (dst[0] = 0, dst[i] = sum_{j < i} src[j]) -- the running sum phi of the log-derivative lookup argument
// [EXT-recalled halo2_proofs (scroll fork) src/plonk/mv_lookup/prover.rs: phi[i + 1] = phi[i] + sum_j 1 / (beta + f_j[i]) - m[i] / (beta + t[i])].
template <int PHASE, bool ADD = false> __global__ void __launch_bounds__(FRSCAN_THREADS) k_fr_prefix_product(const fe_t *__restrict__ src, fe_t *__restrict__ dst, uint64_t n,
                                                                                           fe_t *__restrict__ tile_prod, const fe_t *__restrict__ tile_prefix) {
  extern __shared__ uint32_t sm[];
  uint32_t *tile = sm, *buf = sm;   // the scan buffers (18 KB) reuse the tile (65 KB): the tile is in registers while the scan runs, and two workgroups fit a CU
  const uint64_t base = (uint64_t)blockIdx.x * FRSCAN_TILE;
  fe_t a[FRSCAN_EPT];
  tile_load(src, base, n, tile, a, ADD);
  fe_t run = a[0];
#pragma unroll
  for (uint32_t j = 1; j < FRSCAN_EPT; j++) run = fr_scan_op<ADD>(run, a[j]);
  fe_t total;
  fe_t ex = block_exclusive_mul_scan<false, false, ADD>(run, buf, total);
  if (PHASE == 0) { if (threadIdx.x == 0) g_store(&tile_prod[blockIdx.x], total); return; }
  ex = fr_scan_op<ADD>(ex, g_load(&tile_prefix[blockIdx.x]));
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) {
    lds_put(tile + threadIdx.x * 65 + j * 8, ex);
    ex = fr_scan_op<ADD>(ex, a[j]);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) {
    const uint32_t e = j * FRSCAN_THREADS + threadIdx.x;
    if (base + e < n) g_store(&dst[base + e], lds_get(tile + (e >> 3) * 65 + (e & 7) * 8));
  }
}
// exclusive scan of the m tile products (sums) by one workgroup (thread t owns a contiguous run); total_out = product (sum) of everything
template <bool ADD> __global__ void __launch_bounds__(FRSCAN_THREADS) k_fr_scan_tiles(const fe_t *__restrict__ tile_prod, fe_t *__restrict__ tile_prefix, uint32_t m, fe_t *__restrict__ total_out) {
  __shared__ uint32_t buf[2 * FRSCAN_THREADS * 9];
  const uint32_t per = (m + FRSCAN_THREADS - 1) / FRSCAN_THREADS, lo = min(m, threadIdx.x * per), hi = min(m, lo + per);
  fe_t run = ADD ? Fr::zero() : Fr::one();
  for (uint32_t i = lo; i < hi; i++) run = fr_scan_op<ADD>(run, g_load(&tile_prod[i]));
  fe_t total;
  fe_t ex = block_exclusive_mul_scan<false, false, ADD>(run, buf, total);
  for (uint32_t i = lo; i < hi; i++) { g_store(&tile_prefix[i], ex); ex = fr_scan_op<ADD>(ex, g_load(&tile_prod[i])); }
  if (threadIdx.x == 0) g_store(total_out, total);
}

// ---- first-order linear recurrence with a constant multiplier: P_j = b_j + m * P_(j-1), P_(-1) = 0, j < n.
// halo2's kate_division(a, z) = (a(X) - a(z)) / (X - z) [EXT-recalled halo2_proofs src/arithmetic.rs; the quotient polynomials of the
// multi-open (SHPLONK) argument, SURVEY 3.2 step 10] is this recurrence read from the top coefficient down: q_i = a_(i+1) + z * q_(i+1).
// REVERSE maps recurrence index j to memory index n - 1 - j for loads and stores alike.  Tile scheme as the prefix product: tile totals
// (FINAL = 0) -> the same recurrence over the totals with multiplier m^2048 (host-side recursion) -> per-tile completion with the carried-in
// value folded into the tile's first element (FINAL = 1).  Because every thread's 8-element run has the SAME multiplier m^8, the
// workgroup scan of the affine maps only has to scan the offsets: v_t += v_(t - o) * m^(8 o).
template <int FINAL> __global__ void __launch_bounds__(FRSCAN_THREADS) k_fr_linrec(const fe_t *__restrict__ src, fe_t *__restrict__ dst, uint64_t n, fe_t m, int reverse,
                                                                                   fe_t *__restrict__ tile_tot) {
  extern __shared__ uint32_t sm[];
  uint32_t *tile = sm, *buf = sm;   // scan buffers inside the tile region, as in k_fr_prefix_product
  const uint64_t base = (uint64_t)blockIdx.x * FRSCAN_TILE;
  fe_t a[FRSCAN_EPT];
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) {
    const uint32_t e = j * FRSCAN_THREADS + threadIdx.x;
    const uint64_t jj = base + e;
    const fe_t v = jj < n ? g_load(&src[reverse ? n - 1 - jj : jj]) : Fr::zero();
    lds_put(tile + (e >> 3) * 65 + (e & 7) * 8, v);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) a[j] = lds_get(tile + threadIdx.x * 65 + j * 8);
  __syncthreads();
  if (FINAL && tile_tot && blockIdx.x > 0 && threadIdx.x == 0) a[0] = Fr::add(a[0], fr_mul_ps(m, g_load(&tile_tot[blockIdx.x - 1])));
  fe_t run = a[0];
#pragma unroll
  for (uint32_t j = 1; j < FRSCAN_EPT; j++) run = Fr::add(a[j], fr_mul_ps(m, run));
  // inclusive scan of the thread offsets with multiplier m^8 per thread step
  fe_t pw = fr_sqr_ps(fr_sqr_ps(fr_sqr_ps(m)));   // m^8
  uint32_t *cur = buf, *nxt = buf + FRSCAN_THREADS * 9;
  const uint32_t t = threadIdx.x;
  fe_t v = run;
  lds_put(cur + t * 9, v);
  __syncthreads();
  for (uint32_t o = 1; o < FRSCAN_THREADS; o <<= 1) {
    if (t >= o) v = Fr::add(v, fr_mul_ps(lds_get(cur + (t - o) * 9), pw));
    lds_put(nxt + t * 9, v);
    __syncthreads();
    uint32_t *tmp = cur; cur = nxt; nxt = tmp;
    pw = fr_sqr_ps(pw);
  }
  if (!FINAL) { if (t == FRSCAN_THREADS - 1) g_store(&tile_tot[blockIdx.x], v); return; }
  fe_t P = t ? lds_get(cur + (t - 1) * 9) : Fr::zero();   // value of the recurrence just before this thread's run
  __syncthreads();   // every thread has its carried-in value before the tile region is overwritten with the results
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) { P = Fr::add(a[j], fr_mul_ps(m, P)); lds_put(tile + threadIdx.x * 65 + j * 8, P); }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < FRSCAN_EPT; j++) {
    const uint32_t e = j * FRSCAN_THREADS + threadIdx.x;
    const uint64_t jj = base + e;
    if (jj < n) g_store(&dst[reverse ? n - 1 - jj : jj], lds_get(tile + (e >> 3) * 65 + (e & 7) * 8));
  }
}

#ifndef ZK_FRSCAN_DEVICE_ONLY
// ---- group::Curve::batch_normalize(&[G1], &mut [G1Affine]) [EXT-recalled halo2curves / group crate; create_proof turns each vector of
// projective commitments into affine points with it before they enter the transcript, SURVEY 8f-3]: out[i] = (X / Z^2, Y / Z^3), the
// identity (Z = 0) becomes (0, 0).  Montgomery's trick per 256-point tile: the Z coordinates are scanned from both ends through LDS
// (zeros replaced by one), lane 0 inverts the tile product once (Euclidean inverse), every thread finishes with two multiplications.
__global__ void __launch_bounds__(FRSCAN_THREADS) k_g1_batch_normalize(const g1_jac_t *__restrict__ in, g1_affine_t *__restrict__ out, uint64_t n) {
  __shared__ uint32_t buf[2 * FRSCAN_THREADS * 9];
  __shared__ uint32_t inv_total[8];
  const uint64_t i = (uint64_t)blockIdx.x * FRSCAN_THREADS + threadIdx.x;
  fe_t z = i < n ? g_load(&in[i].z) : Fq::one();
  const bool ident = Fq::is_zero(z);
  if (ident) z = Fq::one();
  fe_t total, total_r;
  const fe_t left = block_exclusive_mul_scan<false, true>(z, buf, total);
  const fe_t right = block_exclusive_mul_scan<true, true>(z, buf, total_r);
  if (threadIdx.x == 0) lds_put(inv_total, Fq::inv_sgcd(total));
  __syncthreads();
  if (i >= n) return;
  g1_affine_t r; r.x = Fq::zero(); r.y = Fq::zero();
  if (!ident) {
    const fe_t zi = fq_mul_ps(fq_mul_ps(lds_get(inv_total), left), right), zi2 = fq_sqr_ps(zi);
    r.x = fq_mul_ps(g_load(&in[i].x), zi2); r.y = fq_mul_ps(g_load(&in[i].y), fq_mul_ps(zi2, zi));
  }
  g_store(&out[i].x, r.x); g_store(&out[i].y, r.y);
}
// ---- narrow uploads (create_proof steps 2-3: witness columns are mostly zeros / bytes / 64-bit words; 32-byte Montgomery words waste the PCIe link)
// packed: src = n little-endian unsigned integers of W bytes (canonical values) -> dst[i] = value * R mod r (one Montgomery product with R^2 per element);
// a lane reads W bytes and writes 32: both sides coalesced
template <int W> __global__ void __launch_bounds__(256) k_expand_packed(fe_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v;
    if (W == 1) v = src[i]; else if (W == 2) v = ((const uint16_t *)src)[i]; else if (W == 4) v = ((const uint32_t *)src)[i]; else v = ((const uint64_t *)src)[i];
    fe_t c = Fr::zero(); c.l[0] = (uint32_t)v; c.l[1] = (uint32_t)(v >> 32);
    g_store(&dst[i], v ? Fr::from_canonical(c) : c);
  }
}
// sparse: dst (n cells) is zero-filled by the caller; dst[idx[j]] = vals[j] (32-byte Montgomery words, as they sit in the caller's column).  An index >= n never reaches memory
// (the pairs come from the caller's host code; the C-ABI documents that such a pair is dropped)
__global__ void __launch_bounds__(256) k_scatter_fr(fe_t *__restrict__ dst, uint64_t n, const uint32_t *__restrict__ idx, const fe_t *__restrict__ vals, uint64_t count) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) { const uint32_t i = idx[j]; if (i < n) g_store(&dst[i], g_load(&vals[j])); }
}
#endif  // ZK_FRSCAN_DEVICE_ONLY
#endif  // __HIPCC__
}  // namespace zk
