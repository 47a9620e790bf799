// ntt.hpp -- BN254 Fr number-theoretic transform for gfx950: LDS-tiled Cooley-Tukey in at most three
// global passes.  Stands in for halo2_proofs::arithmetic::best_fft (natural order in -> natural order out,
// a'[i] = sum_j a[j] w^(ij), no scaling) and the EvaluationDomain wrappers around it.
//
// Decomposition N = M1*M2*M3 (each M <= 2^10).  Level l < L ("strided pass"): every contiguous sub-problem of
// size S = M_l * T is viewed as an M_l x T matrix; a workgroup loads M_l rows x C adjacent columns (C*32 B
// contiguous per row -> coalesced), runs the size-M_l DFT of each column in LDS (radix-2 DIF, bit reversal
// undone on the way out), multiplies by the inter-level twiddle w_S^(col*k) and stores in place.  Level L
// ("final pass"): a workgroup loads C contiguous segments of M_L elements that differ in the FASTEST output
// digit, transforms them in LDS and scatters so that each store instruction writes C*32 B contiguous; this
// pass also applies the digit-reversal permutation, so it runs out of place (scratch -> destination).
// HBM traffic = 64 B per element per pass (2-3 passes; algorithmic minimum is one pass = 64*N bytes).
//
// Element layout in LDS: two 16-byte planes (lo = limbs 0..3, hi = limbs 4..7) so that consecutive lanes touch
// consecutive 16-B slots (conflict-free ds_read_b128 / ds_write_b128).
#pragma once
#include "fp_asm.hpp"

namespace zk {

struct NttLevel {
  uint32_t log_m;        // log2 of this level's DFT size
  uint32_t log_t;        // log2 of the column count T (stride between rows); S = M*T
  const fe_t *tw_m;      // w_M^i, i < M/2
  const fe_t *tw_s_lo;   // w_S^i,            i < 2^split
  const fe_t *tw_s_hi;   // w_S^(i * 2^split), i < S / 2^split
  uint32_t split;
};

ZK_HD uint32_t bitrev32(uint32_t x, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | ((x >> i) & 1); } return r;
#endif
}

#if defined(__HIPCC__)
__device__ __forceinline__ fe_t lds_get(const uint4 *lo, const uint4 *hi, uint32_t i) {
  uint4 a = lo[i], b = hi[i]; fe_t r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; return r;
}
__device__ __forceinline__ void lds_put(uint4 *lo, uint4 *hi, uint32_t i, const fe_t &v) {
  lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
// In-LDS radix-2 DIF over `cols` independent sequences of length M = 2^log_m.
// Element (m, c) lives at index m*sm + c*sc.  Output position m holds X[bitrev(m)].
__device__ __forceinline__ void lds_dif(uint4 *lo, uint4 *hi, uint32_t log_m, uint32_t log_c, uint32_t sm, uint32_t sc,
                                         const fe_t *__restrict__ tw_m, bool col_fast) {
  const uint32_t M = 1u << log_m, C = 1u << log_c, nbf = (M >> 1) << log_c;
  for (uint32_t s = 0; s < log_m; s++) {
    const uint32_t log_h = log_m - 1 - s, h = 1u << log_h;
    for (uint32_t b = threadIdx.x; b < nbf; b += blockDim.x) {
      uint32_t c, j;
      if (col_fast) { c = b & (C - 1); j = b >> log_c; } else { j = b & ((M >> 1) - 1); c = b >> (log_m - 1); }
      const uint32_t jl = j & (h - 1), i0 = ((j >> log_h) << (log_h + 1)) | jl;
      const uint32_t e0 = i0 * sm + c * sc, e1 = e0 + h * sm;
      fe_t u = lds_get(lo, hi, e0), v = lds_get(lo, hi, e1);
      fe_t sum = Fr::add(u, v), dif = Fr::sub(u, v);
      if (jl != 0) dif = fr_mul_ps(dif, g_load(&tw_m[jl << s]));   // w_{2h}^jl = w_M^(jl * M/(2h))
      lds_put(lo, hi, e0, sum); lds_put(lo, hi, e1, dif);
    }
    __syncthreads();
  }
}

// w_S^e from the two-level table
__device__ __forceinline__ fe_t twiddle_s(const NttLevel &L, uint32_t e) {
  fe_t a = g_load(&L.tw_s_lo[e & ((1u << L.split) - 1)]);
  const uint32_t eh = e >> L.split;
  if (eh == 0) return a;
  return fr_mul_ps(a, g_load(&L.tw_s_hi[eh]));
}

// ---- strided pass.  grid.x = (N / S) * (T / C) workgroups; dynamic LDS = 2 * 16 B * M * C.
// src_len: elements of src at index >= src_len read as zero (zero padding of coeff_to_extended);
// pre3: optional {f0, f1, f2}: element i of src is multiplied by pre3[i % 3] (distribute_powers_zeta).
__global__ void __launch_bounds__(1024) k_ntt_strided(const fe_t *__restrict__ src, fe_t *__restrict__ dst, NttLevel L, uint32_t log_c,
                                                      uint64_t src_len, const fe_t *__restrict__ pre3) {
  extern __shared__ uint4 lds[];
  const uint32_t M = 1u << L.log_m, C = 1u << log_c, tile = M << log_c;
  uint4 *lo = lds, *hi = lds + tile;
  const uint32_t cb_per_sub = 1u << (L.log_t - log_c);
  const uint64_t sub = blockIdx.x >> (L.log_t - log_c);
  const uint32_t cb = blockIdx.x & (cb_per_sub - 1);
  const uint64_t base = (sub << (L.log_m + L.log_t)) + ((uint64_t)cb << log_c);
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t c = e & (C - 1), m = e >> log_c;
    const uint64_t gi = base + ((uint64_t)m << L.log_t) + c;
    fe_t v;
    if (gi < src_len) { v = g_load(&src[gi]); if (pre3) { uint32_t r3 = (uint32_t)(gi % 3); if (r3) v = fr_mul_ps(v, g_load(&pre3[r3])); } }
    else v = Fr::zero();
    lds_put(lo, hi, e, v);
  }
  __syncthreads();
  lds_dif(lo, hi, L.log_m, log_c, C, 1, L.tw_m, true);
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t c = e & (C - 1), k = e >> log_c;
    fe_t v = lds_get(lo, hi, (bitrev32(k, L.log_m) << log_c) + c);
    const uint32_t col = (cb << log_c) + c;
    if (k != 0 && col != 0) v = fr_mul_ps(v, twiddle_s(L, col * k));
    g_store(&dst[base + ((uint64_t)k << L.log_t) + c], v);
  }
}

// ---- final pass.  Segment q (M contiguous elements at q*M) = prefix digits (k1, k2) with q = k1*B + k2;
// its outputs go to k1 + A*k2 + (N/M)*k.  grid.x = B * (A / C) workgroups: each takes C consecutive k1.
// post3: optional {f0,f1,f2}: output element i multiplied by post3[i % 3] (ifft divisor / leaving the coset).
__global__ void __launch_bounds__(1024) k_ntt_final(const fe_t *__restrict__ src, fe_t *__restrict__ dst, uint32_t log_m, uint32_t log_a, uint32_t log_b,
                                                    uint32_t log_c, const fe_t *__restrict__ tw_m, uint64_t src_len, const fe_t *__restrict__ pre3,
                                                    const fe_t *__restrict__ post3) {
  extern __shared__ uint4 lds[];
  const uint32_t M = 1u << log_m, C = 1u << log_c, seg = M + 1, tile = M << log_c;
  uint4 *lo = lds, *hi = lds + (seg << log_c);
  const uint32_t k2 = blockIdx.x & ((1u << log_b) - 1);
  const uint32_t k1_0 = (blockIdx.x >> log_b) << log_c;
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t m = e & (M - 1), c = e >> log_m;
    const uint64_t q = ((uint64_t)(k1_0 + c) << log_b) + k2, gi = (q << log_m) + m;
    fe_t v;
    if (gi < src_len) { v = g_load(&src[gi]); if (pre3) { uint32_t r3 = (uint32_t)(gi % 3); if (r3) v = fr_mul_ps(v, g_load(&pre3[r3])); } }
    else v = Fr::zero();
    lds_put(lo, hi, c * seg + m, v);
  }
  __syncthreads();
  lds_dif(lo, hi, log_m, log_c, 1, seg, tw_m, false);
  const uint32_t log_stride = log_a + log_b;  // N / M
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t c = e & (C - 1), k = e >> log_c;
    fe_t v = lds_get(lo, hi, c * seg + bitrev32(k, log_m));
    const uint64_t oi = (uint64_t)(k1_0 + c) + ((uint64_t)k2 << log_a) + ((uint64_t)k << log_stride);
    if (post3) v = fr_mul_ps(v, g_load(&post3[oi % 3]));
    g_store(&dst[oi], v);
  }
}

// out[i] = base^(i * step) for i < count   (twiddle tables; tiny)
__global__ void k_pow_table(fe_t *out, fe_t base, uint64_t step, uint32_t count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe_t b = step == 1 ? base : Fr::pow_u64(base, step);
  g_store(&out[i], Fr::pow_u64(b, i));
}
#endif  // __HIPCC__

}  // namespace zk
