// g1fft.hpp -- DFT over G1 points: best_fft::<Fr, G1> of halo2_proofs [EXT-recalled src/arithmetic.rs], the transform behind
// g_to_lagrange / ParamsKZG::downsize [REF integration/tests/integration.rs:12-22; SURVEY 8a row a3, 8f-2]:
//     a'[i] = sum_j omega^(ij) a[j]      (natural order in, natural order out, no scaling)
//
// Round 2: the butterflies run on the 9 x 29-bit field with the chained multiplier and the GLV endomorphism (g1_xyzz29_mul_scalar below:
// k P = k1 P + k2 phi(P) with 127-bit k1, k2); the work array keeps the saturated XYZZ records.
// Every butterfly costs one 254-bit scalar multiple of a point (~4000 field multiplications) against 192 B of traffic, so the kernels are
// plain radix-2 stages over a work array of XYZZ points in HBM -- there is nothing for LDS tiling to win.  The points are permuted into
// bit-reversed order on load, the stages then run decimation-in-time exactly like the serial reference:
//     stage s (m = 2^s):  t = w^j * a[g + j + m];  a[g + j + m] = a[g + j] - t;  a[g + j] += t,   w = omega^(n / 2m)
// Butterflies with j == 0 (all of stage 0) skip the scalar multiple.
#pragma once
#include "fp_asm.hpp"
#include "g1.hpp"
#include "g1_29.hpp"
#include "glv.hpp"

namespace zk {
#ifdef __HIPCC__

// k * p for a canonical (non-Montgomery) 256-bit k: fixed 2-bit windows from the top, table {p, 2p, 3p} in registers.  Uniform schedule
// across the wavefront (the digit only selects the table entry), the exceptional cases are handled inside g1_xyzz_add_ps.
__device__ __noinline__ g1_xyzz_t g1_xyzz_mul_fr(const g1_xyzz_t &p, const fe_t &k) {
  if (g1_xyzz_is_identity(p)) return p;
  const g1_xyzz_t p2 = g1_xyzz_dbl_ps(p);
  g1_xyzz_t p3 = p2; g1_xyzz_add_ps(p3, p);
  g1_xyzz_t acc = g1_xyzz_identity();
  for (int i = 126; i >= 0; i--) {   // r < 2^254: digit 127 is always zero
    acc = g1_xyzz_dbl_ps(g1_xyzz_dbl_ps(acc));
    const uint32_t d = (k.l[i >> 4] >> ((i & 15) * 2)) & 3u;
    if (d) {
      g1_xyzz_t sel;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        sel.x.l[j] = d == 1 ? p.x.l[j] : d == 2 ? p2.x.l[j] : p3.x.l[j];
        sel.y.l[j] = d == 1 ? p.y.l[j] : d == 2 ? p2.y.l[j] : p3.y.l[j];
        sel.zz.l[j] = d == 1 ? p.zz.l[j] : d == 2 ? p2.zz.l[j] : p3.zz.l[j];
        sel.zzz.l[j] = d == 1 ? p.zzz.l[j] : d == 2 ? p2.zzz.l[j] : p3.zzz.l[j];
      }
      g1_xyzz_add_ps(acc, sel);
    }
  }
  return acc;
}

// The same scalar multiple on the 9 x 29-bit field with the chained multiplier (1.4x the saturated multiplier's rate): signed 2-bit
// digits {-1, 0, 1, 2} (a digit 3 becomes -1 with a carry into the next one), so the table is {p, 2p} plus a negated y -- 72 registers
// less than {p, 2p, 3p}.  p: valid accumulator with TIGHT coordinates (from g1_xyzz29_from_sat); result: valid accumulator.
__device__ __forceinline__ g1_xyzz29_t g1_xyzz29_from_sat(const g1_xyzz_t &p) {
  g1_xyzz29_t r;
  if (g1_xyzz_is_identity(p)) return g1_xyzz29_identity();
  r.x = Fq29::reduce_small(Fq29::from_sat(p.x)); r.y = Fq29::reduce_small(Fq29::from_sat(p.y));
  r.zz = Fq29::reduce_small(Fq29::from_sat(p.zz)); r.zzz = Fq29::reduce_small(Fq29::from_sat(p.zzz));
  return r;
}
__device__ __noinline__ g1_xyzz29_t g1_xyzz29_mul_fr(const g1_xyzz29_t &p, const fe_t &k) {
  if (g1_xyzz29_is_identity(p)) return p;
  uint32_t code[8]; uint32_t carry = 0;   // 16 digits per word, 2 bits each: 0, 1, 2, or 3 meaning -1
#pragma unroll
  for (int w = 0; w < 8; w++) {
    uint32_t out = 0;
    for (int j = 0; j < 16; j++) {
      uint32_t d = ((k.l[w] >> (2 * j)) & 3u) + carry;
      carry = d >= 3 ? 1u : 0u;
      d = d == 4 ? 0u : d;               // 3 stays 3 (= -1, carry 1), 4 = 0 with carry 1
      out |= d << (2 * j);
    }
    code[w] = out;
  }
  // k < r < 2^254: digit 127 is 0 or (after a carry) 1, never a carry out of the top
  const g1_xyzz29_t p2 = g1_xyzz29_dbl(p);
  fe29_t yneg = Fq29::sub4(Fq29::zero(), p.y);           // 4p - y: limbs <= 2^29 + 8, value < 4p (p.y tight)
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (int i = 127; i >= 0; i--) {
    acc = g1_xyzz29_dbl(g1_xyzz29_dbl(acc));
    uint32_t word = code[0];               // static indices + selects: a runtime-indexed register array would live in scratch memory
#pragma unroll
    for (int w = 1; w < 8; w++) word = (i >> 4) == w ? code[w] : word;
    const uint32_t d = (word >> ((i & 15) * 2)) & 3u;
    if (d) {
      g1_xyzz29_t sel;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        sel.x.l[j] = d == 2 ? p2.x.l[j] : p.x.l[j];
        sel.y.l[j] = d == 2 ? p2.y.l[j] : (d == 3 ? yneg.l[j] : p.y.l[j]);
        sel.zz.l[j] = d == 2 ? p2.zz.l[j] : p.zz.l[j];
        sel.zzz.l[j] = d == 2 ? p2.zzz.l[j] : p.zzz.l[j];
      }
      g1_xyzz29_add(acc, sel);
    }
  }
  return acc;
}

// GLV form of the same multiple (glv.hpp): k = k1 + lambda k2, one joint double-and-add over 64 signed 2-bit digits with the tables
// {P, 2P} and their images under phi(x, y) = (beta x, y) -- 128 doublings + ~96 additions instead of 256 + ~96 (-31 % of the field
// multiplications).  ZK_G1FFT_GLV=false keeps the plain ladder for A/B runs.
#ifndef ZK_G1FFT_GLV
#define ZK_G1FFT_GLV true
#endif
__device__ __forceinline__ g1_xyzz29_t g1_xyzz29_mul_scalar(const g1_xyzz29_t &p, const fe_t &k) { return ZK_G1FFT_GLV ? g1_xyzz29_mul_glv(p, k) : g1_xyzz29_mul_fr(p, k); }

__device__ __forceinline__ g1_xyzz_t g1fft_load_xyzz(const g1_xyzz_t *p) {
  g1_xyzz_t r; r.x = g_load(&p->x); r.y = g_load(&p->y); r.zz = g_load(&p->zz); r.zzz = g_load(&p->zzz); return r;
}
__device__ __forceinline__ void g1fft_store_xyzz(g1_xyzz_t *p, const g1_xyzz_t &v) {
  g_store(&p->x, v.x); g_store(&p->y, v.y); g_store(&p->zz, v.zz); g_store(&p->zzz, v.zzz);
}

// work[bitrev(i)] = in[i] as XYZZ.  JAC = 1: 96-byte Jacobian input (any representative); JAC = 0: 64-byte affine input.
template <int JAC> __global__ void __launch_bounds__(256) k_g1fft_load(const void *__restrict__ in, g1_xyzz_t *__restrict__ work, uint32_t log_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  const uint32_t r = log_n ? __brev(i) >> (32 - log_n) : 0;
  g1_xyzz_t v;
  if (JAC) {
    const g1_jac_t *src = static_cast<const g1_jac_t *>(in) + i;
    g1_jac_t q; q.x = g_load(&src->x); q.y = g_load(&src->y); q.z = g_load(&src->z);
    if (Fq::is_zero(q.z)) v = g1_xyzz_identity();
    else { v.x = q.x; v.y = q.y; v.zz = fq_sqr_ps(q.z); v.zzz = fq_mul_ps(v.zz, q.z); }
  } else {
    const g1_affine_t *src = static_cast<const g1_affine_t *>(in) + i;
    g1_affine_t q; q.x = g_load(&src->x); q.y = g_load(&src->y);
    v = g1_xyzz_from_affine(q);
  }
  g1fft_store_xyzz(&work[r], v);
}

// one decimation-in-time stage; tw[i] = omega^i (Montgomery), i < n / 2
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_g1fft_stage(g1_xyzz_t *__restrict__ work, const fe_t *__restrict__ tw, uint32_t log_n, uint32_t s) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (1u << log_n) / 2) return;
  const uint32_t m = 1u << s, j = t & (m - 1), ia = ((t >> s) << (s + 1)) + j, ib = ia + m;
  g1_xyzz29_t wb = g1_xyzz29_from_sat(g1fft_load_xyzz(&work[ib]));
  if (j) {
    fe_t one_c = Fr::zero(); one_c.l[0] = 1;
    const fe_t k = fr_mul_ps(g_load(&tw[(uint64_t)j << (log_n - 1 - s)]), one_c);   // Montgomery -> canonical
    wb = g1_xyzz29_mul_scalar(wb, k);
  }
  // `a` is loaded AFTER the scalar multiple (inlined: no call, no scratch memory): 32 registers less across the ladder, the kernel stays at
  // two waves per SIMD without spilling
  const g1_xyzz_t a = g1fft_load_xyzz(&work[ia]);
  g1_xyzz29_t lo = g1_xyzz29_from_sat(a), hi = lo;
  g1_xyzz29_add(lo, wb);
  if (!g1_xyzz29_is_identity(wb)) wb.y = Fq29::sub8(Fq29::zero(), wb.y);   // -w b: 8p - y (accumulator invariant: y < 6.1 p, limbs <= 2^30 - 2); only ever a multiplication operand below
  g1_xyzz29_add(hi, wb);
  g1fft_store_xyzz(&work[ia], g1_xyzz29_to_sat(lo)); g1fft_store_xyzz(&work[ib], g1_xyzz29_to_sat(hi));
}

// out[i] = scale * work[i], normalised.  JAC = 1: Jacobian (x, y, R) / all-zero identity; JAC = 0: affine, identity (0, 0).
// has_scale == 0: no scalar multiple.  The scale travels by value (a kernel argument is copied at launch: the caller's host copy may be a temporary).
template <int JAC> __global__ void __launch_bounds__(256) k_g1fft_store(const g1_xyzz_t *__restrict__ work, void *__restrict__ out, uint32_t log_n, fe_t scale, int has_scale) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  g1_xyzz_t v = g1fft_load_xyzz(&work[i]);
  if (has_scale) {
    fe_t one_c = Fr::zero(); one_c.l[0] = 1;
    v = g1_xyzz29_to_sat(g1_xyzz29_mul_scalar(g1_xyzz29_from_sat(v), fr_mul_ps(scale, one_c)));
  }
  const g1_jac_t r = g1_xyzz_to_jac_normalised(v);
  if (JAC) { g1_jac_t *dst = static_cast<g1_jac_t *>(out) + i; g_store(&dst->x, r.x); g_store(&dst->y, r.y); g_store(&dst->z, r.z); }
  else { g1_affine_t *dst = static_cast<g1_affine_t *>(out) + i; g_store(&dst->x, r.x); g_store(&dst->y, r.y); }
}

#endif  // __HIPCC__
}  // namespace zk
