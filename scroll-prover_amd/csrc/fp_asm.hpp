// fp_asm.hpp -- product-scanning (FIPS) Montgomery multiplier tuned for the gfx950 ISA.
//
// Why: hipcc turns the textbook CIOS loop of fp.hpp into mad_u64_u32 + v_lshl_add_u64 + 2 x v_mov per
// limb product (~540 VALU instructions per field multiplication).  CDNA4's v_mad_u64_u32 is a VOP3B
// instruction with a carry-out SGPR, so a column accumulator {acc64, ex32} can absorb one 32x32 product in
// exactly two instructions (v_mad_u64_u32 ; v_addc_co_u32) with no register shuffling: ~300 instructions
// per multiplication.  Squaring uses a pre-doubled operand (a < 2^254 so 2a fits 8 limbs): 36 products
// instead of 64 in the a*a half.
//
// The host build of the same functions (plain C, for tests/hostcheck) uses 128-bit integers in `mac`.
#pragma once
#include "fp.hpp"
#include "g1.hpp"
#include "fp_asm_gen.inc"

namespace zk {

// {ex:acc} += a * b      (96-bit column accumulator)
ZK_HD void mac(uint64_t &acc, uint32_t &ex, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ex) : "v"(a), "v"(b) : "vcc");
#else
  unsigned __int128 s = (unsigned __int128)acc + (unsigned __int128)a * b; acc = (uint64_t)s; ex += (uint32_t)(s >> 64);
#endif
}
// same, with a wave-uniform second operand held in an SGPR (modulus limbs)
ZK_HD void mac_s(uint64_t &acc, uint32_t &ex, uint32_t a, uint32_t b_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ex) : "v"(a), "s"(b_uniform) : "vcc");
#else
  mac(acc, ex, a, b_uniform);
#endif
}
ZK_HD void col_shift(uint64_t &acc, uint32_t &ex) { acc = (acc >> 32) | ((uint64_t)ex << 32); ex = 0; }

// Montgomery reduction columns shared by mul and sqr: P-terms for column k given m[0..]
template <class P> ZK_HD fe_t mont_mul_ps(const fe_t &a, const fe_t &b) {
  uint64_t acc = 0; uint32_t ex = 0; uint32_t m[8]; fe_t r;
#if defined(__HIP_DEVICE_COMPILE__)
  ZK_MONT_MUL_COLUMNS
  return Fp<P>::reduce_once(r);
#else
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac(acc, ex, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac_s(acc, ex, m[i], P::mod(k - i));
    m[k] = (uint32_t)acc * P::INV;
    mac_s(acc, ex, m[k], P::mod(0));
    col_shift(acc, ex);
  }
#pragma unroll
  for (int k = 8; k < 15; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac(acc, ex, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac_s(acc, ex, m[i], P::mod(k - i));
    r.l[k - 8] = (uint32_t)acc;
    col_shift(acc, ex);
  }
  r.l[7] = (uint32_t)acc;  // value < 2m < 2^255: nothing above
  return Fp<P>::reduce_once(r);
#endif
}
template <class P> ZK_HD fe_t mont_sqr_ps(const fe_t &a) {
  // 2 * sum_{i<j} a_i a_j 2^(32(i+j)) = sum_i a_i * (2 * (a with limbs <= i cleared)).  Limb j of that doubled
  // upper part is a2[j] = (a_j << 1 | a_{j-1} >> 31) for j > i + 1 and a2lo[j] = a_j << 1 for j == i + 1
  // (the bit shifted in from a_i must not be counted); a < 2^254 so nothing spills into a ninth limb.
  uint32_t a2[8], a2lo[8];
  a2[0] = a2lo[0] = a.l[0] << 1;
#pragma unroll
  for (int i = 1; i < 8; i++) { a2lo[i] = a.l[i] << 1; a2[i] = a2lo[i] | (a.l[i - 1] >> 31); }
  uint64_t acc = 0; uint32_t ex = 0; uint32_t m[8]; fe_t r;
#if defined(__HIP_DEVICE_COMPILE__)
  ZK_MONT_SQR_COLUMNS
  return Fp<P>::reduce_once(r);
#else
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) mac(acc, ex, a.l[i], (k - i == i + 1) ? a2lo[k - i] : a2[k - i]);
    if ((k & 1) == 0) mac(acc, ex, a.l[k / 2], a.l[k / 2]);
#pragma unroll
    for (int i = 0; i < k; i++) mac_s(acc, ex, m[i], P::mod(k - i));
    m[k] = (uint32_t)acc * P::INV;
    mac_s(acc, ex, m[k], P::mod(0));
    col_shift(acc, ex);
  }
#pragma unroll
  for (int k = 8; k < 15; k++) {
#pragma unroll
    for (int i = k - 7; 2 * i < k; i++) mac(acc, ex, a.l[i], (k - i == i + 1) ? a2lo[k - i] : a2[k - i]);
    if ((k & 1) == 0) mac(acc, ex, a.l[k / 2], a.l[k / 2]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac_s(acc, ex, m[i], P::mod(k - i));
    r.l[k - 8] = (uint32_t)acc;
    col_shift(acc, ex);
  }
  r.l[7] = (uint32_t)acc;
  return Fp<P>::reduce_once(r);
#endif
}

ZK_HD fe_t fq_mul_ps(const fe_t &a, const fe_t &b) { return mont_mul_ps<FqP>(a, b); }
ZK_HD fe_t fq_sqr_ps(const fe_t &a) { return mont_sqr_ps<FqP>(a); }
ZK_HD fe_t fr_mul_ps(const fe_t &a, const fe_t &b) { return mont_mul_ps<FrP>(a, b); }
ZK_HD fe_t fr_sqr_ps(const fe_t &a) { return mont_sqr_ps<FrP>(a); }

// XYZZ mixed addition on the tuned multiplier (same formulas and special cases as g1_xyzz_madd)
ZK_HD g1_xyzz_t g1_xyzz_dbl_affine_ps(const g1_affine_t &p) {
  fe_t U = Fq::dbl(p.y), V = fq_sqr_ps(U), W = fq_mul_ps(U, V), S = fq_mul_ps(p.x, V);
  fe_t M = fq_sqr_ps(p.x); M = Fq::add(Fq::dbl(M), M);
  g1_xyzz_t r;
  r.x = Fq::sub(fq_sqr_ps(M), Fq::dbl(S));
  r.y = Fq::sub(fq_mul_ps(M, Fq::sub(S, r.x)), fq_mul_ps(W, p.y));
  r.zz = V; r.zzz = W;
  return r;
}
ZK_HD g1_xyzz_t g1_xyzz_dbl_ps(const g1_xyzz_t &p) {
  if (g1_xyzz_is_identity(p)) return p;
  fe_t U = Fq::dbl(p.y), V = fq_sqr_ps(U), W = fq_mul_ps(U, V), S = fq_mul_ps(p.x, V);
  fe_t M = fq_sqr_ps(p.x); M = Fq::add(Fq::dbl(M), M);
  g1_xyzz_t r;
  r.x = Fq::sub(fq_sqr_ps(M), Fq::dbl(S));
  r.y = Fq::sub(fq_mul_ps(M, Fq::sub(S, r.x)), fq_mul_ps(W, p.y));
  r.zz = fq_mul_ps(V, p.zz); r.zzz = fq_mul_ps(W, p.zzz);
  return r;
}
ZK_HD void g1_xyzz_madd_ps(g1_xyzz_t &acc, const g1_affine_t &q) {
  if (g1_affine_is_identity(q)) return;
  if (g1_xyzz_is_identity(acc)) { acc.x = q.x; acc.y = q.y; acc.zz = Fq::one(); acc.zzz = Fq::one(); return; }
  fe_t U2 = fq_mul_ps(q.x, acc.zz), S2 = fq_mul_ps(q.y, acc.zzz);
  fe_t Pd = Fq::sub(U2, acc.x), Rd = Fq::sub(S2, acc.y);
  if (Fq::is_zero(Pd)) {
    if (Fq::is_zero(Rd)) acc = g1_xyzz_dbl_affine_ps(q); else acc = g1_xyzz_identity();
    return;
  }
  fe_t PP = fq_sqr_ps(Pd), PPP = fq_mul_ps(Pd, PP), Q = fq_mul_ps(acc.x, PP);
  fe_t X3 = Fq::sub(Fq::sub(fq_sqr_ps(Rd), PPP), Fq::dbl(Q));
  fe_t Y3 = Fq::sub(fq_mul_ps(Rd, Fq::sub(Q, X3)), fq_mul_ps(acc.y, PPP));
  acc.x = X3; acc.y = Y3; acc.zz = fq_mul_ps(acc.zz, PP); acc.zzz = fq_mul_ps(acc.zzz, PPP);
}
ZK_HD void g1_xyzz_add_ps(g1_xyzz_t &acc, const g1_xyzz_t &q) {
  if (g1_xyzz_is_identity(q)) return;
  if (g1_xyzz_is_identity(acc)) { acc = q; return; }
  fe_t U1 = fq_mul_ps(acc.x, q.zz), U2 = fq_mul_ps(q.x, acc.zz);
  fe_t S1 = fq_mul_ps(acc.y, q.zzz), S2 = fq_mul_ps(q.y, acc.zzz);
  fe_t Pd = Fq::sub(U2, U1), Rd = Fq::sub(S2, S1);
  if (Fq::is_zero(Pd)) {
    if (Fq::is_zero(Rd)) acc = g1_xyzz_dbl_ps(acc); else acc = g1_xyzz_identity();
    return;
  }
  fe_t PP = fq_sqr_ps(Pd), PPP = fq_mul_ps(Pd, PP), Q = fq_mul_ps(U1, PP);
  fe_t X3 = Fq::sub(Fq::sub(fq_sqr_ps(Rd), PPP), Fq::dbl(Q));
  fe_t Y3 = Fq::sub(fq_mul_ps(Rd, Fq::sub(Q, X3)), fq_mul_ps(S1, PPP));
  acc.x = X3; acc.y = Y3;
  acc.zz = fq_mul_ps(fq_mul_ps(acc.zz, q.zz), PP); acc.zzz = fq_mul_ps(fq_mul_ps(acc.zzz, q.zzz), PPP);
}

}  // namespace zk
