// g1_29.hpp -- XYZZ bucket accumulator on the 9 x 29-bit unsaturated field (fp29.hpp); used by k_msm_accumulate, whose
// time is entirely field multiplications.  Same formulas and exceptional cases as g1.hpp (madd-2008-s / mdbl-2008-s); what
// changes is the bookkeeping of lazy values.  Invariants of an accumulator between additions (p = Fq modulus):
//     x   limbs <= 2^29 + 8, value < 14 p          zz, zzz   tight (mul outputs), value < 1.1 p
//     y   limbs <= 2^30 - 2, value < 6.1 p         identity  <=> all limbs of zz are 0
// Bases arrive in the ABI form (8 x 32, R = 2^256) and are re-sliced on the fly (from_sat: value < 2^259, limbs < 2^29:
// only ever used as a multiplication operand).  Bounds of every intermediate are written next to it.
#pragma once
#include "fp29.hpp"
#include "g1.hpp"

namespace zk {

struct g1_xyzz29_t { fe29_t x, y, zz, zzz; };

ZK_HD g1_xyzz29_t g1_xyzz29_identity() { g1_xyzz29_t r; r.x = Fq29::zero(); r.y = Fq29::zero(); r.zz = Fq29::zero(); r.zzz = Fq29::zero(); return r; }
ZK_HD bool g1_xyzz29_is_identity(const g1_xyzz29_t &p) { uint32_t o = 0; for (int i = 0; i < 9; i++) o |= p.zz.l[i]; return o == 0; }

// 64p with limbs >= 2^29 - 1 (top limb 0xc19139c > 2^27): -y for a from_sat() operand (limbs < 2^29, value < 2^259 = 54.4 p)
ZK_HD fe29_t fq29_neg_loaded(const fe29_t &y) {
  constexpr uint32_t c[9] = {0x3f3f51c0u, 0x21182dafu, 0x3ca8d3c1u, 0x3548b437u, 0x21765e04u, 0x36d0302au, 0x29b85044u, 0x37098d00u, 0xc19139bu};
  fe29_t r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c[i] - y.l[i];
  return r;   // limbs < 2^30, value < 64 p
}

// the full addition / doubling below (reduction tail of the MSM, DFT over G1 points) take the chained multiplier on the device as well
#ifndef ZK_G1_29_CHAIN
#define ZK_G1_29_CHAIN true
#endif
#define FQ29_MUL(a, b) Fq29::mul_t<ZK_G1_29_CHAIN>(a, b)
#define FQ29_SQR(a) Fq29::sqr_t<ZK_G1_29_CHAIN>(a)

// 2 * (affine point) for tight coordinates xt, yt (< 1.1 p): mdbl-2008-s
ZK_HD g1_xyzz29_t g1_xyzz29_dbl_affine(const fe29_t &xt, const fe29_t &yt) {
  const fe29_t U = Fq29::dbl(yt);                         // limbs <= 2^30 - 2, < 2.2 p
  const fe29_t V = FQ29_SQR(U), W = FQ29_MUL(U, V), S = FQ29_MUL(xt, V);   // tight, < 1.1 p
  const fe29_t xx = FQ29_SQR(xt);
  const fe29_t M = Fq29::carry(Fq29::add(Fq29::dbl(xx), xx));                  // 3 x^2, limbs <= 2^29 + 8, < 3.3 p
  g1_xyzz29_t r;
  r.x = Fq29::sub8(FQ29_SQR(M), Fq29::dbl(S));           // < 1.1 p + 8 p
  const fe29_t t = Fq29::sub16(S, r.x);                    // < 17.1 p
  r.y = Fq29::sub4(FQ29_MUL(M, t), FQ29_MUL(W, yt));     // < 1.4 p + 4 p
  r.zz = V; r.zzz = W;
  return r;
}

// acc += (+-) q, q in the ABI form.  madd-2008-s.  CHAIN: limb products as explicitly chained v_mad (fp29.hpp mac_*), bit-identical.
#ifndef ZK_MADD_CHAIN_DEFAULT
#define ZK_MADD_CHAIN_DEFAULT false
#endif
template <bool FUSED_Y3 = true, bool CHAIN = ZK_MADD_CHAIN_DEFAULT> ZK_HD void g1_xyzz29_madd_core(g1_xyzz29_t &acc, const fe29_t &x2, const fe29_t &y2, bool y2_needs_normalise);
template <bool FUSED_Y3 = true, bool CHAIN = ZK_MADD_CHAIN_DEFAULT> ZK_HD void g1_xyzz29_madd(g1_xyzz29_t &acc, const g1_affine_t &q, bool negate) {
  if (g1_affine_is_identity(q)) return;
  const fe29_t x2 = Fq29::from_sat(q.x);
  fe29_t y2 = Fq29::from_sat(q.y);
  if (negate) y2 = fq29_neg_loaded(y2);                    // limbs < 2^30, value < 64 p: multiplication operand only
  g1_xyzz29_madd_core<FUSED_Y3, CHAIN>(acc, x2, y2, negate);
}
// the same addition for an addend that already sits in 29-bit limbs (x2, y2: limbs < 2^30, value < 64 p, not the identity)
template <bool FUSED_Y3, bool CHAIN> ZK_HD void g1_xyzz29_madd_core(g1_xyzz29_t &acc, const fe29_t &x2, const fe29_t &y2, bool y2_needs_normalise) {
  if (g1_xyzz29_is_identity(acc)) {
    // first point of a bucket: taken by every lane at a different iteration (divergent), so keep it multiplication-free
    acc.x = Fq29::reduce_small(x2); acc.y = Fq29::reduce_small(y2_needs_normalise ? Fq29::normalise(y2) : y2);   // tight, < 2p
    acc.zz = Fq29::one(); acc.zzz = Fq29::one();
    return;
  }
  const fe29_t U2 = Fq29::mul_t<CHAIN>(x2, acc.zz), S2 = Fq29::mul_t<CHAIN>(y2, acc.zzz);      // tight, < 1.2 p
  const fe29_t Pd = Fq29::sub16(U2, acc.x);                                    // < 18 p
  const fe29_t Rd = Fq29::sub8(S2, acc.y);                                     // < 10 p
  const fe29_t PP = Fq29::sqr_t<CHAIN>(Pd);                                    // < 3 p
  const fe29_t ZZ3 = Fq29::mul_t<CHAIN>(acc.zz, PP);                           // < 1.1 p : zero iff Pd == 0 (acc.zz != 0)
  if (Fq29::is_zero_tight(ZZ3)) {
    // q == +-acc: doubling or annihilation (rare; taken by repeated / opposite points inside one bucket)
    const fe29_t one = Fq29::one();
    if (Fq29::is_zero_tight(Fq29::mul(Rd, one))) acc = g1_xyzz29_dbl_affine(Fq29::mul(x2, one), Fq29::mul(y2, one));
    else acc = g1_xyzz29_identity();
    return;
  }
  const fe29_t PPP = Fq29::mul_t<CHAIN>(Pd, PP);                               // < 1.4 p
  const fe29_t Q = Fq29::mul_t<CHAIN>(acc.x, PP);                              // < 1.3 p
  const fe29_t X3 = Fq29::sub4_8(Fq29::sqr_t<CHAIN>(Rd), PPP, Fq29::dbl(Q));          // (1.6 + 4 + 8) p = 13.6 p, one carry
  // Y3 = Rd (Q - X3) - Y1 PPP: both products under ONE Montgomery reduction (signed column accumulator): (1.02 - 0.05 .. ) p + p + [0, p) < 3.1 p
  const fe29_t Y3 = FUSED_Y3 ? Fq29::mul_sub_t<CHAIN>(Rd, Fq29::sub16(Q, X3), acc.y, PPP)
                             : Fq29::sub4(Fq29::mul_t<CHAIN>(Rd, Fq29::sub16(Q, X3)), Fq29::mul_t<CHAIN>(acc.y, PPP));   // unfused form: < 2.1 p + 4 p
  acc.x = X3; acc.y = Y3; acc.zz = ZZ3; acc.zzz = Fq29::mul_t<CHAIN>(acc.zzz, PPP);
}

// 2 * acc for an accumulator under the invariants above (dbl-2008-s-1).  Output: x < 9.1 p (limbs <= 2^29 + 8), y < 5.4 p
// (limbs <= 2^29 + 8), zz / zzz tight -- again a valid accumulator.
ZK_HD g1_xyzz29_t g1_xyzz29_dbl(const g1_xyzz29_t &a) {
  if (g1_xyzz29_is_identity(a)) return a;
  const fe29_t xt = Fq29::reduce_small(Fq29::normalise(a.x));   // tight, < 2 p (x^2 of a 14 p value would leave the < 2 p output range)
  const fe29_t yc = Fq29::carry(a.y);                           // limbs <= 2^29 + 2, value < 6.1 p
  const fe29_t U = Fq29::dbl(yc);                               // limbs <= 2^30 + 4, < 12.2 p   (U^2 = 149 p^2 < 2^261 p = 168 p^2)
  const fe29_t V = FQ29_SQR(U), W = FQ29_MUL(U, V), S = FQ29_MUL(xt, V);   // tight, < 1.9 p / 1.2 p / 1.1 p
  const fe29_t xx = FQ29_SQR(xt);
  const fe29_t M = Fq29::carry(Fq29::add(Fq29::dbl(xx), xx));                  // 3 x^2, limbs <= 2^29 + 8, < 3.3 p
  g1_xyzz29_t r;
  r.x = Fq29::sub8(FQ29_SQR(M), Fq29::dbl(S));                 // < 1.1 p + 8 p
  const fe29_t t = Fq29::sub16(S, r.x);                          // < 17.1 p
  r.y = Fq29::sub4(FQ29_MUL(M, t), FQ29_MUL(W, yc));           // < 1.4 p + 4 p
  r.zz = FQ29_MUL(V, a.zz); r.zzz = FQ29_MUL(W, a.zzz);
  return r;
}

// acc += q, both accumulators under the invariants above (add-2008-s).  The loose coordinates only ever enter multiplications
// (U1 = X1 ZZ2, S1 = Y1 ZZZ2, ...), after which the body is the mixed addition's with (U1, S1) in the place of (X1, Y1): same
// expression shapes, same or tighter bounds.  12 M + 2 S.
ZK_HD void g1_xyzz29_add(g1_xyzz29_t &acc, const g1_xyzz29_t &q) {
  if (g1_xyzz29_is_identity(q)) return;
  if (g1_xyzz29_is_identity(acc)) { acc = q; return; }
  const fe29_t U1 = FQ29_MUL(acc.x, q.zz), S1 = FQ29_MUL(acc.y, q.zzz);      // tight, < 1.1 p
  const fe29_t U2 = FQ29_MUL(q.x, acc.zz), S2 = FQ29_MUL(q.y, acc.zzz);      // tight, < 1.1 p
  const fe29_t Pd = Fq29::sub4(U2, U1);                                        // < 5.1 p
  const fe29_t Rd = Fq29::sub4(S2, S1);                                        // < 5.1 p
  const fe29_t PP = FQ29_SQR(Pd);                                             // < 1.2 p
  const fe29_t ZZt = FQ29_MUL(acc.zz, PP);                                    // zero iff Pd == 0 (both zz != 0)
  if (Fq29::is_zero_tight(ZZt)) {
    // q == +-acc: doubling or annihilation
    if (Fq29::is_zero_tight(FQ29_MUL(Rd, Fq29::one()))) acc = g1_xyzz29_dbl(acc);
    else acc = g1_xyzz29_identity();
    return;
  }
  const fe29_t PPP = FQ29_MUL(Pd, PP);                                        // < 1.1 p
  const fe29_t Q = FQ29_MUL(U1, PP);                                          // < 1.1 p
  const fe29_t X3 = Fq29::sub4_8(FQ29_SQR(Rd), PPP, Fq29::dbl(Q));          // (1.2 + 4 + 8) p = 13.2 p, one carry
  const fe29_t Y3 = Fq29::mul_sub_t<ZK_G1_29_CHAIN>(Rd, Fq29::sub16(Q, X3), S1, PPP);           // as in the mixed addition, with the tight S1 for Y1
  acc.x = X3; acc.y = Y3;
  acc.zz = FQ29_MUL(ZZt, q.zz); acc.zzz = FQ29_MUL(FQ29_MUL(acc.zzz, PPP), q.zzz);
}

// accumulator -> the saturated XYZZ record the reduction kernels consume (R = 2^256 Montgomery, fully reduced)
ZK_HD g1_xyzz_t g1_xyzz29_to_sat(const g1_xyzz29_t &a) {
  g1_xyzz_t r;
  if (g1_xyzz29_is_identity(a)) return g1_xyzz_identity();
  r.x = Fq29::to_sat(a.x); r.y = Fq29::to_sat(a.y); r.zz = Fq29::to_sat(a.zz); r.zzz = Fq29::to_sat(a.zzz);
  return r;
}

}  // namespace zk
