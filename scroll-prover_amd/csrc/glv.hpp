// glv.hpp -- the GLV endomorphism of BN254 G1 for the scalar multiples of the G1 DFT (g1fft.hpp: best_fft::<Fr, G1>, g_to_lagrange,
// ParamsKZG::downsize).  phi(x, y) = (beta x, y) acts as multiplication by lambda (lambda^2 + lambda + 1 = 0 mod r; lambda is halo2curves'
// Fr::ZETA), so a 254-bit scalar k = k1 + lambda k2 with |k1|, |k2| < 2^127 turns k P into k1 P + k2 phi(P): one joint double-and-add over
// 127 bits instead of 254 (Gallant, Lambert, Vanstone, CRYPTO 2001).  Constants: tools/gen_glv_constants.py, which derives them from r and p;
// tests/test_device_limb_code_on_host.py re-runs that derivation and holds this header's decomposition and scalar multiple (compiled for the
// host) to the same integers and to big-integer curve arithmetic.
#pragma once
#include "g1_29.hpp"

namespace zk {

struct GlvP {
  // short lattice basis of {(a, b): a + b lambda = 0 mod r}: (a1, b1) = (A1, -B1M), (a2, b2) = (A2, B2), det = +r
  ZK_HD static constexpr uint32_t a1(int i) { constexpr uint32_t m[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u}; return m[i]; }
  ZK_HD static constexpr uint32_t b1m(int i) { constexpr uint32_t m[2] = {0x94d213e3u, 0x89d32568u}; return m[i]; }   // |b1|, b1 < 0
  ZK_HD static constexpr uint32_t a2(int i) { constexpr uint32_t m[2] = {0x94d213e3u, 0x89d32568u}; return m[i]; }
  ZK_HD static constexpr uint32_t b2(int i) { constexpr uint32_t m[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u}; return m[i]; }
  // g1 = floor(2^256 b2 / r) (130 bits), g2 = floor(2^256 |b1| / r) (66 bits)
  ZK_HD static constexpr uint32_t g1(int i) { constexpr uint32_t m[5] = {0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x00000002u}; return m[i]; }
  ZK_HD static constexpr uint32_t g2(int i) { constexpr uint32_t m[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u}; return m[i]; }
  // beta * 2^261 mod p in 29-bit limbs (canonical): Montgomery factor of the 9 x 29-bit field
  ZK_HD static constexpr uint32_t beta29(int i) { constexpr uint32_t m[9] = {0x18ccb791u, 0x175b1c3au, 0xb83d6e2u, 0xe8ed071u, 0x1282bee2u, 0x4220e84u, 0x1fe4017fu, 0x15084d4au, 0x169119u}; return m[i]; }
};

// limbs [lo, lo + NO) of a * b  (a: NA limbs, b: NB limbs, 32-bit little endian); columns below `lo` only feed the carry
template <int NA, int NB, int LO, int NO, class FA, class FB> ZK_HD void glv_mul_window(uint32_t (&out)[NO], FA a, FB b) {
  uint64_t acc = 0, hi = 0;   // (hi, acc) = 96-bit column accumulator kept as two 64-bit halves: acc holds bits 0..63, hi the overflow count
#pragma unroll
  for (int k = 0; k < LO + NO; k++) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int j = k - i;
      if (j < 0 || j >= NB) continue;
      const uint64_t p = (uint64_t)a(i) * b(j);
      acc += p; hi += acc < p ? 1u : 0u;
    }
    if (k >= LO) out[k - LO] = (uint32_t)acc;
    acc = (acc >> 32) | (hi << 32); hi = 0;
  }
}

// k (canonical, < r) -> (|k1|, sign1, |k2|, sign2) with k = k1 + lambda k2 mod r and |k_i| < 2^127
ZK_HD void glv_decompose(const fe_t &k, uint32_t (&k1)[4], bool &neg1, uint32_t (&k2)[4], bool &neg2) {
  uint32_t c1[5], c2[3];
  glv_mul_window<8, 5, 8, 5>(c1, [&](int i) { return k.l[i]; }, [](int i) { return GlvP::g1(i); });   // (k g1) >> 256 < 2^128
  glv_mul_window<8, 3, 8, 3>(c2, [&](int i) { return k.l[i]; }, [](int i) { return GlvP::g2(i); });   // (k g2) >> 256 < 2^64
  // all arithmetic modulo 2^160, two's complement: the results lie in (-2^127, 2^127)
  uint32_t t1[5], t2[5], t3[5], t4[5];
  glv_mul_window<5, 4, 0, 5>(t1, [&](int i) { return c1[i]; }, [](int i) { return GlvP::a1(i); });     // c1 a1
  glv_mul_window<3, 2, 0, 5>(t2, [&](int i) { return c2[i]; }, [](int i) { return GlvP::a2(i); });     // c2 a2
  glv_mul_window<5, 2, 0, 5>(t3, [&](int i) { return c1[i]; }, [](int i) { return GlvP::b1m(i); });    // c1 |b1|
  glv_mul_window<3, 4, 0, 5>(t4, [&](int i) { return c2[i]; }, [](int i) { return GlvP::b2(i); });     // c2 b2
  uint32_t r1[5], r2[5];
  { uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint64_t d = (uint64_t)k.l[i] - t1[i] - borrow; r1[i] = (uint32_t)d; borrow = (d >> 63) & 1u; }
    borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint64_t d = (uint64_t)r1[i] - t2[i] - borrow; r1[i] = (uint32_t)d; borrow = (d >> 63) & 1u; }
    borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint64_t d = (uint64_t)t3[i] - t4[i] - borrow; r2[i] = (uint32_t)d; borrow = (d >> 63) & 1u; } }
  neg1 = (r1[4] >> 31) != 0; neg2 = (r2[4] >> 31) != 0;
  { uint32_t c = neg1 ? 1u : 0u, m = neg1 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint64_t s = (uint64_t)(r1[i] ^ m) + c; k1[i] = (uint32_t)s; c = (uint32_t)(s >> 32); } }
  { uint32_t c = neg2 ? 1u : 0u, m = neg2 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint64_t s = (uint64_t)(r2[i] ^ m) + c; k2[i] = (uint32_t)s; c = (uint32_t)(s >> 32); } }
}

// signed 2-bit digits {-1, 0, 1, 2} of a magnitude < 2^127, 16 per word (code 3 = -1 with a carry into the next digit)
ZK_HD void glv_recode(const uint32_t (&k)[4], uint32_t (&code)[4]) {
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    uint32_t out = 0;
    for (int j = 0; j < 16; j++) {
      uint32_t d = ((k[w] >> (2 * j)) & 3u) + carry;
      carry = d >= 3 ? 1u : 0u;
      d = d == 4 ? 0u : d;
      out |= d << (2 * j);
    }
    code[w] = out;
  }
  // magnitude < 2^127: digit 63 is 0 or 1 before the carry, at most 2 after it -- never a carry out of the top
}

// k P = k1 P + k2 phi(P) by one joint double-and-add over 64 signed 2-bit digits.  p: valid accumulator with TIGHT coordinates
// (g1_xyzz29_from_sat); k canonical (< r).  Tables: {P, 2P} and their images under phi share ZZ / ZZZ (phi only scales x), the signs of
// k1 / k2 and the digit -1 choose between y and its negative.
ZK_HD g1_xyzz29_t g1_xyzz29_mul_glv(const g1_xyzz29_t &p, const fe_t &k) {
  if (g1_xyzz29_is_identity(p)) return p;
  uint32_t k1[4], k2[4], code1[4], code2[4]; bool neg1, neg2;
  glv_decompose(k, k1, neg1, k2, neg2);
  glv_recode(k1, code1); glv_recode(k2, code2);
  fe29_t beta; for (int i = 0; i < 9; i++) beta.l[i] = GlvP::beta29(i);
  const g1_xyzz29_t p2 = g1_xyzz29_dbl(p);                     // x < 9.1 p, y < 5.4 p (limbs <= 2^29 + 8), zz / zzz tight
  const fe29_t bx1 = FQ29_MUL(p.x, beta), bx2 = FQ29_MUL(p2.x, beta);   // tight
  // -y is formed when a digit asks for it (8p - y, limbs <= 2^29 + 8: a multiplication operand of the addition): two fewer table entries in
  // registers keep the kernel at two waves per SIMD
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (int i = 63; i >= 0; i--) {
    acc = g1_xyzz29_dbl(g1_xyzz29_dbl(acc));
    uint32_t wa = code1[0], wb = code2[0];   // static indices + selects: a runtime-indexed register array would live in scratch memory
#pragma unroll
    for (int w = 1; w < 4; w++) { wa = (i >> 4) == w ? code1[w] : wa; wb = (i >> 4) == w ? code2[w] : wb; }
    const uint32_t da = (wa >> ((i & 15) * 2)) & 3u, db = (wb >> ((i & 15) * 2)) & 3u;
    if (da) {
      const bool two = da == 2, minus = (da == 3) != neg1;   // digit -1 or a negative k1, not both
      g1_xyzz29_t sel;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        sel.x.l[j] = two ? p2.x.l[j] : p.x.l[j];
        sel.y.l[j] = two ? p2.y.l[j] : p.y.l[j];
        sel.zz.l[j] = two ? p2.zz.l[j] : p.zz.l[j];
        sel.zzz.l[j] = two ? p2.zzz.l[j] : p.zzz.l[j];
      }
      { const fe29_t yn = Fq29::sub8(Fq29::zero(), sel.y);   // y tight or < 5.4 p, limbs <= 2^29 + 8
#pragma unroll
        for (int j = 0; j < 9; j++) sel.y.l[j] = minus ? yn.l[j] : sel.y.l[j]; }
      g1_xyzz29_add(acc, sel);
    }
    if (db) {
      const bool two = db == 2, minus = (db == 3) != neg2;
      g1_xyzz29_t sel;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        sel.x.l[j] = two ? bx2.l[j] : bx1.l[j];
        sel.y.l[j] = two ? p2.y.l[j] : p.y.l[j];
        sel.zz.l[j] = two ? p2.zz.l[j] : p.zz.l[j];
        sel.zzz.l[j] = two ? p2.zzz.l[j] : p.zzz.l[j];
      }
      { const fe29_t yn = Fq29::sub8(Fq29::zero(), sel.y);   // y tight or < 5.4 p, limbs <= 2^29 + 8
#pragma unroll
        for (int j = 0; j < 9; j++) sel.y.l[j] = minus ? yn.l[j] : sel.y.l[j]; }
      g1_xyzz29_add(acc, sel);
    }
  }
  return acc;
}

}  // namespace zk
