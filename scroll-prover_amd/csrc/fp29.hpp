// fp29.hpp -- BN254 Fq / Fr in 9 x 29-bit UNSATURATED limbs for the compute-bound inner loops on gfx950.
//
// Why: on MI355X v_mad_u64_u32 issues in ~4.5 cycles but so does every carry instruction
// (v_addc_co_u32, v_lshl_add_u64), so in the saturated 8 x 32 form each limb product costs two 4-cycle instructions.
// With 29-bit limbs a 64-bit column accumulator absorbs all 18 products of a Montgomery column without overflow
// (9 * 2^30 * 2^30 + 9 * 2^29 * 2^29 + carry < 2^64), so a product is ONE v_mad_u64_u32 and the compiler schedules
// it freely (no inline asm, no VCC dependency): 162 + 9 multiplies instead of 128 + 128 carry ops + 8.
//
// Representation: value = sum l[i] * 2^(29 i), Montgomery radix R' = 2^261.
//   "tight": every limb < 2^29 (l[8] small), value < 2p      -- what mul/sqr return
//   "loose": limbs < 2^30.3, value < 2^259                     -- what add/sub return and mul/sqr accept
// The ABI (and HBM) keep the saturated 8 x 32, R = 2^256 form; from_sat()/to_sat() convert at load/store:
// x * 2^256 (stored) -> limbs of (stored << 5) = x * 2^261 (loose, no reduction needed).
#pragma once
#include "fp.hpp"

namespace zk {

struct fe29_t { uint32_t l[9]; };
constexpr uint32_t M29 = (1u << 29) - 1;

struct Fq29P {
  static constexpr uint32_t INV = 0x4866389u;  // -p^-1 mod 2^29
  ZK_HD static constexpr uint32_t mod(int i) { constexpr uint32_t m[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu}; return m[i]; }
  ZK_HD static constexpr uint32_t one(int i) { constexpr uint32_t m[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x14c0419u, 0xaa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x52ac7a8u, 0xdc836u}; return m[i]; }      // 2^261 mod p
  ZK_HD static constexpr uint32_t r256(int i) { constexpr uint32_t m[9] = {0x58f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0xe0a77u}; return m[i]; }   // 2^256 mod p (plain integer)
  // multiples of p written with "fat" limbs (2^29 resp. 2^30 borrowed into limbs 0..7) so that limb-wise a + C - b never goes negative
  ZK_HD static constexpr uint32_t fat29_4p(int i) { constexpr uint32_t m[9] = {0x21f3f51cu, 0x241182dau, 0x31ca8d3bu, 0x2b548b42u, 0x361765dfu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0xc19138u}; return m[i]; }   // limbs >= 2^29 - 1
  ZK_HD static constexpr uint32_t fat30_8p(int i) { constexpr uint32_t m[9] = {0x43e7ea38u, 0x482305b4u, 0x43951a76u, 0x56a91685u, 0x4c2ecbbeu, 0x56da0603u, 0x45370a06u, 0x52e1319eu, 0x1832271u}; return m[i]; }  // limbs >= 2^30 - 2
  ZK_HD static constexpr uint32_t fat30_16p(int i) { constexpr uint32_t m[9] = {0x47cfd470u, 0x50460b6au, 0x472a34eeu, 0x4d522d0cu, 0x585d977fu, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x30644e5u}; return m[i]; }
};
struct Fr29P {
  static constexpr uint32_t INV = 0xfffffffu;
  ZK_HD static constexpr uint32_t mod(int i) { constexpr uint32_t m[9] = {0x10000001u, 0x1f0fac9fu, 0xe5c2450u, 0x7d090f3u, 0x1585d283u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu}; return m[i]; }
  ZK_HD static constexpr uint32_t one(int i) { constexpr uint32_t m[9] = {0xfffff57u, 0x1ea70ab4u, 0x52c068bu, 0x17504f49u, 0xaa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x52ac7a8u, 0xdc836u}; return m[i]; }
  ZK_HD static constexpr uint32_t r256(int i) { constexpr uint32_t m[9] = {0xffffffbu, 0x4b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0xe0a77u}; return m[i]; }
  ZK_HD static constexpr uint32_t fat29_4p(int i) { constexpr uint32_t m[9] = {0x20000004u, 0x3c3eb27du, 0x39709142u, 0x3f4243ccu, 0x36174a0bu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0xc19138u}; return m[i]; }
  ZK_HD static constexpr uint32_t fat30_8p(int i) { constexpr uint32_t m[9] = {0x40000008u, 0x587d64fau, 0x52e12285u, 0x5e848799u, 0x4c2e9417u, 0x56da0603u, 0x45370a06u, 0x52e1319eu, 0x1832271u}; return m[i]; }
  ZK_HD static constexpr uint32_t fat30_16p(int i) { constexpr uint32_t m[9] = {0x40000010u, 0x50fac9f6u, 0x45c2450du, 0x5d090f35u, 0x585d2831u, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x30644e5u}; return m[i]; }
};

// Chained multiplier (A/B experiment).  Written as plain C++ the compiler sums every column's products in a fresh
// accumulator and joins it to the carried one with a 64-bit add (a shorter dependency chain, but one more 4-cycle instruction per column:
// 16 per multiplication, ~7 % of its issue slots).  mul_c / sqr_c / mul_sub_c issue the products of a column as one inline-asm block of
// chained v_mad instead; with three waves per SIMD the chain latency is hidden anyway.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI355_FP29_NO_CHAIN)
#define ZK_FP29_CHAIN 1
#include "fp29_asm_gen.inc"
#else
#define ZK_FP29_CHAIN 0
#endif

template <class P> struct Fp29 {
  ZK_HD static fe29_t zero() { fe29_t r; for (int i = 0; i < 9; i++) r.l[i] = 0; return r; }
  ZK_HD static fe29_t one() { fe29_t r; for (int i = 0; i < 9; i++) r.l[i] = P::one(i); return r; }

  // Montgomery product, product-scanning.  Inputs loose (max limb product < 2^60.6), output tight (< 2p when a*b < 2^261 p).
  ZK_HD static fe29_t mul(const fe29_t &a, const fe29_t &b) {
    uint64_t acc = 0; uint32_t m[9]; fe29_t r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::mod(k - i);
      m[k] = ((uint32_t)acc * P::INV) & M29;
      acc += (uint64_t)m[k] * P::mod(0);
      acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P::mod(k - i);
      r.l[k - 9] = (uint32_t)acc & M29;
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
  }
  // (a*b - c*d) * R'^-1 + p  in ONE Montgomery reduction (lazy reduction of a difference of products): the column accumulator is signed.
  // Limb bounds: all four operands <= 2^29 + 8 so that 9 + 9 + 9 products stay below 2^63.  Value: (ab - cd)/R' must lie in (-p, 6p).
  // Result: limbs <= 2^30 - 2 (no carry needed), value = (ab - cd)/R' + p + [0, p).
  ZK_HD static fe29_t mul_sub(const fe29_t &a, const fe29_t &b, const fe29_t &c, const fe29_t &d) {
    int64_t acc = 0; uint32_t m[9]; fe29_t r;
    int32_t nc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) nc[i] = -(int32_t)c.l[i];
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) { acc += (int64_t)(int32_t)a.l[i] * (int32_t)b.l[k - i]; acc += (int64_t)nc[i] * (int32_t)d.l[k - i]; }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (int64_t)(int32_t)m[i] * (int32_t)P::mod(k - i);
      m[k] = ((uint32_t)acc * P::INV) & M29;
      acc += (int64_t)(int32_t)m[k] * (int32_t)P::mod(0);
      acc >>= 29;   // exact: the low 29 bits are zero
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
      for (int i = k - 8; i < 9; i++) { acc += (int64_t)(int32_t)a.l[i] * (int32_t)b.l[k - i]; acc += (int64_t)nc[i] * (int32_t)d.l[k - i]; }
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc += (int64_t)(int32_t)m[i] * (int32_t)P::mod(k - i);
      r.l[k - 9] = ((uint32_t)acc & M29) + P::mod(k - 9);
      acc >>= 29;   // arithmetic shift: floor division keeps the limbs in [0, 2^29)
    }
    r.l[8] = (uint32_t)((int32_t)acc + (int32_t)P::mod(8));
    return r;
  }
  // the same three routines with the limb products of every column issued as ONE inline-asm block of chained v_mad (fp29_asm_gen.inc,
  // written by tools/gen_fp29_asm.py): bit-identical results; on the host they are the plain C++ routines
  ZK_HD static fe29_t mul_c(const fe29_t &a, const fe29_t &b) {
#if ZK_FP29_CHAIN
    uint64_t acc = 0, co; uint32_t m[9]; fe29_t r;
    ZK_FP29_MUL_COLUMNS
    (void)co;
    return r;
#else
    return mul(a, b);
#endif
  }
  ZK_HD static fe29_t sqr_c(const fe29_t &a) {
#if ZK_FP29_CHAIN
    uint64_t acc = 0, co; uint32_t m[9]; fe29_t r;
    uint32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
    ZK_FP29_SQR_COLUMNS
    (void)co;
    return r;
#else
    return sqr(a);
#endif
  }
  ZK_HD static fe29_t mul_sub_c(const fe29_t &a, const fe29_t &b, const fe29_t &c, const fe29_t &d) {
#if ZK_FP29_CHAIN
    int64_t acc = 0; uint64_t co; uint32_t m[9]; fe29_t r;
    int32_t nc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) nc[i] = -(int32_t)c.l[i];
    ZK_FP29_MULSUB_COLUMNS
    (void)co;
    return r;
#else
    return mul_sub(a, b, c, d);
#endif
  }
  template <bool C> ZK_HD static fe29_t mul_t(const fe29_t &a, const fe29_t &b) { return C ? mul_c(a, b) : mul(a, b); }
  template <bool C> ZK_HD static fe29_t sqr_t(const fe29_t &a) { return C ? sqr_c(a) : sqr(a); }
  template <bool C> ZK_HD static fe29_t mul_sub_t(const fe29_t &a, const fe29_t &b, const fe29_t &c, const fe29_t &d) { return C ? mul_sub_c(a, b, c, d) : mul_sub(a, b, c, d); }
  // same product with TWO independent column accumulators (even / odd terms): halves the dependent v_mad_u64_u32 chain
  // at the price of one 64-bit add per column; pays when few waves share a SIMD (experiment, see tools/microbench.hip)
  ZK_HD static fe29_t mul2(const fe29_t &a, const fe29_t &b) {
    uint64_t acc = 0; uint32_t m[9]; fe29_t r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      uint64_t e = acc, o = 0;
#pragma unroll
      for (int i = 0; i <= k; i++) { if (i & 1) o += (uint64_t)a.l[i] * b.l[k - i]; else e += (uint64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
      for (int i = 0; i < k; i++) { if (i & 1) e += (uint64_t)m[i] * P::mod(k - i); else o += (uint64_t)m[i] * P::mod(k - i); }
      acc = e + o;
      m[k] = ((uint32_t)acc * P::INV) & M29;
      acc += (uint64_t)m[k] * P::mod(0);
      acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
      uint64_t e = acc, o = 0;
#pragma unroll
      for (int i = k - 8; i < 9; i++) { if (i & 1) o += (uint64_t)a.l[i] * b.l[k - i]; else e += (uint64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
      for (int i = k - 8; i < 9; i++) { if (i & 1) e += (uint64_t)m[i] * P::mod(k - i); else o += (uint64_t)m[i] * P::mod(k - i); }
      acc = e + o;
      r.l[k - 9] = (uint32_t)acc & M29;
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
  }
  ZK_HD static fe29_t sqr(const fe29_t &a) {
    uint64_t acc = 0; uint32_t m[9]; fe29_t r;
    uint32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;   // limbs < 2^30.3 -> doubled < 2^31.3: a_i * 2a_j < 2^61.6, <= 4 such + 1 square per column
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
      if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::mod(k - i);
      m[k] = ((uint32_t)acc * P::INV) & M29;
      acc += (uint64_t)m[k] * P::mod(0);
      acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
      for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)a.l[i] * a2[k - i];
      if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P::mod(k - i);
      r.l[k - 9] = (uint32_t)acc & M29;
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
  }
  // one carry pass: limbs back below 2^29 + 8 (value unchanged); the top limb absorbs the last carry
  ZK_HD static fe29_t carry(const fe29_t &a) {
    fe29_t r; r.l[0] = a.l[0] & M29;
#pragma unroll
    for (int i = 1; i < 8; i++) r.l[i] = (a.l[i] & M29) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
  }
  // a + b: limb-wise, no carry (tight + tight -> loose)
  ZK_HD static fe29_t add(const fe29_t &a, const fe29_t &b) { fe29_t r; for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i]; return r; }
  // a - b + 4p:  b TIGHT (mul/sqr output: limbs < 2^29, value < 4p).                 result limbs <= 2^29 + 8
  ZK_HD static fe29_t sub4(const fe29_t &a, const fe29_t &b) {
    fe29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P::fat29_4p(i) - b.l[i];
    return carry(r);
  }
  // a - b + 8p:  b limbs <= 2^30 - 2 (carried values, doubled tight values), value(b) < 7.9 p
  ZK_HD static fe29_t sub8(const fe29_t &a, const fe29_t &b) {
    fe29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P::fat30_8p(i) - b.l[i];
    return carry(r);
  }
  // a - b - c + 12p in ONE pass and ONE carry (X3 = R^2 - PPP - 2Q of the point additions): b TIGHT as for sub4, c as for sub8; a limbs <= 2^29 + 8.
  // Limb-wise a + (4p + 8p fat limbs, < 2^31.3) - b - c < 2^31.5: no 32-bit overflow, never negative; after the carry limbs <= 2^29 + 8.
  ZK_HD static fe29_t sub4_8(const fe29_t &a, const fe29_t &b, const fe29_t &c) {
    fe29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (P::fat29_4p(i) + P::fat30_8p(i)) - b.l[i] - c.l[i];
    return carry(r);
  }
  // a - b + 16p: b limbs <= 2^30 - 2, value(b) < 15.9 p
  ZK_HD static fe29_t sub16(const fe29_t &a, const fe29_t &b) {
    fe29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P::fat30_16p(i) - b.l[i];
    return carry(r);
  }
  ZK_HD static fe29_t dbl(const fe29_t &a) { fe29_t r; for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1; return r; }

  // exact integer normalisation: full carry propagation (limbs < 2^29, unique representation of the integer)
  ZK_HD static fe29_t normalise(const fe29_t &a) {
    fe29_t r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint32_t v = a.l[i] + c; r.l[i] = v & M29; c = v >> 29; }
    r.l[8] = a.l[8] + c;
    return r;
  }
  // r = a - p if a >= p else a, for a normalised
  ZK_HD static fe29_t cond_sub_p(const fe29_t &a) {
    fe29_t d; uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { uint32_t v = a.l[i] - P::mod(i) - borrow; borrow = v >> 31; d.l[i] = (i < 8) ? (v & M29) : v; }
    fe29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
  }
  // v normalised (limbs < 2^29), value < 64 p  ->  tight, value < 2p, WITHOUT a multiplication: q ~ floor(v / p) from the top 16 bits
  // (q_est in {q, q-1}, checked exhaustively at the boundaries in tests), then v - q_est * p through a signed carry chain.
  ZK_HD static fe29_t reduce_small(const fe29_t &v) {
    const uint32_t q = (uint32_t)(((uint64_t)(v.l[8] >> 12) * 0x15291u) >> 26);   // 0x15291 = floor(2^270 / p) for both BN254 moduli (equal top bits)
    fe29_t r; int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { c += (int64_t)v.l[i] - (int64_t)((uint64_t)q * P::mod(i)); r.l[i] = (i < 8) ? ((uint32_t)c & M29) : (uint32_t)c; c >>= 29; }
    return r;
  }
  // value == 0 mod p, for a TIGHT mul/sqr output (value < 2p, limbs exact): value is 0 or p
  ZK_HD static bool is_zero_tight(const fe29_t &a) {
    // the low limb decides almost always (it is 0 or p's low limb with probability 2^-28 for a non-zero value): the full comparison
    // sits behind a branch that a wavefront practically never takes
    if (a.l[0] != 0 && a.l[0] != P::mod(0)) return false;
    uint32_t z = 0, q = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= a.l[i]; q |= a.l[i] ^ P::mod(i); }
    return z == 0 || q == 0;
  }
  // saturated R=2^256 Montgomery element (as in HBM) -> loose 29-bit limbs of x * 2^261  (= stored << 5)
  ZK_HD static fe29_t from_sat(const fe_t &s) {
    // bits of (s << 5): limb i covers bits [29 i, 29 i + 29) of the shifted value = bits [29 i - 5, 29 i + 24) of s
    fe29_t r;
    r.l[0] = (s.l[0] << 5) & M29;
#pragma unroll
    for (int i = 1; i < 9; i++) {
      const int bit = 29 * i - 5, w = bit >> 5, sh = bit & 31;
      uint32_t v = s.l[w] >> sh;
      if (sh > 3 && w + 1 < 8) v |= s.l[w + 1] << (32 - sh);
      r.l[i] = v & M29;
    }
    return r;   // top limb: bits 227.. of s -> < 2^29 automatically (s < 2^256)
  }
  // same without the Montgomery re-scaling (plain re-slicing of a 256-bit integer)
  ZK_HD static fe29_t from_sat_plain(const fe_t &s) {
    fe29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
      uint32_t v = s.l[w] >> sh;
      if (sh > 3 && w + 1 < 8) v |= s.l[w + 1] << (32 - sh);
      r.l[i] = v & M29;
    }
    return r;
  }
  // normalised, value < 2^256 -> 8 x 32 re-slicing
  ZK_HD static fe_t to_sat_plain(const fe29_t &a) {
    fe_t s;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const int bit = 32 * w, i = bit / 29, sh = bit - 29 * i;
      uint64_t v = (uint64_t)a.l[i] >> sh;
      v |= (uint64_t)a.l[i + 1] << (29 - sh);
      if (i + 2 < 9) v |= (uint64_t)a.l[i + 2] << (58 - sh);
      s.l[w] = (uint32_t)v;
    }
    return s;
  }
  // loose x * 2^261 -> canonical saturated x * 2^256 (the ABI form): multiply by the plain integer 2^256 (Montgomery: * 2^256 / 2^261), reduce fully
  ZK_HD static fe_t to_sat(const fe29_t &a) {
    fe29_t c; for (int i = 0; i < 9; i++) c.l[i] = P::r256(i);
    fe29_t t = cond_sub_p(normalise(mul(a, c)));   // mul output < 2p and limb-exact
    return to_sat_plain(t);
  }
};

using Fq29 = Fp29<Fq29P>;
using Fr29 = Fp29<Fr29P>;

}  // namespace zk
