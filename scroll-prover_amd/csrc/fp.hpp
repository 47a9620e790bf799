// fp.hpp -- BN254 Fq / Fr arithmetic for gfx950 (MI355X), 8 x 32-bit limbs, Montgomery form R = 2^256.
//
// In-memory element = 32 bytes little-endian = exactly halo2curves' `Fr([u64;4])` / `Fq([u64;4])`
// (fixture KAT A1/A2 prove the Montgomery/LE/fully-reduced convention), so data crosses
// the C-ABI without conversion.  CDNA4 has no 64x64 multiplier: the work-horse is v_mad_u64_u32
// (32x32+64 -> 64).  All functions keep values fully reduced (< modulus) so that results are bit-exact
// with the CPU path in raw Montgomery bytes.
//
// Everything here is `__host__ __device__` so that tests can run the very same limb code on the CPU
// (csrc/host_selftest.cpp) against the oracle before it ever reaches a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

struct alignas(16) fe_t { uint32_t l[8]; };

// ---- modulus parameter packs (constants re-derived in tests/test_oracle_golden.py::test_constants_rederived)
struct FqP {  // base field, p = 0x30644e72...d87cfd47
  static constexpr uint32_t INV = 0xe4866389u;  // -p^-1 mod 2^32
  ZK_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return m[i];
  }
  ZK_HD static constexpr uint32_t one(int i) {  // R mod p
    constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return m[i];
  }
  ZK_HD static constexpr uint32_t r2(int i) {  // R^2 mod p
    constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    return m[i];
  }
};
struct FrP {  // scalar field, r = 0x30644e72...f0000001
  static constexpr uint32_t INV = 0xefffffffu;
  ZK_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return m[i];
  }
  ZK_HD static constexpr uint32_t one(int i) {
    constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return m[i];
  }
  ZK_HD static constexpr uint32_t r2(int i) {
    constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    return m[i];
  }
};

template <class P> struct Fp {
  ZK_HD static fe_t zero() { fe_t r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
  ZK_HD static fe_t one() { fe_t r; for (int i = 0; i < 8; i++) r.l[i] = P::one(i); return r; }
  ZK_HD static bool is_zero(const fe_t &a) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.l[i]; return o == 0; }
  ZK_HD static bool eq(const fe_t &a, const fe_t &b) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i]; return o == 0; }

  // r = a - mod if a >= mod else a   (a < 2*mod)
  ZK_HD static fe_t reduce_once(const fe_t &a) {
    fe_t d; uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)a.l[i] - P::mod(i) - borrow; d.l[i] = (uint32_t)x; borrow = (uint32_t)(x >> 63); }
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
  }
  ZK_HD static fe_t add(const fe_t &a, const fe_t &b) {
    fe_t s; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)a.l[i] + b.l[i] + c; s.l[i] = (uint32_t)x; c = (uint32_t)(x >> 32); }
    return reduce_once(s);  // moduli < 2^254: no carry out of 256 bits
  }
  ZK_HD static fe_t sub(const fe_t &a, const fe_t &b) {
    fe_t d; uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)a.l[i] - b.l[i] - borrow; d.l[i] = (uint32_t)x; borrow = (uint32_t)(x >> 63); }
    uint32_t mask = 0u - borrow, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)d.l[i] + (P::mod(i) & mask) + c; d.l[i] = (uint32_t)x; c = (uint32_t)(x >> 32); }
    return d;
  }
  ZK_HD static fe_t neg(const fe_t &a) { return sub(zero(), a); }
  ZK_HD static fe_t dbl(const fe_t &a) { return add(a, a); }

  // Montgomery product a*b*R^-1 mod m, CIOS over 32-bit limbs.  With m < 2^254 the running value stays
  // < 2m, so the ninth word never survives an iteration.
  ZK_HD static fe_t mul(const fe_t &a, const fe_t &b) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t c = 0;
      const uint32_t bi = b.l[i];
#pragma unroll
      for (int j = 0; j < 8; j++) { uint64_t x = (uint64_t)a.l[j] * bi + t[j] + c; t[j] = (uint32_t)x; c = x >> 32; }
      uint32_t t8 = (uint32_t)c;
      const uint32_t m = t[0] * P::INV;
      uint64_t x = (uint64_t)m * P::mod(0) + t[0]; c = x >> 32;
#pragma unroll
      for (int j = 1; j < 8; j++) { x = (uint64_t)m * P::mod(j) + t[j] + c; t[j - 1] = (uint32_t)x; c = x >> 32; }
      t[7] = t8 + (uint32_t)c;  // < 2^32 because the value is < 2m < 2^255
    }
    fe_t r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = t[j];
    return reduce_once(r);
  }
  ZK_HD static fe_t sqr(const fe_t &a) { return mul(a, a); }
  ZK_HD static fe_t from_canonical(const fe_t &a) { fe_t r2; for (int i = 0; i < 8; i++) r2.l[i] = P::r2(i); return mul(a, r2); }
  ZK_HD static fe_t to_canonical(const fe_t &a) { fe_t o = zero(); o.l[0] = 1; return mul(a, o); }
  // the same value by the Montgomery reduction alone (a * 1 has no partial products to add: 8 rounds of m = t0 * INV, t = (t + m * mod) >> 32):
  // 64 multiply-adds instead of 128; k_msm_digits converts every scalar with it
  ZK_HD static fe_t redc(const fe_t &a) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = a.l[j];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t m = t[0] * P::INV;
      uint64_t c = ((uint64_t)m * P::mod(0) + t[0]) >> 32;
#pragma unroll
      for (int j = 1; j < 8; j++) { const uint64_t x = (uint64_t)m * P::mod(j) + t[j] + c; t[j - 1] = (uint32_t)x; c = x >> 32; }
      t[7] = (uint32_t)c;   // a < 2^256 and m * mod < 2^32 * mod: the running value stays below mod + 2^224, so no ninth word
    }
    fe_t r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = t[j];
    return reduce_once(r);
  }
  // a^e for a 256-bit exponent given as 8 LE words (not constant time; exponents are public)
  ZK_HD static fe_t pow(const fe_t &a, const uint32_t e[8]) {
    fe_t acc = one();
    for (int i = 255; i >= 0; i--) { acc = sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a); }
    return acc;
  }
  ZK_HD static fe_t pow_u64(const fe_t &a, uint64_t e) {
    fe_t acc = one(); bool started = false;
    for (int i = 63; i >= 0; i--) { if (started) acc = sqr(acc); if ((e >> i) & 1) { acc = started ? mul(acc, a) : a; started = true; } }
    return acc;
  }
  ZK_HD static fe_t inv(const fe_t &a) {  // Fermat: a^(m-2); 0 -> 0
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = P::mod(i);
    e[0] -= 2;  // low words of both moduli are >= 2
    return pow(a, e);
  }

  // a^-1 by the binary extended Euclidean algorithm (Montgomery form in and out; 0 -> 0).  Data-dependent control flow: meant for the
  // places where ONE lane normalises a result -- about 20 k simple instructions instead of the ladder's ~380 dependent multiplications
  // (k_msm_final spent 0.4 ms of a 3 ms MSM in the ladder).  Invariants: u = x1 * a, v = x2 * a (mod m); gcd(a, m) = 1.
  ZK_HD static void w_halve(uint32_t *w) { for (int i = 0; i < 7; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31); w[7] >>= 1; }
  ZK_HD static void w_halve_mod(uint32_t *x) {   // x / 2 mod m; x + m < 2^255 never leaves the eight words
    if (x[0] & 1u) { uint32_t c = 0; for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)x[i] + P::mod(i) + c; x[i] = (uint32_t)t; c = (uint32_t)(t >> 32); } }
    w_halve(x);
  }
  ZK_HD static bool w_geq(const uint32_t *a, const uint32_t *b) { for (int i = 7; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i]; return true; }
  ZK_HD static bool w_is_one(const uint32_t *w) { uint32_t o = w[0] ^ 1u; for (int i = 1; i < 8; i++) o |= w[i]; return o == 0; }
  ZK_HD static void w_sub(uint32_t *a, const uint32_t *b) {   // a -= b, a >= b
    uint32_t borrow = 0; for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)a[i] - b[i] - borrow; a[i] = (uint32_t)t; borrow = (uint32_t)(t >> 63); }
  }
  ZK_HD static void w_sub_mod(uint32_t *a, const uint32_t *b) {   // a = a - b mod m, both < m
    uint32_t borrow = 0; for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)a[i] - b[i] - borrow; a[i] = (uint32_t)t; borrow = (uint32_t)(t >> 63); }
    if (borrow) { uint32_t c = 0; for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)a[i] + P::mod(i) + c; a[i] = (uint32_t)t; c = (uint32_t)(t >> 32); } }
  }
  ZK_HD static fe_t inv_bgcd(const fe_t &a) {
    if (is_zero(a)) return a;
    uint32_t u[8], v[8], x1[8], x2[8];
    for (int i = 0; i < 8; i++) { u[i] = a.l[i]; v[i] = P::mod(i); x1[i] = i == 0; x2[i] = 0; }
    while (!w_is_one(u) && !w_is_one(v)) {
      while (!(u[0] & 1u)) { w_halve(u); w_halve_mod(x1); }
      while (!(v[0] & 1u)) { w_halve(v); w_halve_mod(x2); }
      if (w_geq(u, v)) { w_sub(u, v); w_sub_mod(x1, x2); } else { w_sub(v, u); w_sub_mod(x2, x1); }
    }
    fe_t r, r2; const bool from_u = w_is_one(u);
    for (int i = 0; i < 8; i++) { r.l[i] = from_u ? x1[i] : x2[i]; r2.l[i] = P::r2(i); }
    // r = (a_plain R)^-1 as a plain residue; two Montgomery products with R^2 give a_plain^-1 * R
    return mul(mul(r, r2), r2);
  }

  // a^-1 by Bernstein-Yang division steps ("safegcd", as published by Bernstein and Yang, CHES 2019, in the batched form with 30-bit
  // signed limbs and a 2x2 transition matrix per 30 steps).  Montgomery form in and out; 0 -> 0.  The instruction stream does not depend on
  // the data -- 20 batches of 30 steps cover every 256-bit input (bound 590) -- so a whole wavefront can invert 64 different values
  // without divergence, and the dependent chains are 32-bit shifts / adds instead of multiword borrow chains: about a quarter of the
  // binary Euclid's latency in the one-lane normalisations (k_msm_final29, the tile inversions of the scans).
  // State: f = m, g = a, d = 0, e = 1 with d * a = f, e * a = g (mod m) up to the running power of two, which update_de divides out.
  static constexpr int32_t SG_M30 = 0x3fffffff;
  ZK_HD static constexpr uint32_t sg_mod(int i) {   // bits [30 i, 30 i + 30) of the modulus
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    uint32_t v = P::mod(w) >> sh;
    if (sh > 2 && w + 1 < 8) v |= P::mod(w + 1) << (32 - sh);
    return v & (uint32_t)SG_M30;
  }
  ZK_HD static constexpr uint32_t sg_mod_inv30() {   // m^-1 mod 2^30 (Newton iteration on the odd low limb)
    uint32_t m = P::mod(0), x = m;
    for (int i = 0; i < 5; i++) x *= 2u - m * x;
    return x & (uint32_t)SG_M30;
  }
  struct sg_t { int32_t v[9]; };
  // 30 division steps on the low limbs; t = (u, v, q, r) with 2^30 (f', g') = t (f, g)
  ZK_HD static int32_t sg_divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t (&t)[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
      uint32_t c1 = (uint32_t)(zeta >> 31);           // all ones iff zeta < 0
      const uint32_t c2 = 0u - (g & 1u);              // all ones iff g is odd
      const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // (f, u, v), negated when zeta < 0
      g += x & c2; q += y & c2; r += z & c2;
      c1 &= c2;                                       // swap: zeta < 0 and g odd
      zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);   // -zeta - 2, or zeta - 1
      f += g & c1; u += q & c1; v += r & c1;
      g >>= 1; u <<= 1; v <<= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return zeta;
  }
  // (d, e) <- t (d, e) / 2^30 mod m: a multiple of m makes the low 30 bits vanish before they are shifted out
  ZK_HD static void sg_update_de(sg_t &d, sg_t &e, const int32_t (&t)[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0], ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((sg_mod_inv30() * (uint32_t)cd + (uint32_t)md) & (uint32_t)SG_M30);
    me -= (int32_t)((sg_mod_inv30() * (uint32_t)ce + (uint32_t)me) & (uint32_t)SG_M30);
    cd += (int64_t)sg_mod(0) * md; ce += (int64_t)sg_mod(0) * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
      cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i]; ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i];
      cd += (int64_t)sg_mod(i) * md; ce += (int64_t)sg_mod(i) * me;
      d.v[i - 1] = (int32_t)cd & SG_M30; cd >>= 30;
      e.v[i - 1] = (int32_t)ce & SG_M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
  }
  // (f, g) <- t (f, g) / 2^30 (exact)
  ZK_HD static void sg_update_fg(sg_t &f, sg_t &g, const int32_t (&t)[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0], cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
      cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i]; cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
      f.v[i - 1] = (int32_t)cf & SG_M30; cf >>= 30;
      g.v[i - 1] = (int32_t)cg & SG_M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
  }
  ZK_HD static fe_t inv_sgcd(const fe_t &a) {
    sg_t f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
      uint32_t x = a.l[w] >> sh;
      if (sh > 2 && w + 1 < 8) x |= a.l[w + 1] << (32 - sh);
      g.v[i] = (int32_t)(x & (uint32_t)SG_M30); f.v[i] = (int32_t)sg_mod(i); d.v[i] = 0; e.v[i] = i == 0;
    }
    int32_t zeta = -1;
    for (int it = 0; it < 20; it++) {
      int32_t t[4];
      zeta = sg_divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
      sg_update_de(d, e, t);
      sg_update_fg(f, g, t);
    }
    // g = 0 and f = +-1 now (f = +-m for a = 0, where d stays 0): d = sign(f) a^-1 in (-2m, m); bring it to [0, m)
    int32_t r[9];
    const int32_t add1 = d.v[8] >> 31, neg = f.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) { r[i] = d.v[i] + ((int32_t)sg_mod(i) & add1); r[i] = (r[i] ^ neg) - neg; }
#pragma unroll
    for (int i = 0; i < 8; i++) { r[i + 1] += r[i] >> 30; r[i] &= SG_M30; }
    const int32_t add2 = r[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] += (int32_t)sg_mod(i) & add2;
#pragma unroll
    for (int i = 0; i < 8; i++) { r[i + 1] += r[i] >> 30; r[i] &= SG_M30; }
    fe_t o, r2;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const int bit = 32 * w, i = bit / 30, sh = bit - 30 * i;
      uint64_t x = (uint64_t)(uint32_t)r[i] >> sh;
      x |= (uint64_t)(uint32_t)r[i + 1] << (30 - sh);
      if (i + 2 < 9) x |= (uint64_t)(uint32_t)r[i + 2] << (60 - sh);
      o.l[w] = (uint32_t)x; r2.l[w] = P::r2(w);
    }
    // o = (a_plain R)^-1 as a plain residue; two Montgomery products with R^2 give a_plain^-1 * R
    return mul(mul(o, r2), r2);
  }
};

#if defined(__HIPCC__)
// 32-byte element <-> two dwordx4 global accesses
__device__ __forceinline__ fe_t g_load(const fe_t *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p); uint4 a = q[0], b = q[1]; fe_t r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; return r;
}
__device__ __forceinline__ void g_store(fe_t *p, const fe_t &v) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
#endif

using Fq = Fp<FqP>;
using Fr = Fp<FrP>;

}  // namespace zk
