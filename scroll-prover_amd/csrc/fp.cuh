// fp.cuh -- BN254 Fq / Fr arithmetic for gfx950 (MI355X), 8 x 32-bit limbs, Montgomery form R = 2^256.
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
