// g1.hpp -- BN254 G1 (y^2 = x^3 + 3 over Fq) point arithmetic for the MSM kernels.
//
// ABI types (halo2curves bn256 [EXT-recalled]):
//   g1_affine_t  {x, y}      64 B, Montgomery, identity = (0, 0)          == halo2curves G1Affine
//   g1_jac_t     {x, y, z}   96 B, Jacobian (x = X/Z^2, y = Y/Z^3), identity z = 0   == halo2curves G1
// Internal accumulator: XYZZ ("extended Jacobian", x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), identity ZZ = 0.
// XYZZ mixed addition costs 8M + 2S against 7M + 4S for Jacobian, needs no field doubling chains,
// and the accumulator is what the bucket kernels keep in registers (32 VGPRs).
// Formulas: EFD "madd-2008-s", "add-2008-s", "dbl-2008-s-1", "mdbl-2008-s".
#pragma once
#include "fp.hpp"

namespace zk {

struct alignas(16) g1_affine_t { fe_t x, y; };
struct alignas(16) g1_jac_t { fe_t x, y, z; };
struct alignas(16) g1_xyzz_t { fe_t x, y, zz, zzz; };

ZK_HD bool g1_affine_is_identity(const g1_affine_t &p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
ZK_HD bool g1_xyzz_is_identity(const g1_xyzz_t &p) { return Fq::is_zero(p.zz); }
ZK_HD g1_xyzz_t g1_xyzz_identity() { g1_xyzz_t r; r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero(); return r; }
ZK_HD g1_xyzz_t g1_xyzz_from_affine(const g1_affine_t &p) {
  if (g1_affine_is_identity(p)) return g1_xyzz_identity();
  g1_xyzz_t r; r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one(); return r;
}
ZK_HD g1_affine_t g1_affine_neg(const g1_affine_t &p) { g1_affine_t r; r.x = p.x; r.y = Fq::neg(p.y); return r; }  // neg(0) = 0 keeps identity

// 2 * (affine point), mdbl-2008-s.  p must not be the identity.
ZK_HD g1_xyzz_t g1_xyzz_dbl_affine(const g1_affine_t &p) {
  fe_t U = Fq::dbl(p.y), V = Fq::sqr(U), W = Fq::mul(U, V), S = Fq::mul(p.x, V);
  fe_t M = Fq::sqr(p.x); M = Fq::add(Fq::dbl(M), M);
  g1_xyzz_t r;
  r.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
  r.y = Fq::sub(Fq::mul(M, Fq::sub(S, r.x)), Fq::mul(W, p.y));
  r.zz = V; r.zzz = W;
  return r;
}
// dbl-2008-s-1
ZK_HD g1_xyzz_t g1_xyzz_dbl(const g1_xyzz_t &p) {
  if (g1_xyzz_is_identity(p)) return p;
  fe_t U = Fq::dbl(p.y), V = Fq::sqr(U), W = Fq::mul(U, V), S = Fq::mul(p.x, V);
  fe_t M = Fq::sqr(p.x); M = Fq::add(Fq::dbl(M), M);
  g1_xyzz_t r;
  r.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
  r.y = Fq::sub(Fq::mul(M, Fq::sub(S, r.x)), Fq::mul(W, p.y));
  r.zz = Fq::mul(V, p.zz); r.zzz = Fq::mul(W, p.zzz);
  return r;
}
// acc += q (affine), madd-2008-s with the exceptional cases (identity operands, q == +-acc)
ZK_HD void g1_xyzz_madd(g1_xyzz_t &acc, const g1_affine_t &q) {
  if (g1_affine_is_identity(q)) return;
  if (g1_xyzz_is_identity(acc)) { acc.x = q.x; acc.y = q.y; acc.zz = Fq::one(); acc.zzz = Fq::one(); return; }
  fe_t U2 = Fq::mul(q.x, acc.zz), S2 = Fq::mul(q.y, acc.zzz);
  fe_t Pd = Fq::sub(U2, acc.x), Rd = Fq::sub(S2, acc.y);
  if (Fq::is_zero(Pd)) {
    if (Fq::is_zero(Rd)) acc = g1_xyzz_dbl_affine(q); else acc = g1_xyzz_identity();
    return;
  }
  fe_t PP = Fq::sqr(Pd), PPP = Fq::mul(Pd, PP), Q = Fq::mul(acc.x, PP);
  fe_t X3 = Fq::sub(Fq::sub(Fq::sqr(Rd), PPP), Fq::dbl(Q));
  fe_t Y3 = Fq::sub(Fq::mul(Rd, Fq::sub(Q, X3)), Fq::mul(acc.y, PPP));
  acc.x = X3; acc.y = Y3; acc.zz = Fq::mul(acc.zz, PP); acc.zzz = Fq::mul(acc.zzz, PPP);
}
// acc += q (XYZZ), add-2008-s with the exceptional cases
ZK_HD void g1_xyzz_add(g1_xyzz_t &acc, const g1_xyzz_t &q) {
  if (g1_xyzz_is_identity(q)) return;
  if (g1_xyzz_is_identity(acc)) { acc = q; return; }
  fe_t U1 = Fq::mul(acc.x, q.zz), U2 = Fq::mul(q.x, acc.zz);
  fe_t S1 = Fq::mul(acc.y, q.zzz), S2 = Fq::mul(q.y, acc.zzz);
  fe_t Pd = Fq::sub(U2, U1), Rd = Fq::sub(S2, S1);
  if (Fq::is_zero(Pd)) {
    if (Fq::is_zero(Rd)) acc = g1_xyzz_dbl(acc); else acc = g1_xyzz_identity();
    return;
  }
  fe_t PP = Fq::sqr(Pd), PPP = Fq::mul(Pd, PP), Q = Fq::mul(U1, PP);
  fe_t X3 = Fq::sub(Fq::sub(Fq::sqr(Rd), PPP), Fq::dbl(Q));
  fe_t Y3 = Fq::sub(Fq::mul(Rd, Fq::sub(Q, X3)), Fq::mul(S1, PPP));
  acc.x = X3; acc.y = Y3;
  acc.zz = Fq::mul(Fq::mul(acc.zz, q.zz), PP); acc.zzz = Fq::mul(Fq::mul(acc.zzz, q.zzz), PPP);
}
// XYZZ -> affine (one field inversion)
ZK_HD g1_affine_t g1_xyzz_to_affine(const g1_xyzz_t &p) {
  g1_affine_t r;
  if (g1_xyzz_is_identity(p)) { r.x = Fq::zero(); r.y = Fq::zero(); return r; }
  // 1/ZZZ, then 1/ZZ = ZZZ^-1 * ZZZ / ZZ ... cheaper: i = (ZZ*ZZZ)^-1; 1/ZZ = i*ZZZ; 1/ZZZ = i*ZZ
  fe_t i = Fq::inv_sgcd(Fq::mul(p.zz, p.zzz));   // division-step inverse (fp.hpp): no multiword borrow chains, no divergence when every lane converts a point
  r.x = Fq::mul(p.x, Fq::mul(i, p.zzz));
  r.y = Fq::mul(p.y, Fq::mul(i, p.zz));
  return r;
}
// XYZZ -> normalised Jacobian as handed back over the C-ABI: (x, y, 1) or (0, 0, 0) for the identity
ZK_HD g1_jac_t g1_xyzz_to_jac_normalised(const g1_xyzz_t &p) {
  g1_affine_t a = g1_xyzz_to_affine(p);
  g1_jac_t r; r.x = a.x; r.y = a.y;
  r.z = g1_xyzz_is_identity(p) ? Fq::zero() : Fq::one();
  return r;
}
// Jacobian (any representative, as Rust may hand it to us) -> XYZZ: ZZ = Z^2, ZZZ = Z^3
ZK_HD g1_xyzz_t g1_jac_to_xyzz(const g1_jac_t &p) {
  g1_xyzz_t r;
  if (Fq::is_zero(p.z)) return g1_xyzz_identity();
  r.x = p.x; r.y = p.y; r.zz = Fq::sqr(p.z); r.zzz = Fq::mul(r.zz, p.z);
  return r;
}
// k * p for a small unsigned k (double-and-add from the top bit); used by the bucket-reduce kernel
ZK_HD g1_xyzz_t g1_xyzz_mul_small(const g1_xyzz_t &p, uint32_t k) {
  g1_xyzz_t acc = g1_xyzz_identity();
  for (int i = 31; i >= 0; i--) { acc = g1_xyzz_dbl(acc); if ((k >> i) & 1) g1_xyzz_add(acc, p); }
  return acc;
}

}  // namespace zk
