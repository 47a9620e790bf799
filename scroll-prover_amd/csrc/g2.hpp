// g2.hpp -- BN254 G2 (the sextic twist y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1)) for the ONE place the proving path
// touches it: ParamsKZG::setup's s_g2 = tau * G2 (there is no G2 MSM in create_proof).
// g2_affine_t = {x: {c0, c1}, y: {c0, c1}} = 128 B of Montgomery limbs == halo2curves bn256::G2Affine == the `g2` / `s_g2`
// fields of a RawBytes params file; identity = all zero.  Same XYZZ formulas as g1.hpp with Fq2 in the place of Fq
// (madd-2008-s / dbl-2008-s-1 / mdbl-2008-s); one scalar multiplication per SRS, so a single lane and the plain C++ multiplier.
#pragma once
#include "fp.hpp"

namespace zk {

struct fe2_t { fe_t c0, c1; };
struct alignas(16) g2_affine_t { fe2_t x, y; };
struct g2_xyzz_t { fe2_t x, y, zz, zzz; };

struct Fq2 {
  ZK_HD static fe2_t zero() { fe2_t r; r.c0 = Fq::zero(); r.c1 = Fq::zero(); return r; }
  ZK_HD static fe2_t one() { fe2_t r; r.c0 = Fq::one(); r.c1 = Fq::zero(); return r; }
  ZK_HD static bool is_zero(const fe2_t &a) { return Fq::is_zero(a.c0) && Fq::is_zero(a.c1); }
  ZK_HD static fe2_t add(const fe2_t &a, const fe2_t &b) { fe2_t r; r.c0 = Fq::add(a.c0, b.c0); r.c1 = Fq::add(a.c1, b.c1); return r; }
  ZK_HD static fe2_t sub(const fe2_t &a, const fe2_t &b) { fe2_t r; r.c0 = Fq::sub(a.c0, b.c0); r.c1 = Fq::sub(a.c1, b.c1); return r; }
  ZK_HD static fe2_t dbl(const fe2_t &a) { return add(a, a); }
  // (a0 + a1 u)(b0 + b1 u) = a0 b0 - a1 b1 + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u
  ZK_HD static fe2_t mul(const fe2_t &a, const fe2_t &b) {
    const fe_t t0 = Fq::mul(a.c0, b.c0), t1 = Fq::mul(a.c1, b.c1), t2 = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
    fe2_t r; r.c0 = Fq::sub(t0, t1); r.c1 = Fq::sub(Fq::sub(t2, t0), t1); return r;
  }
  // (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u
  ZK_HD static fe2_t sqr(const fe2_t &a) { fe2_t r; r.c0 = Fq::mul(Fq::add(a.c0, a.c1), Fq::sub(a.c0, a.c1)); r.c1 = Fq::dbl(Fq::mul(a.c0, a.c1)); return r; }
  ZK_HD static fe2_t inv(const fe2_t &a) {   // conj(a) / (a0^2 + a1^2); 0 -> 0
    const fe_t n = Fq::inv_sgcd(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
    fe2_t r; r.c0 = Fq::mul(a.c0, n); r.c1 = Fq::mul(Fq::neg(a.c1), n); return r;
  }
};

ZK_HD bool g2_affine_is_identity(const g2_affine_t &p) { return Fq2::is_zero(p.x) && Fq2::is_zero(p.y); }
ZK_HD g2_xyzz_t g2_xyzz_identity() { g2_xyzz_t r; r.x = Fq2::zero(); r.y = Fq2::zero(); r.zz = Fq2::zero(); r.zzz = Fq2::zero(); return r; }
ZK_HD g2_xyzz_t g2_xyzz_dbl_affine(const g2_affine_t &p) {
  const fe2_t U = Fq2::dbl(p.y), V = Fq2::sqr(U), W = Fq2::mul(U, V), S = Fq2::mul(p.x, V);
  fe2_t M = Fq2::sqr(p.x); M = Fq2::add(Fq2::dbl(M), M);
  g2_xyzz_t r;
  r.x = Fq2::sub(Fq2::sqr(M), Fq2::dbl(S));
  r.y = Fq2::sub(Fq2::mul(M, Fq2::sub(S, r.x)), Fq2::mul(W, p.y));
  r.zz = V; r.zzz = W;
  return r;
}
ZK_HD g2_xyzz_t g2_xyzz_dbl(const g2_xyzz_t &p) {
  if (Fq2::is_zero(p.zz)) return p;
  const fe2_t U = Fq2::dbl(p.y), V = Fq2::sqr(U), W = Fq2::mul(U, V), S = Fq2::mul(p.x, V);
  fe2_t M = Fq2::sqr(p.x); M = Fq2::add(Fq2::dbl(M), M);
  g2_xyzz_t r;
  r.x = Fq2::sub(Fq2::sqr(M), Fq2::dbl(S));
  r.y = Fq2::sub(Fq2::mul(M, Fq2::sub(S, r.x)), Fq2::mul(W, p.y));
  r.zz = Fq2::mul(V, p.zz); r.zzz = Fq2::mul(W, p.zzz);
  return r;
}
ZK_HD void g2_xyzz_madd(g2_xyzz_t &acc, const g2_affine_t &q) {
  if (g2_affine_is_identity(q)) return;
  if (Fq2::is_zero(acc.zz)) { acc.x = q.x; acc.y = q.y; acc.zz = Fq2::one(); acc.zzz = Fq2::one(); return; }
  const fe2_t U2 = Fq2::mul(q.x, acc.zz), S2 = Fq2::mul(q.y, acc.zzz);
  const fe2_t Pd = Fq2::sub(U2, acc.x), Rd = Fq2::sub(S2, acc.y);
  if (Fq2::is_zero(Pd)) { if (Fq2::is_zero(Rd)) acc = g2_xyzz_dbl_affine(q); else acc = g2_xyzz_identity(); return; }
  const fe2_t PP = Fq2::sqr(Pd), PPP = Fq2::mul(Pd, PP), Q = Fq2::mul(acc.x, PP);
  const fe2_t X3 = Fq2::sub(Fq2::sub(Fq2::sqr(Rd), PPP), Fq2::dbl(Q));
  const fe2_t Y3 = Fq2::sub(Fq2::mul(Rd, Fq2::sub(Q, X3)), Fq2::mul(acc.y, PPP));
  acc.x = X3; acc.y = Y3; acc.zz = Fq2::mul(acc.zz, PP); acc.zzz = Fq2::mul(acc.zzz, PPP);
}
ZK_HD g2_affine_t g2_xyzz_to_affine(const g2_xyzz_t &p) {
  g2_affine_t r;
  if (Fq2::is_zero(p.zz)) { r.x = Fq2::zero(); r.y = Fq2::zero(); return r; }
  const fe2_t i = Fq2::inv(Fq2::mul(p.zz, p.zzz));
  r.x = Fq2::mul(p.x, Fq2::mul(i, p.zzz)); r.y = Fq2::mul(p.y, Fq2::mul(i, p.zz));
  return r;
}
// k * p, k canonical (8 x 32-bit words); double-and-add from the top bit
ZK_HD g2_affine_t g2_mul_canonical(const g2_affine_t &p, const fe_t &k) {
  g2_xyzz_t acc = g2_xyzz_identity();
  for (int i = 255; i >= 0; i--) { acc = g2_xyzz_dbl(acc); if ((k.l[i >> 5] >> (i & 31)) & 1) g2_xyzz_madd(acc, p); }
  return g2_xyzz_to_affine(acc);
}
// y^2 == x^3 + 3 / (9 + u), or the identity
ZK_HD bool g2_is_on_curve(const g2_affine_t &p) {
  if (g2_affine_is_identity(p)) return true;
  fe_t c = Fq::zero(); fe2_t three = Fq2::zero(), nine_u;
  c.l[0] = 3; three.c0 = Fq::from_canonical(c); c.l[0] = 9; nine_u.c0 = Fq::from_canonical(c); nine_u.c1 = Fq::one();
  const fe2_t b = Fq2::mul(three, Fq2::inv(nine_u));
  const fe2_t lhs = Fq2::sqr(p.y), rhs = Fq2::add(Fq2::mul(Fq2::sqr(p.x), p.x), b);
  return Fq::eq(lhs.c0, rhs.c0) && Fq::eq(lhs.c1, rhs.c1);
}

#if defined(__HIPCC__)
// out = scalar * p (scalar: Fr, Montgomery form as the ABI delivers it).  One lane: 254 doublings + ~127 additions over Fq2.
__global__ void k_g2_mul(const g2_affine_t *__restrict__ p, fe_t scalar_mont, g2_affine_t *__restrict__ out, uint32_t *__restrict__ on_curve) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const g2_affine_t P = *p;
  *on_curve = g2_is_on_curve(P) ? 1u : 0u;
  *out = g2_mul_canonical(P, Fr::to_canonical(scalar_mont));
}
#endif

}  // namespace zk
