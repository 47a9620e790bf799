// host_selftest.cpp -- TEST INFRASTRUCTURE: compiles the product's __host__ __device__ limb code
// (scroll-prover_amd/csrc/fp.hpp, g1.hpp) for the CPU with plain g++ so that the 8x32-bit Montgomery
// and XYZZ formulas can be checked against the oracle in the GPU-less container.  Never shipped,
// never linked into libmi355zk.so.
#include "../../scroll-prover_amd/csrc/g1.hpp"
#include "../../scroll-prover_amd/csrc/fp_asm.hpp"
#include "../../scroll-prover_amd/csrc/fp29.hpp"
#include "../../scroll-prover_amd/csrc/g1_29.hpp"
#include "../../scroll-prover_amd/csrc/glv.hpp"
#include <string.h>
using namespace zk;

template <class F, class P> static void binop(int op, fe_t *o, const fe_t *a, const fe_t *b) {
  switch (op) {
    case 0: *o = F::add(*a, *b); break;
    case 1: *o = F::sub(*a, *b); break;
    case 2: *o = F::mul(*a, *b); break;
    case 3: *o = F::inv(*a); break;
    case 4: *o = F::neg(*a); break;
    case 5: *o = F::from_canonical(*a); break;
    case 6: *o = F::to_canonical(*a); break;
    case 7: *o = mont_mul_ps<P>(*a, *b); break;
    case 8: *o = mont_sqr_ps<P>(*a); break;
    case 9: *o = F::inv_bgcd(*a); break;
    case 10: *o = F::inv_sgcd(*a); break;
    case 11: *o = F::redc(*a); break;
  }
}
extern "C" void hs_f_op(int which, int op, void *o, const void *a, const void *b) {
  if (which) binop<Fr, FrP>(op, (fe_t *)o, (const fe_t *)a, (const fe_t *)b); else binop<Fq, FqP>(op, (fe_t *)o, (const fe_t *)a, (const fe_t *)b);
}
// sum_i k_i * P_i with k_i canonical 256-bit, via XYZZ double-and-add (exercises madd / add / dbl and their special cases)
extern "C" void hs_msm_naive(void *out_jac, const void *scalars_canonical, const void *bases, uint64_t n) {
  const fe_t *k = (const fe_t *)scalars_canonical; const g1_affine_t *b = (const g1_affine_t *)bases;
  g1_xyzz_t total = g1_xyzz_identity();
  for (uint64_t i = 0; i < n; i++) {
    g1_xyzz_t acc = g1_xyzz_identity();
    for (int bit = 255; bit >= 0; bit--) { acc = g1_xyzz_dbl(acc); if ((k[i].l[bit >> 5] >> (bit & 31)) & 1) g1_xyzz_madd(acc, b[i]); }
    g1_xyzz_add(total, acc);
  }
  *(g1_jac_t *)out_jac = g1_xyzz_to_jac_normalised(total);
}
extern "C" void hs_xyzz_madd(void *acc_xyzz, const void *affine) { g1_xyzz_madd(*(g1_xyzz_t *)acc_xyzz, *(const g1_affine_t *)affine); }
extern "C" void hs_xyzz_add(void *acc_xyzz, const void *q) { g1_xyzz_add(*(g1_xyzz_t *)acc_xyzz, *(const g1_xyzz_t *)q); }
extern "C" void hs_xyzz_to_jac(void *out_jac, const void *p) { *(g1_jac_t *)out_jac = g1_xyzz_to_jac_normalised(*(const g1_xyzz_t *)p); }
extern "C" void hs_jac_to_xyzz(void *out, const void *p) { *(g1_xyzz_t *)out = g1_jac_to_xyzz(*(const g1_jac_t *)p); }
extern "C" void hs_xyzz_mul_small(void *out, const void *p, uint32_t k) { *(g1_xyzz_t *)out = g1_xyzz_mul_small(*(const g1_xyzz_t *)p, k); }
extern "C" void hs_xyzz_madd_ps(void *acc_xyzz, const void *affine) { g1_xyzz_madd_ps(*(g1_xyzz_t *)acc_xyzz, *(const g1_affine_t *)affine); }
extern "C" void hs_xyzz_add_ps(void *acc_xyzz, const void *q) { g1_xyzz_add_ps(*(g1_xyzz_t *)acc_xyzz, *(const g1_xyzz_t *)q); }

// ---- 29-bit unsaturated arithmetic (fp29.hpp): every op takes/returns saturated ABI elements so the test can use the oracle
template <class F29> static void op29(int op, fe_t *o, const fe_t *a, const fe_t *b) {
  fe29_t x = F29::from_sat(*a), y = F29::from_sat(*b), r;
  switch (op) {
    case 0: r = F29::mul(x, y); break;                                  // a*b
    case 1: r = F29::sqr(x); break;                                     // a^2
    case 2: r = F29::mul(F29::add(x, y), y); break;                     // (a+b)*b   loose operand
    case 3: r = F29::sub4(x, F29::mul(y, F29::one())); break;           // a - b     (b made tight first)
    case 4: r = F29::sub8(F29::mul(x, F29::one()), F29::dbl(F29::mul(y, F29::one()))); break;   // a - 2b
    case 5: r = F29::sub16(x, F29::sub8(F29::mul(y, F29::one()), F29::dbl(F29::mul(x, F29::one())))); break;  // a - (b - 2a) = 3a - b
    case 6: { fe29_t xt = F29::mul(x, F29::one()), yt = F29::mul(y, F29::one()); r = F29::sqr(F29::sub16(xt, F29::sub4(yt, xt))); } break;  // (2a - b)^2, chained lazy values (from_sat output is < 2^259 and may only feed mul or the minuend)
    case 8: { fe29_t xt = F29::mul(x, F29::one()), yt = F29::mul(y, F29::one()); r = F29::mul_sub(F29::sub16(xt, yt), F29::sub8(yt, F29::dbl(xt)), F29::sub4(yt, xt), xt); } break;  // (a-b)(b-2a) - (b-a)a
    default: r = x;
  }
  *o = F29::to_sat(r);
}
extern "C" void hs_f29_op(int which, int op, void *o, const void *a, const void *b) {
  if (which) op29<Fr29>(op, (fe_t *)o, (const fe_t *)a, (const fe_t *)b); else op29<Fq29>(op, (fe_t *)o, (const fe_t *)a, (const fe_t *)b);
}
extern "C" int hs_f29_is_zero(int which, const void *a, const void *b) {   // is (a - b) == 0 through the tight zero test
  const fe_t *x = (const fe_t *)a, *y = (const fe_t *)b;
  if (which) { fe29_t d = Fr29::sub4(Fr29::from_sat(*x), Fr29::mul(Fr29::from_sat(*y), Fr29::one())); return Fr29::is_zero_tight(Fr29::mul(d, Fr29::one())); }
  fe29_t d = Fq29::sub4(Fq29::from_sat(*x), Fq29::mul(Fq29::from_sat(*y), Fq29::one())); return Fq29::is_zero_tight(Fq29::mul(d, Fq29::one()));
}

// bucket-style accumulation with the 29-bit accumulator: out = sum_i (+-) pts[i] (sign bit = bit 0 of signs[i]), flushed to the saturated XYZZ record
extern "C" void hs_bucket_sum29(void *out_xyzz, const void *pts, const uint8_t *signs, uint64_t n) {
  const g1_affine_t *p = (const g1_affine_t *)pts;
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (uint64_t i = 0; i < n; i++) g1_xyzz29_madd(acc, p[i], signs[i] & 1);
  *(g1_xyzz_t *)out_xyzz = g1_xyzz29_to_sat(acc);
}

// reduce_small on k*p + delta inputs given as plain 9-limb integers (limbs < 2^29): returns 1 when the result is tight, < 2p and congruent
extern "C" void hs_f29_reduce_small(int which, uint32_t *out9, const uint32_t *in9) {
  fe29_t v; for (int i = 0; i < 9; i++) v.l[i] = in9[i];
  fe29_t r = which ? Fr29::reduce_small(v) : Fq29::reduce_small(v);
  for (int i = 0; i < 9; i++) out9[i] = r.l[i];
}

// ---- 29-bit full addition / doubling (g1_29.hpp): groups of mixed additions are combined with g1_xyzz29_add (loose inputs), and
// a double-and-add ladder runs on g1_xyzz29_dbl / g1_xyzz29_add; results leave as saturated XYZZ for the oracle comparison
extern "C" void hs_xyzz29_grouped_sum(void *out_xyzz, const void *affine, const uint8_t *signs, uint64_t n, uint64_t group) {
  const g1_affine_t *p = (const g1_affine_t *)affine;
  g1_xyzz29_t total = g1_xyzz29_identity();
  for (uint64_t g0 = 0; g0 < n; g0 += group) {
    g1_xyzz29_t acc = g1_xyzz29_identity();
    for (uint64_t i = g0; i < n && i < g0 + group; i++) g1_xyzz29_madd(acc, p[i], signs[i] & 1);
    g1_xyzz29_add(total, acc);
  }
  *(g1_xyzz_t *)out_xyzz = g1_xyzz29_to_sat(total);
}
extern "C" void hs_xyzz29_ladder(void *out_xyzz, const void *affine, const uint8_t *signs, uint64_t n, uint32_t k) {
  const g1_affine_t *p = (const g1_affine_t *)affine;
  g1_xyzz29_t base = g1_xyzz29_identity();
  for (uint64_t i = 0; i < n; i++) g1_xyzz29_madd(base, p[i], signs[i] & 1);   // a loose accumulator as the ladder's base point
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (int bit = 31; bit >= 0; bit--) { acc = g1_xyzz29_dbl(acc); if ((k >> bit) & 1) g1_xyzz29_add(acc, base); }
  *(g1_xyzz_t *)out_xyzz = g1_xyzz29_to_sat(acc);
}

// ---- GLV (glv.hpp): decomposition of a canonical scalar and the joint scalar multiple k * P on the 29-bit field
extern "C" void hs_glv_decompose(const void *k_canonical, uint32_t *k1, int *neg1, uint32_t *k2, int *neg2) {
  uint32_t a[4], b[4]; bool n1, n2;
  glv_decompose(*(const fe_t *)k_canonical, a, n1, b, n2);
  for (int i = 0; i < 4; i++) { k1[i] = a[i]; k2[i] = b[i]; }
  *neg1 = n1; *neg2 = n2;
}
extern "C" void hs_g1_mul_glv(void *out_xyzz, const void *affine, const void *k_canonical) {
  const g1_affine_t *q = (const g1_affine_t *)affine;
  g1_xyzz29_t base = g1_xyzz29_identity();
  g1_xyzz29_madd(base, *q, false);                              // tight coordinates, zz = zzz = one (what g1_xyzz29_from_sat hands over)
  *(g1_xyzz_t *)out_xyzz = g1_xyzz29_to_sat(g1_xyzz29_mul_glv(base, *(const fe_t *)k_canonical));
}
