// host_selftest.cpp -- TEST INFRASTRUCTURE: compiles the product's __host__ __device__ limb code
// (scroll-prover_amd/csrc/fp.cuh, g1.cuh) for the CPU with plain g++ so that the 8x32-bit Montgomery
// and XYZZ formulas can be checked against the oracle in the GPU-less container.  Never shipped,
// never linked into libmi355zk.so.
#include "../../scroll-prover_amd/csrc/g1.cuh"
#include "../../scroll-prover_amd/csrc/fp_asm.cuh"
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
