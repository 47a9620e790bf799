// ntt.hpp -- what the NTT kernels of ntt29.hpp share with the rest of the library (bit reversal, the 8 x 32 power-table kernel), and the description of the
// decomposition they implement.  The 8 x 32-limb kernels that lived here (rounds 1-2; an A/B path behind MI355_NTT_SAT since) were removed in round 6.
// BN254 Fr number-theoretic transform for gfx950: LDS-tiled Cooley-Tukey in at most three
// global passes.  Stands in for halo2_proofs::arithmetic::best_fft (natural order in -> natural order out,
// a'[i] = sum_j a[j] w^(ij), no scaling) and the EvaluationDomain wrappers around it.
//
// Decomposition N = M1*M2*M3 (each M <= 2^10).  Level l < L ("strided pass"): every contiguous sub-problem of
// size S = M_l * T is viewed as an M_l x T matrix; a workgroup loads M_l rows x C adjacent columns (C*32 B
// contiguous per row -> coalesced), runs the size-M_l DFT of each column in LDS (radix-2 DIF, bit reversal
// undone on the way out), multiplies by the inter-level twiddle w_S^(col*k) and stores in place.  Level L
// ("final pass"): a workgroup loads C contiguous segments of M_L elements that differ in the FASTEST output
// digit, transforms them in LDS and scatters so that each store instruction writes C*32 B contiguous; this
// pass also applies the digit-reversal permutation, so it runs out of place (scratch -> destination).
// HBM traffic = 64 B per element per pass (2-3 passes; algorithmic minimum is one pass = 64*N bytes).
//
// Element layout in LDS: two 16-byte planes (lo = limbs 0..3, hi = limbs 4..7) so that consecutive lanes touch
// consecutive 16-B slots (conflict-free ds_read_b128 / ds_write_b128).
#pragma once
#include "fp_asm.hpp"

namespace zk {

ZK_HD uint32_t bitrev32(uint32_t x, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | ((x >> i) & 1); } return r;
#endif
}

#if defined(__HIPCC__)
// out[i] = base^(i * step) for i < count   (twiddle tables; tiny)
__global__ void k_pow_table(fe_t *out, fe_t base, uint64_t step, uint32_t count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe_t b = step == 1 ? base : Fr::pow_u64(base, step);
  g_store(&out[i], Fr::pow_u64(b, i));
}
#endif  // __HIPCC__

}  // namespace zk
