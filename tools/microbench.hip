// microbench.hip -- instruction-rate and field-multiplier probes for gfx950 (MI355X).
// Not part of the product: its numbers decide which Montgomery formulation the kernels use
// (DESIGN.md "ALU roofline").  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../scroll-prover_amd/csrc/g1.hpp"
#include "../scroll-prover_amd/csrc/fp_asm.hpp"
#include "../scroll-prover_amd/csrc/fp29.hpp"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;
// ---- raw instruction probes: 8 independent chains per lane, ITERS iterations, 8 ops per iteration per chain set
#define PROBE_KERNEL(name, decl, body, sink)                                        \
  __global__ void name(uint32_t *out, uint32_t seed) {                               \
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; decl;                        \
    for (int i = 0; i < ITERS; i++) { body; }                                        \
    out[t] = sink;                                                                   \
  }

PROBE_KERNEL(k_mad64, uint64_t a0 = t; uint64_t a1 = t + 1; uint64_t a2 = t + 2; uint64_t a3 = t + 3; uint64_t a4 = t + 4; uint64_t a5 = t + 5; uint64_t a6 = t + 6; uint64_t a7 = t + 7; uint32_t x = seed | 1; uint32_t y = t | 3,
  asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
               "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc"),
  (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7))

#define OP8_32(INSN)                                                                                                     \
  asm volatile(INSN " %0, %8, %0\n " INSN " %1, %8, %1\n " INSN " %2, %8, %2\n " INSN " %3, %8, %3\n " INSN " %4, %8, %4\n " \
               INSN " %5, %8, %5\n " INSN " %6, %8, %6\n " INSN " %7, %8, %7\n"                                            \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x))
#define DECL32 uint32_t a0 = t; uint32_t a1 = t + 1; uint32_t a2 = t + 2; uint32_t a3 = t + 3; uint32_t a4 = t + 4; uint32_t a5 = t + 5; uint32_t a6 = t + 6; uint32_t a7 = t + 7; uint32_t x = seed | 1
#define SINK32 (a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
PROBE_KERNEL(k_mul_lo, DECL32, OP8_32("v_mul_lo_u32"), SINK32)
PROBE_KERNEL(k_mul_hi, DECL32, OP8_32("v_mul_hi_u32"), SINK32)
PROBE_KERNEL(k_add_u32, DECL32, OP8_32("v_add_u32"), SINK32)
PROBE_KERNEL(k_mul_u24, DECL32, OP8_32("v_mul_u32_u24"), SINK32)
PROBE_KERNEL(k_mul_hi_u24, DECL32, OP8_32("v_mul_hi_u32_u24"), SINK32)
PROBE_KERNEL(k_xor, DECL32, OP8_32("v_xor_b32"), SINK32)

#define OP8_3(INSN)                                                                                                      \
  asm volatile(INSN " %0, %8, %9, %0\n " INSN " %1, %8, %9, %1\n " INSN " %2, %8, %9, %2\n " INSN " %3, %8, %9, %3\n "       \
               INSN " %4, %8, %9, %4\n " INSN " %5, %8, %9, %5\n " INSN " %6, %8, %9, %6\n " INSN " %7, %8, %9, %7\n"       \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
PROBE_KERNEL(k_mad_u24, DECL32; uint32_t y = t | 3, OP8_3("v_mad_u32_u24"), SINK32)
PROBE_KERNEL(k_add3, DECL32; uint32_t y = t | 3, OP8_3("v_add3_u32"), SINK32)
PROBE_KERNEL(k_fma_f32, float a0 = t; float a1 = t + 1; float a2 = t + 2; float a3 = t + 3; float a4 = t + 4; float a5 = t + 5; float a6 = t + 6; float a7 = t + 7; float x = 1.0001f; float y = 0.5f,
             OP8_3("v_fma_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
#define DECL64F double a0 = t; double a1 = t + 1; double a2 = t + 2; double a3 = t + 3; double a4 = t + 4; double a5 = t + 5; double a6 = t + 6; double a7 = t + 7; double x = 1.0000001; double y = 0.5
PROBE_KERNEL(k_fma_f64, DECL64F, OP8_3("v_fma_f64"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
PROBE_KERNEL(k_add_f64, DECL64F, OP8_32("v_add_f64"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
PROBE_KERNEL(k_lshl_add_u64, uint64_t a0 = t; uint64_t a1 = t + 1; uint64_t a2 = t + 2; uint64_t a3 = t + 3; uint64_t a4 = t + 4; uint64_t a5 = t + 5; uint64_t a6 = t + 6; uint64_t a7 = t + 7; uint64_t x = seed | 1,
  asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
               "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x)),
  (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7))
// mad + addc pair (the product-scanning inner step): 4 chains x 2 instr
PROBE_KERNEL(k_mad_addc, uint64_t a0 = t; uint64_t a1 = t + 1; uint64_t a2 = t + 2; uint64_t a3 = t + 3; uint32_t e0 = 0; uint32_t e1 = 0; uint32_t e2 = 0; uint32_t e3 = 0; uint32_t x = seed | 0x80000001u; uint32_t y = t | 0xc0000003u,
  asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
               "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(x), "v"(y) : "vcc"),
  (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ e0 ^ e1 ^ e2 ^ e3)

// ---- field multiplier probes: dependent chain per lane (latency hidden by occupancy), MULS per lane
constexpr int MULS = 2048;
template <int VARIANT> __global__ void k_fqmul(fe_t *io) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  fe_t a = io[2 * t], b = io[2 * t + 1];
  for (int i = 0; i < MULS; i += 2) {
    if (VARIANT == 0) { a = Fq::mul(a, b); b = Fq::mul(b, a); }
    if (VARIANT == 1) { a = fq_mul_ps(a, b); b = fq_mul_ps(b, a); }
    if (VARIANT == 2) { a = fq_sqr_ps(a); a = fq_mul_ps(a, b); }
  }
  io[2 * t] = a; io[2 * t + 1] = b;
}
template <int VARIANT> __global__ void k_fq29mul(fe_t *io) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  fe29_t a = Fq29::from_sat(io[2 * t]), b = Fq29::from_sat(io[2 * t + 1]);
  for (int i = 0; i < MULS; i += 2) {
    if (VARIANT == 0) { a = Fq29::mul(a, b); b = Fq29::mul(b, a); }
    if (VARIANT == 1) { a = Fq29::sqr(a); a = Fq29::mul(a, b); }
    if (VARIANT == 2) { a = Fq29::mul(Fq29::sub4(a, b), b); b = Fq29::mul(Fq29::add(b, a), a); }   // with lazy add/sub in the chain
    if (VARIANT == 3) { a = Fq29::mul2(a, b); b = Fq29::mul2(b, a); }
    if (VARIANT == 4) { a = Fq29::mul_c(a, b); b = Fq29::mul_c(b, a); }       // chained v_mad (inline asm), no per-column 64-bit add
    if (VARIANT == 5) { a = Fq29::sqr_c(a); a = Fq29::mul_c(a, b); }
  }
  io[2 * t] = Fq29::to_sat(a); io[2 * t + 1] = Fq29::to_sat(b);
}
// madd chain on the 29-bit accumulator, to see the ALU ceiling of k_msm_accumulate at its real occupancy
#include "../scroll-prover_amd/csrc/g1_29.hpp"
template <bool FUSED, bool CHAIN = false> __global__ void __launch_bounds__(256) k_madd29(g1_xyzz_t *accs, const g1_affine_t *pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  g1_xyzz29_t acc = g1_xyzz29_identity();
  for (int i = 0; i < iters; i++) {
    g1_affine_t p = pts[(t * 31 + i) % npts];
    g1_xyzz29_madd<FUSED, CHAIN>(acc, p, (i & 1) != 0);
  }
  accs[t] = g1_xyzz29_to_sat(acc);
}
// ---- batched-affine candidate ("pairs, then madd"): entries are taken two at a time, the pair is added in AFFINE coordinates with one
// shared inversion per batch of B pairs (Montgomery's trick: prefix products kept in LDS, points gathered again in the backward pass, as
// the real kernel would have to -- B x 4 coordinates do not fit in registers or LDS), and the affine sum enters the XYZZ accumulator by
// a mixed addition.  INV = 0: the inversion is skipped (results are wrong; the time is a LOWER bound for any inversion algorithm);
// INV = 1: a Fermat-sized ladder (254 squarings + ~127 multiplications on the 29-bit field; timing only, the exponent is not p - 2).  Compare entries/s with the plain madd chain.
template <int B, int INV> __global__ void __launch_bounds__(256) k_affine_pairs(g1_xyzz_t *accs, const g1_affine_t *pts, int npts, int iters) {
  extern __shared__ uint32_t lds_pref[];                 // [B][9][256]: prefix products, lane-interleaved (conflict-free)
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x;
  g1_xyzz29_t acc = g1_xyzz29_identity();
  auto pref_put = [&](int i, const fe29_t &v) { for (int k = 0; k < 9; k++) lds_pref[(i * 9 + k) * 256 + lane] = v.l[k]; };
  auto pref_get = [&](int i) { fe29_t v; for (int k = 0; k < 9; k++) v.l[k] = lds_pref[(i * 9 + k) * 256 + lane]; return v; };
  for (int it = 0; it < iters; it += 2 * B) {
    fe29_t run = Fq29::one();
    for (int i = 0; i < B; i++) {                        // forward: d_i = x2 - x1, running product
      const g1_affine_t p = pts[(t * 31 + it + 2 * i) % npts], q = pts[(t * 31 + it + 2 * i + 1) % npts];
      const fe29_t d = Fq29::sub16(Fq29::from_sat(q.x), Fq29::reduce_small(Fq29::from_sat(p.x)));
      pref_put(i, run);
      run = Fq29::mul(run, d);
    }
    fe29_t inv = run;
    if (INV == 1) {                                       // run^(p-2): square-and-multiply over the fixed exponent
      fe29_t r = Fq29::one();
      for (int bit = 253; bit >= 0; bit--) { r = Fq29::sqr(r); if (((uint32_t)bit * 2654435761u) >> 31) r = Fq29::mul(r, run); }   // timing only: ~half the exponent bits set, as in p - 2
      inv = r;
    }
    for (int i = B - 1; i >= 0; i--) {                   // backward: inverse of d_i, the affine sum, its mixed addition into the accumulator
      const g1_affine_t p = pts[(t * 31 + it + 2 * i) % npts], q = pts[(t * 31 + it + 2 * i + 1) % npts];
      const fe29_t x1 = Fq29::reduce_small(Fq29::from_sat(p.x)), y1 = Fq29::reduce_small(Fq29::from_sat(p.y)), x2 = Fq29::from_sat(q.x), y2 = Fq29::from_sat(q.y);
      const fe29_t d = Fq29::sub16(x2, x1);
      const fe29_t inv_i = Fq29::mul(inv, pref_get(i));
      inv = Fq29::mul(inv, d);
      const fe29_t lam = Fq29::mul(Fq29::sub16(y2, y1), inv_i);
      const fe29_t x3 = Fq29::sub8(Fq29::sub4(Fq29::sqr(lam), x1), Fq29::carry(x2));
      const fe29_t y3 = Fq29::sub4(Fq29::mul(lam, Fq29::sub16(x1, x3)), y1);
      g1_xyzz29_madd_core<true, false>(acc, x3, y3, false);
    }
  }
  accs[t] = g1_xyzz29_to_sat(acc);
}
// XYZZ mixed-add chain: the real MSM inner loop without memory traffic
template <int VARIANT> __global__ void k_madd(g1_xyzz_t *accs, const g1_affine_t *pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  g1_xyzz_t acc = accs[t];
  for (int i = 0; i < iters; i++) {
    g1_affine_t p = pts[(t * 31 + i) % npts];
    if (VARIANT == 0) g1_xyzz_madd(acc, p); else g1_xyzz_madd_ps(acc, p);
  }
  accs[t] = acc;
}

template <class F> static float time_kernel(F launch, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int blocks = prop.multiProcessorCount * 8, threads = 256;
  const double lanes = (double)blocks * threads;
  uint32_t *out; CK(hipMalloc(&out, lanes * 4));
  const double simds = prop.multiProcessorCount * 4.0, ghz = 2.4;
#define RUN_PROBE(k, ops_per_iter)                                                                       \
  { float ms = time_kernel([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }); \
    double ops = lanes * ITERS * (ops_per_iter); double waveinstr = ops / 64.0;                           \
    printf("%-16s %8.3f ms  %8.2f Tlane-op/s  ~%5.2f cyc/wave-instr/SIMD @2.4GHz\n", #k, ms, ops / ms * 1e-9, ms * 1e-3 * ghz * 1e9 * simds / waveinstr); }
  RUN_PROBE(k_xor, 8) RUN_PROBE(k_add_u32, 8) RUN_PROBE(k_add3, 8) RUN_PROBE(k_fma_f32, 8)
  RUN_PROBE(k_mad64, 8) RUN_PROBE(k_mul_lo, 8) RUN_PROBE(k_mul_hi, 8) RUN_PROBE(k_mul_u24, 8) RUN_PROBE(k_mul_hi_u24, 8) RUN_PROBE(k_mad_u24, 8)
  RUN_PROBE(k_fma_f64, 8) RUN_PROBE(k_add_f64, 8) RUN_PROBE(k_lshl_add_u64, 8) RUN_PROBE(k_mad_addc, 8)

  // field multiplier variants: correctness vs the plain C++ version, then throughput
  size_t nfe = (size_t)lanes * 2;
  std::vector<fe_t> h(nfe);
  uint64_t s = 0x9e3779b97f4a7c15ULL;
  for (size_t i = 0; i < nfe; i++) { for (int j = 0; j < 8; j++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].l[j] = (uint32_t)s; } h[i].l[7] &= 0x1fffffffu; }
  fe_t *d0, *d1, *d2; CK(hipMalloc(&d0, nfe * 32)); CK(hipMalloc(&d1, nfe * 32)); CK(hipMalloc(&d2, nfe * 32));
  CK(hipMemcpy(d0, h.data(), nfe * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, h.data(), nfe * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, h.data(), nfe * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_fqmul<0>, dim3(blocks), dim3(threads), 0, 0, d0);
  hipLaunchKernelGGL(k_fqmul<1>, dim3(blocks), dim3(threads), 0, 0, d1);
  CK(hipDeviceSynchronize());
  std::vector<fe_t> r0(nfe), r1(nfe);
  CK(hipMemcpy(r0.data(), d0, nfe * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d1, nfe * 32, hipMemcpyDeviceToHost));
  size_t bad = 0; for (size_t i = 0; i < nfe; i++) if (memcmp(&r0[i], &r1[i], 32)) bad++;
  printf("fq_mul_ps vs Fq::mul mismatches: %zu of %zu\n", bad, nfe);
  // sqr variant check: a = a^2 * b chain computed with variant 1 style ops on host side is not available; compare GPU variant2 with variant built from mul_ps
  {
    // variant 2 result must equal: repeat { a = a*a; a = a*b } using Fq::mul -> reuse kernel by a tiny checker kernel
  }
#define RUN_MUL(V, name)                                                                                                                \
  { float ms = time_kernel([&] { hipLaunchKernelGGL(k_fqmul<V>, dim3(blocks), dim3(threads), 0, 0, d2); });                                \
    double muls = lanes * MULS; printf("%-24s %8.3f ms  %8.2f G fieldmul/s  ~%6.0f cyc/wave-mul/SIMD @2.4GHz\n", name, ms, muls / ms * 1e-6, ms * 1e-3 * ghz * 1e9 * simds / (muls / 64.0)); }
  RUN_MUL(0, "Fq::mul (C++ CIOS)") RUN_MUL(1, "fq_mul_ps (asm FIPS)") RUN_MUL(2, "fq_sqr_ps+mul_ps")
#define RUN_MUL29(V, name)                                                                                                              \
  { float ms = time_kernel([&] { hipLaunchKernelGGL(k_fq29mul<V>, dim3(blocks), dim3(threads), 0, 0, d2); });                              \
    double muls = lanes * MULS; printf("%-24s %8.3f ms  %8.2f G fieldmul/s  ~%6.0f cyc/wave-mul/SIMD @2.4GHz\n", name, ms, muls / ms * 1e-6, ms * 1e-3 * ghz * 1e9 * simds / (muls / 64.0)); }
  {
    CK(hipMemcpy(d1, h.data(), nfe * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fq29mul<0>, dim3(blocks), dim3(threads), 0, 0, d1); CK(hipDeviceSynchronize());
    CK(hipMemcpy(r1.data(), d1, nfe * 32, hipMemcpyDeviceToHost));
    size_t bad29 = 0; for (size_t i = 0; i < nfe; i++) if (memcmp(&r0[i], &r1[i], 32)) bad29++;
    printf("Fq29::mul chain vs Fq::mul chain mismatches (inputs must be < p for equality; random inputs here are < 2^253): %zu of %zu\n", bad29, nfe);
  }
  RUN_MUL29(0, "Fq29::mul (9x29)") RUN_MUL29(1, "Fq29 sqr+mul") RUN_MUL29(2, "Fq29 mul + lazy add/sub") RUN_MUL29(3, "Fq29::mul2 (dual acc)")
  RUN_MUL29(4, "Fq29::mul_c (chained mad)") RUN_MUL29(5, "Fq29 sqr_c+mul_c")
  {
    CK(hipMemcpy(d1, h.data(), nfe * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, h.data(), nfe * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fq29mul<4>, dim3(blocks), dim3(threads), 0, 0, d1); hipLaunchKernelGGL(k_fq29mul<0>, dim3(blocks), dim3(threads), 0, 0, d2); CK(hipDeviceSynchronize());
    CK(hipMemcpy(r1.data(), d1, nfe * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r0.data(), d2, nfe * 32, hipMemcpyDeviceToHost));
    size_t badc = 0; for (size_t i = 0; i < nfe; i++) if (memcmp(&r0[i], &r1[i], 32)) badc++;
    printf("Fq29::mul_c chain vs Fq29::mul chain mismatches: %zu of %zu\n", badc, nfe);
  }
  for (int bpc : {1, 2, 4}) {
    int b2 = prop.multiProcessorCount * bpc;
    float m0 = time_kernel([&] { hipLaunchKernelGGL(k_fq29mul<0>, dim3(b2), dim3(threads), 0, 0, d2); });
    float m3 = time_kernel([&] { hipLaunchKernelGGL(k_fq29mul<3>, dim3(b2), dim3(threads), 0, 0, d2); });
    printf("Fq29 %d waves/SIMD: mul %8.2f G/s   mul2 %8.2f G/s\n", bpc, (double)b2 * threads * MULS / m0 * 1e-6, (double)b2 * threads * MULS / m3 * 1e-6);
  }
  // occupancy sensitivity: fewer blocks
  for (int bpc : {1, 2, 4}) {
    int b2 = prop.multiProcessorCount * bpc;
    float ms = time_kernel([&] { hipLaunchKernelGGL(k_fqmul<1>, dim3(b2), dim3(threads), 0, 0, d2); });
    printf("fq_mul_ps %d blocks/CU: %8.3f ms %8.2f G fieldmul/s\n", bpc, ms, (double)b2 * threads * MULS / ms * 1e-6);
  }
  // XYZZ madd chain
  {
    const int npts = 4096, iters = 256;
    std::vector<g1_affine_t> hp(npts); for (int i = 0; i < npts; i++) { hp[i].x = h[2 * i]; hp[i].y = h[2 * i + 1]; }  // arbitrary field elements: formulas don't care about curve membership for timing
    g1_affine_t *dp; g1_xyzz_t *da, *db; CK(hipMalloc(&dp, npts * 64)); CK(hipMalloc(&da, (size_t)lanes * 128)); CK(hipMalloc(&db, (size_t)lanes * 128));
    CK(hipMemcpy(dp, hp.data(), npts * 64, hipMemcpyHostToDevice)); CK(hipMemset(da, 0, (size_t)lanes * 128)); CK(hipMemset(db, 0, (size_t)lanes * 128));
    hipLaunchKernelGGL(k_madd<0>, dim3(blocks), dim3(threads), 0, 0, da, dp, npts, iters);
    hipLaunchKernelGGL(k_madd<1>, dim3(blocks), dim3(threads), 0, 0, db, dp, npts, iters);
    CK(hipDeviceSynchronize());
    std::vector<g1_xyzz_t> ra((size_t)lanes), rb((size_t)lanes);
    CK(hipMemcpy(ra.data(), da, (size_t)lanes * 128, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), db, (size_t)lanes * 128, hipMemcpyDeviceToHost));
    size_t badm = 0; for (size_t i = 0; i < (size_t)lanes; i++) if (memcmp(&ra[i], &rb[i], 128)) badm++;
    printf("g1_xyzz_madd_ps vs g1_xyzz_madd mismatches: %zu of %zu\n", badm, (size_t)lanes);
    for (int bpc : {1, 2, 3, 4, 8}) {
      int b2 = prop.multiProcessorCount * bpc;
      float ms = time_kernel([&] { hipLaunchKernelGGL(k_madd29<true>, dim3(b2), dim3(threads), 0, 0, da, dp, npts, iters); });
      float ms0 = time_kernel([&] { hipLaunchKernelGGL(k_madd29<false>, dim3(b2), dim3(threads), 0, 0, da, dp, npts, iters); });
      float msc = time_kernel([&] { hipLaunchKernelGGL((k_madd29<true, true>), dim3(b2), dim3(threads), 0, 0, db, dp, npts, iters); });
      printf("xyzz29 madd chain %d waves/SIMD: fused-Y3 %8.2f G madd/s   unfused %8.2f G madd/s   fused + chained mads %8.2f G madd/s\n", bpc, (double)b2 * threads * iters / ms * 1e-6, (double)b2 * threads * iters / ms0 * 1e-6, (double)b2 * threads * iters / msc * 1e-6);
    }
    // batched-affine candidate: entries per second (2 per pair) against the madd chain's, at the occupancy its LDS use allows
    {
      const int it2 = 512;
#define RUN_AFF(B, INV, bpc)                                                                                                              \
      { int b2 = prop.multiProcessorCount * (bpc); size_t ldsb = (size_t)(B) * 9 * 256 * 4;                                                 \
        CK(hipFuncSetAttribute((const void *)k_affine_pairs<B, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
        float ms = time_kernel([&] { hipLaunchKernelGGL((k_affine_pairs<B, INV>), dim3(b2), dim3(threads), ldsb, 0, db, dp, npts, it2); });   \
        printf("affine pairs + madd  B=%2d inv=%s  %d blocks/CU (LDS %3zu KB/block): %8.2f G entries/s\n", B, INV ? "fermat" : "none  ", bpc, ldsb >> 10, (double)b2 * threads * it2 / ms * 1e-6); }
      RUN_AFF(4, 0, 2) RUN_AFF(8, 0, 2) RUN_AFF(16, 0, 1) RUN_AFF(8, 0, 1)
      RUN_AFF(8, 1, 2) RUN_AFF(16, 1, 1)
      for (int bpc : {1, 2, 3}) { int b2 = prop.multiProcessorCount * bpc; float ms = time_kernel([&] { hipLaunchKernelGGL((k_madd29<true, false>), dim3(b2), dim3(threads), 0, 0, da, dp, npts, it2); });
        printf("plain madd chain (same table, same loop) %d blocks/CU: %8.2f G entries/s\n", bpc, (double)b2 * threads * it2 / ms * 1e-6); }
    }
    for (int v = 0; v < 2; v++) {
      float ms = time_kernel([&] { if (v == 0) hipLaunchKernelGGL(k_madd<0>, dim3(blocks), dim3(threads), 0, 0, da, dp, npts, iters); else hipLaunchKernelGGL(k_madd<1>, dim3(blocks), dim3(threads), 0, 0, db, dp, npts, iters); });
      printf("xyzz madd chain variant %d: %8.3f ms  %8.2f G madd/s\n", v, ms, lanes * iters / ms * 1e-6);
    }
  }
  return 0;
}
