// ntt29.hpp -- the NTT passes of ntt.hpp on the 9 x 29-bit unsaturated field (fp29.hpp).
//
// Same decomposition, tiling and global access pattern as ntt.hpp (strided passes + digit-reversing final pass); what
// changes is the arithmetic inside the tile: one v_mad_u64_u32 per limb product and lazy additions.
//   * data stay in the ABI domain (x * 2^256): they are only re-sliced (from_sat_plain) on load; twiddles are kept as
//     w * 2^261 mod r (canonical, SoA tables), so Montgomery products with R' = 2^261 land back in the x * 2^256 domain
//   * butterfly (DIF): sum = carry(u + v), dif = (u - v + 64 r) * w  -- no branch for w = 1 (table entry 0 is the unit)
//   * value growth: a sum doubles the bound; after stages 5 and 10 of a tile the sums are brought back below 2r with
//     reduce_small (no multiplication); differences come out of a multiplication (< 1.6 r).  Bounds: start < 1.4 r (a pass reads canonical
//     input or the previous pass's multiplication output),
//     <= 44.8 r before a reduction, 64 r is the limit of sub64 / reduce_small.  (Round 3: the trivial-twiddle differences of a tile's LAST stage
//     stay un-reduced, < 103 r, when their consumer allows it -- a pass then starts below 1.62 r and reaches 51.6 r before its reduction.)
//   * elements leave a pass through a multiplication (inter-level twiddle, or the ifft / coset factor) or reduce_small,
//     then (closing pass) one conditional subtraction: everything the caller sees is canonical, so results stay bit-exact; the strided passes
//     leave the tight multiplication output (< 1.4 r) in the scratch buffer as it is.
// LDS: 36 B per element as two 16-byte planes + one 4-byte plane (4096-element tile = 144 KiB of the 160 KiB).
#pragma once
#include "fp29.hpp"
#include "fp_asm.hpp"
#include "ntt.hpp"
#include "ntt_types.hpp"

namespace zk {

#ifndef ZK_NTT_ODD_FIRST
#define ZK_NTT_ODD_FIRST 1   // see lds_dif29
#endif
#ifndef ZK_NTT_LAZY_LAST
#define ZK_NTT_LAZY_LAST true   // trivial-twiddle differences of a tile's last stage stay un-reduced (see lds_dif29_round)
#endif
#ifndef ZK_GATE_CHAIN
#define ZK_GATE_CHAIN true   // k_fr_gate_eval's products as column blocks of chained v_mad (fp29.hpp mul_c), as in the NTT butterflies and the bucket accumulation: 16 instructions fewer per multiplication (round 6 A/B: profiles/r06_gate_chain_ab.json)
#endif
#ifndef ZK_NTT_CHAIN
#define ZK_NTT_CHAIN true    // limb products of the NTT butterflies as column blocks of chained v_mad (fp29.hpp mul_c): 8.61 vs 8.86 ms at 2^26 in round 3 (round 2 measured no gain; false restores the C++ multiplier for A/B builds)
#endif
// tw_in / has_in (round 6, the coset shift folded into the first pass): element (m, column) of the FIRST strided pass is multiplied by tw_in[m] = (f^(2^log_t))^m on load, and the
// pass's inter-level table (direct 2 layout) carries f^column next to w_S^(column k) -- together the f^i of distribute_powers, for one multiplication per element and no pass of its own
struct Ntt29Level { uint32_t log_m, log_t, split; Tw29 tw_m, tw_s_lo, tw_s_hi; uint32_t direct; Tw29 tw_in = {nullptr, nullptr, nullptr}; uint32_t has_in = 0; };   // direct 1: tw_s_lo holds every inter-level twiddle w_S^e (small levels), no lo x hi product; direct 2: tw_s_lo is the table [k][column] = w_S^(column k) of a big level, read like the data (8 adjacent columns per row)

#if defined(__HIPCC__)
__device__ __forceinline__ fe29_t tw29_load(const Tw29 &T, uint32_t i) {
  const uint4 a = T.lo[i], b = T.hi[i]; fe29_t r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = T.top[i]; return r;
}
struct Lds29 { uint4 *lo; uint4 *hi; uint32_t *top; };
__device__ __forceinline__ Lds29 lds29_carve(uint4 *base, uint32_t elems) { Lds29 L; L.lo = base; L.hi = base + elems; L.top = reinterpret_cast<uint32_t *>(base + 2 * elems); return L; }
__device__ __forceinline__ fe29_t lds29_get(const Lds29 &L, uint32_t i) {
  const uint4 a = L.lo[i], b = L.hi[i]; fe29_t r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = L.top[i]; return r;
}
__device__ __forceinline__ void lds29_put(const Lds29 &L, uint32_t i, const fe29_t &v) {
  L.lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); L.hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]); L.top[i] = v.l[8];
}
// u - v + 64 r, limb-wise, no carry: v limbs <= 2^30 - 2, value(v) < 63.9 r; result limbs < 2^31.4 (multiplication operand only)
__device__ __forceinline__ fe29_t fr29_sub64(const fe29_t &u, const fe29_t &v) {
  constexpr uint32_t c[9] = {0x40000040u, 0x43eb27deu, 0x5709143cu, 0x54243cdau, 0x4174a0cdu, 0x56d03029u, 0x49b85043u, 0x57098cffu, 0xc19139au};
  fe29_t r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = u.l[i] + c[i] - v.l[i];
  return r;
}
// canonical ABI element from a tight value (< 2r, exact limbs)
__device__ __forceinline__ fe_t fr29_finish(const fe29_t &t) { return Fr29::to_sat_plain(Fr29::cond_sub_p(t)); }

// One round = R consecutive radix-2 DIF stages done in registers: a work item owns the 2^R elements that differ only in the
// R index bits those stages pair up, so the tile makes one LDS round trip and one barrier per R stages (3 per 9-stage tile
// instead of 9) and each lane carries 2^(R-1) independent multiplications per stage (ILP instead of occupancy).
// LAST (the round that ends the tile, b_lo = 0): the twiddle of a butterfly then depends on its position inside the work item only, and
// 2^R - 1 of the R * 2^(R-1) butterflies have the twiddle 1 (7 of 12 for R = 3): their multiplication is dropped at compile time -- the
// difference is brought back below 2r with reduce_small instead (~40 instructions against ~220).  Over a 2^26 transform this removes
// 2.6 of the 17 multiplications per element.
template <int R, bool LAST> __device__ __forceinline__ void lds_dif29_round(const Lds29 &L, uint32_t log_m, uint32_t log_c, uint32_t sm, uint32_t sc,
                                                                const Tw29 &tw_m, bool col_fast, uint32_t s0, bool lazy_last) {
  constexpr uint32_t Q = 1u << R;
  const uint32_t M = 1u << log_m, C = 1u << log_c, b_lo = log_m - s0 - R, items = (M >> R) << log_c;
  for (uint32_t g = threadIdx.x; g < items; g += blockDim.x) {
    uint32_t c, rest;
    if (col_fast) { c = g & (C - 1); rest = g >> log_c; } else { rest = g & ((M >> R) - 1); c = g >> (log_m - R); }
    const uint32_t low = rest & ((1u << b_lo) - 1), base = ((rest >> b_lo) << (b_lo + R)) | low;
    // all twiddles of the round are fetched first (L1/L2 hits, but ~500 cycles each): R * 2^(R-1) independent loads in flight
    // while the LDS reads and the first multiplications run
    fe29_t tw[R][Q / 2];
#pragma unroll
    for (int t = 0; t < R; t++) {
      const uint32_t stage = s0 + t, bit = R - 1 - t;
      uint32_t n = 0;
#pragma unroll
      for (uint32_t q0 = 0; q0 < Q; q0++) {
        if (q0 & (1u << bit)) continue;
        if (LAST && (q0 & ((1u << bit) - 1)) == 0) { n++; continue; }   // twiddle 1: never loaded, never multiplied
        tw[t][n++] = tw29_load(tw_m, (low | ((q0 & ((1u << bit) - 1)) << b_lo)) << stage);
      }
    }
    fe29_t x[Q];
    // A sum of two CARRIED values needs no carry pass of its own: its limbs stay below 2^30 + 16, which the next stage's sum (32-bit limbs),
    // difference (fr29_sub64: u < 2^30.1, v <= 2^30 + 64) and the pass-closing multiplication all accept.  Only sums with an un-carried operand
    // are carried -- half the carry passes of a round.  Whether the operands of a butterfly are un-carried is a function of the stage and of
    // the index bits already processed in this round (values read from LDS count as un-carried: the previous round stores its last sums as
    // they are; a difference is a multiplication output): loose_t = (bit processed at stage t - 1 is 0) and not loose_(t-1), loose_0 = true.
    // Codegen note: the choice is written as a limb-wise select between the two variants below.  As `if (in_loose) sum = carry(sum)` on the
    // 36-byte value the compiler kept the butterfly operands in scratch memory (ScratchSize 240, 90 registers) and the transform took 29 ms
    // instead of 10; check `hipcc -S` (ScratchSize 0, 122 registers) after touching this loop.
#pragma unroll
    for (uint32_t q = 0; q < Q; q++) x[q] = lds29_get(L, (base | (q << b_lo)) * sm + c * sc);
#pragma unroll
    for (int t = 0; t < R; t++) {
      const uint32_t stage = s0 + t, bit = R - 1 - t;
      const bool reduce_now = (stage == 4 || stage == 9);
      uint32_t n = 0;
#pragma unroll
      for (uint32_t q0 = 0; q0 < Q; q0++) {
        if (q0 & (1u << bit)) continue;
        const uint32_t q1 = q0 | (1u << bit);
        const fe29_t u = x[q0], v = x[q1];
        // both operands share the processed bits, hence one flag; R <= 3: stage 0 carries, stage 1 does not, stage 2 carries the sums of stage-1 sums
        static_assert(R <= 3, "closed form of the recurrence for three stages");
        const bool in_loose = t == 0 ? true : t == 1 ? false : !((q0 >> (R >= 2 ? R - 2 : 0)) & 1u);
        const fe29_t s_raw = Fr29::add(u, v), s_car = Fr29::carry(s_raw);
        fe29_t sum;
#pragma unroll
        for (int i = 0; i < 9; i++) sum.l[i] = in_loose ? s_car.l[i] : s_raw.l[i];   // a compile-time choice per butterfly: the unused variant is dead code
        if (reduce_now) sum = Fr29::reduce_small(Fr29::normalise(s_raw));            // normalise is the full carry propagation
        // u - v + 64 r < 103 r: reduce_small is exact up to 2^261 = 168 r (host-checked), result tight < 2 r.  In the LAST STAGE of the tile
        // (t == R - 1 of the closing round) the difference is left as it is: nothing adds to it any more, and both consumers accept a loose
        // value below 103 r with limbs < 2^31.4 -- the strided pass multiplies every output by a canonical twiddle (9 limb products
        // < 2^60.4 per column plus the reduction terms stay below 2^64; (103 r)(r) / 2^261 + r < 1.7 r), the closing pass ends with
        // reduce_small(normalise(.)) or a multiplication itself.  One reduction (~70 instructions) per butterfly of that stage saved.
        if (LAST && (q0 & ((1u << bit) - 1)) == 0) { x[q1] = (ZK_NTT_LAZY_LAST && lazy_last && t == R - 1) ? fr29_sub64(u, v) : Fr29::reduce_small(Fr29::normalise(fr29_sub64(u, v))); n++; }
        else x[q1] = Fr29::mul_t<ZK_NTT_CHAIN>(fr29_sub64(u, v), tw[t][n++]);
        x[q0] = sum;
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < Q; q++) lds29_put(L, (base | (q << b_lo)) * sm + c * sc, x[q]);
  }
  __syncthreads();
}
// lazy_last: the consumer of the tile accepts loose last-stage differences (< 103 r): true for the strided passes (every output is multiplied
// by an inter-level twiddle that is canonical, or the product of two canonical table entries: (103 r)(1.006 r) / 2^261 + r < 1.62 r) and for a
// closing pass without post-scaling (it ends with reduce_small(normalise(.)), exact below 168 r); a closing pass that multiplies by a
// post-scaling constant (tight, < 2 r) needs the reduced value ((103 r)(2 r) / 2^261 + r would exceed the single conditional subtraction).
template <int RMAX> __device__ __forceinline__ void lds_dif29(const Lds29 &L, uint32_t log_m, uint32_t log_c, uint32_t sm, uint32_t sc, const Tw29 &tw_m, bool col_fast, bool lazy_last) {
  uint32_t s = 0;
  // an odd tile with radix-4 rounds takes its single radix-2 stage FIRST (round 3): the closing round is then a LAST radix-4 round, which drops the
  // unit twiddles of the last TWO stages (0.75 multiplications per element) instead of the last one (0.5) -- the round count stays the same
  if (ZK_NTT_ODD_FIRST && RMAX == 2 && (log_m & 1u) && log_m >= 3) { lds_dif29_round<1, false>(L, log_m, log_c, sm, sc, tw_m, col_fast, 0, false); s = 1; }
  while (s < log_m) {
    const uint32_t left = log_m - s;
    if (RMAX >= 3 && left >= 3 && left != 4) { if (left == 3) lds_dif29_round<3, true>(L, log_m, log_c, sm, sc, tw_m, col_fast, s, lazy_last); else lds_dif29_round<3, false>(L, log_m, log_c, sm, sc, tw_m, col_fast, s, false); s += 3; }
    else if (RMAX >= 2 && left >= 2) { if (left == 2) lds_dif29_round<2, true>(L, log_m, log_c, sm, sc, tw_m, col_fast, s, lazy_last); else lds_dif29_round<2, false>(L, log_m, log_c, sm, sc, tw_m, col_fast, s, false); s += 2; }
    else { if (left == 1) lds_dif29_round<1, true>(L, log_m, log_c, sm, sc, tw_m, col_fast, s, lazy_last); else lds_dif29_round<1, false>(L, log_m, log_c, sm, sc, tw_m, col_fast, s, false); s += 1; }
  }
}
__device__ __forceinline__ fe29_t load_input29(const fe_t *__restrict__ src, uint64_t gi, uint64_t src_len, const fe_t *__restrict__ pre3) {
  if (gi >= src_len) return Fr29::zero();
  fe29_t v = Fr29::from_sat_plain(g_load(&src[gi]));
  if (pre3) { const uint32_t r3 = (uint32_t)(gi % 3); if (r3) v = Fr29::mul(v, Fr29::from_sat(g_load(&pre3[r3]))); }   // factor (c_sat << 5) = c * 2^261 < 2^259: output < 1.3 r
  return v;
}

// Raw scratch (round 4 experiment, MI355_NTT_RAW_SCRATCH=1): between passes an element stays in its 9 x 29-bit limbs, as three planes (two 16-byte, one
// 4-byte: 36 B per element) instead of being re-sliced to 8 x 32 bits on the way out and back on the way in (~55 VALU instructions per element and
// pass boundary).  The strided passes' multiplication output is normalised (limbs < 2^29), so from_sat_plain(to_sat_plain(t)) == t: same bits either way.
struct Raw29 { uint4 *lo; uint4 *hi; uint32_t *top; };
// Several equal-size transforms in one launch (round 4): blockIdx.y picks the vector.  Small transforms (2^19 .. 2^22: the many-column layers run
// thousands per proof) are one round of workgroups each, so between two launches the device ramps down and up again; batched, workgroups of the next
// vector start as those of the previous one finish.  srcs == nullptr: the single-vector launch.
struct NttBatch { const fe_t *const *srcs; fe_t *const *dsts; };
__device__ __forceinline__ fe29_t raw29_load(const Raw29 &R, uint64_t i) {
  const uint4 a = R.lo[i], b = R.hi[i]; fe29_t r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = R.top[i]; return r;
}
__device__ __forceinline__ void raw29_store(const Raw29 &R, uint64_t i, const fe29_t &v) {
  R.lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); R.hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]); R.top[i] = v.l[8];
}
// MODE 0: source and destination in the ABI form (8 x 32); 1: ABI source, raw destination (first pass); 2: raw source and destination (second strided pass, in place)
template <int RMAX, int MODE = 0> __global__ void __launch_bounds__(RMAX >= 2 ? 512 : 1024) k_ntt29_strided(const fe_t *__restrict__ src, fe_t *__restrict__ dst, Ntt29Level L, uint32_t log_c,
                                                        uint64_t src_len, const fe_t *__restrict__ pre3, Raw29 raw = Raw29{nullptr, nullptr, nullptr}, NttBatch batch = NttBatch{nullptr, nullptr}) {
  extern __shared__ uint4 lds[];
  if (batch.srcs) { src = batch.srcs[blockIdx.y]; dst = batch.dsts[blockIdx.y]; }
  const uint32_t M = 1u << L.log_m, C = 1u << log_c, tile = M << log_c;
  const Lds29 S = lds29_carve(lds, tile);
  const uint32_t cb_per_sub = 1u << (L.log_t - log_c);
  const uint64_t sub = blockIdx.x >> (L.log_t - log_c);
  const uint32_t cb = blockIdx.x & (cb_per_sub - 1);
  const uint64_t base = (sub << (L.log_m + L.log_t)) + ((uint64_t)cb << log_c);
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t c = e & (C - 1), m = e >> log_c;
    fe29_t v = MODE == 2 ? raw29_load(raw, base + ((uint64_t)m << L.log_t) + c) : load_input29(src, base + ((uint64_t)m << L.log_t) + c, src_len, pre3);
    // canonical input (< r, exact limbs) times a canonical table entry: tight (< 1.4 r), what a pass may start from (header); uniform branch (kernel argument)
    if (L.has_in) v = Fr29::mul_t<ZK_NTT_CHAIN>(v, tw29_load(L.tw_in, m));
    lds29_put(S, e, v);
  }
  __syncthreads();
  lds_dif29<RMAX>(S, L.log_m, log_c, C, 1, L.tw_m, true, true);
  const uint32_t smask = (1u << L.split) - 1;
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t c = e & (C - 1), k = e >> log_c;
    const fe29_t v = lds29_get(S, (bitrev32(k, L.log_m) << log_c) + c);
    const uint32_t ex = ((cb << log_c) + c) * k;     // inter-level twiddle w_S^(col * k); entry 0 of both tables is the unit
    // the passes are ALU-bound and leave most of the HBM bandwidth idle: a big level reads its twiddles from a 2^log_s-entry table laid out
    // like the data (coalesced) instead of multiplying two half-size table entries -- one multiplication per element less in the first pass
    const fe29_t w = L.direct == 2 ? tw29_load(L.tw_s_lo, ((uint32_t)k << L.log_t) + (cb << log_c) + c)
                   : L.direct ? tw29_load(L.tw_s_lo, ex) : Fr29::mul_t<ZK_NTT_CHAIN>(tw29_load(L.tw_s_lo, ex & smask), tw29_load(L.tw_s_hi, ex >> L.split));
    // the strided passes only ever write the library's scratch buffer: their outputs stay the multiplication's tight value (< 1.4 r < 2^256, exact
    // limbs) re-sliced to 8 x 32 bits -- no conditional subtraction; the next pass starts from < 1.4 r (five doublings: < 45 r < the 64 r limit) and
    // only the closing pass, whose output the caller sees, makes everything canonical
    if (MODE == 0) g_store(&dst[base + ((uint64_t)k << L.log_t) + c], Fr29::to_sat_plain(Fr29::mul_t<ZK_NTT_CHAIN>(v, w)));
    else raw29_store(raw, base + ((uint64_t)k << L.log_t) + c, Fr29::mul_t<ZK_NTT_CHAIN>(v, w));
  }
}

template <int RMAX, int MODE = 0> __global__ void __launch_bounds__(RMAX >= 2 ? 512 : 1024) k_ntt29_final(const fe_t *__restrict__ src, fe_t *__restrict__ dst, uint32_t log_m, uint32_t log_a, uint32_t log_b,
                                                      uint32_t log_c, Tw29 tw_m, uint64_t src_len, const fe_t *__restrict__ pre3, const fe_t *__restrict__ post3, Raw29 raw = Raw29{nullptr, nullptr, nullptr}, NttBatch batch = NttBatch{nullptr, nullptr}) {
  extern __shared__ uint4 lds[];
  if (batch.srcs) { src = batch.srcs[blockIdx.y]; dst = batch.dsts[blockIdx.y]; }
  const uint32_t M = 1u << log_m, C = 1u << log_c, seg = M + 1, tile = M << log_c;
  const Lds29 S = lds29_carve(lds, seg << log_c);
  const uint32_t k2 = blockIdx.x & ((1u << log_b) - 1);
  const uint32_t k1_0 = (blockIdx.x >> log_b) << log_c;
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t m = e & (M - 1), c = e >> log_m;
    const uint64_t q = ((uint64_t)(k1_0 + c) << log_b) + k2;
    lds29_put(S, c * seg + m, MODE == 2 ? raw29_load(raw, (q << log_m) + m) : load_input29(src, (q << log_m) + m, src_len, pre3));
  }
  __syncthreads();
  lds_dif29<RMAX>(S, log_m, log_c, 1, seg, tw_m, false, post3 == nullptr);
  const uint32_t log_stride = log_a + log_b;  // N / M
  fe29_t post0 = Fr29::zero(), post1 = post0, post2 = post0;   // named (not an array): runtime-indexed arrays go to scratch
  if (post3) { post0 = Fr29::reduce_small(Fr29::from_sat(g_load(&post3[0]))); post1 = Fr29::reduce_small(Fr29::from_sat(g_load(&post3[1]))); post2 = Fr29::reduce_small(Fr29::from_sat(g_load(&post3[2]))); }   // c * 2^261, tight
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t c = e & (C - 1), k = e >> log_c;
    const fe29_t v = lds29_get(S, c * seg + bitrev32(k, log_m));
    const uint64_t oi = (uint64_t)(k1_0 + c) + ((uint64_t)k2 << log_a) + ((uint64_t)k << log_stride);
    fe29_t t;
    if (post3) { const uint32_t r3 = (uint32_t)(oi % 3); t = Fr29::mul(v, r3 == 0 ? post0 : (r3 == 1 ? post1 : post2)); }
    else t = Fr29::reduce_small(Fr29::normalise(v));
    g_store(&dst[oi], fr29_finish(t));
  }
}

// ---- eval_polynomial (halo2_proofs::arithmetic::eval_polynomial, step 9 of create_proof: evaluations at x * omega^rot):
// p(x) = sum_i c_i x^i as a streaming reduction: each thread runs Horner over a contiguous run of EVAL_RUN coefficients
// (one multiplication per 32-byte coefficient read: the one kernel of the path that is close to HBM-bound), scales by
// x^(start of run), then the block sums its values (wavefront shuffles + LDS) into one partial per block.
constexpr uint32_t EVAL_RUN = 64;   // coefficients per thread; a block covers 256 * EVAL_RUN consecutive coefficients
__device__ __forceinline__ fe29_t shfl_down_fe29(const fe29_t &v, uint32_t o) { fe29_t r; for (int i = 0; i < 9; i++) r.l[i] = __shfl_down(v.l[i], o); return r; }
__device__ __forceinline__ fe29_t fr29_pow_u64(const fe29_t &x, uint64_t e) {
  fe29_t r = Fr29::one(), sq = x;
  while (e) { if (e & 1) r = Fr29::mul(r, sq); e >>= 1; if (e) sq = Fr29::sqr(sq); }
  return r;
}
// Thread t of a block takes the coefficients base + t + 256 k (k < EVAL_RUN): loads are coalesced (consecutive lanes, consecutive
// 32-byte coefficients) and every thread runs Horner in the SAME y = x^256, so the per-coefficient cost is one multiplication;
// the thread-specific factor x^t and the block factor x^base (computed once per block, broadcast through LDS) are applied at the end.
__device__ __forceinline__ void eval_poly_partial_body(const fe_t *__restrict__ poly, uint64_t n, const fe_t &x_sat, fe_t *__restrict__ partial) {
  __shared__ uint32_t lds[5][9];
  const uint64_t base = (uint64_t)blockIdx.x * 256 * EVAL_RUN;
  const fe29_t x = Fr29::reduce_small(Fr29::from_sat(x_sat));          // x * 2^261, tight
  if (threadIdx.x < 64) {                                               // wave 0: x^base for the whole block (all lanes compute the same value)
    const fe29_t xb = fr29_pow_u64(x, base);
    if (threadIdx.x == 0) for (int k = 0; k < 9; k++) lds[4][k] = xb.l[k];
  }
  fe29_t y = x;
#pragma unroll
  for (int i = 0; i < 8; i++) y = Fr29::sqr(y);                        // x^256
  fe29_t acc = Fr29::zero();
  bool any = false;
  for (int k = EVAL_RUN - 1; k >= 0; k--) {
    const uint64_t i = base + threadIdx.x + 256ull * (uint32_t)k;
    if (any) acc = Fr29::mul(acc, y);                                  // tight, < 1.3 r
    if (i < n) {
      const fe29_t c = Fr29::from_sat_plain(g_load(&poly[i]));         // ABI domain (c * 2^256); products with x-powers keep it
      for (int q = 0; q < 9; q++) acc.l[q] += c.l[q];                  // lazy add: limbs < 2^30, value < 2.3 r
      any = true;
    }
  }
  acc = Fr29::mul(acc, fr29_pow_u64(x, threadIdx.x));                  // * x^t (<= 8 squarings + multiplications)
  __syncthreads();
  { fe29_t xb; for (int k = 0; k < 9; k++) xb.l[k] = lds[4][k]; acc = Fr29::mul(acc, xb); }   // * x^base, tight
  for (uint32_t o = 32; o >= 1; o >>= 1) {
    const fe29_t other = shfl_down_fe29(acc, o);
    acc = Fr29::carry(Fr29::add(acc, other));
    if (o == 4) acc = Fr29::reduce_small(Fr29::normalise(acc));         // after 4 doublings: < 16 * 1.3 r -> < 2 r
  }
  acc = Fr29::reduce_small(Fr29::normalise(acc));                       // < 8 * 2 r -> < 2 r
  if ((threadIdx.x & 63) == 0) for (int k = 0; k < 9; k++) lds[threadIdx.x >> 6][k] = acc.l[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < 4; w++) { fe29_t o; for (int k = 0; k < 9; k++) o.l[k] = lds[w][k]; acc = Fr29::carry(Fr29::add(acc, o)); }
    g_store(&partial[blockIdx.x], fr29_finish(Fr29::reduce_small(Fr29::normalise(acc))));   // canonical, ABI domain
  }
}
__global__ void __launch_bounds__(256) k_eval_poly_partial(const fe_t *__restrict__ poly, uint64_t n, fe_t x_sat, fe_t *__restrict__ partial) { eval_poly_partial_body(poly, n, x_sat, partial); }
// blockIdx.y = evaluation: step 9 of create_proof evaluates thousands of (polynomial, point) pairs at k = 20, each a 64-block launch whose ~100 us are
// latency (a serial Horner chain of 64 multiplications per thread behind a power ladder); one launch over all pairs is throughput-bound instead
__global__ void __launch_bounds__(256) k_eval_poly_partial_batch(const fe_t *const *__restrict__ polys, uint64_t n, const fe_t *__restrict__ points, fe_t *__restrict__ partial, uint64_t stride) {
  eval_poly_partial_body(polys[blockIdx.y], n, g_load(&points[blockIdx.y]), partial + stride * blockIdx.y);
}
// a[i] *= f^i  (halo2_proofs distribute_powers: the coset shift of coeff_to_extended_part / general coset FFTs).
// Same tiling as k_eval_poly_partial: thread t walks i = base + t + 256 k with a running power stepped by f^256.
// src == a: in place; otherwise a = src scaled (the coset transforms write the scaled copy straight into their destination: no separate copy)
// DistFactors: f, f^256 and f^(256 EVAL_RUN) in Montgomery (ABI) form, the two powers computed by the host (round 4: the in-kernel ladder
// f^base with a 64-bit exponent cost every block ~40 dependent multiplications before its first store -- 108 us for a 2^16-element call)
struct DistFactors { fe_t f, f256, fblock; };
__device__ __forceinline__ void distribute_powers_body(const fe_t *src, fe_t *a, uint64_t n, const DistFactors &F) {
  __shared__ uint32_t lds[9];
  const uint64_t base = (uint64_t)blockIdx.x * 256 * EVAL_RUN;
  const fe29_t f = Fr29::reduce_small(Fr29::from_sat(F.f));
  if (threadIdx.x < 64) { const fe29_t fb = fr29_pow_u64(Fr29::reduce_small(Fr29::from_sat(F.fblock)), blockIdx.x); if (threadIdx.x == 0) for (int k = 0; k < 9; k++) lds[k] = fb.l[k]; }   // (f^(256 EVAL_RUN))^block: a short exponent
  const fe29_t y = Fr29::reduce_small(Fr29::from_sat(F.f256));         // f^256
  fe29_t pw = fr29_pow_u64(f, threadIdx.x);
  __syncthreads();
  { fe29_t fb; for (int k = 0; k < 9; k++) fb.l[k] = lds[k]; pw = Fr29::mul(pw, fb); }   // f^(base + t), tight
  for (uint32_t k = 0; k < EVAL_RUN; k++) {
    const uint64_t i = base + threadIdx.x + 256ull * k;
    if (i >= n) break;
    g_store(&a[i], fr29_finish(Fr29::mul(Fr29::from_sat_plain(g_load(&src[i])), pw)));
    pw = Fr29::mul(pw, y);
  }
}
__global__ void __launch_bounds__(256) k_distribute_powers(const fe_t *src, fe_t *a, uint64_t n, DistFactors F) { distribute_powers_body(src, a, n, F); }
// blockIdx.y = polynomial: the coset shift of every polynomial of a coset part in ONE launch (mi355_coset_ntt_fr_batch_dev)
__global__ void __launch_bounds__(256) k_distribute_powers_batch(const fe_t *const *srcs, fe_t *const *dsts, uint64_t n, DistFactors F) { distribute_powers_body(srcs[blockIdx.y], dsts[blockIdx.y], n, F); }

// element-wise vector operations on device-resident polynomials (the pointwise steps between the transforms of the quotient
// construction): op 0 add, 1 sub, 2 mul; and data[i] *= table[i mod period] (division by the vanishing polynomial on
// the extended coset: halo2's t_evaluations have period 2^(extended_k - k)).  Streaming, 16 B/lane accesses, grid-stride.
// no __restrict__: dst may be one of the operands (each thread reads element i of both operands before it writes element i)
__global__ void __launch_bounds__(256) k_fr_vec_op(int op, fe_t *dst, const fe_t *a, const fe_t *b, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const fe_t x = g_load(&a[i]), y = g_load(&b[i]);
    fe_t r;
    if (op == 0) r = Fr::add(x, y);
    else if (op == 1) r = Fr::sub(x, y);
    else r = fr29_finish(Fr29::mul(Fr29::from_sat_plain(x), Fr29::from_sat(y)));   // (x 2^256)(y 2^261) / 2^261
    g_store(&dst[i], r);
  }
}
// dst = a + s * b (s a scalar): the linear combinations sum_i v^i p_i(X) of the multi-open argument, one polynomial at a time
__global__ void __launch_bounds__(256) k_fr_vec_axpy(fe_t *dst, const fe_t *a, const fe_t *b, fe_t s_sat, uint64_t n) {   // dst may alias a or b
  const fe29_t s = Fr29::from_sat(s_sat);   // s * 2^261: the product with b * 2^256 lands back in the ABI domain
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const fe_t sb = fr29_finish(Fr29::mul(Fr29::from_sat_plain(g_load(&b[i])), s));
    g_store(&dst[i], a ? Fr::add(g_load(&a[i]), sb) : sb);
  }
}
__global__ void __launch_bounds__(256) k_fr_vec_mul_periodic(fe_t *__restrict__ data, uint64_t n, const fe_t *__restrict__ table, uint32_t period_mask) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    g_store(&data[i], fr29_finish(Fr29::mul(Fr29::from_sat_plain(g_load(&data[i])), Fr29::from_sat(g_load(&table[i & period_mask])))));
}

// ---- gate-shaped fused evaluation (the operand shape of halo2's evaluate_h [EXT-recalled halo2_proofs src/plonk/evaluation.rs: GraphEvaluator /
// get_rotation_idx]): dst[i] (+)= sum_j c_j * prod_k p_{jk}[(i + r_jk) mod n] for a small term list, rotations included, in
// ONE pass -- every operand is read once per use and nothing but dst is written, instead of one full HBM round trip per add / mul of a chain of
// k_fr_vec_op launches.  The term list travels as a kernel argument (scalar loads, uniform across the wavefront).  n is a power of two (the
// extended domain, or one 2^k coset part of the scroll fork); rotations arrive already scaled (rot * 2^(extended_k - k) on the extended domain).
// Arithmetic: the first factor is re-sliced in the ABI domain (x 2^256), the coefficient and every further factor enter as y 2^261, so each
// Montgomery product (R' = 2^261) lands back in the ABI domain; term values (< 2 r) are summed lazily, carried every fourth term, and reduced
// once (<= 16 terms + dst: < 34 r, below reduce_small's 64 r).
constexpr uint32_t GATE_MAX_TERMS = 16, GATE_MAX_FACTORS = 48, GATE_MAX_POLYS = 24, GATE_MAX_TERM_LEN = 16;   // a degree-9 gate of the inner circuit (selector, coefficient, seven cells) is ONE term
struct GatePlan {
  const fe_t *poly[GATE_MAX_POLYS];
  fe_t coeff[GATE_MAX_TERMS];            // Montgomery (ABI) form (constant terms)
  fe29_t coeff29[GATE_MAX_TERMS];        // the same coefficient as c * 2^261 in 29-bit limbs (Fr29::from_sat, done once on the host: round 4 -- the kernel used to re-slice every general coefficient for every row)
  int32_t factor_rot[GATE_MAX_FACTORS];
  uint8_t factor_poly[GATE_MAX_FACTORS];
  uint8_t term_len[GATE_MAX_TERMS];      // factors per term (0: the constant c_j)
  uint8_t coeff_kind[GATE_MAX_TERMS];    // 0: general coefficient, 1: c_j = 1, 2: c_j = -1 (set by the host from the coefficient bytes)
  uint32_t n_terms, accumulate;
};
// dst carries no __restrict__: it may be one of the operands (un-rotated, checked by the host) and is read when G.accumulate is set
__global__ void __launch_bounds__(256) k_fr_gate_eval(fe_t *dst, GatePlan G, uint64_t n) {
  const uint64_t mask = n - 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    fe29_t acc = Fr29::zero();
    uint32_t f = 0;
    for (uint32_t j = 0; j < G.n_terms; j++) {
      const uint32_t len = G.term_len[j];
      fe29_t t;
      if (len == 0) t = Fr29::from_sat_plain(G.coeff[j]);
      else {
        const fe_t x0 = g_load(&G.poly[G.factor_poly[f]][(i + (uint64_t)(int64_t)G.factor_rot[f]) & mask]);
        // unit coefficients (the common case in halo2 gates: a - b, z(wX) prod - z(X) prod): no multiplication by c_j; -1 negates the canonical first
        // factor instead (r - x, zero stays zero), so the term value stays a tight non-negative representative (< r) like every other
        const uint32_t kind = G.coeff_kind[j];
        if (kind == 0) t = Fr29::mul_t<ZK_GATE_CHAIN>(Fr29::from_sat_plain(x0), G.coeff29[j]);
        else t = Fr29::from_sat_plain(kind == 2 ? Fr::neg(x0) : x0);
        for (uint32_t q = 1; q < len; q++)
          t = Fr29::mul_t<ZK_GATE_CHAIN>(t, Fr29::from_sat(g_load(&G.poly[G.factor_poly[f + q]][(i + (uint64_t)(int64_t)G.factor_rot[f + q]) & mask])));
      }
      f += len;
      acc = Fr29::add(acc, t);
      if ((j & 3) == 3) acc = Fr29::carry(acc);
    }
    if (G.accumulate) acc = Fr29::add(acc, Fr29::from_sat_plain(g_load(&dst[i])));
    g_store(&dst[i], fr29_finish(Fr29::reduce_small(Fr29::normalise(acc))));
  }
}

// dst[i * Q + q] = parts[q][i]: the Q coset parts of the scroll fork's evaluate_h (part q = the evaluations at zeta * omega_ext^(q + Q i), i < n)
// laid out as the extended domain's natural order, which is what extended_to_coeff inverts.  Q <= 8 pointers travel as a kernel argument; a lane
// reads one 32-byte element per part (consecutive lanes, consecutive elements) and writes Q consecutive elements: both sides coalesced.
struct InterleavePlan { const fe_t *part[8]; uint32_t q; };
__global__ void __launch_bounds__(256) k_fr_interleave(fe_t *__restrict__ dst, InterleavePlan P, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    for (uint32_t q = 0; q < P.q; q++) g_store(&dst[i * P.q + q], g_load(&P.part[q][i]));
}

// sum of m canonical field elements (the per-block partials) by one workgroup; blockIdx.x = which vector (stride elements apart) of a batch
__global__ void __launch_bounds__(256) k_fr_sum(const fe_t *__restrict__ in_all, uint64_t m, fe_t *__restrict__ out_all, uint64_t stride = 0) {
  const fe_t *__restrict__ in = in_all + stride * blockIdx.x; fe_t *__restrict__ out = out_all + blockIdx.x;
  __shared__ fe_t lds[4];
  fe_t acc = Fr::zero();
  for (uint64_t i = threadIdx.x; i < m; i += blockDim.x) acc = Fr::add(acc, g_load(&in[i]));
  for (uint32_t o = 32; o >= 1; o >>= 1) { fe_t other; for (int k = 0; k < 8; k++) other.l[k] = __shfl_down(acc.l[k], o); acc = Fr::add(acc, other); }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { for (uint32_t w = 1; w < 4; w++) acc = Fr::add(acc, lds[w]); g_store(out, acc); }
}

// table [k][col] (col < 2^log_t) of base^(col k) * 2^261 mod r: the inter-level twiddles of a big level in the order the pass reads them
// use_col: the entry also carries colbase^col (the column part f^col of a folded coset shift; colbase in Montgomery form)
__global__ void k_pow_table29_2d(uint4 *lo, uint4 *hi, uint32_t *top, fe_t base, uint32_t log_t, uint64_t count, fe_t colbase = fe_t{}, int use_col = 0) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t k = i >> log_t, col = i & ((1ull << log_t) - 1);
  fe_t m32; { constexpr uint32_t c[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0xdc83629u}; for (int q = 0; q < 8; q++) m32.l[q] = c[q]; }   // 32 in Montgomery form
  fe_t e = Fr::pow_u64(base, k * col);
  if (use_col) e = fr_mul_ps(e, Fr::pow_u64(colbase, col));
  const fe29_t w = Fr29::from_sat_plain(fr_mul_ps(e, m32));
  lo[i] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]); hi[i] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]); top[i] = w.l[8];
}
// dst[i] = src[i] * d (d: Montgomery form of the ABI): a twiddle table with a constant folded in -- the inverse transform's divisor rides on
// the inter-level twiddles of its last strided pass instead of costing the closing pass one more multiplication per element
__global__ void k_scale_table29(const uint4 *slo, const uint4 *shi, const uint32_t *stop, uint4 *lo, uint4 *hi, uint32_t *top, fe_t d, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Tw29 S; S.lo = slo; S.hi = shi; S.top = stop;
  const fe29_t w = Fr29::cond_sub_p(Fr29::normalise(Fr29::mul(tw29_load(S, i), Fr29::from_sat(d))));   // (w 2^261)(d 2^261) / 2^261, canonical
  lo[i] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]); hi[i] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]); top[i] = w.l[8];
}
// SoA twiddle table: entry i = (base^step)^i * 2^261 mod r, canonical 29-bit limbs
__global__ void k_pow_table29(uint4 *lo, uint4 *hi, uint32_t *top, fe_t base, uint64_t step, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const fe_t b = step == 1 ? base : Fr::pow_u64(base, step);
  fe_t m32; { constexpr uint32_t c[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0xdc83629u}; for (int k = 0; k < 8; k++) m32.l[k] = c[k]; }   // 32 in Montgomery form
  const fe29_t w = Fr29::from_sat_plain(fr_mul_ps(Fr::pow_u64(b, i), m32));   // (w * 2^256) * 32 = w * 2^261 mod r, canonical
  lo[i] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]); hi[i] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]); top[i] = w.l[8];
}
#endif  // __HIPCC__

}  // namespace zk
