// mi355zk_transcript.hpp -- the Fiat-Shamir transcripts of create_proof on the host side: halo2's stock Blake2b one and the Poseidon one the reference proves its layers 0-5 with.
//
// create_proof owns the transcript: commitments and evaluations go in, challenges come out.  Both kinds WRITE the same proof bytes (a point in its 32-byte compressed form, a scalar
// as 32 canonical little-endian bytes: the reference's layout); they differ in what they hash:
//   Blake2b  [EXT-recalled halo2_proofs src/transcript/blake2b.rs: Blake2bWrite<_, _, Challenge255<_>>]: Blake2b-512, personalisation "Halo2-Transcript"; a point is absorbed as
//            0x01 | x | y (canonical little-endian coordinates), a scalar as 0x02 | bytes; a challenge = the 64-byte digest of (state | 0x00) reduced mod r.
//   Poseidon [EXT-recalled snark-verifier system/halo2/transcript/halo2.rs PoseidonTranscript<NativeLoader>, util/hash/poseidon.rs; snark-verifier-sdk: T = 5, RATE = 4, R_F = 8,
//            R_P = 60; constants from the Grain LFSR of the Poseidon paper's reference script, as the `poseidon` crate [REF Cargo.lock:2927-2929] generates them]: a scalar is absorbed
//            as itself, a point as (x mod r, y mod r); `update` only buffers; a challenge = one squeeze: the buffer is absorbed RATE words at a time into state words 1.., a short
//            chunk is followed by a 1, an exact multiple of RATE by one more permutation of an empty chunk, and state word 1 is the answer (a full field element).
//            The Python restatement of the same sponge (oracle/poseidon.py) makes the reference's RELEASED chunk and batch proofs verify
//            (tests/test_plonk_protocol.py::test_reference_released_proofs_verify); this C++ one is compared with it word for word (--transcript-selftest).
//   Evm      [EXT-recalled snark-verifier system/halo2/transcript/evm.rs EvmTranscript; spelled out by REF release-v0.13.1/evm_verifier.yul:66-100], layer 6: everything is 32-byte
//            BIG-endian words appended to a buffer -- a scalar one word, a point two (x, y), and the PROOF carries points uncompressed (64 bytes) and scalars big-endian --; a
//            challenge = Keccak-256 of the buffer (plus one byte 0x01 when the buffer is exactly one word: a squeeze right after a squeeze) mod r; the hash is the new buffer.
//            oracle/keccak.py + oracle/plonk.py EvmTranscript make the RELEASED BUNDLE PROOF verify (test_released_bundle_evm_proof_verifies); compared word for word as above.
// All refuse the identity (halo2's common_point fails on it).
// The GPU library sees none of this: 96-byte commitments and 32-byte evaluations arrive from the C-ABI and are hashed here, on the calling thread.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "mi355zk_halo2.hpp"

namespace mi355zk {
namespace plonk {

// RFC 7693 BLAKE2b, sequential mode, no key, 64-byte digest, 16-byte personalisation
struct Blake2b {
  uint64_t h[8]; uint64_t t0 = 0, t1 = 0; uint8_t buf[128]; size_t buflen = 0;
  static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
  explicit Blake2b(const char personal[16]) {
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010040ull;                         // digest length 64, no key, fanout 1, depth 1
    uint64_t p0, p1; std::memcpy(&p0, personal, 8); std::memcpy(&p1, personal + 8, 8);
    h[6] ^= p0; h[7] ^= p1;
  }
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(const uint8_t *block, bool last) {
    static const uint8_t S[12][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}, {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
                                      {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8}, {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10}, {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
                                      {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    std::memcpy(m, block, 128);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t0; v[13] ^= t1; if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
      const uint8_t *s = S[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  }
  void update(const void *data, size_t len) {
    const uint8_t *p = static_cast<const uint8_t *>(data);
    while (len) {
      if (buflen == 128) { t0 += 128; if (t0 < 128) t1++; compress(buf, false); buflen = 0; }   // a full buffer is only compressed once more input follows
      const size_t take = std::min(len, size_t(128) - buflen);
      std::memcpy(buf + buflen, p, take); buflen += take; p += take; len -= take;
    }
  }
  // digest of everything absorbed so far; the state itself is not disturbed (halo2 clones the hasher to squeeze)
  std::array<uint8_t, 64> digest() const {
    Blake2b c = *this;
    c.t0 += c.buflen; if (c.t0 < c.buflen) c.t1++;
    std::memset(c.buf + c.buflen, 0, 128 - c.buflen);
    c.compress(c.buf, true);
    std::array<uint8_t, 64> out; std::memcpy(out.data(), c.h, 64); return out;
  }
};

// Fr::from_uniform_bytes: 512-bit little-endian integer mod r, returned in Montgomery form
inline halo2::Fr fr_from_uniform_bytes(const std::array<uint8_t, 64> &b) {
  auto reduce256 = [](const uint8_t *p) {
    zk::fe_t a; std::memcpy(&a, p, 32);
    uint32_t m[8]; for (int i = 0; i < 8; i++) m[i] = zk::FrP::mod(i);
    while (zk::Fr::w_geq(a.l, m)) zk::Fr::w_sub(a.l, m);      // 2^256 < 6 r
    return zk::Fr::from_canonical(a);
  };
  const zk::fe_t lo = reduce256(b.data()), hi = reduce256(b.data() + 32);
  zk::fe_t r2; for (int i = 0; i < 8; i++) r2.l[i] = zk::FrP::r2(i);   // R^2 mod r as a Montgomery-form value IS (R mod r) = 2^256 mod r
  return halo2::detail::from_fe(zk::Fr::add(lo, zk::Fr::mul(hi, r2)));
}

// ---- Poseidon: parameters from the Grain LFSR (80-bit state: field type 1 (2 bits) | s-box 0 (4) | field bits 254 (12) | t (12) | R_F (10) | R_P (10) | thirty 1s; 160 warm-up
// clocks; self-shrinking output), round constants by rejection sampling of 254-bit draws, the Cauchy matrix 1 / (x_i + y_j) from 2 t draws reduced mod r
struct PoseidonSpec {
  static constexpr int T = 5, RATE = 4, RF = 8, RP = 60;
  std::vector<halo2::Fr> rc; halo2::Fr mds[T][T];
  static const PoseidonSpec &get() { static const PoseidonSpec s; return s; }
 private:
  PoseidonSpec() {
    std::vector<uint8_t> st;
    auto push_bits = [&](uint32_t v, int nb) { for (int i = nb - 1; i >= 0; i--) st.push_back((v >> i) & 1); };
    push_bits(1, 2); push_bits(0, 4); push_bits(254, 12); push_bits(T, 12); push_bits(RF, 10); push_bits(RP, 10); for (int i = 0; i < 30; i++) st.push_back(1);
    size_t head = 0;   // st[head .. head + 80) is the register; clocking appends and advances
    auto clock = [&]() { const uint8_t nb = st[head + 62] ^ st[head + 51] ^ st[head + 38] ^ st[head + 23] ^ st[head + 13] ^ st[head]; st.push_back(nb); head++; return nb; };
    for (int i = 0; i < 160; i++) clock();
    auto next_bit = [&]() { uint8_t nb = clock(); while (nb == 0) { clock(); nb = clock(); } return clock(); };
    uint32_t modw[8]; for (int i = 0; i < 8; i++) modw[i] = zk::FrP::mod(i);
    auto draw = [&](bool reject, bool &ok) {   // 254 bits, most significant first
      zk::fe_t a = zk::Fr::zero();
      for (int b = 253; b >= 0; b--) if (next_bit()) a.l[b / 32] |= 1u << (b % 32);
      ok = !zk::Fr::w_geq(a.l, modw);
      if (!ok && !reject) { while (zk::Fr::w_geq(a.l, modw)) zk::Fr::w_sub(a.l, modw); ok = true; }
      return halo2::detail::from_fe(zk::Fr::from_canonical(a));
    };
    while (rc.size() < (size_t)(RF + RP) * T) { bool ok; const halo2::Fr v = draw(true, ok); if (ok) rc.push_back(v); }
    halo2::Fr xs[T], ys[T]; bool ok;
    for (auto &x : xs) x = draw(false, ok);
    for (auto &y : ys) y = draw(false, ok);
    for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) mds[i][j] = halo2::detail::fr_inv(halo2::detail::from_fe(zk::Fr::add(halo2::detail::to_fe(xs[i]), halo2::detail::to_fe(ys[j]))));
  }
};
// Host arithmetic of the sponge: 4 x 64-bit Montgomery words (the ABI form: R = 2^256, what halo2curves holds), one CIOS product = 32 64x64 multiplications.  The library's
// 8 x 32 host code costs 280 us per permutation here; a layer-0 proof absorbs ~3 300 words = 830 permutations, a quarter of a second on the calling thread.  This form: ~45 us
// (round 6, MDS rows with one reduction each: see mac / redc).
struct Fr64 {
  uint64_t l[4];
  static constexpr uint64_t M[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;   // -r^-1 mod 2^64
  static bool geq_m(const uint64_t *a) { for (int i = 3; i >= 0; i--) if (a[i] != M[i]) return a[i] > M[i]; return true; }
  static void sub_m(uint64_t *a) { unsigned __int128 b = 0; for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)a[i] - M[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; } }
  static Fr64 add(const Fr64 &a, const Fr64 &b) {
    Fr64 r; unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_m(r.l)) sub_m(r.l);      // a, b < r < 2^254: no carry out; kept for form
    return r;
  }
  static Fr64 mul(const Fr64 &a, const Fr64 &b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * INV;
      c = (unsigned __int128)m * M[0] + t[0]; c >>= 64;
      for (int j = 1; j < 4; j++) { c += (unsigned __int128)m * M[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr64 r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_m(r.l)) sub_m(r.l);
    return r;
  }
  // t[0..8] += a * b (plain 512-bit product, no reduction) and the Montgomery reduction of a sum of up to FIVE such products (5 r^2 < r 2^256, so the result is below 2 r):
  // a row of the MDS product costs five multiplications and ONE reduction instead of five (round 6: 0.62 of the 64-bit multiplications of the sponge's permutation)
  static void mac(uint64_t *t, const Fr64 &a, const Fr64 &b) {
    for (int i = 0; i < 4; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
      for (int k = i + 4; c && k < 9; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
    }
  }
  static Fr64 redc(uint64_t *t) {
    for (int i = 0; i < 4; i++) {
      const uint64_t m = t[i] * INV;
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) { c += (unsigned __int128)m * M[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
      for (int k = i + 4; c && k < 9; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
    }
    Fr64 r{{t[4], t[5], t[6], t[7]}};
    if (t[8] || geq_m(r.l)) sub_m(r.l);
    return r;
  }
  static Fr64 from(const halo2::Fr &a) { Fr64 r; std::memcpy(r.l, a.data(), 32); return r; }
  halo2::Fr fr() const { halo2::Fr r; std::memcpy(r.data(), l, 32); return r; }
};
struct PoseidonSponge {
  static constexpr int T = PoseidonSpec::T, RATE = PoseidonSpec::RATE;
  Fr64 state[T]; std::vector<Fr64> buf;
  PoseidonSponge() { for (auto &w : state) w = Fr64{{0, 0, 0, 0}}; zk::fe_t c = zk::Fr::zero(); c.l[2] = 1; state[0] = Fr64::from(halo2::detail::from_fe(zk::Fr::from_canonical(c))); }   // [2^64, 0, 0, 0, 0]
  // Eager absorption (round 6): a full RATE-word chunk is permuted as soon as it is complete instead of at the next squeeze -- the same chunks in the same order, so the same
  // state -- which moves the hashing of a many-column layer's commitments (layer 0: 1 600 words before theta) from one serial block at the squeeze, with the device idle, into the
  // gaps between the commitment batches while the witness still crosses PCIe
  void update(const halo2::Fr &w) { buf.push_back(Fr64::from(w)); if (buf.size() == (size_t)RATE) { absorb_and_permute(buf.data(), RATE); buf.clear(); } }
  static Fr64 pow5(const Fr64 &a) { const Fr64 a2 = Fr64::mul(a, a); return Fr64::mul(Fr64::mul(a2, a2), a); }
  static Fr64 sub(const Fr64 &a, const Fr64 &b) { Fr64 nb{{0, 0, 0, 0}}; bool z = !(b.l[0] | b.l[1] | b.l[2] | b.l[3]); if (!z) { unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)Fr64::M[i] - b.l[i] - (uint64_t)br; nb.l[i] = (uint64_t)d; br = (d >> 64) & 1; } } return Fr64::add(a, nb); }
  static Fr64 inv(const Fr64 &a) { return Fr64::from(halo2::detail::fr_inv(a.fr())); }
  // Partial rounds with SPARSE matrices (round 6; the schedule of the Poseidon paper's appendix B, which the `poseidon` crate the reference links also runs): a dense matrix D
  // splits as D = D' D'' with D' = [[1, 0], [0, D^]] and D'' = [[D00, D[0, 1:]], [w^, I]], w^ = D^^-1 D[1:, 0]; D' leaves element 0 alone, so it commutes with the partial
  // round's S-box and is pushed through the next round's constants into the next round's matrix (M D', split again).  Round k then costs the S-box, ONE dot product and T - 1
  // multiply-adds instead of T dot products; what is left of D' after the last partial round is applied once, densely.  The tables are checked against the plain schedule on a
  // fixed state when they are built (and the transcript tests compare every squeeze with the Python restatement that verifies the reference's released proofs).
  struct Tables { std::vector<Fr64> rc; Fr64 mds[T][T]; std::vector<Fr64> prc /* RP x T adjusted constants */, row0 /* RP x T */, what /* RP x (T - 1) */; Fr64 left[T][T]; };
  static void plain_rounds(const Tables &S, Fr64 *st, int r0, int r1) {
    for (int r = r0; r < r1; r++) {
      for (int i = 0; i < T; i++) st[i] = Fr64::add(st[i], S.rc[(size_t)r * T + i]);
      if (r < PoseidonSpec::RF / 2 || r >= PoseidonSpec::RF / 2 + PoseidonSpec::RP) { for (int i = 0; i < T; i++) st[i] = pow5(st[i]); } else st[0] = pow5(st[0]);
      Fr64 nx[T];
      for (int i = 0; i < T; i++) { uint64_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; for (int j = 0; j < T; j++) Fr64::mac(t, S.mds[i][j], st[j]); nx[i] = Fr64::redc(t); }   // canonical operands: five products, one reduction
      for (int i = 0; i < T; i++) st[i] = nx[i];
    }
  }
  static void sparse_rounds(const Tables &S, Fr64 *st) {
    constexpr int RP = PoseidonSpec::RP;
    for (int k = 0; k < RP; k++) {
      for (int i = 0; i < T; i++) st[i] = Fr64::add(st[i], S.prc[(size_t)k * T + i]);
      st[0] = pow5(st[0]);
      uint64_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; for (int j = 0; j < T; j++) Fr64::mac(t, S.row0[(size_t)k * T + j], st[j]);
      const Fr64 s0 = st[0];
      for (int i = 1; i < T; i++) st[i] = Fr64::add(st[i], Fr64::mul(S.what[(size_t)k * (T - 1) + (i - 1)], s0));
      st[0] = Fr64::redc(t);
    }
    Fr64 nx[T];
    for (int i = 0; i < T; i++) { uint64_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; for (int j = 0; j < T; j++) Fr64::mac(t, S.left[i][j], st[j]); nx[i] = Fr64::redc(t); }
    for (int i = 0; i < T; i++) st[i] = nx[i];
  }
  static const Tables &tables() {
    static const Tables t = [] {
      Tables x; const PoseidonSpec &S = PoseidonSpec::get();
      for (const auto &c : S.rc) x.rc.push_back(Fr64::from(c));
      for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) x.mds[i][j] = Fr64::from(S.mds[i][j]);
      constexpr int RP = PoseidonSpec::RP, R0 = PoseidonSpec::RF / 2, N = T - 1;
      const Fr64 zero{{0, 0, 0, 0}}, one = Fr64::from(halo2::detail::from_fe(zk::Fr::one()));
      Fr64 D[T][T]; for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) D[i][j] = x.mds[i][j];
      Fr64 hinv[N][N];                                           // D^^-1 of the PREVIOUS round's split: applied to this round's constants
      x.prc.assign((size_t)RP * T, zero); x.row0.assign((size_t)RP * T, zero); x.what.assign((size_t)RP * N, zero);
      for (int k = 0; k < RP; k++) {
        // this round's constants: c_0 as they are; later ones with D'^-1 of the previous split pushed through: [c_0, D^^-1 c_1..]
        for (int i = 0; i < T; i++) x.prc[(size_t)k * T + i] = x.rc[(size_t)(R0 + k) * T + i];
        if (k > 0) for (int i = 0; i < N; i++) { Fr64 acc = zero; for (int j = 0; j < N; j++) acc = Fr64::add(acc, Fr64::mul(hinv[i][j], x.rc[(size_t)(R0 + k) * T + 1 + j])); x.prc[(size_t)k * T + 1 + i] = acc; }
        // invert D^ = D[1:, 1:] (Gauss-Jordan over the field; N = 4)
        Fr64 A[N][2 * N];
        for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { A[i][j] = D[1 + i][1 + j]; A[i][N + j] = i == j ? one : zero; }
        for (int c = 0; c < N; c++) {
          int piv = c; while (piv < N && !(A[piv][c].l[0] | A[piv][c].l[1] | A[piv][c].l[2] | A[piv][c].l[3])) piv++;
          if (piv == N) throw std::runtime_error("poseidon: singular sub-matrix in the sparse schedule");
          if (piv != c) for (int j = 0; j < 2 * N; j++) std::swap(A[piv][j], A[c][j]);
          const Fr64 iv = inv(A[c][c]);
          for (int j = 0; j < 2 * N; j++) A[c][j] = Fr64::mul(A[c][j], iv);
          for (int i = 0; i < N; i++) if (i != c) { const Fr64 f = A[i][c]; for (int j = 0; j < 2 * N; j++) A[i][j] = sub(A[i][j], Fr64::mul(f, A[c][j])); }
        }
        for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) hinv[i][j] = A[i][N + j];
        for (int j = 0; j < T; j++) x.row0[(size_t)k * T + j] = D[0][j];
        for (int i = 0; i < N; i++) { Fr64 acc = zero; for (int j = 0; j < N; j++) acc = Fr64::add(acc, Fr64::mul(hinv[i][j], D[1 + j][0])); x.what[(size_t)k * N + i] = acc; }
        // D' = [[1, 0], [0, D^]]: the next round's dense matrix is M D'; after the last round D' itself is what is left
        Fr64 Dp[T][T]; for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) Dp[i][j] = (i == 0 || j == 0) ? (i == j ? one : zero) : D[i][j];
        if (k + 1 < RP) { for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) { Fr64 acc = zero; for (int q = 0; q < T; q++) acc = Fr64::add(acc, Fr64::mul(x.mds[i][q], Dp[q][j])); D[i][j] = acc; } }
        else for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) x.left[i][j] = Dp[i][j];
      }
      // self-check against the plain schedule on a fixed state
      Fr64 a[T], b[T]; for (int i = 0; i < T; i++) a[i] = b[i] = x.rc[(size_t)(7 * i + 3) % x.rc.size()];
      plain_rounds(x, a, R0, R0 + RP); sparse_rounds(x, b);
      for (int i = 0; i < T; i++) if (std::memcmp(a[i].l, b[i].l, 32) != 0) throw std::runtime_error("poseidon: the sparse partial-round schedule disagrees with the plain one");
      return x;
    }();
    return t;
  }
  void permute() {
    const Tables &S = tables();
    constexpr int R0 = PoseidonSpec::RF / 2, RP = PoseidonSpec::RP;
    plain_rounds(S, state, 0, R0);
    sparse_rounds(S, state);
    plain_rounds(S, state, R0 + RP, PoseidonSpec::RF + RP);
  }
  void absorb_and_permute(const Fr64 *chunk, size_t len) {
    for (size_t i = 0; i < len; i++) state[1 + i] = Fr64::add(state[1 + i], chunk[i]);
    if (len < (size_t)RATE) state[len + 1] = Fr64::add(state[len + 1], Fr64::from(halo2::detail::from_fe(zk::Fr::one())));
    permute();
  }
  halo2::Fr squeeze() {
    // what is left is the partial last chunk (padded), or nothing -- then the words since the last squeeze were a multiple of RATE (possibly zero) and the sponge takes the empty chunk
    std::vector<Fr64> b; b.swap(buf);
    absorb_and_permute(b.data(), b.size());
    return state[1].fr();
  }
};

// Keccak-256 (original padding 0x01 ... 0x80, rate 136): one-shot over a byte vector
inline std::array<uint8_t, 32> keccak256(const std::vector<uint8_t> &data) {
  static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                                  0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                                  0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
  auto rol = [](uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; };
  std::vector<uint8_t> p(data); p.push_back(0x01); while (p.size() % 136) p.push_back(0); p.back() |= 0x80;
  uint64_t A[5][5] = {};   // [x][y]
  for (size_t off = 0; off < p.size(); off += 136) {
    for (int i = 0; i < 17; i++) { uint64_t w; std::memcpy(&w, p.data() + off + 8 * i, 8); A[i % 5][i / 5] ^= w; }
    for (int rnd = 0; rnd < 24; rnd++) {
      uint64_t C[5], D[5], B[5][5];
      for (int x = 0; x < 5; x++) C[x] = A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4];
      for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
      for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) A[x][y] ^= D[x];
      for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y][(2 * x + 3 * y) % 5] = rol(A[x][y], ROT[x][y]);
      for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) A[x][y] = B[x][y] ^ (~B[(x + 1) % 5][y] & B[(x + 2) % 5][y]);
      A[0][0] ^= RC[rnd];
    }
  }
  std::array<uint8_t, 32> out; for (int i = 0; i < 4; i++) std::memcpy(out.data() + 8 * i, &A[i % 5][i / 5], 8);
  return out;
}

enum class TranscriptKind { Blake2b, Poseidon, Evm, ByLayer /* not a transcript: ProofOptions' default, resolved by plonk::reference_transcript */ };
inline TranscriptKind transcript_kind_from_name(const std::string &s) {
  if (s == "blake2b") return TranscriptKind::Blake2b;
  if (s == "poseidon") return TranscriptKind::Poseidon;
  if (s == "evm") return TranscriptKind::Evm;
  throw std::invalid_argument("transcript: blake2b, poseidon or evm expected, got " + s);
}

struct Transcript {
  TranscriptKind kind;
  Blake2b state{"Halo2-Transcript"};
  PoseidonSponge sponge;
  std::vector<uint8_t> evm_buf;                                 // Evm: the words waiting for the next Keccak
  std::vector<uint8_t> proof;                                   // what the transcript's writer receives: the proof, in the reference's layout
  explicit Transcript(TranscriptKind k = TranscriptKind::Blake2b) : kind(k) { if (k == TranscriptKind::ByLayer) throw std::invalid_argument("transcript: ByLayer must be resolved with reference_transcript(protocol) first"); }
  static void be32(const zk::fe_t &canonical, uint8_t out[32]) { const uint8_t *le = reinterpret_cast<const uint8_t *>(&canonical); for (int i = 0; i < 32; i++) out[i] = le[31 - i]; }
  halo2::Fr squeeze_challenge() {
    if (kind == TranscriptKind::Poseidon) return sponge.squeeze();
    if (kind == TranscriptKind::Evm) {
      if (evm_buf.size() == 32) evm_buf.push_back(1);
      const std::array<uint8_t, 32> h = keccak256(evm_buf);
      evm_buf.assign(h.begin(), h.end());
      zk::fe_t c; uint8_t *le = reinterpret_cast<uint8_t *>(&c); for (int i = 0; i < 32; i++) le[i] = h[31 - i];      // the hash as a big-endian 256-bit integer, mod r (2^256 < 6 r)
      uint32_t m[8]; for (int i = 0; i < 8; i++) m[i] = zk::FrP::mod(i);
      while (zk::Fr::w_geq(c.l, m)) zk::Fr::w_sub(c.l, m);
      return halo2::detail::from_fe(zk::Fr::from_canonical(c));
    }
    const uint8_t z = 0; state.update(&z, 1); return fr_from_uniform_bytes(state.digest());
  }
  void common_scalar(const halo2::Fr &s) {
    if (kind == TranscriptKind::Poseidon) { sponge.update(s); return; }
    const zk::fe_t c = zk::Fr::to_canonical(halo2::detail::to_fe(s));
    if (kind == TranscriptKind::Evm) { uint8_t w[32]; be32(c, w); evm_buf.insert(evm_buf.end(), w, w + 32); return; }
    const uint8_t tag = 2; state.update(&tag, 1); state.update(&c, 32);
  }
  void write_scalar(const halo2::Fr &s) {
    common_scalar(s);
    const zk::fe_t c = zk::Fr::to_canonical(halo2::detail::to_fe(s));
    if (kind == TranscriptKind::Evm) { uint8_t w[32]; be32(c, w); proof.insert(proof.end(), w, w + 32); return; }
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&c); proof.insert(proof.end(), p, p + 32);
  }
  // g: a commitment as the C-ABI returns it (normalised Jacobian: x, y, z = R; all-zero = identity)
  void write_point(const halo2::G1 &g) {
    halo2::G1Affine a; std::memcpy(a.data(), g.data(), 64);
    bool ident = true; for (auto w : g) ident = ident && w == 0;
    if (ident) throw std::invalid_argument("transcript: the identity has no coordinates (halo2's common_point fails on it)");
    zk::fe_t x, y; std::memcpy(&x, a.data(), 32); std::memcpy(&y, a.data() + 4, 32);
    const zk::fe_t xc = zk::Fq::to_canonical(x), yc = zk::Fq::to_canonical(y);
    if (kind == TranscriptKind::Evm) {
      uint8_t w[64]; be32(xc, w); be32(yc, w + 32);
      evm_buf.insert(evm_buf.end(), w, w + 64); proof.insert(proof.end(), w, w + 64);
      return;
    }
    if (kind == TranscriptKind::Poseidon) {
      auto base_to_scalar = [](zk::fe_t c) {   // a base-field coordinate as a scalar: its value mod r (q < 2 r: one subtraction at most)
        uint32_t m[8]; for (int i = 0; i < 8; i++) m[i] = zk::FrP::mod(i);
        while (zk::Fr::w_geq(c.l, m)) zk::Fr::w_sub(c.l, m);
        return halo2::detail::from_fe(zk::Fr::from_canonical(c));
      };
      sponge.update(base_to_scalar(xc)); sponge.update(base_to_scalar(yc));
    } else {
      const uint8_t tag = 1; state.update(&tag, 1); state.update(&xc, 32); state.update(&yc, 32);
    }
    const halo2::G1Bytes b = halo2::g1_to_bytes(a); proof.insert(proof.end(), b.begin(), b.end());
  }
};

// the verifying key's scalar in the transcript: halo2 hashes the Debug rendering of the pinned key with the personalisation "Halo2-Verify-Key"; that string does
// not exist outside Rust, so the bytes hashed here are the .vkey serialisation (u32 BE k | u32 BE fixed columns | compressed commitments,
// the layout of [REF release-v0.13.1/vk_chunk.vkey])
inline halo2::Fr vk_transcript_repr(const std::vector<uint8_t> &vk_bytes) {
  Blake2b hsh("Halo2-Verify-Key"); hsh.update(vk_bytes.data(), vk_bytes.size());
  return fr_from_uniform_bytes(hsh.digest());
}

}  // namespace plonk
}  // namespace mi355zk
