// mi355zk_transcript.hpp -- halo2's Blake2b transcript on the host side of create_proof.
//
// create_proof owns the transcript: commitments and evaluations go in, challenges come out [EXT-recalled halo2_proofs src/transcript/blake2b.rs:
// Blake2bWrite<_, _, Challenge255<_>>]: Blake2b-512 with the personalisation "Halo2-Transcript"; a point is absorbed as 0x01 | x | y (canonical 32-byte
// little-endian coordinates; the identity is refused) and WRITTEN to the proof in its 32-byte compressed form; a scalar as 0x02 | canonical bytes, written
// the same way; a challenge = the 64-byte digest of (state | 0x00), read as a 512-bit little-endian integer and reduced mod r (Fr::from_uniform_bytes).
// The reference's inner layers hash with Poseidon and layer 6 with Keccak [EXT-recalled snark-verifier-sdk]: their parameters are not in the checkout,
// so this stock halo2 transcript stands in; the proof's BYTE LAYOUT is the reference's either way.
// The GPU library sees none of this: 96-byte commitments and 32-byte evaluations arrive from the C-ABI and are hashed here, on the calling thread.
// Cross-checked against Python's hashlib.blake2b in tests/test_plonk_host.py (same personalisation, same byte stream).
#pragma once
#include <cstring>

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

struct Transcript {
  Blake2b state{"Halo2-Transcript"};
  std::vector<uint8_t> proof;                                   // what Blake2bWrite's writer receives: the proof, in the reference's layout
  halo2::Fr squeeze_challenge() { const uint8_t z = 0; state.update(&z, 1); return fr_from_uniform_bytes(state.digest()); }
  void common_scalar(const halo2::Fr &s) {
    const uint8_t tag = 2; state.update(&tag, 1);
    const zk::fe_t c = zk::Fr::to_canonical(halo2::detail::to_fe(s)); state.update(&c, 32);
  }
  void write_scalar(const halo2::Fr &s) {
    common_scalar(s);
    const zk::fe_t c = zk::Fr::to_canonical(halo2::detail::to_fe(s)); const uint8_t *p = reinterpret_cast<const uint8_t *>(&c); proof.insert(proof.end(), p, p + 32);
  }
  // g: a commitment as the C-ABI returns it (normalised Jacobian: x, y, z = R; all-zero = identity)
  void write_point(const halo2::G1 &g) {
    halo2::G1Affine a; std::memcpy(a.data(), g.data(), 64);
    bool ident = true; for (auto w : g) ident = ident && w == 0;
    if (ident) throw std::invalid_argument("transcript: the identity has no coordinates (halo2's common_point fails on it)");
    zk::fe_t x, y; std::memcpy(&x, a.data(), 32); std::memcpy(&y, a.data() + 4, 32);
    const zk::fe_t xc = zk::Fq::to_canonical(x), yc = zk::Fq::to_canonical(y);
    const uint8_t tag = 1; state.update(&tag, 1); state.update(&xc, 32); state.update(&yc, 32);
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
