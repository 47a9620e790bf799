// mi355zk_halo2.hpp -- C++ host-side mirror of the halo2_proofs operator surface on top of the C-ABI (include/mi355zk.h).
//
// The reference's host language is Rust; no Rust toolchain exists in this repository's container, so next to
// the Rust binding that a maintainer adds (rust_shim/mi355zk.rs) this header gives compiled-code callers the same names, argument
// meaning and error behaviour as the functions scroll-prover reaches through gen_halo2_chunk_proof / gen_batch_proof
// [REF integration/src/prove.rs:37,67,96] in halo2_proofs@e5ddf67 [EXT-recalled]:
//
//   best_multiexp(coeffs, bases) -> G1           src/arithmetic.rs     (length mismatch: the Rust code panics; here std::invalid_argument)
//   best_fft(a, omega, log_n)                    src/arithmetic.rs     in place, natural order in and out
//   eval_polynomial(poly, point)                 src/arithmetic.rs
//   EvaluationDomain::new(j, k) + lagrange_to_coeff / coeff_to_lagrange / coeff_to_extended / extended_to_coeff / get_omega ...
//   ParamsKZG { k, n, g, g_lagrange } + commit / commit_lagrange (bases registered once, resident in HBM)
//
// Header-only; link with -lmi355zk.  Types are the in-memory forms of halo2curves::bn256 (4 x u64 LE Montgomery limbs).
// Host arithmetic here is limited to the domain constants (omega, n^-1, ...), exactly what EvaluationDomain::new computes;
// it reuses the product's own limb code (scroll-prover_amd/csrc/fp.hpp compiles for the host).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "mi355zk.h"
#include "../scroll-prover_amd/csrc/fp.hpp"

namespace mi355zk {
namespace halo2 {

using Fr = std::array<uint64_t, 4>;
using G1Affine = std::array<uint64_t, 8>;   // {x, y}, identity (0, 0)
using G1 = std::array<uint64_t, 12>;        // Jacobian {x, y, z}; results come back normalised (z = R or the all-zero identity)

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error("mi355zk error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) { if (rc != MI355_OK) throw Error(rc, mi355_last_error()); }
inline void init(int device_id = 0) { check(mi355_init(device_id)); }

// ---- helpers on Fr (host, for constants only)
namespace detail {
inline zk::fe_t to_fe(const Fr &a) { zk::fe_t r; std::memcpy(&r, a.data(), 32); return r; }
inline Fr from_fe(const zk::fe_t &a) { Fr r; std::memcpy(r.data(), &a, 32); return r; }
inline Fr fr_from_u64(uint64_t v) { zk::fe_t c = zk::Fr::zero(); c.l[0] = (uint32_t)v; c.l[1] = (uint32_t)(v >> 32); return from_fe(zk::Fr::from_canonical(c)); }
inline Fr fr_mul(const Fr &a, const Fr &b) { return from_fe(zk::Fr::mul(to_fe(a), to_fe(b))); }
inline Fr fr_inv(const Fr &a) { return from_fe(zk::Fr::inv(to_fe(a))); }
inline Fr fr_pow(const Fr &a, uint64_t e) { return from_fe(zk::Fr::pow_u64(to_fe(a), e)); }
// halo2curves bn256::Fr::ROOT_OF_UNITY = 7^((r-1)/2^28) and ZETA (cube root of unity), canonical values [EXT-recalled src/bn256/fr.rs];
// both are re-derived / checked in tests/test_oracle_golden.py
inline Fr root_of_unity() {
  zk::fe_t c; const uint32_t w[8] = {0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du, 0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u};
  for (int i = 0; i < 8; i++) c.l[i] = w[i];
  return from_fe(zk::Fr::from_canonical(c));
}
inline Fr zeta() {
  zk::fe_t c; const uint32_t w[8] = {0x36636f23u, 0xb8ca0b2du, 0xec2bc5e9u, 0xcc37a73fu, 0x3fd84104u, 0x048b6e19u, 0xe131a029u, 0x30644e72u};
  for (int i = 0; i < 8; i++) c.l[i] = w[i];
  return from_fe(zk::Fr::from_canonical(c));
}
}  // namespace detail

constexpr uint32_t FR_S = 28;

// ------------------------------------------------------------------------------------------------ arithmetic.rs
inline G1 best_multiexp(const std::vector<Fr> &coeffs, const std::vector<G1Affine> &bases) {
  if (coeffs.size() != bases.size()) throw std::invalid_argument("best_multiexp: coeffs.len() != bases.len()");
  G1 out;
  check(mi355_msm_g1_adhoc_host(bases.data(), coeffs.data(), coeffs.size(), out.data()));
  return out;
}
inline void best_fft(std::vector<Fr> &a, const Fr &omega, uint32_t log_n) {
  if (a.size() != (size_t(1) << log_n)) throw std::invalid_argument("best_fft: a.len() != 1 << log_n");
  check(mi355_ntt_fr_host(a.data(), log_n, omega.data()));
}
// best_fft::<Fr, G1>: the same transform over curve points (Jacobian, in place); its caller in halo2 is g_to_lagrange
inline void best_fft(std::vector<G1> &a, const Fr &omega, uint32_t log_n) {
  if (a.size() != (size_t(1) << log_n)) throw std::invalid_argument("best_fft: a.len() != 1 << log_n");
  check(mi355_g1_fft_host(a.data(), log_n, omega.data()));
}
// group::Curve::batch_normalize(p, q): Jacobian -> affine with one shared inversion per 256 points; panics on unequal lengths like the original
inline void batch_normalize(const std::vector<G1> &p, std::vector<G1Affine> &q) {
  if (p.size() != q.size()) throw std::invalid_argument("batch_normalize: p.len() != q.len()");
  check(mi355_g1_batch_normalize_host(p.data(), q.data(), p.size()));
}
inline Fr eval_polynomial(const std::vector<Fr> &poly, const Fr &point) {
  Fr out;
  check(mi355_eval_polynomial_host(poly.data(), poly.size(), point.data(), out.data()));
  return out;
}

// ------------------------------------------------------------------------------------------------ halo2curves G1Affine::to_bytes / from_bytes
// The 32-byte compressed form written to transcripts, proofs and .vkey files [EXT-recalled halo2curves derive/curve.rs; pinned by the
// fixture KAT A4, SURVEY 8a-0]: little-endian canonical x, bit 254 (0x40 of byte 31) = parity of canonical y, identity = all zero.
// Host arithmetic (a few field operations per point): proofs carry a dozen points, nothing here is a hot path.
using G1Bytes = std::array<uint8_t, 32>;
inline G1Bytes g1_to_bytes(const G1Affine &p) {
  G1Bytes out{};
  bool ident = true; for (auto w : p) ident = ident && w == 0;
  if (ident) return out;
  zk::fe_t x, y; std::memcpy(&x, p.data(), 32); std::memcpy(&y, p.data() + 4, 32);
  const zk::fe_t xc = zk::Fq::to_canonical(x), yc = zk::Fq::to_canonical(y);
  std::memcpy(out.data(), &xc, 32);
  out[31] |= (uint8_t)((yc.l[0] & 1u) << 6);
  return out;
}
// returns false for encodings that are not a curve point (x >= p, or x^3 + 3 a non-residue), as from_bytes' CtOption does
inline bool g1_from_bytes(const G1Bytes &b, G1Affine &out) {
  G1Bytes t = b; const unsigned sign = (t[31] >> 6) & 1u; t[31] &= 0x3f;
  zk::fe_t xc; std::memcpy(&xc, t.data(), 32);
  if (zk::Fq::is_zero(xc)) { out.fill(0); return sign == 0; }
  for (int i = 7; i >= 0; i--) { const uint32_t m = zk::FqP::mod(i); if (xc.l[i] != m) { if (xc.l[i] > m) return false; break; } if (i == 0) return false; }
  const zk::fe_t x = zk::Fq::from_canonical(xc);
  zk::fe_t three = zk::Fq::zero(); three.l[0] = 3; three = zk::Fq::from_canonical(three);
  const zk::fe_t y2 = zk::Fq::add(zk::Fq::mul(zk::Fq::sqr(x), x), three);
  uint32_t e[8]; { uint64_t c = 1; for (int i = 0; i < 8; i++) { c += zk::FqP::mod(i); e[i] = (uint32_t)c; c >>= 32; } }   // p + 1
  for (int i = 0; i < 8; i++) e[i] = (e[i] >> 2) | (i < 7 ? e[i + 1] << 30 : 0);                                          // (p + 1) / 4: p = 3 mod 4
  zk::fe_t y = zk::Fq::pow(y2, e);
  if (!zk::Fq::eq(zk::Fq::sqr(y), y2)) return false;
  if ((zk::Fq::to_canonical(y).l[0] & 1u) != sign) y = zk::Fq::neg(y);
  std::memcpy(out.data(), &x, 32); std::memcpy(out.data() + 4, &y, 32);
  return true;
}

// ------------------------------------------------------------------------------------------------ resident polynomials
// One vector of Fr resident in HBM: a mi355_buf_alloc block (the C++ twin of the Rust shim's DevicePoly).  Move-only; the destructor hands the
// block back to the library's pool (no hipFree, no device synchronisation; work already queued on it stays valid).  `slot` picks the device of an
// mi355_init_multi process.  create_proof uploads each witness column ONCE (from_host: the DMA overlaps whatever the device computes for other
// threads) and then works on it through the `*_dev` entry points; host memory sees the 96-byte commitments and 32-byte evaluations only.
struct DevicePoly {
  void *p = nullptr; uint64_t n = 0; int slot = 0;
  DevicePoly() = default;
  DevicePoly(uint64_t n_, int slot_ = 0) : n(n_), slot(slot_) { check(mi355_buf_alloc(n_ * 32, slot_, &p)); }
  DevicePoly(const DevicePoly &) = delete;
  DevicePoly &operator=(const DevicePoly &) = delete;
  DevicePoly(DevicePoly &&o) noexcept : p(o.p), n(o.n), slot(o.slot) { o.p = nullptr; }
  DevicePoly &operator=(DevicePoly &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; slot = o.slot; o.p = nullptr; } return *this; }
  ~DevicePoly() { release(); }
  void release() noexcept { if (p) { (void)mi355_buf_free(p); p = nullptr; } }
  static DevicePoly from_host(const std::vector<Fr> &v, int slot = 0) { DevicePoly d(v.size(), slot); check(mi355_buf_upload(d.p, v.data(), v.size() * 32)); return d; }
  std::vector<Fr> to_host() const { std::vector<Fr> v(n); check(mi355_buf_download(v.data(), p, n * 32)); return v; }
  // element offset into the block: a device pointer like any other
  void *at(uint64_t i) const { return static_cast<char *>(p) + i * 32; }
  Fr eval(const Fr &point) const { Fr out; check(mi355_eval_polynomial_dev(p, n, point.data(), out.data())); return out; }
};


// A host column whose storage is either ordinary memory or page-locked memory of the library (mi355_host_alloc): the choice is the caller's, per column
// vector, at run time.  A first copy out of pageable memory crosses PCIe at ~34 GB/s, out of page-locked memory at ~56 GB/s; for the many-column
// layers that is the difference between create_proof's steps 2-3 being bound by the link or by their commitments.
template <class T> struct ColumnAllocator {
  using value_type = T; bool pinned = false;
  using propagate_on_container_move_assignment = std::true_type;   // `column = Column(alloc)` must carry the page-locked choice with it
  using propagate_on_container_copy_assignment = std::true_type;
  using propagate_on_container_swap = std::true_type;
  ColumnAllocator() = default; explicit ColumnAllocator(bool p) : pinned(p) {}
  template <class U> ColumnAllocator(const ColumnAllocator<U> &o) : pinned(o.pinned) {}
  T *allocate(size_t n) {
    if (!pinned) return static_cast<T *>(::operator new(n * sizeof(T)));
    void *p = nullptr; check(mi355_host_alloc(n * sizeof(T), &p)); return static_cast<T *>(p);
  }
  // a block that outlives mi355_shutdown cannot be returned to the library any more: the error is dropped, the pages go back with the process
  void deallocate(T *p, size_t) noexcept { if (pinned) (void)mi355_host_free(p); else ::operator delete(p); }
  template <class U> bool operator==(const ColumnAllocator<U> &o) const { return pinned == o.pinned; }
  template <class U> bool operator!=(const ColumnAllocator<U> &o) const { return pinned != o.pinned; }
};
using Column = std::vector<Fr, ColumnAllocator<Fr>>;

// ------------------------------------------------------------------------------------------------ poly/domain.rs
class EvaluationDomain {
 public:
  uint32_t k, extended_k, quotient_poly_degree;
  uint64_t n;
  Fr omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;

  // EvaluationDomain::new(j, k): n = 2^k, quotient_poly_degree = j - 1, extended_k minimal with 2^extended_k >= n * (j - 1)
  EvaluationDomain(uint32_t j, uint32_t k_) : k(k_), quotient_poly_degree(j - 1), n(uint64_t(1) << k_) {
    extended_k = k;
    while ((uint64_t(1) << extended_k) < n * quotient_poly_degree) extended_k++;
    if (extended_k > FR_S) throw std::invalid_argument("EvaluationDomain: extended_k exceeds the two-adicity of Fr");
    extended_omega = detail::root_of_unity();
    for (uint32_t i = extended_k; i < FR_S; i++) extended_omega = detail::fr_mul(extended_omega, extended_omega);
    omega = extended_omega;
    for (uint32_t i = k; i < extended_k; i++) omega = detail::fr_mul(omega, omega);
    omega_inv = detail::fr_inv(omega); extended_omega_inv = detail::fr_inv(extended_omega);
    g_coset = detail::zeta(); g_coset_inv = detail::fr_mul(g_coset, g_coset);
    ifft_divisor = detail::fr_inv(detail::fr_from_u64(n));
    extended_ifft_divisor = detail::fr_inv(detail::fr_from_u64(uint64_t(1) << extended_k));
  }
  uint64_t extended_len() const { return uint64_t(1) << extended_k; }
  const Fr &get_omega() const { return omega; }
  const Fr &get_extended_omega() const { return extended_omega; }

  void coeff_to_lagrange(std::vector<Fr> &a) const { best_fft(a, omega, k); }
  // ifft: best_fft(a, omega_inv, k) then a[i] *= n^-1
  void lagrange_to_coeff(std::vector<Fr> &a) const {
    if (a.size() != n) throw std::invalid_argument("lagrange_to_coeff: wrong length");
    check(mi355_intt_fr_host(a.data(), k, omega_inv.data(), ifft_divisor.data()));
  }
  void lagrange_to_coeff(DevicePoly &a) const {
    if (a.n != n) throw std::invalid_argument("lagrange_to_coeff: wrong length");
    check(mi355_intt_fr_dev(a.p, k, omega_inv.data(), ifft_divisor.data()));
  }
  std::vector<Fr> coeff_to_extended(const std::vector<Fr> &a) const {
    if (a.size() != n) throw std::invalid_argument("coeff_to_extended: wrong length");
    std::vector<Fr> out(extended_len());
    check(mi355_coeff_to_extended_host(out.data(), a.data(), k, extended_k, g_coset.data(), g_coset_inv.data(), extended_omega.data()));
    return out;
  }
  // inverse transform, truncated to n * quotient_poly_degree coefficients as halo2 does
  std::vector<Fr> extended_to_coeff(std::vector<Fr> a) const {
    if (a.size() != extended_len()) throw std::invalid_argument("extended_to_coeff: wrong length");
    check(mi355_extended_to_coeff_host(a.data(), extended_k, g_coset.data(), g_coset_inv.data(), extended_omega_inv.data(), extended_ifft_divisor.data()));
    a.resize(n * quotient_poly_degree);
    return a;
  }
};

// ------------------------------------------------------------------------------------------------ poly/kzg/commitment.rs
class ParamsKZG {
 public:
  uint32_t k;
  uint64_t n;
  // what Prover::load_params hands down: both bases are registered once and stay resident (params_map outlives every prover)
  ParamsKZG(uint32_t k_, const std::vector<G1Affine> &g, const std::vector<G1Affine> &g_lagrange, bool window_tables = false) : k(k_), n(uint64_t(1) << k_) {
    if (g.size() != n || g_lagrange.size() != n) throw std::invalid_argument("ParamsKZG: bases must have 2^k points");
    check(mi355_srs_register_host(g.data(), n, &g_));
    check(mi355_srs_register_host(g_lagrange.data(), n, &gl_));
    if (window_tables) { check(mi355_srs_precompute(g_, 0, 0)); check(mi355_srs_precompute(gl_, 0, 0)); }
  }
  // Prover::load_params for one degree: the RawBytes file is streamed into HBM by the library (exact-length rule; validate = check every
  // point on the device); g2 / s_g2 are kept as the raw 128-byte encodings
  static std::unique_ptr<ParamsKZG> read(const std::string &path, bool validate = false) {
    uint32_t kk = 0; uint64_t hg = 0, hl = 0; std::array<uint8_t, 128> g2{}, sg2{};
    check(mi355_srs_load_params_file(path.c_str(), validate ? 1u : 0u, &kk, &hg, &hl, g2.data(), sg2.data()));
    std::unique_ptr<ParamsKZG> p(new ParamsKZG(kk, hg, hl));
    p->g2 = g2; p->s_g2 = sg2;
    return p;
  }
  std::array<uint8_t, 128> g2{}, s_g2{};
  ParamsKZG(const ParamsKZG &) = delete;
  ParamsKZG &operator=(const ParamsKZG &) = delete;
  ~ParamsKZG() { if (g_) mi355_srs_release(g_); if (gl_) mi355_srs_release(gl_); }

  // commit(poly: Coeff) = best_multiexp(poly, &g[..poly.len()]);  commit_lagrange(poly) = best_multiexp(poly, &g_lagrange[..n])
  G1 commit(const std::vector<Fr> &poly) const {
    if (poly.size() > n) throw std::invalid_argument("commit: polynomial longer than the basis");
    G1 out; check(mi355_msm_g1_host(g_, 0, poly.data(), poly.size(), out.data())); return out;
  }
  G1 commit_lagrange(const std::vector<Fr> &poly) const {
    if (poly.size() != n) throw std::invalid_argument("commit_lagrange: polynomial must have exactly n evaluations");
    G1 out; check(mi355_msm_g1_host(gl_, 0, poly.data(), poly.size(), out.data())); return out;
  }
  // commit / commit_lagrange of a resident polynomial: the scalars never leave HBM
  G1 commit(const DevicePoly &poly) const {
    if (poly.n > n) throw std::invalid_argument("commit: polynomial longer than the basis");
    G1 out; check(mi355_msm_g1_dev(g_, 0, poly.p, poly.n, out.data())); return out;
  }
  G1 commit_lagrange(const DevicePoly &poly) const {
    if (poly.n != n) throw std::invalid_argument("commit_lagrange: polynomial must have exactly n evaluations");
    G1 out; check(mi355_msm_g1_dev(gl_, 0, poly.p, poly.n, out.data())); return out;
  }
  // ParamsKZG::downsize(k): g.truncate(2^k); g_lagrange = g_to_lagrange(g, k) -- an inverse DFT over G1 points, run on the device
  void downsize(uint32_t new_k) {
    if (new_k > k) throw std::invalid_argument("downsize: k must not exceed the current degree");
    if (new_k == k) return;
    Fr w = detail::root_of_unity();
    for (uint32_t i = new_k; i < FR_S; i++) w = detail::fr_mul(w, w);
    const Fr w_inv = detail::fr_inv(w), n_inv = detail::fr_inv(detail::fr_from_u64(uint64_t(1) << new_k));
    uint64_t h = 0;
    check(mi355_srs_downsize(g_, new_k, w_inv.data(), n_inv.data(), &h));
    mi355_srs_release(gl_); gl_ = h; k = new_k; n = uint64_t(1) << new_k;
  }
  // the bases as ParamsKZG::write serialises them (first n points)
  std::vector<G1Affine> get_g() const { std::vector<G1Affine> v(n); check(mi355_srs_read_host(g_, 0, n, v.data())); return v; }
  std::vector<G1Affine> get_g_lagrange() const { std::vector<G1Affine> v(n); check(mi355_srs_read_host(gl_, 0, n, v.data())); return v; }

  // the per-column loop of create_proof (advice / lookup / permutation commitments of one phase) as one call: equal-length polynomials,
  // results in input order, identical to calling commit / commit_lagrange on each
  std::vector<G1> commit_many(const std::vector<const std::vector<Fr> *> &polys, bool lagrange = false) const {
    std::vector<G1> out(polys.size());
    if (polys.empty()) return out;
    const size_t len = polys[0]->size();
    std::vector<const void *> p(polys.size());
    for (size_t m = 0; m < polys.size(); m++) {
      if (polys[m]->size() != len) throw std::invalid_argument("commit_many: polynomials must have equal length");
      p[m] = polys[m]->data();
    }
    if (len > n || (lagrange && len != n)) throw std::invalid_argument("commit_many: polynomial length does not fit the basis");
    check(mi355_msm_g1_batch_host(lagrange ? gl_ : g_, 0, p.data(), (uint32_t)polys.size(), len, out.data()));
    return out;
  }

 private:
  ParamsKZG(uint32_t k_, uint64_t hg, uint64_t hl) : k(k_), n(uint64_t(1) << k_), g_(hg), gl_(hl) {}
  uint64_t g_ = 0, gl_ = 0;
};

}  // namespace halo2
}  // namespace mi355zk

// plonk::create_proof for a PlonkProtocol over resident polynomials: include "mi355zk_plonk.hpp"
