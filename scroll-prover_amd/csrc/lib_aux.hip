// lib_aux.hip -- libmi355zk.so, the translation unit of the kernels either side of MSM / NTT: the DFT over G1 points
// (g1fft.hpp: g_to_lagrange, ParamsKZG::downsize), the multiplicative scans of the permutation / lookup arguments and kate_division
// (frscan.hpp), Curve::batch_normalize, and the one G2 scalar multiple of ParamsKZG::setup (g2.hpp).  Host logic only.
// kernel headers first: lib_common.hpp defines the macro `g` (the calling thread's device context), a name the kernels use for locals
#include "g1fft.hpp"
#include "frscan.hpp"
#include "g2.hpp"
#include "lib_common.hpp"
#include <thread>

namespace mi355 {

static_assert(sizeof(g2_affine_t) == 128, "G2Affine is 128 bytes (x.c0, x.c1, y.c0, y.c1)");

int aux_tu_init_device() {
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_prefix_product<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_prefix_product<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)(k_fr_prefix_product<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)(k_fr_prefix_product<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_linrec<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void *)k_fr_linrec<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return MI355_OK;
}

// DFT over G1 points (g1fft.hpp).  in: n x (96 B Jacobian | 64 B affine), out likewise (may alias in); scale: optional Fr (Montgomery).
int g1fft_impl(const void *in, int in_jac, void *out, int out_jac, uint32_t log_n, const void *omega, const void *scale_host) {
  const uint32_t n = 1u << log_n, half = std::max(1u, n / 2);
  g1_xyzz_t *work; fe_t *tw; fe_t scale = Fr::zero(); if (scale_host) memcpy(&scale, scale_host, 32);
  CHK(ws_get("g1fft.work", (size_t)n * sizeof(g1_xyzz_t), (void **)&work));
  CHK(ws_get("g1fft.tw", (size_t)half * sizeof(fe_t), (void **)&tw));
  hipStream_t s = g.stream;
  fe_t w; memcpy(&w, omega, 32);
  Scope total("g1_fft");
  CHK(launch_pow_table(tw, w, 1, half));   // kernel of ntt.hpp, launched by lib_ntt.hip on this context's stream
  if (in_jac) hipLaunchKernelGGL(k_g1fft_load<1>, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, work, log_n);
  else hipLaunchKernelGGL(k_g1fft_load<0>, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, work, log_n);
  for (uint32_t st = 0; st < log_n; st++) hipLaunchKernelGGL(k_g1fft_stage, dim3(ceil_div(n / 2, 256)), dim3(256), 0, s, work, tw, log_n, st);
  if (out_jac) hipLaunchKernelGGL(k_g1fft_store<1>, dim3(ceil_div(n, 256)), dim3(256), 0, s, work, out, log_n, scale, scale_host ? 1 : 0);
  else hipLaunchKernelGGL(k_g1fft_store<0>, dim3(ceil_div(n, 256)), dim3(256), 0, s, work, out, log_n, scale, scale_host ? 1 : 0);
  HIPCHK(hipGetLastError());
  return MI355_OK;
}

// data[i] <- data[i]^-1 (zeros kept): tile products -> (recursively) their inverses -> per-tile completion
int batch_invert_impl(fe_t *data, uint64_t n, int level) {
  const uint32_t tiles = (uint32_t)ceil_div(n, FRSCAN_TILE);
  hipStream_t s = g.stream;
  if (tiles <= 32) { hipLaunchKernelGGL(k_fr_batch_invert<0>, dim3(tiles), dim3(FRSCAN_THREADS), 0, s, data, n, (fe_t *)nullptr); return MI355_OK; }
  fe_t *tile_prod; const std::string role = "frscan.inv_tiles" + std::to_string(level);
  CHK(ws_get(role.c_str(), (size_t)tiles * sizeof(fe_t), (void **)&tile_prod));
  hipLaunchKernelGGL(k_fr_batch_invert<1>, dim3(tiles), dim3(FRSCAN_THREADS), 0, s, data, n, tile_prod);
  CHK(batch_invert_impl(tile_prod, tiles, level + 1));
  hipLaunchKernelGGL(k_fr_batch_invert<2>, dim3(tiles), dim3(FRSCAN_THREADS), 0, s, data, n, tile_prod);
  return MI355_OK;
}

// P_j = src_j + m * P_(j-1) over n elements (dst may alias src); reverse: index j lives at memory position n - 1 - j
int linrec_impl(const fe_t *src, fe_t *dst, uint64_t n, const fe_t &m, bool reverse, int level) {
  const uint32_t tiles = (uint32_t)ceil_div(n, FRSCAN_TILE);
  const size_t lds = (size_t)(FRSCAN_THREADS * 65) * 4;
  hipStream_t s = g.stream;
  if (tiles <= 1) { hipLaunchKernelGGL(k_fr_linrec<1>, dim3(1), dim3(FRSCAN_THREADS), lds, s, src, dst, n, m, reverse ? 1 : 0, (fe_t *)nullptr); return MI355_OK; }
  fe_t *tile_tot; const std::string role = "frscan.linrec" + std::to_string(level);
  CHK(ws_get(role.c_str(), (size_t)tiles * sizeof(fe_t), (void **)&tile_tot));
  hipLaunchKernelGGL(k_fr_linrec<0>, dim3(tiles), dim3(FRSCAN_THREADS), lds, s, src, dst, n, m, reverse ? 1 : 0, tile_tot);
  CHK(linrec_impl(tile_tot, tile_tot, tiles, Fr::pow_u64(m, FRSCAN_TILE), false, level + 1));   // the value of the recurrence at every tile end
  hipLaunchKernelGGL(k_fr_linrec<1>, dim3(tiles), dim3(FRSCAN_THREADS), lds, s, src, dst, n, m, reverse ? 1 : 0, tile_tot);
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

// dst[i] = product (ADD: sum) of src[j], j < i: tile totals -> scan of the totals by one workgroup -> per-tile completion
template <bool ADD> int prefix_scan_entry(void *dst_dev, const void *src_dev, uint64_t n, void *total_out_host, const char *what) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({dst_dev, src_dev}, &slot, what)); DevGuard lk(slot);
  CHK(need_init(slot));
  if (n && (!dst_dev || !src_dev)) return fail(MI355_EBADARG, std::string(what) + ": null pointer");
  if (n >= (1ull << 40)) return fail(MI355_EBADARG, std::string(what) + ": n too large");
  const uint32_t tiles = (uint32_t)ceil_div(n, FRSCAN_TILE);
  fe_t *tile_prod, *tile_prefix, *total;
  CHK(ws_get("frscan.tile_prod", ((size_t)tiles + 1) * sizeof(fe_t), (void **)&tile_prod));
  CHK(ws_get("frscan.tile_prefix", ((size_t)tiles + 1) * sizeof(fe_t), (void **)&tile_prefix));
  CHK(ws_get("frscan.total", sizeof(fe_t), (void **)&total));
  const size_t lds = (size_t)(FRSCAN_THREADS * 65) * 4;
  hipStream_t s = g.stream;
  if (tiles) hipLaunchKernelGGL((k_fr_prefix_product<0, ADD>), dim3(tiles), dim3(FRSCAN_THREADS), lds, s, (const fe_t *)src_dev, (fe_t *)dst_dev, n, tile_prod, (const fe_t *)tile_prefix);
  hipLaunchKernelGGL(k_fr_scan_tiles<ADD>, dim3(1), dim3(FRSCAN_THREADS), 0, s, (const fe_t *)tile_prod, tile_prefix, tiles, total);
  if (tiles) hipLaunchKernelGGL((k_fr_prefix_product<1, ADD>), dim3(tiles), dim3(FRSCAN_THREADS), lds, s, (const fe_t *)src_dev, (fe_t *)dst_dev, n, tile_prod, (const fe_t *)tile_prefix);
  HIPCHK(hipGetLastError());
  if (total_out_host) { HIPCHK(hipMemcpyAsync(total_out_host, total, sizeof(fe_t), hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); }
  return MI355_OK;
  });
}

extern "C" {

// Curve::batch_normalize: n Jacobian points (96 B, any representative) -> n affine points (64 B); k_g1_batch_normalize (frscan.hpp)
int mi355_g1_batch_normalize_dev(const void *jac_dev, void *affine_dev, uint64_t n) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({jac_dev, affine_dev}, &slot, "g1_batch_normalize")); DevGuard lk(slot);
  CHK(need_init(slot));
  if (n && (!jac_dev || !affine_dev)) return fail(MI355_EBADARG, "g1_batch_normalize: null pointer");
  if (n >= (1ull << 31)) return fail(MI355_EBADARG, "g1_batch_normalize: n must be < 2^31");
  if (n) {
    const char *a = (const char *)jac_dev, *b = (const char *)affine_dev;
    if (a < b + n * sizeof(g1_affine_t) && b < a + n * sizeof(g1_jac_t)) return fail(MI355_EBADARG, "g1_batch_normalize: input and output must not overlap");
    hipLaunchKernelGGL(k_g1_batch_normalize, dim3(ceil_div(n, FRSCAN_THREADS)), dim3(FRSCAN_THREADS), 0, g.stream, (const g1_jac_t *)jac_dev, (g1_affine_t *)affine_dev, n);
    HIPCHK(hipGetLastError());
  }
  return finish_async();
  });
}
int mi355_g1_batch_normalize_host(const void *jac_host, void *affine_host, uint64_t n) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);
  CHK(need_init(slot));
  if (n && (!jac_host || !affine_host)) return fail(MI355_EBADARG, "g1_batch_normalize: null pointer");
  if (n >= (1ull << 31)) return fail(MI355_EBADARG, "g1_batch_normalize: n must be < 2^31");
  if (!n) return MI355_OK;
  char *dev; CHK(ws_get("io.g1norm", n * (sizeof(g1_jac_t) + sizeof(g1_affine_t)), (void **)&dev));
  g1_jac_t *in = (g1_jac_t *)dev; g1_affine_t *out = (g1_affine_t *)(dev + n * sizeof(g1_jac_t));
  HIPCHK(hipMemcpyAsync(in, jac_host, n * sizeof(g1_jac_t), hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_g1_batch_normalize, dim3(ceil_div(n, FRSCAN_THREADS)), dim3(FRSCAN_THREADS), 0, g.stream, (const g1_jac_t *)in, out, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(affine_host, out, n * sizeof(g1_affine_t), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return MI355_OK;
  });
}
// ---- DFT over G1 points (best_fft::<Fr, G1>, g_to_lagrange)
int mi355_g1_fft_dev(void *points_jac_dev, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({points_jac_dev}, &slot, "g1_fft")); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(points_jac_dev, log_n, omega));
  CHK(g1fft_impl(points_jac_dev, 1, points_jac_dev, 1, log_n, omega, nullptr));
  return finish_async();
  });
}
int mi355_g1_fft_host(void *points_jac_host, uint32_t log_n, const void *omega) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(points_jac_host, log_n, omega));
  NttHostArgs a{log_n, omega, nullptr};
  const size_t bytes = sizeof(g1_jac_t) << log_n;
  return with_host_io(points_jac_host, bytes, bytes, bytes, "io.g1fft", [](void *dev, void *ud) { auto *a = (NttHostArgs *)ud; return g1fft_impl(dev, 1, dev, 1, a->log_n, a->omega, nullptr); }, &a);
  });
}
int mi355_g_to_lagrange_dev(const void *g_affine_dev, void *g_lagrange_affine_dev, uint32_t log_n, const void *omega_inv, const void *n_inv) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({g_affine_dev, g_lagrange_affine_dev}, &slot, "g_to_lagrange")); DevGuard lk(slot);
  CHK(need_init(slot)); CHK(check_ntt_args(g_affine_dev, log_n, omega_inv));
  if (!g_lagrange_affine_dev || !n_inv) return fail(MI355_EBADARG, "g_to_lagrange: null pointer");
  CHK(g1fft_impl(g_affine_dev, 0, g_lagrange_affine_dev, 0, log_n, omega_inv, n_inv));
  return finish_async();
  });
}
int mi355_srs_downsize(uint64_t g_handle, uint32_t k, const void *omega_inv, const void *n_inv, uint64_t *g_lagrange_handle_out) {
  return guarded([&]() -> int {
  AllGuard lk;
  CHK(need_init());
  Srs *sp; CHK(srs_find(g_handle, &sp, "srs_downsize"));
  if (!omega_inv || !n_inv || !g_lagrange_handle_out || k > 28 || (1ull << k) > sp->n) return fail(MI355_EBADARG, "srs_downsize: bad argument (2^k must not exceed the registered basis)");
  const uint64_t n = 1ull << k;
  for (int sl = 1; sl < g_ndev; sl++) { CHK(bind_ctx(sl)); HIPCHK(hipStreamSynchronize(g.stream)); }
  CHK(bind_ctx(0));
  const g1_affine_t *src; CHK(srs_gather_to_primary(*sp, n, &src));
  g1_affine_t *res; CHK(dev_malloc((void **)&res, n * sizeof(g1_affine_t), "srs_downsize"));
  int rc = g1fft_impl(src, 0, res, 0, k, omega_inv, n_inv);
  if (rc == MI355_OK) rc = finish_async();
  if (rc == MI355_OK && hipStreamSynchronize(g.stream) != hipSuccess) rc = fail(MI355_EHIP, "srs_downsize: stream synchronize failed");
  Srs s; s.n = n; s.mem = std::make_shared<SrsMem>();
  if (rc == MI355_OK) {
    if (plan_shards(n).size() == 1) { Shard one; one.slot = 0; one.lo = 0; one.n = n; one.dev = res; one.owned = true; s.mem->sh.push_back(one); res = nullptr; }
    else rc = srs_scatter_from_primary(*s.mem, res, n, false);
  }
  if (res) { (void)bind_ctx(0); (void)hipFree(res); }
  if (rc != MI355_OK) return rc;
  *g_lagrange_handle_out = srs_insert(s); return MI355_OK;
  });
}
int mi355_fr_kate_division_dev(void *dst_dev, const void *poly_dev, uint64_t n, const void *z) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({dst_dev, poly_dev}, &slot, "fr_kate_division")); DevGuard lk(slot);
  CHK(need_init(slot));
  if (!z || n == 0 || !poly_dev || (n > 1 && !dst_dev)) return fail(MI355_EBADARG, "fr_kate_division: null pointer or empty polynomial");
  if (n >= (1ull << 40)) return fail(MI355_EBADARG, "fr_kate_division: n too large");
  if (n == 1) return MI355_OK;   // a constant: the quotient is empty
  fe_t m; memcpy(&m, z, 32);
  CHK(linrec_impl((const fe_t *)poly_dev + 1, (fe_t *)dst_dev, n - 1, m, true, 0));
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_batch_invert_dev(void *data_dev, uint64_t n) {
  return guarded([&]() -> int {
  int slot; CHK(common_slot({data_dev}, &slot, "fr_batch_invert")); DevGuard lk(slot);
  CHK(need_init(slot));
  if (n && !data_dev) return fail(MI355_EBADARG, "fr_batch_invert: null pointer");
  if (n >= (1ull << 40)) return fail(MI355_EBADARG, "fr_batch_invert: n too large");
  if (n == 0) return MI355_OK;
  CHK(batch_invert_impl((fe_t *)data_dev, n, 0));
  HIPCHK(hipGetLastError());
  return MI355_OK;
  });
}
int mi355_fr_prefix_product_dev(void *dst_dev, const void *src_dev, uint64_t n, void *total_out_host) { return prefix_scan_entry<false>(dst_dev, src_dev, n, total_out_host, "fr_prefix_product"); }
int mi355_fr_prefix_sum_dev(void *dst_dev, const void *src_dev, uint64_t n, void *total_out_host) { return prefix_scan_entry<true>(dst_dev, src_dev, n, total_out_host, "fr_prefix_sum"); }
// ---- G2: s_g2 = tau * G2 of ParamsKZG::setup (the only G2 arithmetic on the path)
int mi355_g2_mul_host(const void *p_affine_host, const void *scalar, void *out_affine_host) {
  return guarded([&]() -> int {
  const int slot = pick_replica_slot(); DevGuard lk(slot);
  CHK(need_init(slot));
  if (!p_affine_host || !scalar || !out_affine_host) return fail(MI355_EBADARG, "g2_mul: null pointer");
  g2_affine_t *dev; CHK(ws_get("g2.io", 2 * sizeof(g2_affine_t) + 16, (void **)&dev));
  uint32_t *flag = reinterpret_cast<uint32_t *>(dev + 2);
  fe_t k; memcpy(&k, scalar, 32);
  HIPCHK(hipMemcpyAsync(dev, p_affine_host, sizeof(g2_affine_t), hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_g2_mul, dim3(1), dim3(64), 0, g.stream, (const g2_affine_t *)dev, k, dev + 1, flag);
  HIPCHK(hipGetLastError());
  uint32_t ok = 0; g2_affine_t res;
  HIPCHK(hipMemcpyAsync(&res, dev + 1, sizeof res, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipMemcpyAsync(&ok, flag, 4, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  if (!ok) return fail(MI355_EBADARG, "g2_mul: the point is not on the twist y^2 = x^3 + 3 / (9 + u)");
  memcpy(out_affine_host, &res, sizeof res);
  return MI355_OK;
  });
}

// ---- narrow uploads: a witness column as W-byte integers or as (index, value) pairs of its non-zero cells instead of n 32-byte words.  The narrow data crosses PCIe through
// mi355_buf_upload (copy stream, no device lock) into a pooled staging block; the expansion kernel is queued on the owner's compute stream WITHOUT the device lock (it touches
// no context state; HIP streams accept work from several threads) and the call returns without waiting for it: calls issued afterwards on that device are ordered behind it.
// staging blocks come in size classes (powers of two up to 1 MiB, then whole MiB): column after column of slightly different size -- the pairs of a sparse column -- then hits
// the exact-size pool instead of missing it and recycling the pool (which synchronises the copy stream and the free events on the uploader thread)
static uint64_t stage_class(uint64_t bytes) {
  if (bytes <= (1ull << 20)) { uint64_t c = 4096; while (c < bytes) c <<= 1; return c; }
  return (bytes + (1ull << 20) - 1) & ~((1ull << 20) - 1);
}
int mi355_buf_upload_packed(void *dst_dev, const void *src_host, uint64_t n, uint32_t width_bytes) {
  return guarded([&]() -> int {
  if (n == 0) return MI355_OK;
  if (!dst_dev || !src_host) return fail(MI355_EBADARG, "buf_upload_packed: null pointer");
  if (width_bytes != 1 && width_bytes != 2 && width_bytes != 4 && width_bytes != 8) return fail(MI355_EBADARG, "buf_upload_packed: width must be 1, 2, 4 or 8 bytes");
  if (n > (1ull << 32)) return fail(MI355_EBADARG, "buf_upload_packed: more than 2^32 cells");   // also keeps n * width_bytes and n * 32 far from overflow
  CHK(buf_check_range(dst_dev, n * sizeof(fe_t), "buf_upload_packed"));
  const int slot = slot_of(dst_dev);
  void *stage = nullptr; CHK(mi355_buf_alloc(stage_class(n * width_bytes), slot, &stage));
  int rc = mi355_buf_upload(stage, src_host, n * width_bytes);
  if (rc == MI355_OK) rc = need_init(slot);
  if (rc == MI355_OK) {
    hipStream_t s = g_ctx[slot].stream; const dim3 grid((uint32_t)std::min<uint64_t>(ceil_div(n, 256), 65535u * 4));
    switch (width_bytes) {
      case 1: hipLaunchKernelGGL(k_expand_packed<1>, grid, dim3(256), 0, s, (fe_t *)dst_dev, (const uint8_t *)stage, n); break;
      case 2: hipLaunchKernelGGL(k_expand_packed<2>, grid, dim3(256), 0, s, (fe_t *)dst_dev, (const uint8_t *)stage, n); break;
      case 4: hipLaunchKernelGGL(k_expand_packed<4>, grid, dim3(256), 0, s, (fe_t *)dst_dev, (const uint8_t *)stage, n); break;
      default: hipLaunchKernelGGL(k_expand_packed<8>, grid, dim3(256), 0, s, (fe_t *)dst_dev, (const uint8_t *)stage, n); break;
    }
    if (hipGetLastError() != hipSuccess) rc = fail(MI355_EHIP, "buf_upload_packed: kernel launch failed");
  }
  (void)mi355_buf_free(stage);   // back to the pool; its reuse waits for the kernel queued above
  return rc;
  });
}
int mi355_buf_upload_sparse(void *dst_dev, uint64_t n, const uint32_t *idx_host, const void *vals_host, uint64_t count) {
  return guarded([&]() -> int {
  if (n == 0) return MI355_OK;
  if (!dst_dev || (count && (!idx_host || !vals_host))) return fail(MI355_EBADARG, "buf_upload_sparse: null pointer");
  if (count > n || n > (1ull << 32)) return fail(MI355_EBADARG, "buf_upload_sparse: more pairs than cells, or more than 2^32 cells");
  CHK(buf_check_range(dst_dev, n * sizeof(fe_t), "buf_upload_sparse"));
  const int slot = slot_of(dst_dev);
  CHK(need_init(slot));
  hipStream_t s = g_ctx[slot].stream;
  if (count == 0) { HIPCHK(hipMemsetAsync(dst_dev, 0, n * sizeof(fe_t), s)); return MI355_OK; }
  const uint64_t idx_bytes = (count * 4 + 31) & ~31ull;
  void *stage = nullptr; CHK(mi355_buf_alloc(stage_class(idx_bytes + count * sizeof(fe_t)), slot, &stage));
  int rc = mi355_buf_upload(stage, idx_host, count * 4);
  if (rc == MI355_OK) rc = mi355_buf_upload((char *)stage + idx_bytes, vals_host, count * sizeof(fe_t));
  if (rc == MI355_OK) rc = need_init(slot);
  if (rc == MI355_OK) {
    if (hipMemsetAsync(dst_dev, 0, n * sizeof(fe_t), s) != hipSuccess) rc = fail(MI355_EHIP, "buf_upload_sparse: memset failed");
    else {
      hipLaunchKernelGGL(k_scatter_fr, dim3((uint32_t)std::min<uint64_t>(ceil_div(count, 256), 65535u * 4)), dim3(256), 0, s, (fe_t *)dst_dev, n, (const uint32_t *)stage, (const fe_t *)((char *)stage + idx_bytes), count);
      if (hipGetLastError() != hipSuccess) rc = fail(MI355_EHIP, "buf_upload_sparse: kernel launch failed");
    }
  }
  (void)mi355_buf_free(stage);
  return rc;
  });
}
// host helper for the sparse form: the non-zero cells of a column of n 32-byte words as (index, value) pairs, in index order; `threads` workers scan disjoint ranges
// (zero is zero in Montgomery form too, so the scan needs no arithmetic).  idx_out / vals_out must hold n entries in the worst case; pure host code, no device.
int mi355_host_compact_nonzero(const void *src_host, uint64_t n, uint32_t *idx_out, void *vals_out, uint64_t *count_out, int threads) {
  return guarded([&]() -> int {
  if (!count_out || (n && (!src_host || !idx_out || !vals_out))) return fail(MI355_EBADARG, "host_compact_nonzero: null pointer");
  if (n > (1ull << 32)) return fail(MI355_EBADARG, "host_compact_nonzero: more than 2^32 cells");
  const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, threads), std::max<uint64_t>(1, n >> 16)));
  const uint64_t *w = (const uint64_t *)src_host; uint64_t *vo = (uint64_t *)vals_out;
  std::vector<uint64_t> cnt(T + 1, 0);
  auto nz = [&](uint64_t i) { return (w[4 * i] | w[4 * i + 1] | w[4 * i + 2] | w[4 * i + 3]) != 0; };
  auto range = [&](int t, uint64_t &lo, uint64_t &hi) { lo = n * t / T; hi = n * (t + 1) / T; };
  auto count_job = [&](int t) { uint64_t lo, hi; range(t, lo, hi); uint64_t c = 0; for (uint64_t i = lo; i < hi; i++) c += nz(i); cnt[t + 1] = c; };
  { std::vector<std::thread> th; for (int t = 1; t < T; t++) th.emplace_back(count_job, t); count_job(0); for (auto &x : th) x.join(); }
  for (int t = 0; t < T; t++) cnt[t + 1] += cnt[t];
  auto write_job = [&](int t) { uint64_t lo, hi; range(t, lo, hi); uint64_t o = cnt[t]; for (uint64_t i = lo; i < hi; i++) if (nz(i)) { idx_out[o] = (uint32_t)i; memcpy(vo + 4 * o, w + 4 * i, 32); o++; } };
  { std::vector<std::thread> th; for (int t = 1; t < T; t++) th.emplace_back(write_job, t); write_job(0); for (auto &x : th) x.join(); }
  *count_out = cnt[T];
  return MI355_OK;
  });
}

}  // extern "C"
