// pmc_calib.hip -- calibrates rocprofv3 FETCH_SIZE on gfx950 for the two access patterns of the MSM:
//  (a) k_stream: coalesced 16 B/lane streaming read of B bytes; (b) k_gather: random 64-byte record gathers
//  (two 32-byte halves, each as two dwordx4 loads -- exactly load_affine() of msm.hpp) with a known record count.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k_stream(const uint4 *in, uint4 *out, size_t n) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678) out[0] = acc;
}
__global__ void k_gather(const uint4 *recs, uint4 *out, uint32_t nrec_mask, size_t gathers) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < gathers; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint4 *p = recs + (size_t)(h & nrec_mask) * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc.x == 0x12345678) out[0] = acc;
}
int main() {
  size_t bytes = (size_t)4 << 30; uint4 *buf, *out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64); hipMemset(buf, 1, bytes);
  hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, buf, out, bytes / 16);
  hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, buf, out, (uint32_t)(bytes / 64 - 1), (size_t)1 << 26);
  hipDeviceSynchronize();
  printf("stream bytes %zu; gather records %zu x 64 B = %zu bytes\n", bytes, (size_t)1 << 26, ((size_t)1 << 26) * 64);
  return 0;
}
