#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void mac(uint64_t &acc, uint32_t &ex, uint32_t a, uint32_t b) {
  uint64_t p = (uint64_t)a * b; uint64_t s; 
  ex += __builtin_uaddll_overflow(acc, p, (unsigned long long*)&s); acc = s;
}
extern "C" __global__ void k(uint32_t* io) {
  uint32_t t = threadIdx.x; uint64_t acc = io[t]; uint32_t ex = 0;
  uint32_t a[8], b[8];
  for (int i = 0; i < 8; i++) { a[i] = io[t + 64 * (i + 1)]; b[i] = io[t + 64 * (i + 9)]; }
#pragma unroll
  for (int i = 0; i < 8; i++) mac(acc, ex, a[i], b[7 - i]);
  io[t] = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ ex;
}
