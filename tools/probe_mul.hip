#include "../scroll-prover_amd/csrc/g1.cuh"
using namespace zk;
extern "C" __global__ void k_mulchain(fe_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  fe_t a = io[t], b = io[t + 1];
  for (int i = 0; i < iters; i++) { a = Fq::mul(a, b); }
  io[t] = a;
}
