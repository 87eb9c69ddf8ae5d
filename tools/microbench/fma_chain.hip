// fp64 fma issue/latency probe for gfx950: ILP independent chains per wave, W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fma_chain.hip -o fma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int ILP>
__global__ __launch_bounds__(64) void k(double* out, double a, double b, int iters) {
  double v[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) v[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) v[i] = __builtin_fma(v[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int ILP>
void run(int waves_per_simd, double* d) {
  const int iters = 20000;
  const int blocks = 256 * 4 * waves_per_simd;  // one 64-thread block = one wave
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, d, 0.999999, 1e-7, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, d, 0.999999, 1e-7, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fmas = (double)blocks * iters * 16 * ILP;         // wave-instructions
  const double per_simd_per_us = fmas / 1024.0 / (ms * 1e3);      // wave-fma per SIMD per us
  printf("ILP %d waves/SIMD %d: %.3f ms, %.1f wave-fma/SIMD/us (= cycles/instr %.2f at 2.4 GHz), %.1f TFLOP/s\n",
         ILP, waves_per_simd, ms, per_simd_per_us, 2400.0 / per_simd_per_us, fmas * 128 / (ms * 1e-3) / 1e12);
}
int main() {
  double* d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(double));
  for (int w : {1, 2, 4, 8}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
  return 0;
}
