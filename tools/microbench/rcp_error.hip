// How good is v_rcp_f64 / v_rsq_f64 as a seed?  Max relative error over 2^24 arguments in [1, 4)
// (and after one quadratic, one cubic Newton step).  hipcc --offload-arch=gfx950 -O2 rcp_error.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = 1.0 + 3.0 * ((double)i + 0.37) / (double)n;
  double y0 = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y0, 1.0);
  double y1 = __builtin_fma(y0, e, y0);                       // quadratic
  double y3 = __builtin_fma(y0, __builtin_fma(e, e, e), y0);  // cubic
  out[4 * i + 0] = d; out[4 * i + 1] = y0; out[4 * i + 2] = y1; out[4 * i + 3] = y3;
}
int main() {
  const int n = 1 << 24;
  double* d; hipMalloc(&d, sizeof(double) * 4 * n);
  k<<<n / 256, 256>>>(d, n);
  double* h = new double[4 * (size_t)n];
  hipMemcpy(h, d, sizeof(double) * 4 * n, hipMemcpyDeviceToHost);
  long double m0 = 0, m1 = 0, m3 = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / (long double)h[4 * i];
    long double e0 = fabsl((h[4 * i + 1] - t) / t), e1 = fabsl((h[4 * i + 2] - t) / t), e3 = fabsl((h[4 * i + 3] - t) / t);
    if (e0 > m0) m0 = e0; if (e1 > m1) m1 = e1; if (e3 > m3) m3 = e3;
  }
  printf("v_rcp_f64 max rel err 2^%.2f; after a quadratic step 2^%.2f; after a cubic step 2^%.2f\n",
         (double)log2l(m0), (double)log2l(m1), (double)log2l(m3));
  return 0;
}
