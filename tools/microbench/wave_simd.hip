// wave_simd.hip -- which SIMD does wave w of a workgroup run on?  (group_logpost deals a
// workgroup's proposals to its wave slots in a snake over the SIMDs and counts on slot s sitting on
// SIMD s % 4.)  Prints, for a few workgroups of 1024 and of 512 threads, the SIMD_ID field of
// HW_REG_HW_ID of every wave.   hipcc --offload-arch=gfx950 -O2 -o wave_simd wave_simd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}

int main() {
  for (int threads : {1024, 512}) {
    const int waves = threads / 64, blocks = 600;
    unsigned* d = nullptr;
    hipMalloc(&d, sizeof(unsigned) * waves * blocks);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d);
    std::vector<unsigned> h(waves * blocks);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int agree = 0;
    for (int b = 0; b < blocks; ++b) {
      bool ok = true;
      const unsigned s0 = (h[b * waves] >> 4) & 3;
      for (int w = 0; w < waves; ++w) ok = ok && (((h[b * waves + w] >> 4) & 3) == ((s0 + w) & 3));
      agree += ok;
    }
    printf("%d threads: wave w on SIMD (s0 + w) %% 4 in %d of %d workgroups\n", threads, agree, blocks);
    for (int b = 0; b < 3; ++b) {
      printf("  workgroup %d (CU %u): SIMD of waves 0..%d:", b, (h[b * waves] >> 8) & 15, waves - 1);
      for (int w = 0; w < waves; ++w) printf(" %u", (h[b * waves + w] >> 4) & 3);
      printf("\n");
    }
    hipFree(d);
  }
  return 0;
}
