// simd_map.hip — which SIMD does wave w of a 512-thread (and a 384-thread) workgroup land on?  (round 6: the 48-channel
// convolution kernel balances unequal wave jobs per SIMD and needs the wave -> SIMD rule, not an assumption)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/simd_map.hip -o /tmp/simd_map && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_ids(unsigned* out) {
  extern __shared__ unsigned char smem[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
  if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 4);
  for (int nt : {512, 384, 256}) {
    for (int lds : {150 * 1024, 70 * 1024}) {
      hipFuncSetAttribute((const void*)k_ids, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      const int nb = 16;
      hipLaunchKernelGGL(k_ids, dim3(nb), dim3(nt), lds, 0, d);
      hipDeviceSynchronize();
      unsigned h[4096]; hipMemcpy(h, d, nb * (nt / 64) * 4, hipMemcpyDeviceToHost);
      printf("threads %d lds %d KiB: wave -> (simd, cu, se) for the first 3 workgroups\n", nt, lds / 1024);
      for (int b = 0; b < 3; ++b) {
        for (int w = 0; w < nt / 64; ++w) {
          const unsigned v = h[b * (nt / 64) + w];
          printf("  w%d:s%u cu%u se%u wv%u |", w, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7, v & 15);
        }
        printf("\n");
      }
    }
  }
  return 0;
}
