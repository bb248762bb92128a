// LDS atomic throughput on gfx950: ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_write_b32 / ds_read_b32, 64 distinct
// addresses per wave instruction (stride pattern like the window-attention histogram).  Prints cycles per wave
// instruction per CU with 8 waves resident.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic.hip -o gpurun_out/lds_atomic && gpurun -- gpurun_out/lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) float lf;
typedef __attribute__((address_space(3))) unsigned lu;
typedef __attribute__((address_space(3))) unsigned long long lull;
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters, int stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 16384; i += 512) ((float*)smem)[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192;
  unsigned a[16];
  for (int r = 0; r < 16; ++r) a[r] = base + (((lane * stride + r * 37) & 1023) * (MODE == 2 ? 8 : 4));
  float v = 1.0f + lane;
  float acc = 0.f;
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MODE == 0) __hip_atomic_fetch_add((lf*)(uintptr_t)a[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) __hip_atomic_fetch_add((lu*)(uintptr_t)a[r], (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 2) __hip_atomic_fetch_add((lull*)(uintptr_t)a[r], (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 3) *(volatile lf*)(uintptr_t)a[r] = v;
      if (MODE == 4) acc += *(volatile lf*)(uintptr_t)a[r];
      if (MODE == 5) { float x = *(volatile lf*)(uintptr_t)a[r]; *(volatile lf*)(uintptr_t)a[r] = x + v; }
    }
  }
  long long t1 = wall_clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = acc + ((float*)smem)[threadIdx.x];
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
  const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_write_b32", "ds_read_b32", "read+add+write"};
  int iters = 2000;
  for (int stride = 1; stride <= 33; stride += 16)
  for (int mode = 0; mode < 6; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, out, cyc, iters, stride); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, out, cyc, iters, stride); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 65536, 0, out, cyc, iters, stride); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 65536, 0, out, cyc, iters, stride); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 65536, 0, out, cyc, iters, stride); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 65536, 0, out, cyc, iters, stride); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 8 waves x iters x 16 wave-instructions per CU
    double per = ms * 1e-3 * 2.4e9 / (8.0 * iters * 16);
    printf("stride %2d  %-16s %8.3f ms  -> %.1f clk (2.4 GHz) per wave instruction per CU\n", stride, names[mode], ms, per);
  }
  return 0;
}
