// LDS-DMA on gfx950: (A) semantics of `buffer_load_dwordx4 ... offen lds` (out-of-range lanes, num_records = 0, soffset and
// the range check, LDS addresses above 64 KiB); (B) achieved rate of 1 KiB pieces in 64-byte rows from an L2-resident and
// from a streaming source, by waves issuing and pieces in flight; (C) the k_conv3_r32 loop skeleton (27 weight fragments in
// registers, 8 n-tiles, ring of fragment reads, one barrier per tile) with its parts switched on one at a time.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma.hip -o tools/ubench/_bin/lds_dma ; gpurun -- tools/ubench/_bin/lds_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void buf_dma16(unsigned voff, i32x4 rs, unsigned soff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void glb_dma16(const unsigned char* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(g) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned nrec) {
  const unsigned long long b = (unsigned long long)base;
  i32x4 r = {(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)nrec, 0x00020000};
  r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
  r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
  return r;
}

// ---------------------------------------------------------------------------------------------------------------- (A)
__global__ void k_sem(const unsigned* g, unsigned* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  for (int i = threadIdx.x; i < 40960; i += 64) ((unsigned*)smem)[i] = 0xABABABABu;
  __syncthreads();
  const unsigned lane = threadIdx.x;
  unsigned voff = lane * 16, soff = 0, nrec = 0x80000000u, lds = lds0;
  const unsigned* base = g;
  if (mode == 0) { if (lane & 1) voff = 0xFFFFFF00u; }                 // odd lanes out of range
  if (mode == 1) nrec = 0;                                             // everything out of range
  if (mode == 2) { nrec = 1024; soff = 4096; }                         // soffset beyond num_records: range-checked or not?
  if (mode == 3) lds = lds0 + 100 * 1024;                              // LDS address above 64 KiB
  if (mode == 4) { base = g + 64; soff = 0; voff = lane * 16; nrec = 512; }   // lanes 32.. beyond num_records = 512 B
  if (mode == 5) { base = g - 64; soff = 256 + 1024; }                 // base below the allocation, soffset brings it back
  buf_dma16(voff, make_rsrc(base, nrec), __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(lds));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned* src = (const unsigned*)(smem + (mode == 3 ? 100 * 1024 : 0));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = src[lane * 4 + j];
}

// ---------------------------------------------------------------------------------------------------------------- (B)
// every issuing wave: `iters` pieces; piece k of wave w reads 16 rows of 64 B at row stride RS from region (hot: the same
// `hot_rows` rows per workgroup again and again; cold: fresh rows); at most `depth` pieces in flight per wave
template <int DEPTH, bool BUF>
__global__ void __launch_bounds__(512) k_rate(const unsigned char* g, size_t wg_bytes, int rs, int hot_pieces, int iters, int waves_on,
                                               unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < waves_on) {
    const unsigned char* wg = g + (size_t)blockIdx.x * wg_bytes;
    const i32x4 rsrc = make_rsrc(wg, 0x80000000u);
    const unsigned lane_off = (unsigned)(lane >> 2) * (unsigned)rs + (unsigned)(lane & 3) * 16u;
    const unsigned piece_bytes = 16u * (unsigned)rs;
    int k = wave;                                      // piece index, interleaved over the issuing waves
    for (int it = 0; it < iters; ++it) {
      const unsigned soff = (unsigned)(k % hot_pieces) * piece_bytes;
      const unsigned lds = lds0 + (unsigned)(((it & 7) * 8 + wave) * 1024);
      if (BUF) buf_dma16(lane_off, rsrc, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(lds));
      else glb_dma16(wg + soff + lane_off, __builtin_amdgcn_readfirstlane(lds));
      k += waves_on;
      if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      if (DEPTH == 32) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((unsigned*)smem)[5];
}

// ---------------------------------------------------------------------------------------------------------------- (C)
// flags: 1 fragment reads from LDS (else registers only), 2 LDS-DMA of the next halo (10 pieces per wave 0..6 and tile),
// 4 one barrier per tile (+ vmcnt(0)), 8 DMA source streams through HBM (else a 64 KiB box per workgroup, L2 resident),
// 16 the 27 weight fragments are re-loaded from global memory during the tile (3 per (kh,kw) step), 32: two pieces per step in
// steps 0..4 instead of the spread 2,1,1,...
template <int FLAGS>
__global__ void __launch_bounds__(512, 1) k_skel(const unsigned char* g, size_t wg_bytes, const u32x4* w, int tiles, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lv = lane & 15, lq = lane >> 4;
  const int vg = wave >> 1;
  u32x4 wf[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) wf[t] = w[(t * 8 + wave) * 64 + lane];
#pragma unroll
  for (int t = 0; t < 27; ++t) asm volatile("" : "+v"(wf[t]));
  for (int i = tid; i < 32768; i += 512) ((unsigned*)smem)[i] = 0x3f803f80u;   // bf16 1.0 pairs
  __syncthreads();
  unsigned fb[3];
  const int th = 2 * vg + (lv >> 3), tw = lv & 7;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) fb[kh] = (unsigned)((th + kh) * 10 + tw) * 64u + (((unsigned)lq ^ (((unsigned)(th + kh) & 1u) << 1)) << 4);
  const unsigned char* wg = g + (size_t)blockIdx.x * wg_bytes;
  const i32x4 rsrc = make_rsrc(wg, 0x80000000u);
  const int r0 = wave < 6 ? 16 * wave : 84, r = r0 + (lane >> 2);
  const unsigned lane_off = (unsigned)((r / 10) * 128 + (r % 10)) * 64u + (unsigned)(lane & 3) * 16u;   // box row (hh, hw) of a 128-wide plane
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 xr[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xr[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int t = 0; t < tiles; ++t) {
    const unsigned buf = (unsigned)(t & 1) * 64000u, obuf = 64000u - buf;
    const unsigned tile_off = (FLAGS & 8) ? (unsigned)t * (8u * 64u) : 0u;     // next tile along w: 8 voxels further
    constexpr int PLN = 10, SEQ = 9 * PLN, RING = 5;
    auto fa = [&](int e) -> unsigned { const int i = e % PLN, s = e / PLN; return buf + fb[s / 3] + (unsigned)((s % 3) * 64) + (unsigned)(i * 6400); };
    if (FLAGS & 1) {
#pragma unroll
      for (int e = 0; e < RING - 1; ++e) xr[e] = *(const u32x4*)(smem + fa(e));
    }
    int piece = 0;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int kh = s / 3, kw = s % 3;
      if ((FLAGS & 2) && wave < 7) {
        const int cnt = (FLAGS & 32) ? (s < 5 ? 2 : 0) : (s == 0 ? 2 : 1);
#pragma unroll
        for (int c = 0; c < cnt; ++c) {
          const int p = piece + c;
          const unsigned soff = tile_off + (unsigned)p * (128u * 128u * 64u);      // plane p of the box
          buf_dma16(lane_off, rsrc, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(lds0 + obuf + (unsigned)(p * 6400 + r0 * 64)));
        }
        piece += cnt;
      }
#pragma unroll
      for (int i = 0; i < PLN; ++i) {
        const int e = s * PLN + i;
        if ((FLAGS & 1) && e + RING - 1 < SEQ) xr[(e + RING - 1) % RING] = *(const u32x4*)(smem + fa(e + RING - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int pl = i - kd;
          if (pl >= 0 && pl < 8)
            acc[pl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[(kd * 3 + kh) * 3 + kw]),
                                                              __builtin_bit_cast(bf16x8, xr[e % RING]), acc[pl], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (FLAGS & 16) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) wf[(kd * 3 + kh) * 3 + kw] = w[(((kd * 3 + kh) * 3 + kw) * 8 + wave) * 64 + lane];
      }
    }
    if (FLAGS & 4) {
      if (FLAGS & 16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 512 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static float time_ms(F f, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "abc";
  // ---- (A) ----------------------------------------------------------------------------------------------------
  if (strchr(only, 'a')) {
    unsigned* g; unsigned* out;
    CK(hipMalloc(&g, 1 << 20)); CK(hipMalloc(&out, 4096));
    std::vector<unsigned> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;
    CK(hipMemcpy(g, h.data(), 1 << 20, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_sem, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char* names[] = {"odd lanes voffset 0xFFFFFF00", "num_records 0", "soffset 4096 > num_records 1024", "LDS base 100 KiB",
                           "num_records 512 (lanes >= 32 beyond)", "base below allocation + soffset"};
    for (int mode = 0; mode < 6; ++mode) {
      hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 160 * 1024, 0, g + 1024, out, mode);
      CK(hipDeviceSynchronize());
      unsigned o[256];
      CK(hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost));
      printf("[A%d] %-40s lane0: %08x %08x  lane1: %08x %08x  lane2: %08x  lane33: %08x  lane63: %08x\n", mode, names[mode], o[0], o[1], o[4], o[5],
             o[8], o[33 * 4], o[63 * 4]);
    }
    printf("      (source dword index = 1024 + byte offset / 4; 0xabababab = LDS untouched; 0 = zero written)\n");
    CK(hipFree(g)); CK(hipFree(out));
  }
  // ---- (B) ----------------------------------------------------------------------------------------------------
  if (strchr(only, 'b')) {
    const size_t total = (size_t)3 << 30;
    unsigned char* g; unsigned* out;
    CK(hipMalloc(&g, total)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(g, 1, total));
    const size_t wg_bytes = total / 256;   // 12 MiB per workgroup
#define RATE(DEPTH, BUF, rs, hot, waves)                                                                                   \
    do {                                                                                                                   \
      CK(hipFuncSetAttribute((const void*)k_rate<DEPTH, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));     \
      const int per_wave = 4096 / waves;                                                                                   \
      const int hot_pieces = hot ? 64 : (int)(wg_bytes / (16 * rs)) - 8;                                                   \
      float ms = time_ms([&] { hipLaunchKernelGGL((k_rate<DEPTH, BUF>), dim3(256), dim3(512), 64 * 1024, 0, g, wg_bytes, rs, hot_pieces, per_wave, waves, out); }); \
      const double bytes = 256.0 * waves * per_wave * 1024.0;                                                              \
      printf("[B] %s depth %2d  row stride %3d  %s  waves %d : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", BUF ? "buffer" : "global", \
             DEPTH, rs, hot ? "L2-hot (64 KiB/WG)" : "streaming        ", waves, ms * 1e3, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9)); \
    } while (0)
    for (int hot = 1; hot >= 0; --hot) {
      RATE(8, true, 64, hot, 8); RATE(16, true, 64, hot, 8); RATE(32, true, 64, hot, 8);
      RATE(16, true, 64, hot, 4); RATE(16, true, 64, hot, 2); RATE(32, true, 64, hot, 1);
      RATE(16, true, 192, hot, 8); RATE(16, false, 64, hot, 8); RATE(16, false, 192, hot, 8); RATE(4, true, 64, hot, 8);
    }
    CK(hipFree(g)); CK(hipFree(out));
  }
  // ---- (C) ----------------------------------------------------------------------------------------------------
  if (strchr(only, 'c')) {
    const size_t total = (size_t)2 << 30;
    unsigned char* g; u32x4* w; float* out;
    CK(hipMalloc(&g, total + (32u << 20))); CK(hipMalloc(&w, 27 * 8 * 64 * 16)); CK(hipMalloc(&out, 256 * 512 * 4));
    {
      std::vector<unsigned short> h(total / 2 > (64u << 20) ? (64u << 20) : total / 2);
      for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3f00 + (rand() & 0xff));      // bf16 in [0.5, 1)
      for (size_t o = 0; o < total; o += h.size() * 2) CK(hipMemcpy(g + o, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(g + total, h.data(), 32u << 20, hipMemcpyHostToDevice));
      std::vector<unsigned short> hw(27 * 8 * 64 * 8);
      for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3c00 + (rand() & 0xff));
      CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    const size_t wg_bytes = total / 256;
    const int tiles = 64;
#define SKEL(FL)                                                                                                           \
    do {                                                                                                                   \
      CK(hipFuncSetAttribute((const void*)k_skel<FL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));            \
      float ms = time_ms([&] { hipLaunchKernelGGL((k_skel<FL>), dim3(256), dim3(512), 130 * 1024, 0, g, wg_bytes, w, tiles, out); }); \
      const double fl = 256.0 * tiles * 512 * 32 * 32 * 27 * 2;                                                            \
      printf("[C] flags %2d (%s%s%s%s%s%s): %7.1f us  %7.1f TFLOP/s  %4.1f %% of 2.5 PF   %5.2f us/tile\n", FL, (FL & 1) ? "reads " : "", (FL & 2) ? "dma " : "", \
             (FL & 4) ? "barrier " : "", (FL & 8) ? "hbm " : "", (FL & 16) ? "wstream " : "", (FL & 32) ? "burst " : "", ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0, ms * 1e3 / tiles); \
    } while (0)
    SKEL(0); SKEL(1); SKEL(4); SKEL(5); SKEL(7); SKEL(15); SKEL(39); SKEL(47); SKEL(23); SKEL(31); SKEL(6); SKEL(14);
    CK(hipFree(g)); CK(hipFree(w)); CK(hipFree(out));
  }
  return 0;
}
