// cbim_common.h — shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Activations live in HBM as channels-last NDHWC ("[voxel][channel]") tensors of T = bf16
// (fast mode) or f32 (parity mode); every kernel moves them in 16-byte channel chunks
// (8 bf16 / 4 f32 per lane) so a wave touches 1 KiB of contiguous memory per instruction.
#pragma once
#ifdef CBIM_EMU
#include "hip_emu.h"   // tests/emu: host-side executor used only by the CPU test-suite
#define CBIM_LAST_LAUNCH() (cbim_emu::last_launch_err())
#else
#include <hip/hip_runtime.h>
#include <tuple>
#include <utility>
// Launches go through hipLaunchKernel so that the status checked afterwards is the return code of
// THIS launch (hipGetLastError() is a sticky per-thread value that unrelated runtime calls of the host
// process can leave set).
namespace cbim {
inline thread_local hipError_t g_launch_err = hipSuccess;
template <typename... P, size_t... I>
inline hipError_t launch_impl(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, std::tuple<P...>& t,
                              std::index_sequence<I...>) {
  void* ptr[sizeof...(P) + 1] = {(void*)&std::get<I>(t)..., nullptr};
  return hipLaunchKernel((const void*)k, g, b, ptr, sh, st);
}
template <typename... P, typename... A>
inline void launch(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, A&&... a) {
  std::tuple<P...> t{static_cast<P>(a)...};
  g_launch_err = launch_impl(k, g, b, sh, st, t, std::index_sequence_for<P...>{});
}
}  // namespace cbim
#define CBIM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  cbim::launch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)
#define CBIM_LAST_LAUNCH() (cbim::g_launch_err)
#endif
#include <stdint.h>

#include "../../include/cbim_hip.h"

namespace cbim {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct bf16_tag {};  // element-type tags (bf16 carried as raw bits)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16x2 (RNE): one v_cvt_pk_bf16_f32 on gfx950 (the bit-twiddling form costs ~16 VALU per
// pair and made the conv staging/epilogue VALU-bound).  Written as a vector conversion the compiler selects itself,
// NOT as inline asm: the hazard recognizer does not see VALU writes made inside an asm statement, and gfx950 needs 2
// wait states between a VALU write of a VGPR and an MFMA reading it as SrcA/B — with asm the k_attn_bwd_mfma operands
// packed right before their MFMA were read stale (inf/NaN rows on hardware only).
#ifndef CBIM_EMU
typedef __bf16 cbim_bf2_t __attribute__((ext_vector_type(2)));
typedef float cbim_f2_t __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
#ifdef CBIM_EMU
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
#else
  cbim_f2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, cbim_bf2_t));
#endif
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  typedef float type;
  static constexpr int CPC = 4;  // channels per 16-byte chunk
  static constexpr int SIZE = 4;
  static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) {
    u32x4 v;
    v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]);
    v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
    return v;
  }
  static __device__ __forceinline__ float load1(const void* p, size_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store1(void* p, size_t i, float v) { ((float*)p)[i] = v; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_tag> {
  typedef bf16_t type;
  static constexpr int CPC = 8;
  static constexpr int SIZE = 2;
  static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) {
    u32x4 v;
    v.x = pk_bf16(f[0], f[1]);
    v.y = pk_bf16(f[2], f[3]);
    v.z = pk_bf16(f[4], f[5]);
    v.w = pk_bf16(f[6], f[7]);
    return v;
  }
  static __device__ __forceinline__ float load1(const void* p, size_t i) { return bf2f(((const bf16_t*)p)[i]); }
  static __device__ __forceinline__ void store1(void* p, size_t i, float v) { ((bf16_t*)p)[i] = (bf16_t)pk_bf16(v, 0.f); }
  static __device__ __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};

// 16-byte chunk load/store at ELEMENT index `e` (must be a multiple of CPC) of a T tensor.
template <typename T>
__device__ __forceinline__ u32x4 ld_chunk(const void* base, size_t e) {
  return *(const u32x4*)((const char*)base + e * Elem<T>::SIZE);
}
template <typename T>
__device__ __forceinline__ void st_chunk(void* base, size_t e, const u32x4& v) {
  *(u32x4*)((char*)base + e * Elem<T>::SIZE) = v;
}

// ---- activations (model/dim3/utils.py:23-30 of the reference) ---------------------------------
__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case CBIM_ACT_RELU: return x > 0.f ? x : 0.f;
    case CBIM_ACT_LRELU: return x > 0.f ? x : 0.01f * x;
    case CBIM_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case CBIM_ACT_SILU: return x / (1.f + __expf(-x));
    case CBIM_ACT_ELU: return x > 0.f ? x : __expf(x) - 1.f;
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case CBIM_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case CBIM_ACT_LRELU: return x > 0.f ? 1.f : 0.01f;
    case CBIM_ACT_GELU: {
      float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
      float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case CBIM_ACT_SILU: {
      float s = 1.f / (1.f + __expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    case CBIM_ACT_ELU: return x > 0.f ? 1.f : __expf(x);
    default: return 1.f;
  }
}

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Running (count, mean, M2) statistics and Chan's pairwise merge: InstanceNorm variances are
// formed from centred second moments, never from E[x^2]-mean^2 (which cancels catastrophically
// when |mean| >> std, e.g. after residual adds).
struct Moments { float n, mean, m2; };
__device__ __forceinline__ Moments moments_merge(const Moments& a, const Moments& b) {
  Moments r;
  r.n = a.n + b.n;
  if (r.n <= 0.f) { r.n = 0.f; r.mean = 0.f; r.m2 = 0.f; return r; }
  float d = b.mean - a.mean;
  float fb = b.n / r.n;
  r.mean = a.mean + d * fb;
  r.m2 = a.m2 + b.m2 + d * d * a.n * fb;
  return r;
}
// moments of values given as shifted sums: n values, sum d, sum d^2 with d = x - shift
__device__ __forceinline__ Moments moments_from_shifted(float n, float shift, float sd, float sd2) {
  Moments r;
  r.n = n;
  if (n <= 0.f) { r.mean = 0.f; r.m2 = 0.f; return r; }
  float md = sd / n;
  r.mean = shift + md;
  r.m2 = sd2 - sd * md;
  if (r.m2 < 0.f) r.m2 = 0.f;
  return r;
}

// XCD-aware block remap: consecutive logical tiles land on the same XCD (its private 4 MiB L2),
// block b runs on XCD b % 8 (MI355X_MICROARCH.md, observed placement; speed only, never correctness).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned nx = 8;
  unsigned q = nblk / nx, r = nblk % nx;
  unsigned xcd = bid % nx, idx = bid / nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace cbim

// ---- host-side error plumbing (cbim_api.cpp) -----------------------------------------------------
void cbim_set_error(const char* fmt, ...);
#define CBIM_CHECK(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      cbim_set_error(__VA_ARGS__);             \
      return (code);                           \
    }                                          \
  } while (0)

// ---- per-code-object warm-up ------------------------------------------------------------------------------
// Every .hip file is its own code object, loaded lazily at its first launch.  CBIM_DEFINE_WARM(tag) gives the
// file a no-op kernel and `int cbim_warm_<tag>(void* stream)`, which launches it (re-selecting the device and
// retrying if that first launch fails); cbim_runtime_warmup() runs all of them once, so module loading is not
// paid inside a timed or captured region and a broken runtime binding (e.g. this library bound to a second HIP
// runtime instance, see _lib.py) is reported at start-up with the failing code object named.
#ifdef CBIM_EMU
#define CBIM_DEFINE_WARM(tag) extern "C" int cbim_warm_##tag(void*) { return 0; }
#else
#define CBIM_DEFINE_WARM(tag)                                                                           \
  namespace cbim { __global__ void k_warm_##tag() {} }                                                  \
  extern "C" int cbim_warm_##tag(void* stream) {                                                        \
    for (int attempt = 0; attempt < 4; ++attempt) {                                                     \
      cbim::launch(cbim::k_warm_##tag, dim3(1), dim3(64), 0, (hipStream_t)stream);                      \
      if (cbim::g_launch_err == hipSuccess) return CBIM_OK;                                             \
      int n = 0, d = 0;                                                                                 \
      (void)hipGetDeviceCount(&n);                                                                      \
      (void)hipGetDevice(&d);                                                                           \
      (void)hipSetDevice(d);                                                                            \
      (void)hipFree(nullptr);                                                                           \
    }                                                                                                   \
    cbim_set_error("first launch from code object '" #tag "': %s", hipGetErrorString(cbim::g_launch_err)); \
    return CBIM_ELAUNCH;                                                                                \
  }
#endif
