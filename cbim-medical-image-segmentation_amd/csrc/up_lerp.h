// up_lerp.h — the one trilinear interpolation arithmetic every up-sampling kernel shares (pool_up_kernels.hip,
// up_tile_kernels.hip): results are bit-identical whichever kernel forms them (tests compare the paths with torch.equal).
//
// Eight corner chunks (16 bytes each: CPC channels of one coarse voxel), order [(a * 2 + b) * 2 + c] with a / b / c = lower or
// upper neighbour along D / H / W; weights (l0, l1) per axis.  W first, then H, then D, each as fma(l1, upper, l0 * lower) on
// packed f32 pairs (v_pk_mul_f32 / v_pk_fma_f32): 7 x CPC/2 x 2 packed instructions per chunk — the nested scalar form it
// replaces was 22 x CPC scalar instructions under -ffp-contract=off and made the up-path kernels vector-ALU bound
// (1.5 TB/s at the 128^3 level).
#pragma once
#include "cbim_common.h"

namespace cbim {

typedef float up_f2 __attribute__((ext_vector_type(2)));

template <typename T> struct UpPairs;
template <> struct UpPairs<bf16_tag> {
  static constexpr int NP = 4;    // f32 pairs per 16-byte chunk
  static __device__ __forceinline__ void unpack(const u32x4& v, up_f2* f) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = up_f2{__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
  }
};
template <> struct UpPairs<float> {
  static constexpr int NP = 2;
  static __device__ __forceinline__ void unpack(const u32x4& v, up_f2* f) {
    f[0] = up_f2{__uint_as_float(v.x), __uint_as_float(v.y)};
    f[1] = up_f2{__uint_as_float(v.z), __uint_as_float(v.w)};
  }
};

// out[CPC] = the interpolated chunk (f32, not yet rounded to the storage type)
template <typename T>
__device__ __forceinline__ void trilerp(const u32x4* c, float d0, float d1, float h0, float h1, float w0, float w1, float* out) {
  constexpr int NP = UpPairs<T>::NP;
  const up_f2 W0 = {w0, w0}, W1 = {w1, w1}, H0 = {h0, h0}, H1 = {h1, h1}, D0 = {d0, d0}, D1 = {d1, d1};
  up_f2 u[2][NP];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    up_f2 t[2][NP];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      up_f2 f0[NP], f1[NP];
      UpPairs<T>::unpack(c[(a * 2 + b) * 2], f0);
      UpPairs<T>::unpack(c[(a * 2 + b) * 2 + 1], f1);
#pragma unroll
      for (int j = 0; j < NP; ++j) t[b][j] = __builtin_elementwise_fma(W1, f1[j], W0 * f0[j]);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) u[a][j] = __builtin_elementwise_fma(H1, t[1][j], H0 * t[0][j]);
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const up_f2 r = __builtin_elementwise_fma(D1, u[1][j], D0 * u[0][j]);
    out[2 * j] = r.x;
    out[2 * j + 1] = r.y;
  }
}

}  // namespace cbim
