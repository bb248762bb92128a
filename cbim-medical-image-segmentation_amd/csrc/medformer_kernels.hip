// medformer_kernels.hip — the MedFormer-specific pieces of the model/dim3 hot path.
//
//   k_dwconv / k_dwconv_wgrad   depthwise k^3 convolution (DepthwiseSeparableConv.depthwise and the MBConv
//                               depthwise ConvNormAct, /root/reference/model/dim3/conv_layers.py:126-157,211)
//                               with the pre-activation InstanceNorm(+act) applied on the input load
//   k_space_to_depth            PatchMerging's 8 strided slices + channel concat (medformer_utils.py:163-171)
//   k_attn_fwd / k_attn_bwd     BidirectionAttention core (medformer_utils.py:63-97): logits q_f·q_m^T*scale
//                               [L x M], row softmax -> feature side, column softmax over all L voxels ->
//                               map side; one pass over q,v per direction, column softmax merged from
//                               per-block online-softmax records in a fixed order (deterministic)
//   k_mappool_fwd / _bwd        SemanticMapGeneration tail (medformer_utils.py:218-228): softmax over L per
//                               map code + einsum('bij,bkj->bik')
//
// All of these are HBM/latency-bound (M=64 codes, d_head=32: ~4 GF per layer); they run on the vector
// ALUs with the small map-side operands read through the scalar cache (uniform addresses).
// Layout: feature maps channels-last [N][L][C]; head split is the reference's "(dim_head heads)":
// channel c = d*heads + h (medformer_utils.py:43-51).  Map-side tensors are float32 [N][M][inner].
#include "cbim_common.h"
#include "conv_wgrad_r32.h"
#include <stdlib.h>

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

namespace cbim {

static constexpr int NT = 256;
static constexpr int AT = 128;  // voxels (= threads) per attention block
static constexpr int MM = 64;   // max map codes

static inline int grid_for(int64_t items) {
  int64_t b = (items + NT - 1) / NT;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- depthwise convolution ------------------------------------------------------------------------
// y[n,l,c] = sum_t w[c][t] * a(x[n,l+off(t),c]),  a = act((x-mean)*rstd) when in_stats else x, zero outside
// the volume; `bias` (float [N][C], optional) is added to in-range inputs (used for dy + dmean/S in the
// backward); flip=1 uses w[c][T-1-t] (data gradient of a same-padded stride-1 depthwise conv).
template <typename T>
__global__ void __launch_bounds__(NT) k_dwconv(const void* __restrict__ x, int64_t xs,
                                               const float* __restrict__ in_stats, int act,
                                               const float* __restrict__ bias, const float* __restrict__ w,
                                               int flip, void* __restrict__ y, int64_t ys, int D, int H, int W,
                                               int C, int kD, int kH, int kW, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  const int TT = kD * kH * kW, pD = kD / 2, pH = kH / 2, pW = kW / 2;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int cc = (int)(i % cch);
    int64_t r = i / cch;
    int wo = (int)(r % W); r /= W;
    int ho = (int)(r % H); r /= H;
    int dz = (int)(r % D);
    int64_t n = r / D;
    const int c0 = cc * CPC;
    float mean[CPC], rstd[CPC], bs[CPC], acc[CPC];
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      mean[j] = in_stats ? in_stats[((size_t)n * C + c0 + j) * 2] : 0.f;
      rstd[j] = in_stats ? in_stats[((size_t)n * C + c0 + j) * 2 + 1] : 1.f;
      bs[j] = bias ? bias[(size_t)n * C + c0 + j] : 0.f;
      acc[j] = 0.f;
    }
    for (int a = 0; a < kD; ++a) {
      int dd = dz + a - pD;
      if (dd < 0 || dd >= D) continue;
      for (int b = 0; b < kH; ++b) {
        int hh = ho + b - pH;
        if (hh < 0 || hh >= H) continue;
        for (int c = 0; c < kW; ++c) {
          int ww = wo + c - pW;
          if (ww < 0 || ww >= W) continue;
          int tap = (a * kH + b) * kW + c;
          int wt = flip ? TT - 1 - tap : tap;
          size_t row = (((size_t)n * D + dd) * H + hh) * W + ww;
          float f[CPC];
          Elem<T>::unpack(ld_chunk<T>(x, row * xs + c0), f);
#pragma unroll
          for (int j = 0; j < CPC; ++j) {
            float v = f[j];
            if (in_stats) v = act_fwd((v - mean[j]) * rstd[j], act);
            v += bs[j];
            acc[j] = fmaf(v, w[(size_t)(c0 + j) * TT + wt], acc[j]);   // explicit: -ffp-contract=off build
          }
        }
      }
    }
    size_t orow = (((size_t)n * D + dz) * H + ho) * W + wo;
    st_chunk<T>(y, orow * ys + c0, Elem<T>::pack(acc));
  }
}

// dw[c][t] = sum_{n,l} a(x[n,l+off(t),c]) * (dy[n,l,c] + dy_bias[n,c]).  Thread = (channel chunk, tap);
// block (bx, by) covers voxels [bx*vpb, +vpb) and chunks [by*G, +G), G = NT / T.  Per-block partial sums
// go to part[bx][C][T]; k_dwconv_wgrad_reduce adds them in block order (deterministic).
template <typename T>
__global__ void __launch_bounds__(NT) k_dwconv_wgrad(const void* __restrict__ x, int64_t xs,
                                                     const float* __restrict__ in_stats, int act,
                                                     const void* __restrict__ dy, int64_t dys,
                                                     const float* __restrict__ dy_bias, float* __restrict__ part,
                                                     int N, int D, int H, int W, int C, int kD, int kH, int kW,
                                                     int vpb) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  const int TT = kD * kH * kW, G = NT / TT;
  const int cl = threadIdx.x / TT, tap = threadIdx.x % TT;
  const int cc = blockIdx.y * G + cl;
  if (cl >= G || cc >= cch) return;
  const int a = tap / (kH * kW) - kD / 2, b = (tap / kW) % kH - kH / 2, c = tap % kW - kW / 2;
  const int c0 = cc * CPC;
  const int64_t S = (int64_t)D * H * W, tot = (int64_t)N * S;
  int64_t v0 = (int64_t)blockIdx.x * vpb, v1 = v0 + vpb;
  if (v1 > tot) v1 = tot;
  float acc[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) acc[j] = 0.f;
  for (int64_t v = v0; v < v1; ++v) {
    int64_t r = v;
    int wo = (int)(r % W); r /= W;
    int ho = (int)(r % H); r /= H;
    int dz = (int)(r % D);
    int64_t n = r / D;
    int dd = dz + a, hh = ho + b, ww = wo + c;
    if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
    float g[CPC], f[CPC];
    Elem<T>::unpack(ld_chunk<T>(dy, (size_t)v * dys + c0), g);
    size_t row = (((size_t)n * D + dd) * H + hh) * W + ww;
    Elem<T>::unpack(ld_chunk<T>(x, row * xs + c0), f);
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      float xv = f[j];
      if (in_stats) {
        float mean = in_stats[((size_t)n * C + c0 + j) * 2], rstd = in_stats[((size_t)n * C + c0 + j) * 2 + 1];
        xv = act_fwd((xv - mean) * rstd, act);
      }
      float gv = g[j] + (dy_bias ? dy_bias[(size_t)n * C + c0 + j] : 0.f);
      acc[j] = fmaf(xv, gv, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < CPC; ++j) part[((size_t)blockIdx.x * C + c0 + j) * TT + tap] = acc[j];
}


// ---- depthwise convolution, kW == 3 fast path -------------------------------------------------------
// Thread = (channel chunk, strip of WT consecutive outputs along W): every (kd,kh) input row segment of
// WT+2 chunks is loaded and normalised ONCE and feeds 3*WT multiply-adds per channel; the weights of the
// block's chunk group sit in LDS as [tap][channel].  blockIdx.y = chunk group (DG chunks).
static constexpr int DG = 32;   // channel chunks per block (one 512-byte bf16 row segment per voxel)
static constexpr int WT = 4;
// MODE: 0 raw input (dgrad; optional bias) | 1 InstanceNorm + ReLU on load | 2 InstanceNorm only | 3 InstanceNorm + run-time
// activation.  Packed f32 pairs throughout (v_pk_add / v_pk_mul / v_pk_fma_f32): the kernel is vector-ALU bound (27 taps x 8
// channels per output against 32 bytes of traffic), and as scalar fmaf + a run-time activation switch per element it spent
// ~750 lane-instructions per output chunk (52 us per call on the 4x-expanded MBConv tensors of MedFormer).
typedef float dw_f2 __attribute__((ext_vector_type(2)));
template <typename T> struct DwPairs;
template <> struct DwPairs<bf16_tag> {
  static constexpr int NP = 4;
  static __device__ __forceinline__ void unpack(const u32x4& v, dw_f2* f) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = dw_f2{__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
  }
};
template <> struct DwPairs<float> {
  static constexpr int NP = 2;
  static __device__ __forceinline__ void unpack(const u32x4& v, dw_f2* f) {
    f[0] = dw_f2{__uint_as_float(v.x), __uint_as_float(v.y)};
    f[1] = dw_f2{__uint_as_float(v.z), __uint_as_float(v.w)};
  }
};

template <typename T, int MODE>
__global__ void __launch_bounds__(NT) k_dwconv3(const void* __restrict__ x, int64_t xs,
                                                const float* __restrict__ in_stats, int act,
                                                const float* __restrict__ bias, const float* __restrict__ w,
                                                int flip, void* __restrict__ y, int64_t ys, int N, int D, int H, int W,
                                                int C, int kD, int kH) {
  constexpr int CPC = Elem<T>::CPC, NP = DwPairs<T>::NP;
  __shared__ float w_s[27 * DG * CPC];
  const int cch = C / CPC;
  const int g0 = blockIdx.y * DG;
  const int G = cch - g0 < DG ? cch - g0 : DG;
  const int TT = kD * kH * 3, pD = kD / 2, pH = kH / 2;
  // 8 loads in flight per thread (one load per trip of a rolled loop: up to 27 serial memory round trips, ~20 us of every
  // launch — the whole duration of the calls on the 8^3 .. 16^3 tensors of the deep stages)
  for (int base = threadIdx.x; base < TT * G * CPC; base += NT * 8) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * NT;
      wv[u] = 0.f;
      if (i < TT * G * CPC) {
        const int tap = i / (G * CPC), ch = i % (G * CPC);
        wv[u] = w[(size_t)(g0 * CPC + ch) * TT + (flip ? TT - 1 - tap : tap)];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * NT;
      if (i < TT * G * CPC) w_s[(i / (G * CPC)) * (DG * CPC) + i % (G * CPC)] = wv[u];
    }
  }
  __syncthreads();
  const int cl = threadIdx.x % G, sl = threadIdx.x / G, SL = NT / G;
  if (sl >= SL) return;
  const int c0 = (g0 + cl) * CPC;
  const int strips = (W + WT - 1) / WT;
  const int64_t total = (int64_t)N * D * H * strips;
  for (int64_t it = (int64_t)blockIdx.x * SL + sl; it < total; it += (int64_t)gridDim.x * SL) {
    // 32-bit index decode (the host checks total < 2^31; 64-bit divisions by run-time values cost ~150
    // instructions each, four of them per 864 multiply-adds)
    unsigned r = (unsigned)it;
    const int w0 = (int)(r % (unsigned)strips) * WT; r /= (unsigned)strips;
    const int ho = (int)(r % (unsigned)H); r /= (unsigned)H;
    const int dz = (int)(r % (unsigned)D);
    const int64_t n = r / (unsigned)D;
    dw_f2 nmean[NP], rstd[NP], bs[NP], acc[WT][NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      nmean[j] = dw_f2{0.f, 0.f}; rstd[j] = dw_f2{1.f, 1.f}; bs[j] = dw_f2{0.f, 0.f};
      if (MODE != 0) {
        const float* st = in_stats + ((size_t)n * C + c0 + 2 * j) * 2;
        nmean[j] = dw_f2{-st[0], -st[2]};
        rstd[j] = dw_f2{st[1], st[3]};
      }
      if (bias) bs[j] = dw_f2{bias[(size_t)n * C + c0 + 2 * j], bias[(size_t)n * C + c0 + 2 * j + 1]};
#pragma unroll
      for (int o = 0; o < WT; ++o) acc[o][j] = dw_f2{0.f, 0.f};
    }
    for (int a = 0; a < kD; ++a) {
      const int dd = dz + a - pD;
      if (dd < 0 || dd >= D) continue;
      for (int b = 0; b < kH; ++b) {
        const int hh = ho + b - pH;
        if (hh < 0 || hh >= H) continue;
        const size_t rbase = (((size_t)n * D + dd) * H + hh) * W;
        // one 64-bit row pointer per (kd,kh); the WT+2 chunks are xs elements apart
        const unsigned char* rp = (const unsigned char*)x + ((int64_t)(rbase + w0 - 1) * xs + c0) * (int64_t)Elem<T>::SIZE;
        const unsigned xsb = (unsigned)xs * Elem<T>::SIZE;
        dw_f2 in[WT + 2][NP];
#pragma unroll
        for (int q = 0; q < WT + 2; ++q) {
          const int ww = w0 - 1 + q;
          if (ww >= 0 && ww < W) {
            DwPairs<T>::unpack(*(const u32x4*)(rp + (size_t)q * xsb), in[q]);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
              dw_f2 v = in[q][j];
              if (MODE != 0) {
                v = (v + nmean[j]) * rstd[j];
                if (MODE == 1) v = __builtin_elementwise_max(v, dw_f2{0.f, 0.f});
                if (MODE == 3) v = dw_f2{act_fwd(v.x, act), act_fwd(v.y, act)};
              }
              in[q][j] = v + bs[j];
            }
          } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) in[q][j] = dw_f2{0.f, 0.f};
          }
        }
        const float* wt = w_s + ((a * kH + b) * 3) * (DG * CPC) + cl * CPC;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dw_f2 wv[NP];
#pragma unroll
          for (int j = 0; j < NP; ++j) wv[j] = dw_f2{wt[c * (DG * CPC) + 2 * j], wt[c * (DG * CPC) + 2 * j + 1]};
#pragma unroll
          for (int o = 0; o < WT; ++o)
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[o][j] = __builtin_elementwise_fma(in[o + c][j], wv[j], acc[o][j]);
        }
      }
    }
    const size_t obase = (((size_t)n * D + dz) * H + ho) * W;
#pragma unroll
    for (int o = 0; o < WT; ++o)
      if (w0 + o < W) {
        float f[CPC];
#pragma unroll
        for (int j = 0; j < NP; ++j) { f[2 * j] = acc[o][j].x; f[2 * j + 1] = acc[o][j].y; }
        st_chunk<T>(y, (obase + w0 + o) * ys + c0, Elem<T>::pack(f));
      }
  }
}

// ---- depthwise 3x3x3 (kD, kH <= 3, kW == 3), LDS-tiled (round 4) ---------------------------------------------------------
// k_dwconv3 reads every input row segment straight from global memory once per (kd, kh) it feeds and normalises it there:
// nine dependent round trips per strip and the InstanceNorm + activation of every input value nine times over (276 us on the
// 64^3 x 256 PatchMerging tensor against 45 us of HBM time, and ~20 us of serial latency on the 8^3 .. 16^3 tensors of the deep
// stages: 114 of the 120 launches of a MedFormer step).  Here a workgroup owns a 4 x 8 x 8 output tile of LG channel chunks:
// the (4+kD-1) x (8+kH-1) x 10 input halo is requested in ONE sweep (every load of a thread in flight together), transformed
// once — InstanceNorm, activation, bias, literal zeros for the padding — and kept in LDS in the storage type; thread = (chunk,
// strip of 4 outputs along W) as before, its 6-chunk row segments and the tap weights now come from LDS.
static constexpr int LG = 4;                          // channel chunks per workgroup (64 contiguous bytes per halo voxel; 42 KB of LDS: three workgroups per CU)
static constexpr int LTD = 4, LTH = 8, LTW = 8;
template <typename T, int MODE>
__global__ void __launch_bounds__(NT) k_dwconv3_lds(const void* __restrict__ x, int64_t xs, const float* __restrict__ in_stats, int act,
                                                    const float* __restrict__ bias, const float* __restrict__ w, int flip,
                                                    void* __restrict__ y, int64_t ys, int N, int D, int H, int W, int C, int kD, int kH,
                                                    int tiles_d, int tiles_h, int tiles_w) {
  constexpr int CPC = Elem<T>::CPC, NP = DwPairs<T>::NP;
  __shared__ __attribute__((aligned(16))) unsigned char halo_s[(LTD + 2) * (LTH + 2) * (LTW + 2) * LG * 16];
  __shared__ __attribute__((aligned(16))) float w_s[27 * LG * CPC];
  const int cch = C / CPC;
  const int g0 = blockIdx.y * LG;
  const int G = cch - g0 < LG ? cch - g0 : LG;
  const int TT = kD * kH * 3, pD = kD / 2, pH = kH / 2;
  const int hD = LTD + kD - 1, hH = LTH + kH - 1, hW = LTW + 2;
  unsigned bt = blockIdx.x;
  const int tw = (int)(bt % (unsigned)tiles_w); bt /= (unsigned)tiles_w;
  const int th = (int)(bt % (unsigned)tiles_h); bt /= (unsigned)tiles_h;
  const int td = (int)(bt % (unsigned)tiles_d);
  const int n = (int)(bt / (unsigned)tiles_d);
  const int d0 = td * LTD, h0 = th * LTH, w0t = tw * LTW;
  const int tid = threadIdx.x;
  // ---- weights of the chunk group: [tap][LG * CPC] -------------------------------------------------------------------
  {
    constexpr int WU = (27 * LG * CPC + NT - 1) / NT;   // every weight load of the thread in flight together
    float wv[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = tid + u * NT;
      const int tap = i / (LG * CPC), ch = i % (LG * CPC);
      wv[u] = (i < TT * LG * CPC && ch < G * CPC) ? w[(size_t)(g0 * CPC + ch) * TT + (flip ? TT - 1 - tap : tap)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < WU; ++u)
      if (tid + u * NT < TT * LG * CPC) w_s[tid + u * NT] = wv[u];
  }
  // ---- halo: item = (halo voxel, chunk); a thread always meets the same chunk (NT % LG == 0) -----------------------
  const int cl = tid % LG, v0 = tid / LG;
  constexpr int VPT = NT / LG;                        // halo voxels per sweep of the workgroup
  const bool c_ok = cl < G;
  const int c0 = (g0 + (c_ok ? cl : 0)) * CPC;
  dw_f2 nmean[NP], rstd[NP], bs[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    nmean[j] = dw_f2{0.f, 0.f}; rstd[j] = dw_f2{1.f, 1.f}; bs[j] = dw_f2{0.f, 0.f};
    if (MODE != 0) {
      const float* st = in_stats + ((size_t)n * C + c0 + 2 * j) * 2;
      nmean[j] = dw_f2{-st[0], -st[2]};
      rstd[j] = dw_f2{st[1], st[3]};
    }
    if (bias) bs[j] = dw_f2{bias[(size_t)n * C + c0 + 2 * j], bias[(size_t)n * C + c0 + 2 * j + 1]};
  }
  const int hV = hD * hH * hW;
  constexpr int UL = 10;                              // loads in flight per thread and trip: the 600-voxel halo in one sweep
  for (int base = v0; base < hV; base += VPT * UL) {
    u32x4 raw[UL];
    bool in[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int hv = base + u * VPT;
      const int hw = hv % hW, r = hv / hW, hh = r % hH, hd = r / hH;
      const int dd = d0 - pD + hd, hy = h0 - pH + hh, wx = w0t - 1 + hw;
      in[u] = c_ok && hv < hV && dd >= 0 && dd < D && hy >= 0 && hy < H && wx >= 0 && wx < W;
      raw[u] = u32x4{0u, 0u, 0u, 0u};
      if (in[u]) raw[u] = *(const u32x4*)((const unsigned char*)x + (((((int64_t)n * D + dd) * H + hy) * W + wx) * xs + c0) * (int64_t)Elem<T>::SIZE);
    }
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int hv = base + u * VPT;
      if (hv < hV) {
        u32x4 o = u32x4{0u, 0u, 0u, 0u};
        if (in[u]) {
          dw_f2 f[NP];
          DwPairs<T>::unpack(raw[u], f);
          float g[CPC];
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            dw_f2 v = f[j];
            if (MODE != 0) {
              v = (v + nmean[j]) * rstd[j];
              if (MODE == 1) v = __builtin_elementwise_max(v, dw_f2{0.f, 0.f});
              if (MODE == 3) v = dw_f2{act_fwd(v.x, act), act_fwd(v.y, act)};
            }
            v = v + bs[j];
            g[2 * j] = v.x; g[2 * j + 1] = v.y;
          }
          o = Elem<T>::pack(g);
        }
        *(u32x4*)(halo_s + ((size_t)hv * LG + cl) * 16) = o;
      }
    }
  }
  __syncthreads();
  if (!c_ok) return;
  // ---- outputs: thread = (chunk cl, strips v0, v0 + VPT, ...) of the 64 strips of the tile ---------------------------
  constexpr int STRIPS = LTD * LTH * (LTW / WT);
  for (int sidx = v0; sidx < STRIPS; sidx += VPT) {
    const int sw = sidx % (LTW / WT), r = sidx / (LTW / WT), sh = r % LTH, sd = r / LTH;
    const int dz = d0 + sd, ho = h0 + sh, wo = w0t + sw * WT;
    if (dz >= D || ho >= H || wo >= W) continue;
    dw_f2 acc[WT][NP];
#pragma unroll
    for (int o = 0; o < WT; ++o)
#pragma unroll
      for (int j = 0; j < NP; ++j) acc[o][j] = dw_f2{0.f, 0.f};
    for (int a = 0; a < kD; ++a)
      for (int b = 0; b < kH; ++b) {
        const unsigned char* rp = halo_s + ((size_t)(((sd + a) * hH + sh + b) * hW + sw * WT) * LG + cl) * 16;
        dw_f2 inr[WT + 2][NP];
#pragma unroll
        for (int q = 0; q < WT + 2; ++q) DwPairs<T>::unpack(*(const u32x4*)(rp + (size_t)q * LG * 16), inr[q]);
        const float* wt = w_s + ((a * kH + b) * 3) * (LG * CPC) + cl * CPC;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dw_f2 wv[NP];
#pragma unroll
          for (int j = 0; j < NP; ++j) wv[j] = dw_f2{wt[c * (LG * CPC) + 2 * j], wt[c * (LG * CPC) + 2 * j + 1]};
#pragma unroll
          for (int o = 0; o < WT; ++o)
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[o][j] = __builtin_elementwise_fma(inr[o + c][j], wv[j], acc[o][j]);
        }
      }
    const size_t obase = (((size_t)n * D + dz) * H + ho) * W;
#pragma unroll
    for (int o = 0; o < WT; ++o)
      if (wo + o < W) {
        float f[CPC];
#pragma unroll
        for (int j = 0; j < NP; ++j) { f[2 * j] = acc[o][j].x; f[2 * j + 1] = acc[o][j].y; }
        st_chunk<T>(y, (obase + wo + o) * ys + (size_t)(g0 + cl) * CPC, Elem<T>::pack(f));
      }
  }
}

template <typename T>
__global__ void __launch_bounds__(NT) k_dwconv3_wgrad(const void* __restrict__ x, int64_t xs,
                                                      const float* __restrict__ in_stats, int act,
                                                      const void* __restrict__ dy, int64_t dys,
                                                      const float* __restrict__ dy_bias, float* __restrict__ part,
                                                      int N, int D, int H, int W, int C, int kD, int kH, int rpb) {
  constexpr int CPC = Elem<T>::CPC;
  __shared__ float red[NT * 3 * CPC];
  const int cch = C / CPC;
  const int g0 = blockIdx.y * DG;
  const int G = cch - g0 < DG ? cch - g0 : DG;
  const int TT = kD * kH * 3, pD = kD / 2, pH = kH / 2;
  const int cl = threadIdx.x % G, rl = threadIdx.x / G, RL = NT / G;
  const bool active = rl < RL;
  const int c0 = (g0 + cl) * CPC;
  const int64_t rows = (int64_t)N * D * H;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  int64_t r1 = r0 + rpb;
  if (r1 > rows) r1 = rows;
  for (int pr = 0; pr < kD * kH; ++pr) {
    const int a = pr / kH - pD, b = pr % kH - pH;
    float acc[3][CPC];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < CPC; ++j) acc[c][j] = 0.f;
    if (active) {
      for (int64_t row = r0 + rl; row < r1; row += RL) {
        int64_t r = row;
        const int ho = (int)(r % H); r /= H;
        const int dz = (int)(r % D);
        const int64_t n = r / D;
        const int dd = dz + a, hh = ho + b;
        if (dd < 0 || dd >= D || hh < 0 || hh >= H) continue;
        float mean[CPC], rstd[CPC], bs[CPC];
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          mean[j] = in_stats ? in_stats[((size_t)n * C + c0 + j) * 2] : 0.f;
          rstd[j] = in_stats ? in_stats[((size_t)n * C + c0 + j) * 2 + 1] : 1.f;
          bs[j] = dy_bias ? dy_bias[(size_t)n * C + c0 + j] : 0.f;
        }
        const size_t xb = (((size_t)n * D + dd) * H + hh) * W, gb = (size_t)row * W;
        float xm[CPC], xc[CPC], xp[CPC];   // a(x) at w-1, w, w+1
#pragma unroll
        for (int j = 0; j < CPC; ++j) xm[j] = 0.f;
        Elem<T>::unpack(ld_chunk<T>(x, xb * xs + c0), xc);
#pragma unroll
        for (int j = 0; j < CPC; ++j)
          if (in_stats) xc[j] = act_fwd((xc[j] - mean[j]) * rstd[j], act);
        for (int wv = 0; wv < W; ++wv) {
          if (wv + 1 < W) {
            Elem<T>::unpack(ld_chunk<T>(x, (xb + wv + 1) * xs + c0), xp);
#pragma unroll
            for (int j = 0; j < CPC; ++j)
              if (in_stats) xp[j] = act_fwd((xp[j] - mean[j]) * rstd[j], act);
          } else {
#pragma unroll
            for (int j = 0; j < CPC; ++j) xp[j] = 0.f;
          }
          float g[CPC];
          Elem<T>::unpack(ld_chunk<T>(dy, (gb + wv) * dys + c0), g);
#pragma unroll
          for (int j = 0; j < CPC; ++j) {
            float gv = g[j] + bs[j];
            acc[0][j] = fmaf(xm[j], gv, acc[0][j]);
            acc[1][j] = fmaf(xc[j], gv, acc[1][j]);
            acc[2][j] = fmaf(xp[j], gv, acc[2][j]);
            xm[j] = xc[j];
            xc[j] = xp[j];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < CPC; ++j) red[(threadIdx.x * 3 + c) * CPC + j] = acc[c][j];
    __syncthreads();
    if (active && rl == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          float sacc = 0.f;
          for (int q = 0; q < RL; ++q) sacc += red[((q * G + cl) * 3 + c) * CPC + j];
          part[((size_t)blockIdx.x * C + c0 + j) * TT + pr * 3 + c] = sacc;
        }
    }
  }
}


// depthwise weight gradient, 3x3x3, LDS-tiled: a workgroup stages the (transformed) input halo and the output
// gradient of TH rows x all W of one depth slice for WG_CH channel chunks ONCE, then thread (chunk, tap)
// accumulates its tap over the tile from LDS — every global byte is read once per tile instead of once per
// (kd,kh) pair.  Partials [tile][C][27] are summed in tile order by k_dwconv_wgrad_reduce.
static constexpr int WG_CH_MAX = 8;   // channel chunks per workgroup: 8, or 4 / 2 where a row of 8 does not fit the LDS budget (W >= 64)
template <typename T>
__global__ void __launch_bounds__(NT) k_dwconv3_wgrad_lds(const void* __restrict__ x, int64_t xs,
                                                          const float* __restrict__ in_stats, int act,
                                                          const void* __restrict__ dy, int64_t dys,
                                                          const float* __restrict__ dy_bias, float* __restrict__ part,
                                                          int N, int D, int H, int W, int C, int TH, int WG_CH) {
  constexpr int CPC = Elem<T>::CPC;
  CBIM_DYN_SMEM(smem);
  const int cch = C / CPC;
  const int g0 = blockIdx.y * WG_CH;
  const int G = cch - g0 < WG_CH ? cch - g0 : WG_CH;
  const int htiles = (H + TH - 1) / TH;
  const int tile = blockIdx.x;
  const int ht = tile % htiles, dz = (tile / htiles) % D, n = tile / (htiles * D);
  const int h0 = ht * TH;
  const int hW = W + 2, hH = TH + 2;
  // LDS: x_s [3][hH][hW][WG_CH] chunks of float[CPC] ; g_s [TH][W][WG_CH]
  float* x_s = (float*)smem;
  float* g_s = x_s + (size_t)3 * hH * hW * WG_CH * CPC;
  const int xitems = 3 * hH * hW * G, gitems = TH * W * G;
  // Staging with SB loads in flight per thread: the one-item-per-trip loop it replaces waited a full memory round trip per
  // item (13 items per thread on a 32^3 x 512-channel tensor: 26 us per workgroup at one workgroup per CU, 423 us for a
  // launch that moves 67 MB).  With G dividing the workgroup size a thread always stages the same channel chunk, so its
  // statistics / bias live in registers.
  constexpr int SB = 8;
  const bool fixed_cl = NT % G == 0;
  float mean[CPC], rstd[CPC], bias[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) { mean[j] = 0.f; rstd[j] = 1.f; bias[j] = 0.f; }
  if (fixed_cl) {
    const int c0 = (g0 + threadIdx.x % G) * CPC;
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      if (in_stats) { mean[j] = in_stats[((size_t)n * C + c0 + j) * 2]; rstd[j] = in_stats[((size_t)n * C + c0 + j) * 2 + 1]; }
      if (dy_bias) bias[j] = dy_bias[(size_t)n * C + c0 + j];
    }
  }
  for (int base = threadIdx.x; base < xitems; base += NT * SB) {
    u32x4 raw[SB];
    bool in[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int i = base + u * NT;
      raw[u] = u32x4{0u, 0u, 0u, 0u};
      in[u] = false;
      if (i < xitems) {
        const int cl = i % G, v = i / G;
        const int ww = v % hW - 1, hh = (v / hW) % hH - 1 + h0, dd = v / (hW * hH) - 1 + dz;
        in[u] = dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W;
        if (in[u]) raw[u] = ld_chunk<T>(x, ((((size_t)n * D + dd) * H + hh) * W + ww) * xs + (size_t)(g0 + cl) * CPC);
      }
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int i = base + u * NT;
      if (i < xitems) {
        const int cl = i % G, v = i / G;
        float f[CPC];
        Elem<T>::unpack(raw[u], f);
        if (in[u] && in_stats) {
          if (fixed_cl) {
#pragma unroll
            for (int j = 0; j < CPC; ++j) f[j] = act_fwd((f[j] - mean[j]) * rstd[j], act);
          } else {
            const int c0 = (g0 + cl) * CPC;
#pragma unroll
            for (int j = 0; j < CPC; ++j)
              f[j] = act_fwd((f[j] - in_stats[((size_t)n * C + c0 + j) * 2]) * in_stats[((size_t)n * C + c0 + j) * 2 + 1], act);
          }
        }
#pragma unroll
        for (int j = 0; j < CPC; ++j) x_s[((size_t)v * WG_CH + cl) * CPC + j] = f[j];
      }
    }
  }
  for (int base = threadIdx.x; base < gitems; base += NT * SB) {
    u32x4 raw[SB];
    bool in[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int i = base + u * NT;
      raw[u] = u32x4{0u, 0u, 0u, 0u};
      in[u] = false;
      if (i < gitems) {
        const int cl = i % G, v = i / G;
        const int ww = v % W, hh = v / W + h0;
        in[u] = hh < H;
        if (in[u]) raw[u] = ld_chunk<T>(dy, ((((size_t)n * D + dz) * H + hh) * W + ww) * dys + (size_t)(g0 + cl) * CPC);
      }
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int i = base + u * NT;
      if (i < gitems) {
        const int cl = i % G, v = i / G;
        float f[CPC];
        Elem<T>::unpack(raw[u], f);
        if (in[u] && dy_bias) {
          if (fixed_cl) {
#pragma unroll
            for (int j = 0; j < CPC; ++j) f[j] += bias[j];
          } else {
            const int c0 = (g0 + cl) * CPC;
#pragma unroll
            for (int j = 0; j < CPC; ++j) f[j] += dy_bias[(size_t)n * C + c0 + j];
          }
        }
#pragma unroll
        for (int j = 0; j < CPC; ++j) g_s[((size_t)v * WG_CH + cl) * CPC + j] = f[j];
      }
    }
  }
  __syncthreads();
  // thread = (channel chunk, (kd,kh) pair, row third): the three kw taps of a pair share every dy value and slide
  // over the same x row — one x chunk and one dy chunk are read from LDS per 24 multiply-adds (one read pair per 8
  // before: the kernel was LDS-bandwidth bound at 253 GB/s of HBM traffic); the three row thirds are added at the end
  const int cl = threadIdx.x / 27, rem = threadIdx.x % 27;
  const int ab = rem / 3, rs = rem % 3;
  const int ka = ab / 3, kb = ab % 3;
  const bool active = cl < G;
  // packed f32 pairs (v_pk_fma_f32): 12 instead of 24 multiply-add instructions per LDS read pair
  constexpr int NP2 = CPC / 2;
  dw_f2 acc2[3][NP2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < NP2; ++j) acc2[c][j] = dw_f2{0.f, 0.f};
  if (active) {
    for (int r = rs; r < TH; r += 3) {
      const dw_f2* xr = (const dw_f2*)(x_s + ((size_t)((ka * hH + r + kb) * hW) * WG_CH + cl) * CPC);
      const dw_f2* gr = (const dw_f2*)(g_s + ((size_t)(r * W) * WG_CH + cl) * CPC);
      const int RS = WG_CH * CPC / 2;                     // pairs per LDS voxel row
      dw_f2 x0[NP2], x1[NP2];
#pragma unroll
      for (int j = 0; j < NP2; ++j) { x0[j] = xr[j]; x1[j] = xr[RS + j]; }
      for (int w = 0; w < W; ++w) {
        dw_f2 x2[NP2], g[NP2];
#pragma unroll
        for (int j = 0; j < NP2; ++j) {
          x2[j] = xr[(size_t)(w + 2) * RS + j];
          g[j] = gr[(size_t)w * RS + j];
        }
#pragma unroll
        for (int j = 0; j < NP2; ++j) {
          acc2[0][j] = __builtin_elementwise_fma(x0[j], g[j], acc2[0][j]);
          acc2[1][j] = __builtin_elementwise_fma(x1[j], g[j], acc2[1][j]);
          acc2[2][j] = __builtin_elementwise_fma(x2[j], g[j], acc2[2][j]);
          x0[j] = x1[j];
          x1[j] = x2[j];
        }
      }
    }
  }
  float acc[3][CPC];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < NP2; ++j) { acc[c][2 * j] = acc2[c][j].x; acc[c][2 * j + 1] = acc2[c][j].y; }
  __syncthreads();                       // x_s is dead: reuse it for the row-third partials [NT][3][CPC]
  float* red = x_s;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < CPC; ++j) red[((size_t)threadIdx.x * 3 + c) * CPC + j] = acc[c][j];
  __syncthreads();
  if (active && rs == 0) {
    const int c0 = (g0 + cl) * CPC;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        const float v = (red[((size_t)threadIdx.x * 3 + c) * CPC + j] + red[((size_t)(threadIdx.x + 1) * 3 + c) * CPC + j]) +
                        red[((size_t)(threadIdx.x + 2) * 3 + c) * CPC + j];
        part[((size_t)tile * C + c0 + j) * 27 + ab * 3 + c] = v;
      }
  }
}

__global__ void __launch_bounds__(NT) k_dwconv_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw,
                                                            int nblk, int CT) {
  int i = blockIdx.x * NT + threadIdx.x;
  if (i >= CT) return;
  // 8 independent chains, fixed association (one chain = one memory latency per slab)
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 7 < nblk; b += 8)
#pragma unroll
    for (int q = 0; q < 8; ++q) s8[q] += part[(size_t)(b + q) * CT + i];
  for (; b < nblk; ++b) s8[0] += part[(size_t)b * CT + i];
  dw[i] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
}

// ---- PatchMerging gather ---------------------------------------------------------------------------
// merged[n,d',h',w', ((i*sH+j)*sW+k)*C + c] = x[n, d'*sD+i, h'*sH+j, w'*sW+k, c]; inverse=1 copies back.
template <typename T>
__global__ void __launch_bounds__(NT) k_space_to_depth(const void* __restrict__ src, void* __restrict__ dst,
                                                       int D, int H, int W, int C, int sD, int sH, int sW,
                                                       int inverse, int64_t total, int64_t xs) {   // xs: row stride of the fine tensor
  constexpr int CPC = Elem<T>::CPC;
  const int Do = D / sD, Ho = H / sH, Wo = W / sW;
  const int Cm = C * sD * sH * sW, mch = Cm / CPC;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int mc = (int)(i % mch);
    int64_t r = i / mch;
    int wo = (int)(r % Wo); r /= Wo;
    int ho = (int)(r % Ho); r /= Ho;
    int dz = (int)(r % Do);
    int64_t n = r / Do;
    int cm = mc * CPC, oct = cm / C, c = cm % C;
    int kk = oct % sW, jj = (oct / sW) % sH, ii = oct / (sW * sH);
    size_t xrow = (((size_t)n * D + dz * sD + ii) * H + ho * sH + jj) * W + wo * sW + kk;
    size_t mrow = (((size_t)n * Do + dz) * Ho + ho) * Wo + wo;
    if (inverse)
      st_chunk<T>(dst, xrow * xs + c, ld_chunk<T>(src, mrow * Cm + cm));
    else
      st_chunk<T>(dst, mrow * Cm + cm, ld_chunk<T>(src, xrow * xs + c));
  }
}

// ---- bidirectional attention ------------------------------------------------------------------------
// part record per (n, h, block, j): [colmax, colsum, acc[DH]]
template <typename T, int DH>
__global__ void __launch_bounds__(AT) k_attn_fwd(const void* __restrict__ qv, int64_t rs,
                                                 const float* __restrict__ mq, const float* __restrict__ mv,
                                                 void* __restrict__ fo, float* __restrict__ part, int L, int heads,
                                                 int M, float scale, int nblk) {
  __shared__ float E_s[AT][MM + 1];
  __shared__ float v_s[AT][DH + 1];
  __shared__ float red[2][MM];
  __shared__ float colm[MM];
  const int t = threadIdx.x, blk = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int inner = heads * DH;
  const int l = blk * AT + t;
  const bool valid = l < L;
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs;
  float q[DH], v[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = valid ? Elem<T>::load1(qv, row + d * heads + h) : 0.f;
    v[d] = valid ? Elem<T>::load1(qv, row + inner + d * heads + h) : 0.f;
  }
  const float* mqh = mq + (size_t)n * M * inner + h;
  const float* mvh = mv + (size_t)n * M * inner + h;
  float a[MM];
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    float s = -INFINITY;
    if (j < M) {
      s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], mqh[(size_t)j * inner + d * heads], s);   // -ffp-contract=off build
      s *= scale;
    }
    a[j] = s;
  }
  // feature side: softmax over the M codes of this voxel
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MM; ++j) mx = fmaxf(mx, a[j]);
  float sum = 0.f, o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    if (j < M) {
      float e = expf(a[j] - mx);
      sum += e;
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(e, mvh[(size_t)j * inner + d * heads], o[d]);
    }
  }
  if (valid) {
    float inv = 1.f / sum;
    size_t orow = ((size_t)n * L + l) * inner;
#pragma unroll
    for (int d = 0; d < DH; ++d) Elem<T>::store1(fo, orow + d * heads + h, o[d] * inv);
  }
  // map side: online-softmax record of this block's AT voxels per code
#pragma unroll
  for (int j = 0; j < MM; ++j) E_s[t][j] = valid ? a[j] : -INFINITY;
#pragma unroll
  for (int d = 0; d < DH; ++d) v_s[t][d] = v[d];
  __syncthreads();
  {
    int j = t & 63, hf = t >> 6;
    float m = -INFINITY;
    for (int r = 0; r < 64; ++r) m = fmaxf(m, E_s[hf * 64 + r][j]);
    red[hf][j] = m;
  }
  __syncthreads();
  if (t < MM) colm[t] = fmaxf(red[0][t], red[1][t]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < MM; ++j) E_s[t][j] = (valid && j < M) ? expf(a[j] - colm[j]) : 0.f;
  __syncthreads();
  {
    constexpr int HD = DH / 2;
    int j = t >> 1, d0 = (t & 1) * HD;
    float acc[HD], s = 0.f;
#pragma unroll
    for (int k = 0; k < HD; ++k) acc[k] = 0.f;
    for (int r = 0; r < AT; ++r) {
      float e = E_s[r][j];
      s += e;
#pragma unroll
      for (int k = 0; k < HD; ++k) acc[k] = fmaf(e, v_s[r][d0 + k], acc[k]);
    }
    if (j < M) {
      float* p = part + ((((size_t)n * heads + h) * nblk + blk) * M + j) * (DH + 2);
      if ((t & 1) == 0) { p[0] = colm[j]; p[1] = s; }
#pragma unroll
      for (int k = 0; k < HD; ++k) p[2 + d0 + k] = acc[k];
    }
  }
}

// merge the per-block records: map_out[n][j][d*heads+h], colstat[n][h][j] = (max, sum)
__global__ void __launch_bounds__(256) k_attn_merge(const float* __restrict__ part, float* __restrict__ map_out,
                                                    float* __restrict__ colstat, int heads, int M, int DH, int nblk) {
  // one 256-thread block per (n, h, code j): thread (g, d) = (record group, output channel).  The 8 groups walk
  // interleaved records (the matrix-core path writes one record per 32 voxels: 1024 of them at 32^3), partial
  // results are combined through LDS in group order — deterministic.
  __shared__ float mx_s[8], S_s[8], A_s[8][33];
  const int j = blockIdx.x % M, h = (blockIdx.x / M) % heads, n = blockIdx.x / (M * heads);
  const int inner = heads * DH;
  const int d = threadIdx.x & 31, g = threadIdx.x >> 5;
  const float* base = part + ((size_t)n * heads + h) * nblk * M * (DH + 2);
  float mx = -INFINITY;
  for (int b = g; b < nblk; b += 8) mx = fmaxf(mx, base[((size_t)b * M + j) * (DH + 2)]);
  if (d == 0) mx_s[g] = mx;
  __syncthreads();
  mx = mx_s[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) mx = fmaxf(mx, mx_s[i]);
  float S = 0.f, A = 0.f;
  for (int b = g; b < nblk; b += 8) {
    const float* p = base + ((size_t)b * M + j) * (DH + 2);
    float sc = expf(p[0] - mx);
    S += p[1] * sc;
    if (d < DH) A += p[2 + d] * sc;
  }
  if (d == 0) S_s[g] = S;
  A_s[g][d] = A;
  __syncthreads();
  if (g == 0) {
    S = S_s[0]; A = A_s[0][d];
#pragma unroll
    for (int i = 1; i < 8; ++i) { S += S_s[i]; A += A_s[i][d]; }
    if (d < DH) map_out[((size_t)n * M + j) * inner + d * heads + h] = A / S;
    if (d == 0) {
      colstat[(((size_t)n * heads + h) * M + j) * 2] = mx;
      colstat[(((size_t)n * heads + h) * M + j) * 2 + 1] = S;
    }
  }
}

// backward: part record per (n, h, block): [2][M][DH] = (dmv partial, dmq partial)
template <typename T, int DH>
__global__ void __launch_bounds__(AT) k_attn_bwd(const void* __restrict__ qv, int64_t rs,
                                                 const float* __restrict__ mq, const float* __restrict__ mv,
                                                 const float* __restrict__ colstat, const float* __restrict__ map_out,
                                                 const void* __restrict__ dfo, const float* __restrict__ dmo,
                                                 void* __restrict__ dqv, float* __restrict__ part, int L, int heads,
                                                 int M, float scale, int nblk) {
  __shared__ float A_s[AT][MM + 1];
  __shared__ float B_s[AT][DH + 1];
  __shared__ float cj[MM], cM[MM], cIS[MM];
  const int t = threadIdx.x, blk = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int inner = heads * DH;
  const int l = blk * AT + t;
  const bool valid = l < L;
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs;
  const size_t grow = ((size_t)n * L + (valid ? l : 0)) * inner;
  float q[DH], v[DH], g[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = valid ? Elem<T>::load1(qv, row + d * heads + h) : 0.f;
    v[d] = valid ? Elem<T>::load1(qv, row + inner + d * heads + h) : 0.f;
    g[d] = valid ? Elem<T>::load1(dfo, grow + d * heads + h) : 0.f;
  }
  const float* mqh = mq + (size_t)n * M * inner + h;
  const float* mvh = mv + (size_t)n * M * inner + h;
  const float* dmh = dmo + (size_t)n * M * inner + h;
  const float* moh = map_out + (size_t)n * M * inner + h;
  if (t < MM) {
    float c = 0.f, m = 0.f, is = 0.f;
    if (t < M) {
      for (int d = 0; d < DH; ++d) c += moh[(size_t)t * inner + d * heads] * dmh[(size_t)t * inner + d * heads];
      m = colstat[(((size_t)n * heads + h) * M + t) * 2];
      is = 1.f / colstat[(((size_t)n * heads + h) * M + t) * 2 + 1];
    }
    cj[t] = c; cM[t] = m; cIS[t] = is;
  }
  __syncthreads();
  float a[MM], dA[MM], dv[DH], dq[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { dv[d] = 0.f; dq[d] = 0.f; }
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    float s = -INFINITY, da = 0.f;
    if (j < M) {
      s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], mqh[(size_t)j * inner + d * heads], s);
      s *= scale;
      // map side: P2 = exp(a - colmax)/colsum; dA2 = P2*(v.dmo_j - <map_out_j, dmo_j>)
      float p2 = expf(s - cM[j]) * cIS[j], dp2 = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        float w = dmh[(size_t)j * inner + d * heads];
        dp2 = fmaf(v[d], w, dp2);
        dv[d] = fmaf(p2, w, dv[d]);
      }
      da = p2 * (dp2 - cj[j]);
    }
    a[j] = s;
    dA[j] = da;
  }
  // feature side: P1 = row softmax; dA1 = P1*(g.mv_j - sum_j P1 g.mv_j)
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MM; ++j) mx = fmaxf(mx, a[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MM; ++j) { a[j] = j < M ? expf(a[j] - mx) : 0.f; sum += a[j]; }
  float inv = 1.f / sum, rr = 0.f;
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    a[j] *= inv;
    float dp = 0.f;
    if (j < M) {
#pragma unroll
      for (int d = 0; d < DH; ++d) dp = fmaf(g[d], mvh[(size_t)j * inner + d * heads], dp);
    }
    A_s[t][j] = dp;  // own row only
    rr += a[j] * dp;
  }
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    dA[j] += a[j] * (A_s[t][j] - rr);
    if (j < M) {
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(dA[j], mqh[(size_t)j * inner + d * heads], dq[d]);
    }
  }
  if (valid) {
    size_t drow = ((size_t)n * L + l) * (2 * inner);
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      Elem<T>::store1(dqv, drow + d * heads + h, dq[d] * scale);
      Elem<T>::store1(dqv, drow + inner + d * heads + h, dv[d]);
    }
  }
  float* pbase = part + (((size_t)n * heads + h) * nblk + blk) * 2 * M * DH;
  constexpr int HD = DH / 2;
  for (int stage = 0; stage < 2; ++stage) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MM; ++j) A_s[t][j] = stage == 0 ? a[j] : dA[j] * scale;
#pragma unroll
    for (int d = 0; d < DH; ++d) B_s[t][d] = stage == 0 ? g[d] : q[d];
    __syncthreads();
    int j = t >> 1, d0 = (t & 1) * HD;
    float acc[HD];
#pragma unroll
    for (int k = 0; k < HD; ++k) acc[k] = 0.f;
    for (int r = 0; r < AT; ++r) {
      float e = A_s[r][j];
#pragma unroll
      for (int k = 0; k < HD; ++k) acc[k] = fmaf(e, B_s[r][d0 + k], acc[k]);
    }
    if (j < M) {
#pragma unroll
      for (int k = 0; k < HD; ++k) pbase[((size_t)stage * M + j) * DH + d0 + k] = acc[k];
    }
  }
}

__global__ void __launch_bounds__(NT) k_attn_bwd_reduce(const float* __restrict__ part, float* __restrict__ dmq,
                                                        float* __restrict__ dmv, int heads, int M, int DH, int nblk) {
  const int h = blockIdx.x % heads, n = blockIdx.x / heads;
  const int inner = heads * DH;
  const float* base = part + ((size_t)n * heads + h) * nblk * 2 * M * DH;
  for (int idx = blockIdx.y * NT + threadIdx.x; idx < 2 * M * DH; idx += gridDim.y * NT) {
    int st = idx / (M * DH), j = (idx / DH) % M, d = idx % DH;
    // 8 independent chains keep 8 loads in flight (a single chain pays the full memory latency per record);
    // fixed association -> deterministic
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 7 < nblk; b += 8)
#pragma unroll
      for (int i = 0; i < 8; ++i) s8[i] += base[(size_t)(b + i) * 2 * M * DH + idx];
    for (; b < nblk; ++b) s8[0] += base[(size_t)b * 2 * M * DH + idx];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    (st == 0 ? dmv : dmq)[((size_t)n * M + j) * inner + d * heads + h] = s;
  }
}

// ---- semantic map generation: column softmax over L + pooling ---------------------------------------
// fw rows = [feat (C) | weight logits (M)]; part record per (n, block): [M][2 + C]
template <typename T>
// blockIdx.y = (code group, channel group of `cw` channels): low-resolution stages have a handful of row blocks (8^3: two), the
// channel groups give the chip something to do (each recomputes the 128 x 64 exponentials of its rows: cheap)
__global__ void __launch_bounds__(AT) k_mappool_fwd(const void* __restrict__ fw, int64_t rs, float* __restrict__ part,
                                                    int L, int C, int M, int nblk, int cw) {
  __shared__ float E_s[AT][MM + 1];
  __shared__ float f_s[AT][33];
  __shared__ float red[2][MM];
  __shared__ float colm[MM];
  const int t = threadIdx.x, blk = blockIdx.x, n = blockIdx.z;
  const int n_jg = (M + MM - 1) / MM;
  const int j0 = (int)(blockIdx.y % (unsigned)n_jg) * MM;   // code group (column softmaxes of different codes are independent)
  const int cg = (int)(blockIdx.y / (unsigned)n_jg);
  const int c_lo = cg * cw, c_hi = c_lo + cw < C ? c_lo + cw : C;
  const int l = blk * AT + t;
  const bool valid = l < L;
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs;
  for (int j = 0; j < MM; ++j) E_s[t][j] = (valid && j0 + j < M) ? Elem<T>::load1(fw, row + C + j0 + j) : -INFINITY;
  __syncthreads();
  {
    int j = t & 63, hf = t >> 6;
    float m = -INFINITY;
    for (int r = 0; r < 64; ++r) m = fmaxf(m, E_s[hf * 64 + r][j]);
    red[hf][j] = m;
  }
  __syncthreads();
  if (t < MM) colm[t] = fmaxf(red[0][t], red[1][t]);
  __syncthreads();
  for (int j = 0; j < MM; ++j) E_s[t][j] = (valid && j0 + j < M) ? expf(E_s[t][j] - colm[j]) : 0.f;
  __syncthreads();
  float* pb = part + ((size_t)n * nblk + blk) * M * (2 + C);
  {
    int j = t & 63, hf = t >> 6;
    float s = 0.f;
    for (int r = 0; r < 64; ++r) s += E_s[hf * 64 + r][j];
    red[hf][j] = s;
  }
  __syncthreads();
  if (cg == 0 && t < MM && j0 + t < M) { pb[(size_t)(j0 + t) * (2 + C)] = colm[t]; pb[(size_t)(j0 + t) * (2 + C) + 1] = red[0][t] + red[1][t]; }
  const int jq = t & 15, cq = t >> 4;  // 4 codes x 4 channels per thread
  for (int c0 = c_lo; c0 < c_hi; c0 += 32) {
    __syncthreads();
    for (int k = 0; k < 32; ++k) f_s[t][k] = (valid && c0 + k < C) ? Elem<T>::load1(fw, row + c0 + k) : 0.f;
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) acc[u][w] = 0.f;
    for (int r = 0; r < AT; ++r) {
      float e[4], f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { e[u] = E_s[r][jq * 4 + u]; f[u] = f_s[r][cq * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[u][w] = fmaf(e[u], f[w], acc[u][w]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int j = j0 + jq * 4 + u, c = c0 + cq * 4 + w;
        if (j < M && c < C) pb[(size_t)j * (2 + C) + 2 + c] = acc[u][w];
      }
  }
}

// map[n][c][j], colstat[n][j] = (max, sum).  One workgroup per (code j, image): the column maximum and the rescale factors
// exp(max_b - max) of the nblk partial records are formed by the threads in parallel (fixed tree), kept in LDS, and thread c then
// walks the records with coalesced reads of p[2 + c], eight loads in flight.  (The first form gave one thread one (c, j) pair
// and walked the records twice with a strided load -> expf -> add chain: 208 us for 128 records.)
static constexpr int MRG_MAXB = 8192;
__global__ void __launch_bounds__(NT) k_mappool_merge(const float* __restrict__ part, float* __restrict__ map,
                                                      float* __restrict__ colstat, int C, int M, int nblk) {
  __shared__ float sc_s[MRG_MAXB];
  __shared__ float red[NT];
  const int j = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const float* base = part + ((size_t)n * nblk * M + j) * (2 + C);
  const size_t bstride = (size_t)M * (2 + C);
  float m = -INFINITY;
  for (int b = t; b < nblk; b += NT) m = fmaxf(m, base[(size_t)b * bstride]);
  red[t] = m;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (t < s) red[t] = fmaxf(red[t], red[t + s]);
    __syncthreads();
  }
  const float mx = red[0];
  __syncthreads();
  float S = 0.f;
  for (int b = t; b < nblk; b += NT) {
    const float sc = expf(base[(size_t)b * bstride] - mx);
    sc_s[b] = sc;
    S += base[(size_t)b * bstride + 1] * sc;
  }
  red[t] = S;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  S = red[0];
  if (t == 0) { colstat[((size_t)n * M + j) * 2] = mx; colstat[((size_t)n * M + j) * 2 + 1] = S; }
  for (int c = t; c < C; c += NT) {
    float A = 0.f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(b + u) * bstride + 2 + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) A = fmaf(v[u], sc_s[b + u], A);
    }
    for (; b < nblk; ++b) A = fmaf(base[(size_t)b * bstride + 2 + c], sc_s[b], A);
    map[((size_t)n * C + c) * M + j] = A / S;
  }
}

// dfeat[l,c] = sum_j P[l,j] dmap[c,j];  dlogit[l,j] = P[l,j] (sum_c feat[l,c] dmap[c,j] - <map_j, dmap_j>)
template <typename T>
__global__ void __launch_bounds__(AT) k_mappool_bwd(const void* __restrict__ fw, int64_t rs,
                                                    const float* __restrict__ map, const float* __restrict__ colstat,
                                                    const float* __restrict__ dmap, void* __restrict__ dfw, int64_t drs,
                                                    int L, int C, int M) {
  __shared__ float cj[MM];
  const int t = threadIdx.x, n = blockIdx.z;
  const int l = blockIdx.x * AT + t;
  const bool valid = l < L;
  if (t < MM) {
    float c = 0.f;
    if (t < M)
      for (int k = 0; k < C; ++k) c += map[((size_t)n * C + k) * M + t] * dmap[((size_t)n * C + k) * M + t];
    cj[t] = c;
  }
  __syncthreads();
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs, drow = ((size_t)n * L + (valid ? l : 0)) * drs;
  float P[MM], tt[MM];
  constexpr int CPC = Elem<T>::CPC;
  // whole 16-byte channel chunks in and out (2-byte element stores are read-modify-write in the memory system and
  // made this kernel 5x slower); element-wise only for code counts that are not a multiple of the chunk
  const bool chunked = (M % CPC) == 0 && (C % CPC) == 0 && (rs % CPC) == 0 && (drs % CPC) == 0;
  if (chunked) {
#pragma unroll
    for (int j0 = 0; j0 < MM; j0 += CPC) {
      float f[CPC];
      if (j0 < M) Elem<T>::unpack(ld_chunk<T>(fw, row + C + j0), f);
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        P[j0 + j] = j0 < M ? expf(f[j] - colstat[((size_t)n * M + j0 + j) * 2]) / colstat[((size_t)n * M + j0 + j) * 2 + 1] : 0.f;
        tt[j0 + j] = 0.f;
      }
    }
    if (valid) {
      for (int c0 = 0; c0 < C; c0 += CPC) {
        float f[CPC], g[CPC];
        Elem<T>::unpack(ld_chunk<T>(fw, row + c0), f);
#pragma unroll
        for (int k = 0; k < CPC; ++k) {
          const float* dm = dmap + ((size_t)n * C + c0 + k) * M;
          float gsum = 0.f;
#pragma unroll
          for (int j = 0; j < MM; ++j)
            if (j < M) { tt[j] = fmaf(f[k], dm[j], tt[j]); gsum = fmaf(P[j], dm[j], gsum); }
          g[k] = gsum;
        }
        st_chunk<T>(dfw, drow + c0, Elem<T>::pack(g));
      }
    }
    if (!valid) return;
#pragma unroll
    for (int j0 = 0; j0 < MM; j0 += CPC)
      if (j0 < M) {
        float g[CPC];
#pragma unroll
        for (int j = 0; j < CPC; ++j) g[j] = P[j0 + j] * (tt[j0 + j] - cj[j0 + j]);
        st_chunk<T>(dfw, drow + C + j0, Elem<T>::pack(g));
      }
    return;
  }
  if (!valid) return;
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    P[j] = j < M ? expf(Elem<T>::load1(fw, row + C + j) - colstat[((size_t)n * M + j) * 2]) /
                       colstat[((size_t)n * M + j) * 2 + 1]
                 : 0.f;
    tt[j] = 0.f;
  }
  for (int c = 0; c < C; ++c) {
    float f = Elem<T>::load1(fw, row + c), gsum = 0.f;
    const float* dm = dmap + ((size_t)n * C + c) * M;
#pragma unroll
    for (int j = 0; j < MM; ++j)
      if (j < M) { tt[j] = fmaf(f, dm[j], tt[j]); gsum = fmaf(P[j], dm[j], gsum); }
    Elem<T>::store1(dfw, drow + c, gsum);
  }
#pragma unroll
  for (int j = 0; j < MM; ++j)
    if (j < M) Elem<T>::store1(dfw, drow + C + j, P[j] * (tt[j] - cj[j]));
}


// Same contract as k_mappool_bwd for the all-chunked case (M, C and both row strides multiples of the 16-byte chunk,
// M <= 64), with FOUR waves per 64 voxels: wave w takes the channel chunks c0 = (w + 4i)*CPC — dfeat chunks are
// independent, dmap rows stay wave-uniform (scalar loads) — and holds a partial tt over its channels; the partials
// meet once in LDS and thread (w, voxel) finishes codes 16w..16w+15.  k_mappool_bwd had one thread per voxel: 512 waves
// for the 32^3 level of the AMOS configuration, two SIMDs of every CU idle and nothing to hide latency with
// (729 us per call, ~3 % of the vector-ALU peak).
static constexpr int MP4_T = 256;
template <typename T>
__global__ void __launch_bounds__(MP4_T) k_mappool_bwd4(const void* __restrict__ fw, int64_t rs,
                                                        const float* __restrict__ map, const float* __restrict__ colstat,
                                                        const float* __restrict__ dmap, void* __restrict__ dfw, int64_t drs,
                                                        int L, int C, int M) {
  CBIM_DYN_SMEM(smem);
  float* TTs = (float*)smem;                 // [4][64][MM + 1]
  float* cj = TTs + 4 * 64 * (MM + 1);       // [MM]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, n = blockIdx.z;
  const int l = blockIdx.x * 64 + lane;
  const bool valid = l < L;
  {
    // cj[j] = sum_k map[k][j] dmap[k][j]: thread (code j = lane, quarter w of the channels), eight products in flight, the four
    // quarters added in wave order.  (64 threads walking all C channels with a dependent load -> fma chain were 260 of the
    // kernel's 340 us at C = 256.)
    float c = 0.f;
    if (lane < M) {
      const float* mp_ = map + (size_t)n * C * M + lane;
      const float* dp_ = dmap + (size_t)n * C * M + lane;
      int k = w;
      for (; k + 28 < C; k += 32) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = mp_[(size_t)(k + 4 * u) * M]; b[u] = dp_[(size_t)(k + 4 * u) * M]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) c = fmaf(a[u], b[u], c);
      }
      for (; k < C; k += 4) c = fmaf(mp_[(size_t)k * M], dp_[(size_t)k * M], c);
    }
    TTs[w * 64 + lane] = c;
    __syncthreads();
    if (t < MM) cj[t] = ((TTs[t] + TTs[64 + t]) + TTs[128 + t]) + TTs[192 + t];
    __syncthreads();       // TTs is reused for the tt partials below
  }
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs, drow = ((size_t)n * L + (valid ? l : 0)) * drs;
  constexpr int CPC = Elem<T>::CPC;
  float P[MM], tt[MM];
#pragma unroll
  for (int j0 = 0; j0 < MM; j0 += CPC) {
    float f[CPC];
    if (j0 < M) Elem<T>::unpack(ld_chunk<T>(fw, row + C + j0), f);
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      P[j0 + j] = j0 < M ? expf(f[j] - colstat[((size_t)n * M + j0 + j) * 2]) / colstat[((size_t)n * M + j0 + j) * 2 + 1] : 0.f;
      tt[j0 + j] = 0.f;
    }
  }
  if (valid) {
    // the wave index as a SCALAR: the dmap row pointer is then provably wave-uniform and its 64 values arrive by scalar loads
    // (as a function of threadIdx the compiler issued 64 vector loads of one address each per channel: 4096 per thread at
    // C = 256, most of the kernel's time)
#ifdef CBIM_EMU
    const int wu = w;
#else
    const int wu = __builtin_amdgcn_readfirstlane(w);
#endif
    for (int c0 = wu * CPC; c0 < C; c0 += 4 * CPC) {
      float f[CPC], g[CPC];
      Elem<T>::unpack(ld_chunk<T>(fw, row + c0), f);
#pragma unroll
      for (int k = 0; k < CPC; ++k) {
        const float* dm = dmap + ((size_t)n * C + c0 + k) * M;
        float gsum = 0.f;
#pragma unroll
        for (int j = 0; j < MM; ++j)
          if (j < M) { tt[j] = fmaf(f[k], dm[j], tt[j]); gsum = fmaf(P[j], dm[j], gsum); }
        g[k] = gsum;
      }
      st_chunk<T>(dfw, drow + c0, Elem<T>::pack(g));
    }
  }
  float* mine = TTs + (size_t)(w * 64 + lane) * (MM + 1);
#pragma unroll
  for (int j = 0; j < MM; ++j) mine[j] = tt[j];
  __syncthreads();
  if (!valid) return;
#pragma unroll
  for (int jj = 0; jj < 16; jj += CPC) {
    const int j0 = 16 * w + jj;
    if (j0 < M) {
      float f[CPC], g[CPC];
      Elem<T>::unpack(ld_chunk<T>(fw, row + C + j0), f);
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        const float p = expf(f[j] - colstat[((size_t)n * M + j0 + j) * 2]) / colstat[((size_t)n * M + j0 + j) * 2 + 1];
        const float* col = TTs + (size_t)lane * (MM + 1) + j0 + j;
        const float s = ((col[0] + col[64 * (MM + 1)]) + col[2 * 64 * (MM + 1)]) + col[3 * 64 * (MM + 1)];   // wave order
        g[j] = p * (s - cj[j0 + j]);
      }
      st_chunk<T>(dfw, drow + C + j0, Elem<T>::pack(g));
    }
  }
}

}  // namespace cbim

using namespace cbim;

#define DISPATCH_T(dtype, KERNEL, grid, st, ...)                                        \
  do {                                                                                  \
    if ((dtype) == CBIM_BF16)                                                           \
      CBIM_LAUNCH((KERNEL<bf16_tag>), grid, dim3(NT), 0, st, __VA_ARGS__);              \
    else                                                                                \
      CBIM_LAUNCH((KERNEL<float>), grid, dim3(NT), 0, st, __VA_ARGS__);                 \
  } while (0)

static int launch_ok(const char* what) {
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "%s launch: %s", what, hipGetErrorString(e));
  return CBIM_OK;
}

static int check_c(int dtype, int C, const char* what) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(C > 0 && C % cpc == 0, CBIM_EUNSUPPORTED, "%s channel count %d is not a multiple of %d", what, C, cpc);
  return 0;
}

static int check_k(int kD, int kH, int kW) {
  CBIM_CHECK(kD >= 1 && kH >= 1 && kW >= 1 && (kD & 1) && (kH & 1) && (kW & 1) && kD * kH * kW <= NT, CBIM_EUNSUPPORTED,
             "depthwise kernel %dx%dx%d unsupported (odd extents, <= %d taps)", kD, kH, kW, NT);
  return 0;
}

static int g_dw_lds = 1;
/* process-wide switch (tests, A/B): 0 = the streaming depthwise kernel also where the LDS-tiled one applies; < 0 only queries;
 * returns the old value */
extern "C" int cbim_dwconv_lds_enable(int on) {
  const int old = g_dw_lds;
  if (on >= 0) g_dw_lds = on ? 1 : 0;
  return old;
}

extern "C" int cbim_dwconv3d(int dtype, const void* x, int64_t x_stride, const float* in_stats, int act,
                             const float* bias, const float* w, int flip, void* y, int64_t y_stride, int N, int D,
                             int H, int W, int C, int kD, int kH, int kW, void* stream) {
  if (int e = check_c(dtype, C, "dwconv")) return e;
  if (int e = check_k(kD, kH, kW)) return e;
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  if (kW == 3 && kD <= 3 && kH <= 3) {
    int cch = C / cpc, groups = (cch + DG - 1) / DG, G = cch < DG ? cch : DG, SL = NT / G;
    int64_t items = (int64_t)N * D * H * ((W + WT - 1) / WT);
    CBIM_CHECK(items < ((int64_t)1 << 31) && x_stride * 4 < ((int64_t)1 << 28), CBIM_EUNSUPPORTED,
               "dwconv3d: %lld strips / row stride %lld exceed the 32-bit index arithmetic", (long long)items, (long long)x_stride);
    int64_t bx = (items + SL - 1) / SL;
    int64_t cap = 4096 / groups > 1 ? 4096 / groups : 1;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    const int mode = !in_stats ? 0 : act == CBIM_ACT_RELU ? 1 : act == CBIM_ACT_NONE ? 2 : 3;
    static const int lds_on = 1;
    if (lds_on && g_dw_lds) {      // LDS-tiled form (round 4): one sweep of loads, the input transformed once
      const int tiles_d = (D + LTD - 1) / LTD, tiles_h = (H + LTH - 1) / LTH, tiles_w = (W + LTW - 1) / LTW;
      const int64_t tiles = (int64_t)N * tiles_d * tiles_h * tiles_w;
      const int lgroups = (cch + LG - 1) / LG;
      if (tiles < ((int64_t)1 << 31) && lgroups <= 65535) {
#define DW3L_LAUNCH(TT, MM) CBIM_LAUNCH((k_dwconv3_lds<TT, MM>), dim3((unsigned)tiles, (unsigned)lgroups), dim3(NT), 0, (hipStream_t)stream, x, x_stride, \
                                        in_stats, act, bias, w, flip, y, y_stride, N, D, H, W, C, kD, kH, tiles_d, tiles_h, tiles_w)
        if (dtype == CBIM_BF16) { if (mode == 0) DW3L_LAUNCH(bf16_tag, 0); else if (mode == 1) DW3L_LAUNCH(bf16_tag, 1); else if (mode == 2) DW3L_LAUNCH(bf16_tag, 2); else DW3L_LAUNCH(bf16_tag, 3); }
        else { if (mode == 0) DW3L_LAUNCH(float, 0); else if (mode == 1) DW3L_LAUNCH(float, 1); else if (mode == 2) DW3L_LAUNCH(float, 2); else DW3L_LAUNCH(float, 3); }
#undef DW3L_LAUNCH
        return launch_ok("dwconv3d (LDS)");
      }
    }
#define DW3_LAUNCH(TT, MM) CBIM_LAUNCH((k_dwconv3<TT, MM>), dim3((unsigned)bx, groups), dim3(NT), 0, (hipStream_t)stream, x, x_stride, in_stats, act, \
                                       bias, w, flip, y, y_stride, N, D, H, W, C, kD, kH)
    if (dtype == CBIM_BF16) { if (mode == 0) DW3_LAUNCH(bf16_tag, 0); else if (mode == 1) DW3_LAUNCH(bf16_tag, 1); else if (mode == 2) DW3_LAUNCH(bf16_tag, 2); else DW3_LAUNCH(bf16_tag, 3); }
    else { if (mode == 0) DW3_LAUNCH(float, 0); else if (mode == 1) DW3_LAUNCH(float, 1); else if (mode == 2) DW3_LAUNCH(float, 2); else DW3_LAUNCH(float, 3); }
#undef DW3_LAUNCH
    return launch_ok("dwconv3d");
  }
  int64_t total = (int64_t)N * D * H * W * (C / cpc);
  DISPATCH_T(dtype, k_dwconv, dim3(grid_for(total)), (hipStream_t)stream, x, x_stride, in_stats, act, bias, w, flip, y,
             y_stride, D, H, W, C, kD, kH, kW, total);
  return launch_ok("dwconv3d");
}

// LDS-tiled 3x3x3 path: rows of h per tile so that the fp32 halo + gradient tiles fit ~120 KiB
static size_t dw3_lds_bytes(int W, int TH, int cpc, int WG_CH) {
  size_t b = ((size_t)3 * (TH + 2) * (W + 2) + (size_t)TH * W) * WG_CH * cpc * 4;
  const size_t red = (size_t)NT * 3 * cpc * 4;     // row-third partials re-use the front of the buffer
  return b > red ? b : red;
}
static constexpr int64_t DW3_LDS_BUDGET = 120 * 1024;   // of 160 KiB per CU
static int dw3_lds_th(int H, int W, int cpc, int WG_CH) {
  // signed: two halo rows alone exceed the budget for W >= 79 (bf16) / W >= 159 (fp32) -> 0 -> streaming kernel
  const int64_t per_row_x = (int64_t)3 * (W + 2) * WG_CH * cpc * 4, per_row_g = (int64_t)W * WG_CH * cpc * 4;
  const int64_t room = DW3_LDS_BUDGET - 2 * per_row_x;
  if (room < per_row_x + per_row_g) return 0;
  int64_t th = room / (per_row_x + per_row_g);
  if (th > H) th = H;
  if (th > 8) th = 8;
  if (th < 1 || (int64_t)dw3_lds_bytes(W, (int)th, cpc, WG_CH) > 160 * 1024) return 0;
  return (int)th;   // 0: does not fit, use the streaming kernel
}

// kW == 3 path: rows (n,d,h) per block so that ~1024 blocks exist, at least 8 rows each
static void dw3_wgrad_cfg(int64_t rows, int groups, int* nblk, int* rpb) {
  int64_t want = 1024 / groups > 1 ? 1024 / groups : 1;
  int64_t r = (rows + want - 1) / want;
  if (r < 8) r = 8;
  *rpb = (int)r;
  *nblk = (int)((rows + r - 1) / r);
}

static void dw_wgrad_cfg(int64_t vox, int* nblk, int* vpb) {
  int64_t b = (vox + 255) / 256;  // >= 256 voxels per block
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  *vpb = (int)((vox + b - 1) / b);
  *nblk = (int)((vox + *vpb - 1) / *vpb);
}

extern "C" size_t cbim_dwconv3d_wgrad_workspace(int N, int D, int H, int W, int C, int kD, int kH, int kW) {
  int nblk, vpb;
  dw_wgrad_cfg((int64_t)N * D * H * W, &nblk, &vpb);
  int nb3, rpb;
  dw3_wgrad_cfg((int64_t)N * D * H, (C / 4 + DG - 1) / DG, &nb3, &rpb);   // upper bound over both dtypes
  if (nb3 > nblk) nblk = nb3;
  if (kD == 3 && kH == 3 && kW == 3) {   // LDS-tiled path: one partial slab per (n, d, h-tile), TH >= 1
    int64_t tiles = (int64_t)N * D * H;
    if (tiles > nblk) nblk = (int)tiles;
  }
  return (size_t)nblk * C * kD * kH * kW * sizeof(float);
}

extern "C" int cbim_dwconv3d_wgrad(int dtype, const void* x, int64_t x_stride, const float* in_stats, int act,
                                   const void* dy, int64_t dy_stride, const float* dy_bias, float* dw, int N, int D,
                                   int H, int W, int C, int kD, int kH, int kW, void* workspace, size_t ws_bytes,
                                   void* stream) {
  if (int e = check_c(dtype, C, "dwconv wgrad")) return e;
  if (int e = check_k(kD, kH, kW)) return e;
  CBIM_CHECK(workspace && ws_bytes >= cbim_dwconv3d_wgrad_workspace(N, D, H, W, C, kD, kH, kW), CBIM_EWORKSPACE,
             "dwconv wgrad workspace too small");
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int nblk, vpb;
  dw_wgrad_cfg((int64_t)N * D * H * W, &nblk, &vpb);
  int TT = kD * kH * kW, G = NT / TT, cch = C / cpc;
  hipStream_t st = (hipStream_t)stream;
  // matrix-core form (k_wgrad_r32, diagonal of the 32-channel groups): raw input, no gradient bias (functional.DWConvFn
  // materialises act(IN(x)) and dy + bias first), bf16, extents >= 8
  static const int dw_mfma = 1;
  if (dw_mfma && dtype == CBIM_BF16 && !in_stats && !dy_bias && kD == 3 && kH == 3 && kW == 3 && C % 32 == 0) {
    cbim_conv_desc cd = {};
    cd.dtype = CBIM_BF16; cd.N = N; cd.Di = D; cd.Hi = H; cd.Wi = W; cd.Cin = C; cd.Do = D; cd.Ho = H; cd.Wo = W; cd.Cout = C;
    cd.kD = 3; cd.kH = 3; cd.kW = 3; cd.pD = 1; cd.pH = 1; cd.pW = 1; cd.act = 0;
    if (cbim_wgrad_r32_dw_eligible(&cd)) {
      const int strips = cbim_wgrad_r32_dw_strips(&cd);
      CBIM_CHECK((size_t)strips * C * 27 * sizeof(float) <= ws_bytes, CBIM_EWORKSPACE, "dwconv wgrad workspace too small for %d strips", strips);
      if (int e = cbim_wgrad_r32_dw_launch(&cd, x, x_stride, dy, dy_stride, (float*)workspace, stream)) return e;
      CBIM_LAUNCH(k_dwconv_wgrad_reduce, dim3((C * 27 + NT - 1) / NT), dim3(NT), 0, st, (const float*)workspace, dw, strips, C * 27);
      return launch_ok("dwconv3d_wgrad_reduce");
    }
  }
  // channel chunks per workgroup: 8.  (Narrower groups would let 64-wide rows into the LDS-tiled kernel — the 64^3 x 256-channel
  // PatchMerging depthwise of MedFormer runs on the streaming kernel at 897 us — but with 4 chunks and 2-row tiles the tiled
  // kernel took 1.64 ms on that call: measured and rejected, profiles/r03_q_medformer_kernels.txt.)
  int wch = WG_CH_MAX, th_lds = 0;
  if (kD == 3 && kH == 3 && kW == 3) th_lds = dw3_lds_th(H, W, cpc, wch);
  if (th_lds >= 1) {
    const int htiles = (H + th_lds - 1) / th_lds;
    nblk = N * D * htiles;
    const size_t smem = dw3_lds_bytes(W, th_lds, cpc, wch);
    dim3 grid(nblk, (cch + wch - 1) / wch);
#ifndef CBIM_EMU
    static bool attr_done[2] = {false, false};
    if (!attr_done[dtype == CBIM_BF16]) {
      hipError_t e = dtype == CBIM_BF16
                         ? hipFuncSetAttribute((const void*)k_dwconv3_wgrad_lds<bf16_tag>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                         : hipFuncSetAttribute((const void*)k_dwconv3_wgrad_lds<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
      attr_done[dtype == CBIM_BF16] = true;
    }
#endif
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_dwconv3_wgrad_lds<bf16_tag>), grid, dim3(NT), smem, st, x, x_stride, in_stats, act, dy, dy_stride, dy_bias,
                  (float*)workspace, N, D, H, W, C, th_lds, wch);
    else
      CBIM_LAUNCH((k_dwconv3_wgrad_lds<float>), grid, dim3(NT), smem, st, x, x_stride, in_stats, act, dy, dy_stride, dy_bias,
                  (float*)workspace, N, D, H, W, C, th_lds, wch);
  } else if (kW == 3 && kD <= 3 && kH <= 3) {
    int groups = (cch + DG - 1) / DG, rpb;
    dw3_wgrad_cfg((int64_t)N * D * H, groups, &nblk, &rpb);
    DISPATCH_T(dtype, k_dwconv3_wgrad, dim3(nblk, groups), st, x, x_stride, in_stats, act, dy, dy_stride, dy_bias,
               (float*)workspace, N, D, H, W, C, kD, kH, rpb);
  } else {
    dim3 grid(nblk, (cch + G - 1) / G);
    DISPATCH_T(dtype, k_dwconv_wgrad, grid, st, x, x_stride, in_stats, act, dy, dy_stride, dy_bias, (float*)workspace, N,
               D, H, W, C, kD, kH, kW, vpb);
  }
  if (int e = launch_ok("dwconv3d_wgrad")) return e;
  CBIM_LAUNCH(k_dwconv_wgrad_reduce, dim3((C * TT + NT - 1) / NT), dim3(NT), 0, st, (const float*)workspace, dw, nblk,
              C * TT);
  return launch_ok("dwconv3d_wgrad_reduce");
}

extern "C" int cbim_space_to_depth(int dtype, const void* src, void* dst, int N, int D, int H, int W, int C, int sD,
                                   int sH, int sW, int inverse, void* stream) {
  return cbim_space_to_depth_strided(dtype, src, dst, N, D, H, W, C, sD, sH, sW, inverse, C, stream);
}

extern "C" int cbim_space_to_depth_strided(int dtype, const void* src, void* dst, int N, int D, int H, int W, int C, int sD,
                                           int sH, int sW, int inverse, int64_t fine_stride, void* stream) {
  if (int e = check_c(dtype, C, "space_to_depth")) return e;
  CBIM_CHECK(fine_stride >= C && fine_stride % (dtype == CBIM_BF16 ? 8 : 4) == 0, CBIM_EINVAL, "space_to_depth: row stride %lld of the "
             "fine tensor (C = %d)", (long long)fine_stride, C);
  CBIM_CHECK(sD >= 1 && sH >= 1 && sW >= 1 && D % sD == 0 && H % sH == 0 && W % sW == 0, CBIM_EINVAL,
             "space_to_depth: %dx%dx%d not divisible by %dx%dx%d", D, H, W, sD, sH, sW);
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int64_t total = (int64_t)N * D * H * W * (C / cpc);
  DISPATCH_T(dtype, k_space_to_depth, dim3(grid_for(total)), (hipStream_t)stream, src, dst, D, H, W, C, sD, sH, sW,
             inverse, total, fine_stride);
  return launch_ok("space_to_depth");
}

static constexpr int MWIDE_CODES = 128;   // attn_wide.hip
static int attn_check(int dtype, int L, int heads, int dh, int M) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(dh >= 1 && dh <= 4096, CBIM_EUNSUPPORTED, "attention dim_head %d (supported: 1..4096)", dh);
  CBIM_CHECK(M >= 1 && M <= MWIDE_CODES, CBIM_EUNSUPPORTED, "attention map codes %d (supported: 1..%d)", M, MWIDE_CODES);
  CBIM_CHECK(L >= 1 && heads >= 1 && heads <= 65535, CBIM_EINVAL, "bad attention extents");
  return 0;
}

extern "C" size_t cbim_bidir_attn_workspace(int N, int L, int heads, int dh, int M) {
  size_t nblk = (size_t)(L + 31) / 32;   // the matrix-core path writes one record per 32 voxels (the vector path per 128)
  size_t fwd = (size_t)N * heads * nblk * M * (dh + 2), bwd = (size_t)N * heads * nblk * 2 * M * dh;
  return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

#define ATTN_DISPATCH(KERNEL, grid, st, ...)                                                           \
  do {                                                                                                 \
    if (dtype == CBIM_BF16) {                                                                          \
      if (dh == 32) CBIM_LAUNCH((KERNEL<bf16_tag, 32>), grid, dim3(AT), 0, st, __VA_ARGS__);           \
      else if (dh == 16) CBIM_LAUNCH((KERNEL<bf16_tag, 16>), grid, dim3(AT), 0, st, __VA_ARGS__);      \
      else CBIM_LAUNCH((KERNEL<bf16_tag, 8>), grid, dim3(AT), 0, st, __VA_ARGS__);                     \
    } else {                                                                                           \
      if (dh == 32) CBIM_LAUNCH((KERNEL<float, 32>), grid, dim3(AT), 0, st, __VA_ARGS__);              \
      else if (dh == 16) CBIM_LAUNCH((KERNEL<float, 16>), grid, dim3(AT), 0, st, __VA_ARGS__);         \
      else CBIM_LAUNCH((KERNEL<float, 8>), grid, dim3(AT), 0, st, __VA_ARGS__);                        \
    }                                                                                                  \
  } while (0)

// (k_attn_merge grid = N*heads*M blocks of 64 threads; k_attn_bwd_reduce grid = (N*heads, ceil(2*M*dh/NT)))
extern "C" int cbim_attn_fwd_mfma_launch(const void* qv, int64_t qv_stride, const float* mq, const float* mv, void* feat_out,
                                         float* part, int N, int L, int heads, float scale, void* stream);
extern "C" int cbim_attn_bwd_mfma_launch(const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                         const float* colstat, const float* map_out, const void* d_feat_out,
                                         const float* d_map_out, void* d_qv, float* part, int N, int L, int heads, float scale,
                                         void* stream);
// attn_wide.hip: any d_head, up to MWIDE codes (one record per 64 voxels)
extern "C" int cbim_attn_fwd_wide_launch(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                         void* feat_out, float* map_out, float* colstat, float* part, int N, int L,
                                         int heads, int dh, int M, float scale, void* stream);
extern "C" int cbim_attn_bwd_wide_launch(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                         const float* colstat, const float* map_out, const void* d_feat_out,
                                         const float* d_map_out, void* d_qv, float* part, int N, int L, int heads, int dh,
                                         int M, float scale, void* stream);
extern "C" int cbim_mappool_bwd_wide_launch(int dtype, const void* fw, int64_t fw_stride, const float* map,
                                            const float* colstat, const float* dmap, void* dfw, int64_t dfw_stride, int N,
                                            int L, int C, int M, void* stream);
static bool attn_wide(int dh, int M) { return !(dh == 8 || dh == 16 || dh == 32) || M > MM; }
static bool attn_mfma_on() {
  static const int on = 1;
  return on != 0;
}

extern "C" int cbim_bidir_attn_fwd(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                   void* feat_out, float* map_out, float* colstat, int N, int L, int heads, int dh,
                                   int M, float scale, void* workspace, size_t ws_bytes, void* stream) {
  if (int e = attn_check(dtype, L, heads, dh, M)) return e;
  CBIM_CHECK(workspace && ws_bytes >= cbim_bidir_attn_workspace(N, L, heads, dh, M), CBIM_EWORKSPACE,
             "attention workspace too small");
  if (attn_wide(dh, M))
    return cbim_attn_fwd_wide_launch(dtype, qv, qv_stride, mq, mv, feat_out, map_out, colstat, (float*)workspace, N, L,
                                     heads, dh, M, scale, stream);
  int nblk = (L + AT - 1) / AT;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CBIM_BF16 && dh == 32 && M == 64 && attn_mfma_on()) {
    // matrix-core path: one online-softmax record per wave (32 voxels)
    nblk = (L + 31) / 32;
    if (int e = cbim_attn_fwd_mfma_launch(qv, qv_stride, mq, mv, feat_out, (float*)workspace, N, L, heads, scale, stream)) return e;
  } else {
    dim3 grid(nblk, heads, N);
    ATTN_DISPATCH(k_attn_fwd, grid, st, qv, qv_stride, mq, mv, feat_out, (float*)workspace, L, heads, M, scale, nblk);
    if (int e = launch_ok("bidir_attn_fwd")) return e;
  }
  CBIM_LAUNCH(k_attn_merge, dim3(N * heads * M), dim3(256), 0, st, (const float*)workspace, map_out, colstat, heads, M,
              dh, nblk);
  return launch_ok("bidir_attn_merge");
}

extern "C" int cbim_bidir_attn_bwd(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                   const float* colstat, const float* map_out, const void* d_feat_out,
                                   const float* d_map_out, void* d_qv, float* d_mq, float* d_mv, int N, int L,
                                   int heads, int dh, int M, float scale, void* workspace, size_t ws_bytes,
                                   void* stream) {
  if (int e = attn_check(dtype, L, heads, dh, M)) return e;
  CBIM_CHECK(workspace && ws_bytes >= cbim_bidir_attn_workspace(N, L, heads, dh, M), CBIM_EWORKSPACE,
             "attention workspace too small");
  int nblk = (L + AT - 1) / AT;
  hipStream_t st = (hipStream_t)stream;
  if (attn_wide(dh, M)) {
    nblk = (L + 63) / 64;
    if (int e = cbim_attn_bwd_wide_launch(dtype, qv, qv_stride, mq, mv, colstat, map_out, d_feat_out, d_map_out, d_qv,
                                          (float*)workspace, N, L, heads, dh, M, scale, stream)) return e;
  } else if (dtype == CBIM_BF16 && dh == 32 && M == 64 && attn_mfma_on()) {
    nblk = (L + 31) / 32;
    if (int e = cbim_attn_bwd_mfma_launch(qv, qv_stride, mq, mv, colstat, map_out, d_feat_out, d_map_out, d_qv,
                                          (float*)workspace, N, L, heads, scale, stream)) return e;
  } else {
    dim3 grid(nblk, heads, N);
    ATTN_DISPATCH(k_attn_bwd, grid, st, qv, qv_stride, mq, mv, colstat, map_out, d_feat_out, d_map_out, d_qv,
                  (float*)workspace, L, heads, M, scale, nblk);
    if (int e = launch_ok("bidir_attn_bwd")) return e;
  }
  CBIM_LAUNCH(k_attn_bwd_reduce, dim3(N * heads, (2 * M * dh + NT - 1) / NT), dim3(NT), 0, st, (const float*)workspace,
              d_mq, d_mv, heads, M, dh, nblk);
  return launch_ok("bidir_attn_bwd_reduce");
}

extern "C" size_t cbim_colsoftmax_pool_workspace(int N, int L, int C, int M) {
  size_t nblk = (size_t)(L + AT - 1) / AT;
  return (size_t)N * nblk * M * (2 + C) * sizeof(float);
}

extern "C" int cbim_colsoftmax_pool_fwd(int dtype, const void* fw, int64_t fw_stride, float* map, float* colstat,
                                        int N, int L, int C, int M, void* workspace, size_t ws_bytes, void* stream) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(M >= 1 && M <= MWIDE_CODES && C >= 1 && L >= 1, CBIM_EUNSUPPORTED, "colsoftmax_pool: M=%d C=%d L=%d", M, C, L);
  CBIM_CHECK(workspace && ws_bytes >= cbim_colsoftmax_pool_workspace(N, L, C, M), CBIM_EWORKSPACE,
             "colsoftmax_pool workspace too small");
  int nblk = (L + AT - 1) / AT;
  hipStream_t st = (hipStream_t)stream;
  const int n_jg = (M + MM - 1) / MM;
  // channels per workgroup: all of them when the row blocks alone fill the chip, else groups of 32 (multiples of 32: the kernel's
  // channel step)
  int cw = ((C + 31) / 32) * 32;
  if ((int64_t)nblk * n_jg * N < 512) {
    const int64_t want = 512 / ((int64_t)nblk * n_jg * N);
    int groups = (int)(want < (C + 31) / 32 ? want : (C + 31) / 32);
    if (groups < 1) groups = 1;
    cw = (((C + 31) / 32 + groups - 1) / groups) * 32;
  }
  const int n_cg = (C + cw - 1) / cw;
  dim3 grid(nblk, n_jg * n_cg, N);   // y = (group of 64 codes, group of cw channels)
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_mappool_fwd<bf16_tag>), grid, dim3(AT), 0, st, fw, fw_stride, (float*)workspace, L, C, M, nblk, cw);
  else
    CBIM_LAUNCH((k_mappool_fwd<float>), grid, dim3(AT), 0, st, fw, fw_stride, (float*)workspace, L, C, M, nblk, cw);
  if (int e = launch_ok("colsoftmax_pool_fwd")) return e;
  CBIM_CHECK(nblk <= MRG_MAXB, CBIM_EUNSUPPORTED, "colsoftmax_pool: %d partial records per image (max %d)", nblk, MRG_MAXB);
  CBIM_LAUNCH(k_mappool_merge, dim3(M, N), dim3(NT), 0, st, (const float*)workspace, map, colstat, C, M, nblk);
  return launch_ok("colsoftmax_pool_merge");
}

extern "C" int cbim_colsoftmax_pool_bwd(int dtype, const void* fw, int64_t fw_stride, const float* map,
                                        const float* colstat, const float* dmap, void* dfw, int64_t dfw_stride, int N,
                                        int L, int C, int M, void* stream) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(M >= 1 && M <= MWIDE_CODES && C >= 1 && L >= 1, CBIM_EUNSUPPORTED, "colsoftmax_pool: M=%d C=%d L=%d", M, C, L);
  if (M > MM)
    return cbim_mappool_bwd_wide_launch(dtype, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride, N, L, C, M, stream);
  int nblk = (L + AT - 1) / AT;
  hipStream_t st = (hipStream_t)stream;
  const int cpc = dtype == CBIM_BF16 ? 8 : 4;
  static const int four = 1;
  if (four && M % cpc == 0 && C % cpc == 0 && fw_stride % cpc == 0 && dfw_stride % cpc == 0) {
    const size_t smem = (size_t)(4 * 64 * (MM + 1) + MM) * sizeof(float);
#ifndef CBIM_EMU
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e1 = hipFuncSetAttribute((const void*)k_mappool_bwd4<bf16_tag>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipError_t e2 = hipFuncSetAttribute((const void*)k_mappool_bwd4<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      CBIM_CHECK(e1 == hipSuccess && e2 == hipSuccess, CBIM_ELAUNCH, "colsoftmax_pool_bwd: hipFuncSetAttribute failed");
      attr_done = true;
    }
#endif
    dim3 grid4((L + 63) / 64, 1, N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_mappool_bwd4<bf16_tag>), grid4, dim3(MP4_T), smem, st, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride, L,
                  C, M);
    else
      CBIM_LAUNCH((k_mappool_bwd4<float>), grid4, dim3(MP4_T), smem, st, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride, L, C,
                  M);
    return launch_ok("colsoftmax_pool_bwd (4 waves per 64 voxels)");
  }
  dim3 grid(nblk, 1, N);
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_mappool_bwd<bf16_tag>), grid, dim3(AT), 0, st, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride, L,
                C, M);
  else
    CBIM_LAUNCH((k_mappool_bwd<float>), grid, dim3(AT), 0, st, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride, L, C,
                M);
  return launch_ok("colsoftmax_pool_bwd");
}

CBIM_DEFINE_WARM(medformer)
