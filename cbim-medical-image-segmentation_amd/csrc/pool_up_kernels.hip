// pool_up_kernels.hip — MaxPool3d and trilinear(align_corners=True)-upsample + concat.
//
// HBM-bound gather kernels on channels-last tensors; one work item = one 16-byte channel chunk
// of one output voxel, so every global access of a wave is a run of whole 16-byte chunks.
//
// Reference call sites (/root/reference): nn.MaxPool3d(down_scale) unet_utils.py:36;
// F.interpolate(x1, size=x2.shape[2:], mode='trilinear', align_corners=True) + torch.cat([x2,x1],1)
// unet_utils.py:69-71.  Index/weight arithmetic follows ATen's align_corners rule:
// scale=(in-1)/(out-1) (0 if out==1), src=scale*dst, i0=(int)src, i1=min(i0+1,in-1), l1=src-i0.
#include "cbim_common.h"
#include "up_lerp.h"

namespace cbim {

static constexpr int NT = 256;

template <typename T>
__global__ void __launch_bounds__(NT) k_maxpool_fwd(const void* __restrict__ x, void* __restrict__ y,
                                                    uint8_t* __restrict__ idx, int D, int H, int W, int C,
                                                    int sD, int sH, int sW, int Do, int Ho, int Wo,
                                                    int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int cc = (int)(i % cch);
    int64_t r = i / cch;
    int wo = (int)(r % Wo); r /= Wo;
    int ho = (int)(r % Ho); r /= Ho;
    int dz = (int)(r % Do);
    int64_t n = r / Do;
    float m[CPC];
    uint8_t am[CPC];
#pragma unroll
    for (int j = 0; j < CPC; ++j) { m[j] = -INFINITY; am[j] = 0; }
    int pos = 0;
    for (int a = 0; a < sD; ++a)
      for (int b = 0; b < sH; ++b)
        for (int c = 0; c < sW; ++c, ++pos) {
          size_t row = (((size_t)n * D + (dz * sD + a)) * H + (ho * sH + b)) * W + (wo * sW + c);
          float f[CPC];
          Elem<T>::unpack(ld_chunk<T>(x, row * C + (size_t)cc * CPC), f);
#pragma unroll
          for (int j = 0; j < CPC; ++j)
            if (f[j] > m[j] || f[j] != f[j]) { m[j] = f[j]; am[j] = (uint8_t)pos; }
        }
    size_t orow = (((size_t)n * Do + dz) * Ho + ho) * Wo + wo;
    st_chunk<T>(y, orow * C + (size_t)cc * CPC, Elem<T>::pack(m));
#pragma unroll
    for (int j = 0; j < CPC; ++j) idx[orow * C + (size_t)cc * CPC + j] = am[j];
  }
}

// grid = (chunk blocks of one input plane, N * D): the (n, d) plane comes from blockIdx.y, (h, w, chunk) from two 32-bit
// divisions (the flat 64-bit index cost four 64-bit divisions per 16-byte chunk: 47 % of the copy bandwidth); the chunk's
// 8 (4) arg-max codes are one 8- (4-) byte load
template <typename T>
__global__ void __launch_bounds__(NT) k_maxpool_bwd(const void* __restrict__ dy,
                                                    const uint8_t* __restrict__ idx, void* __restrict__ dx,
                                                    int D, int H, int W, int C, int sD, int sH, int sW,
                                                    int Do, int Ho, int Wo, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const unsigned cch = (unsigned)(C / CPC);
  const unsigned per_plane = (unsigned)H * (unsigned)W * cch;
  // gridDim.y is capped at 65535: a block strides over the (image, depth) planes (N * D of any size)
  for (unsigned plane = blockIdx.y; plane < (unsigned)total; plane += gridDim.y) {
  const unsigned n = plane / (unsigned)D, d = plane % (unsigned)D;
  const unsigned dz = d / (unsigned)sD;
  for (unsigned j = blockIdx.x * NT + threadIdx.x; j < per_plane; j += gridDim.x * NT) {
    const unsigned cc = j % cch, hw = j / cch;
    const unsigned w = hw % (unsigned)W, h = hw / (unsigned)W;
    float f[CPC];
#pragma unroll
    for (int q = 0; q < CPC; ++q) f[q] = 0.f;
    const unsigned ho = h / (unsigned)sH, wo = w / (unsigned)sW;
    if (dz < (unsigned)Do && ho < (unsigned)Ho && wo < (unsigned)Wo) {
      const unsigned pos = ((d - dz * sD) * sH + (h - ho * sH)) * sW + (w - wo * sW);
      const size_t orow = (((size_t)n * Do + dz) * Ho + ho) * Wo + wo;
      float g[CPC];
      Elem<T>::unpack(ld_chunk<T>(dy, orow * C + (size_t)cc * CPC), g);
      uint8_t code[CPC];
      if (CPC == 8) *(u32x2*)code = *(const u32x2*)(idx + orow * C + (size_t)cc * CPC);
      else *(unsigned*)code = *(const unsigned*)(idx + orow * C + (size_t)cc * CPC);
#pragma unroll
      for (int q = 0; q < CPC; ++q)
        if (code[q] == (uint8_t)pos) f[q] = g[q];
    }
    const size_t row = (((size_t)n * D + d) * H + h) * W + w;
    st_chunk<T>(dx, row * C + (size_t)cc * CPC, Elem<T>::pack(f));
  }
  }
}

// ---- trilinear align_corners source index ------------------------------------------------------
struct Lin { int i0, i1; float l0, l1; };
__device__ __forceinline__ float lin_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
__device__ __forceinline__ Lin lin_src(int dst, float scale, int in) {
  float src = scale * (float)dst;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  float l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  Lin r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in - 1 ? 1 : 0);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}

template <typename T>
__global__ void __launch_bounds__(NT) k_upcat_fwd(const void* __restrict__ low, const void* __restrict__ skip,
                                                  void* __restrict__ out, int Dl, int Hl, int Wl, int Cl,
                                                  int D, int H, int W, int Cs, int skip_first,
                                                  int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const int Ct = Cs + Cl;
  const int cch = Ct / CPC;
  const float sd = lin_scale(Dl, D), sh = lin_scale(Hl, H), sw = lin_scale(Wl, W);
  const int skip_lo = skip_first ? 0 : Cl;
  const int low_lo = skip_first ? Cs : 0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int cc = (int)(i % cch);
    int64_t r = i / cch;  // output row
    int c0 = cc * CPC;
    bool is_skip = c0 >= skip_lo && c0 < skip_lo + Cs;
    if (is_skip) {
      st_chunk<T>(out, (size_t)r * Ct + c0, ld_chunk<T>(skip, (size_t)r * Cs + (c0 - skip_lo)));
      continue;
    }
    int64_t q = r;
    int w = (int)(q % W); q /= W;
    int h = (int)(q % H); q /= H;
    int d = (int)(q % D);
    int64_t n = q / D;
    Lin ld = lin_src(d, sd, Dl), lh = lin_src(h, sh, Hl), lw = lin_src(w, sw, Wl);
    int cl = c0 - low_lo;
    u32x4 cn[8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        size_t rbase = (((size_t)n * Dl + (a ? ld.i1 : ld.i0)) * Hl + (b ? lh.i1 : lh.i0)) * Wl;
        cn[(a * 2 + b) * 2] = ld_chunk<T>(low, (rbase + lw.i0) * Cl + cl);
        cn[(a * 2 + b) * 2 + 1] = ld_chunk<T>(low, (rbase + lw.i1) * Cl + cl);
      }
    float acc[CPC];
    trilerp<T>(cn, ld.l0, ld.l1, lh.l0, lh.l1, lw.l0, lw.l1, acc);
    st_chunk<T>(out, (size_t)r * Ct + c0, Elem<T>::pack(acc));
  }
}

// Same output as k_upcat_fwd plus the InstanceNorm statistics partials of the concatenated tensor (the first block of
// the decoder level normalises it: a separate statistics pass re-read the 403 MB concat of the 128^3 level).
// grid = (parts, N); thread = (fixed channel chunk, voxel lane) so per-channel sums stay in registers; 32-bit index
// arithmetic.  partials: float [N][P][Ct][3] = (n, mean, M2) records merged by k_stats_finalize.
template <typename T>
__global__ void __launch_bounds__(NT) k_upcat_fwd_stats(const void* __restrict__ low, const void* __restrict__ skip,
                                                        void* __restrict__ out, int Dl, int Hl, int Wl, int Cl,
                                                        int D, int H, int W, int Cs, int skip_first, int P,
                                                        float* __restrict__ partials) {
  constexpr int CPC = Elem<T>::CPC;
  const int Ct = Cs + Cl;
  const int cch = Ct / CPC, vlc = NT / cch;       // host guarantees cch <= NT
  const int t = threadIdx.x, cc = t % cch, vl = t / cch;
  const int part = blockIdx.x, n = blockIdx.y;
  const int S = D * H * W, per = (S + P - 1) / P;
  const int v0 = part * per, v1 = v0 + per < S ? v0 + per : S;
  const float sd = lin_scale(Dl, D), sh = lin_scale(Hl, H), sw = lin_scale(Wl, W);
  const int skip_lo = skip_first ? 0 : Cl, low_lo = skip_first ? Cs : 0;
  const int c0 = cc * CPC;
  const bool is_skip = c0 >= skip_lo && c0 < skip_lo + Cs;
  const bool active = vl < vlc;
  float s0[CPC], s1[CPC], shift[CPC];
  float cnt = 0.f;
#pragma unroll
  for (int j = 0; j < CPC; ++j) { s0[j] = 0.f; s1[j] = 0.f; shift[j] = 0.f; }
  if (active) {
    const size_t nrow = (size_t)n * S;
    for (int v = v0 + vl; v < v1; v += vlc) {
      u32x4 o;
      if (is_skip) {
        o = ld_chunk<T>(skip, (nrow + v) * Cs + (c0 - skip_lo));
      } else {
        const int w = v % W, q = v / W, h = q % H, d = q / H;
        const Lin ld = lin_src(d, sd, Dl), lh = lin_src(h, sh, Hl), lw = lin_src(w, sw, Wl);
        const int cl = c0 - low_lo;
        u32x4 cn[8];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const size_t rbase = (((size_t)n * Dl + (a ? ld.i1 : ld.i0)) * Hl + (b ? lh.i1 : lh.i0)) * Wl;
            cn[(a * 2 + b) * 2] = ld_chunk<T>(low, (rbase + lw.i0) * Cl + cl);
            cn[(a * 2 + b) * 2 + 1] = ld_chunk<T>(low, (rbase + lw.i1) * Cl + cl);
          }
        float acc[CPC];
        trilerp<T>(cn, ld.l0, ld.l1, lh.l0, lh.l1, lw.l0, lw.l1, acc);
        o = Elem<T>::pack(acc);
      }
      if (out) st_chunk<T>(out, (nrow + v) * Ct + c0, o);   // out == nullptr: statistics of the virtual tensor only
      float f[CPC];
      Elem<T>::unpack(o, f);   // statistics of the STORED (rounded) values, like a pass over the tensor
      if (cnt == 0.f) {
#pragma unroll
        for (int j = 0; j < CPC; ++j) shift[j] = f[j];
      }
      cnt += 1.f;
#pragma unroll
      for (int j = 0; j < CPC; ++j) { const float dlt = f[j] - shift[j]; s0[j] += dlt; s1[j] += dlt * dlt; }
    }
  }
  __shared__ float red[NT * 3 * 8];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    const Moments m = moments_from_shifted(cnt, shift[j], s0[j], s1[j]);
    red[(t * CPC + j) * 3 + 0] = m.n;
    red[(t * CPC + j) * 3 + 1] = m.mean;
    red[(t * CPC + j) * 3 + 2] = m.m2;
  }
  __syncthreads();
  for (int ch = t; ch < cch * CPC; ch += NT) {     // one thread per channel merges the voxel lanes in fixed order
    const int c2 = ch / CPC, j = ch % CPC;
    Moments acc = {0.f, 0.f, 0.f};
    for (int q = 0; q < vlc; ++q) {
      const float* r = red + ((q * cch + c2) * CPC + j) * 3;
      const Moments b = {r[0], r[1], r[2]};
      acc = moments_merge(acc, b);
    }
    const size_t o = (((size_t)n * P + part) * Ct + c2 * CPC + j) * 3;
    partials[o] = acc.n; partials[o + 1] = acc.mean; partials[o + 2] = acc.m2;
  }
}


// trilinear(align_corners) value of one channel chunk of `low` at fine voxel (n, d, h, w) in float32: the fused kernels
// below never store the up-sampled tensor, so it is never rounded to the storage type either — they normalise the fp32
// interpolation with the statistics of the fp32 interpolation (k_up_gram_stats)
template <typename T>
__device__ __forceinline__ void up_chunk(const void* __restrict__ low, int n, int d, int h, int w, float sd, float sh,
                                         float sw, int Dl, int Hl, int Wl, int Cl, int cl, float* f) {
  constexpr int CPC = Elem<T>::CPC;
  const Lin ld = lin_src(d, sd, Dl), lh = lin_src(h, sh, Hl), lw = lin_src(w, sw, Wl);
  u32x4 cn[8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const size_t rbase = (((size_t)n * Dl + (a ? ld.i1 : ld.i0)) * Hl + (b ? lh.i1 : lh.i0)) * Wl;
      cn[(a * 2 + b) * 2] = ld_chunk<T>(low, (rbase + lw.i0) * Cl + cl);
      cn[(a * 2 + b) * 2 + 1] = ld_chunk<T>(low, (rbase + lw.i1) * Cl + cl);
    }
  trilerp<T>(cn, ld.l0, ld.l1, lh.l0, lh.l1, lw.l0, lw.l1, f);
}

// a = act(IN([skip | up(low)])) in ONE pass: the decoder level's first block reads the activated concatenation, the raw
// concatenation is never written (round 2 wrote it, 403 MB at the 128^3 level, and a second pass normalised it).
// grid = (parts, N); thread = (fixed channel chunk, voxel lane): its 2 x CPC statistics stay in registers.
template <typename T>
__global__ void __launch_bounds__(NT) k_upcat_act_fwd(const void* __restrict__ low, const void* __restrict__ skip,
                                                      const float* __restrict__ stats, void* __restrict__ out, int Dl,
                                                      int Hl, int Wl, int Cl, int D, int H, int W, int Cs, int skip_first,
                                                      int P, int act) {
  constexpr int CPC = Elem<T>::CPC;
  const int Ct = Cs + Cl;
  const int cch = Ct / CPC, vlc = NT / cch;
  const int t = threadIdx.x, cc = t % cch, vl = t / cch;
  if (vl >= vlc) return;
  const int part = blockIdx.x, n = blockIdx.y;
  const int S = D * H * W, per = (S + P - 1) / P;
  const int v0 = part * per, v1 = v0 + per < S ? v0 + per : S;
  const float sd = lin_scale(Dl, D), sh = lin_scale(Hl, H), sw = lin_scale(Wl, W);
  const int skip_lo = skip_first ? 0 : Cl, low_lo = skip_first ? Cs : 0;
  const int c0 = cc * CPC;
  const bool is_skip = c0 >= skip_lo && c0 < skip_lo + Cs;
  float mean[CPC], rstd[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    mean[j] = stats[((size_t)n * Ct + c0 + j) * 2];
    rstd[j] = stats[((size_t)n * Ct + c0 + j) * 2 + 1];
  }
  const size_t nrow = (size_t)n * S;
  for (int v = v0 + vl; v < v1; v += vlc) {
    float f[CPC];
    if (is_skip) Elem<T>::unpack(ld_chunk<T>(skip, (nrow + v) * Cs + (c0 - skip_lo)), f);
    else {
      const int w = v % W, q = v / W, h = q % H, d = q / H;
      up_chunk<T>(low, n, d, h, w, sd, sh, sw, Dl, Hl, Wl, Cl, c0 - low_lo, f);
    }
#pragma unroll
    for (int j = 0; j < CPC; ++j) f[j] = act_fwd((f[j] - mean[j]) * rstd[j], act);
    st_chunk<T>(out, (nrow + v) * Ct + c0, Elem<T>::pack(f));
  }
}

// InstanceNorm backward of the virtual concatenation, split on the way out: with xh = (x - mean) * rstd of
// x = [skip | up(low)] (re-formed here) dx = rstd * (g - m1 - xh * m2); the skip channels go to dskip, the others to
// dup (fine resolution; k_upcat_bwd_low then gathers it into dlow).  Replaces k_norm_bwd_apply over the stored
// concatenation + k_slice_copy, and lets the gather read dense Cl-channel rows.
template <typename T>
__global__ void __launch_bounds__(NT) k_upcat_norm_bwd(const void* __restrict__ g, const void* __restrict__ low,
                                                       const void* __restrict__ skip, const float* __restrict__ stats,
                                                       const float* __restrict__ sums, void* __restrict__ dskip,
                                                       void* __restrict__ dup, int Dl, int Hl, int Wl, int Cl, int D, int H,
                                                       int W, int Cs, int skip_first, int P) {
  constexpr int CPC = Elem<T>::CPC;
  const int Ct = Cs + Cl;
  const int cch = Ct / CPC, vlc = NT / cch;
  const int t = threadIdx.x, cc = t % cch, vl = t / cch;
  if (vl >= vlc) return;
  const int part = blockIdx.x, n = blockIdx.y;
  const int S = D * H * W, per = (S + P - 1) / P;
  const int v0 = part * per, v1 = v0 + per < S ? v0 + per : S;
  const float sd = lin_scale(Dl, D), sh = lin_scale(Hl, H), sw = lin_scale(Wl, W);
  const int skip_lo = skip_first ? 0 : Cl, low_lo = skip_first ? Cs : 0;
  const int c0 = cc * CPC;
  const bool is_skip = c0 >= skip_lo && c0 < skip_lo + Cs;
  float mean[CPC], rstd[CPC], m1[CPC], m2[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    mean[j] = stats[((size_t)n * Ct + c0 + j) * 2];
    rstd[j] = stats[((size_t)n * Ct + c0 + j) * 2 + 1];
    m1[j] = sums[((size_t)n * Ct + c0 + j) * 2];
    m2[j] = sums[((size_t)n * Ct + c0 + j) * 2 + 1];
  }
  const size_t nrow = (size_t)n * S;
  for (int v = v0 + vl; v < v1; v += vlc) {
    float f[CPC], gg[CPC];
    Elem<T>::unpack(ld_chunk<T>(g, (nrow + v) * Ct + c0), gg);
    if (is_skip) Elem<T>::unpack(ld_chunk<T>(skip, (nrow + v) * Cs + (c0 - skip_lo)), f);
    else {
      const int w = v % W, q = v / W, h = q % H, d = q / H;
      up_chunk<T>(low, n, d, h, w, sd, sh, sw, Dl, Hl, Wl, Cl, c0 - low_lo, f);
    }
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      const float xh = (f[j] - mean[j]) * rstd[j];
      gg[j] = rstd[j] * (gg[j] - m1[j] - xh * m2[j]);
    }
    if (is_skip) st_chunk<T>(dskip, (nrow + v) * Cs + (c0 - skip_lo), Elem<T>::pack(gg));
    else st_chunk<T>(dup, (nrow + v) * Cl + (c0 - low_lo), Elem<T>::pack(gg));
  }
}

// candidate range of destination indices whose interpolation footprint can touch source index l
__device__ __forceinline__ void dst_range(int l, float scale, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  float a = ((float)l - 1.f) / scale, b = ((float)l + 1.f) / scale;
  lo = (int)floorf(a) - 1;
  hi = (int)ceilf(b) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float lin_weight_to(int dst, float scale, int in, int l) {
  Lin s = lin_src(dst, scale, in);
  float w = 0.f;
  if (s.i0 == l) w += s.l0;
  if (s.i1 == l) w += s.l1;
  return w;
}

template <typename T>
__global__ void __launch_bounds__(NT) k_upcat_bwd_low(const void* __restrict__ dout, void* __restrict__ dlow,
                                                      int Dl, int Hl, int Wl, int Cl, int D, int H, int W,
                                                      int Ct, int low_lo, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = Cl / CPC;
  const float sd = lin_scale(Dl, D), sh = lin_scale(Hl, H), sw = lin_scale(Wl, W);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int cc = (int)(i % cch);
    int64_t q = i / cch;
    int wl = (int)(q % Wl); q /= Wl;
    int hl = (int)(q % Hl); q /= Hl;
    int dl = (int)(q % Dl);
    int64_t n = q / Dl;
    int d0, d1, h0, h1, w0, w1;
    dst_range(dl, sd, D, d0, d1);
    dst_range(hl, sh, H, h0, h1);
    dst_range(wl, sw, W, w0, w1);
    float acc[CPC];
#pragma unroll
    for (int j = 0; j < CPC; ++j) acc[j] = 0.f;
    for (int d = d0; d <= d1; ++d) {
      float wd = lin_weight_to(d, sd, Dl, dl);
      if (wd == 0.f) continue;
      for (int h = h0; h <= h1; ++h) {
        float wh = lin_weight_to(h, sh, Hl, hl);
        if (wh == 0.f) continue;
        float wdh = wd * wh;
        size_t rbase = (((size_t)n * D + d) * H + h) * W;
        for (int w = w0; w <= w1; ++w) {
          float ww = lin_weight_to(w, sw, Wl, wl);
          if (ww == 0.f) continue;
          float f[CPC];
          Elem<T>::unpack(ld_chunk<T>(dout, (rbase + w) * Ct + low_lo + (size_t)cc * CPC), f);
          float wt = wdh * ww;
#pragma unroll
          for (int j = 0; j < CPC; ++j) acc[j] += wt * f[j];
        }
      }
    }
    size_t row = (((size_t)n * Dl + dl) * Hl + hl) * Wl + wl;
    st_chunk<T>(dlow, row * Cl + (size_t)cc * CPC, Elem<T>::pack(acc));
  }
}

// ---- transposed trilinear interpolation as three separable 1-D passes ------------------------------------------------------
// dlow = up^T(dup): the adjoint of F.interpolate(trilinear, align_corners=True) factorises over the axes, so the gradient of
// the up-sampled tensor is reduced along W, then H, then D.  Each pass reads its input once (a coarse index gathers <= 4 fine
// neighbours that are adjacent in memory along the reduced axis) and writes half / a quarter / an eighth of it; the one-pass
// gather (k_upcat_bwd_low, k_trilinear_planes_bwd) reads every fine value 8 times through 64 weighted candidates per output
// (358 us at the 128^3 -> 64^3 level, 477 us for the 16-class aux head of MedFormer).
// src [outer][F][inner] -> dst [outer][L][inner]; VEC elements (one 16-byte chunk, or 1 float) per item.
template <typename T, int VEC>
__global__ void __launch_bounds__(NT) k_lin_adjoint_axis(const void* __restrict__ src, void* __restrict__ dst, int F, int L,
                                                         int64_t inner_items, int64_t src_row, int64_t c_off, int64_t total) {
  const float scale = lin_scale(L, F);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t ic = i % inner_items;
    int64_t q = i / inner_items;
    const int l = (int)(q % L);
    const int64_t o = q / L;
    int lo, hi;
    dst_range(l, scale, F, lo, hi);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int f = lo; f <= hi; ++f) {
      const float w = lin_weight_to(f, scale, L, l);
      if (w == 0.f) continue;
      const size_t e = (size_t)(o * F + f) * src_row + c_off + (size_t)ic * VEC;
      if (VEC == 1) acc[0] += w * ((const float*)src)[e];
      else {
        float v[VEC];
        Elem<T>::unpack(ld_chunk<T>(src, e), v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += w * v[j];
      }
    }
    const size_t e = ((size_t)(o * L + l) * inner_items + ic) * VEC;
    if (VEC == 1) ((float*)dst)[e] = acc[0];
    else st_chunk<T>(dst, e, Elem<T>::pack(acc));
  }
}

// ---- statistics of the virtual up-sampled tensor from the COARSE grid --------------------------------------------------------
// up = (W_d (x) W_h (x) W_w) low with 1-D interpolation matrices whose rows hold <= 2 adjacent weights, so
//     sum_f up_f   = sum_l cw_d[l_d] cw_h[l_h] cw_w[l_w] low_l                      (cw = column sums)
//     sum_f up_f^2 = low^T (G_d (x) G_h (x) G_w) low,   G = W^T W tridiagonal      (a 27-point stencil on the coarse grid)
// — 1/8 of the voxels of the fine grid and no interpolation at all (k_up_tile<0> recomputes the whole up-sampled tensor for
// its statistics: 70 M vector instructions, 150 us at 64^3 -> 128^3 x 64 channels for 34 MB of input).  Both sums are taken
// of low - shift[c] (interpolation reproduces constants, so up(low - s) = up(low) - s): E[x^2] - E[x]^2 stays well
// conditioned.  A workgroup owns 4x8x8 coarse tiles (+1 halo, fp32 in LDS, 32 channels at a time); records (shift, sum,
// sum of squares) per workgroup, combined in fp64 by k_stats_finalize mode 2.  The statistics are those of the fp32
// interpolation; the tensor the next pass normalises is the same values rounded to the storage type (rounding noise only).
struct GramAx { float gm, g0, gp, cw; };
static constexpr int GR_D = 4, GR_H = 8, GR_W = 8, GR_ROWS = (GR_D + 2) * (GR_H + 2) * (GR_W + 2), GR_C = 32;
static constexpr size_t GR_SMEM = (size_t)GR_ROWS * GR_C * 4 + (GR_D + GR_H + GR_W) * sizeof(GramAx);
template <typename T>
__global__ void __launch_bounds__(NT) k_up_gram_stats(const void* __restrict__ low, int Dl, int Hl, int Wl, int Cl, int D, int H,
                                                      int W, int P, float* __restrict__ partials) {
  constexpr int CPC = Elem<T>::CPC, GCH = GR_C / CPC, VLC = NT / GCH;
#ifdef CBIM_EMU
  unsigned char* smem = cbim_emu::dyn_smem();
#else
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
  float* box = (float*)smem;                                         // [6][10][10][32]
  GramAx* ax = (GramAx*)(smem + (size_t)GR_ROWS * GR_C * 4);         // d[4] | h[8] | w[8]
  const int tid = threadIdx.x, cc = tid % GCH, vl = tid / GCH, n = blockIdx.y;
  const int tiles_d = (Dl + GR_D - 1) / GR_D, tiles_h = (Hl + GR_H - 1) / GR_H, tiles_w = (Wl + GR_W - 1) / GR_W;
  const int tiles = tiles_d * tiles_h * tiles_w;
  const float sd = lin_scale(Dl, D), sh = lin_scale(Hl, H), sw = lin_scale(Wl, W);
  {
    const int cg0 = (int)blockIdx.z * GR_C;          // one 32-channel group per workgroup (grid.z)
    const int c0 = cg0 + cc * CPC;
    const bool cok = c0 < Cl;
    float shift[CPC], s1[CPC], s2[CPC];
#pragma unroll
    for (int j = 0; j < CPC; ++j) { shift[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    if (cok) Elem<T>::unpack(ld_chunk<T>(low, (size_t)n * Dl * Hl * Wl * Cl + c0), shift);   // the image's first voxel
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
      const int tw = t % tiles_w, q = t / tiles_w, th = q % tiles_h, td = q / tiles_h;
      const int d0 = td * GR_D, h0 = th * GR_H, w0 = tw * GR_W;
      __syncthreads();
      // stage low - shift (zeros outside the grid / past Cl), five loads in flight per thread (bf16: two trips per tile)
      for (int base = tid; base < GR_ROWS * GCH; base += NT * 5) {
        u32x4 raw[5];
        int ok[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = base + u * NT;
          ok[u] = 0;
          raw[u] = u32x4{0u, 0u, 0u, 0u};
          if (i < GR_ROWS * GCH) {
            const int r = i / GCH;
            const int bw = r % (GR_W + 2), q2 = r / (GR_W + 2), bh = q2 % (GR_H + 2), bd = q2 / (GR_H + 2);
            const int ld = d0 - 1 + bd, lh = h0 - 1 + bh, lw = w0 - 1 + bw;
            ok[u] = 1;
            if (cok && ld >= 0 && ld < Dl && lh >= 0 && lh < Hl && lw >= 0 && lw < Wl) {
              raw[u] = ld_chunk<T>(low, ((((size_t)n * Dl + ld) * Hl + lh) * Wl + lw) * Cl + c0);
              ok[u] = 2;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          if (!ok[u]) continue;
          const int i = base + u * NT;
          float f[CPC];
          Elem<T>::unpack(raw[u], f);
#pragma unroll
          for (int j = 0; j < CPC; ++j) box[(size_t)(i / GCH) * GR_C + cc * CPC + j] = ok[u] == 2 ? f[j] - shift[j] : 0.f;
        }
      }
      if (tid < 192) {                                   // (whole waves: the shuffles below need every lane of a wave)
        // axis tables: entry e = tid / 8 (20 of the 24 exist), its fine indices dealt to 8 lanes
        const int e = tid >> 3, k = tid & 7;
        const bool ev = e < GR_D + GR_H + GR_W;
        const int a = e < GR_D ? 0 : e < GR_D + GR_H ? 1 : 2;
        const int l = (a == 0 ? d0 + e : a == 1 ? h0 + e - GR_D : w0 + e - GR_D - GR_H);
        const int L = a == 0 ? Dl : a == 1 ? Hl : Wl, F = a == 0 ? D : a == 1 ? H : W;
        const float sc = a == 0 ? sd : a == 1 ? sh : sw;
        GramAx r = {0.f, 0.f, 0.f, 0.f};
        if (ev && l < L) {
          int lo, hi;
          dst_range(l, sc, F, lo, hi);
          for (int f = lo + k; f <= hi; f += 8) {
            const float w0f = lin_weight_to(f, sc, L, l);
            if (w0f == 0.f) continue;
            r.g0 += w0f * w0f;
            r.cw += w0f;
            r.gm += w0f * lin_weight_to(f, sc, L, l - 1);
            r.gp += w0f * lin_weight_to(f, sc, L, l + 1);
          }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
          r.g0 += __shfl_xor(r.g0, m, 8); r.cw += __shfl_xor(r.cw, m, 8);
          r.gm += __shfl_xor(r.gm, m, 8); r.gp += __shfl_xor(r.gp, m, 8);
        }
        if (ev && k == 0) ax[e] = r;
      }
      __syncthreads();
      for (int v = vl; v < GR_D * GR_H * GR_W; v += VLC) {
        const int bw = v % GR_W, bh = (v / GR_W) % GR_H, bd = v / (GR_W * GR_H);
        if (d0 + bd >= Dl || h0 + bh >= Hl || w0 + bw >= Wl) continue;
        const GramAx gd = ax[bd], gh = ax[GR_D + bh], gw = ax[GR_D + GR_H + bw];
        const float* ctr = box + (size_t)(((bd + 1) * (GR_H + 2) + bh + 1) * (GR_W + 2) + bw + 1) * GR_C + cc * CPC;
        float tt[CPC];
#pragma unroll
        for (int j = 0; j < CPC; ++j) tt[j] = 0.f;
#pragma unroll
        for (int a = -1; a <= 1; ++a) {
          const float wa = a < 0 ? gd.gm : a == 0 ? gd.g0 : gd.gp;
          float ua[CPC];
#pragma unroll
          for (int j = 0; j < CPC; ++j) ua[j] = 0.f;
#pragma unroll
          for (int b = -1; b <= 1; ++b) {
            const float wb = b < 0 ? gh.gm : b == 0 ? gh.g0 : gh.gp;
            const float* row = ctr + (a * (GR_H + 2) + b) * (GR_W + 2) * GR_C;
            float ub[CPC];
#pragma unroll
            for (int j = 0; j < CPC; ++j) ub[j] = fmaf(gw.gp, row[GR_C + j], fmaf(gw.g0, row[j], gw.gm * row[-GR_C + j]));
#pragma unroll
            for (int j = 0; j < CPC; ++j) ua[j] = fmaf(wb, ub[j], ua[j]);
          }
#pragma unroll
          for (int j = 0; j < CPC; ++j) tt[j] = fmaf(wa, ua[j], tt[j]);
        }
        const float cw3 = gd.cw * gh.cw * gw.cw;
#pragma unroll
        for (int j = 0; j < CPC; ++j) { s1[j] = fmaf(cw3, ctr[j], s1[j]); s2[j] = fmaf(ctr[j], tt[j], s2[j]); }
      }
    }
    // voxel lanes -> one record per channel and workgroup (fixed order)
    __syncthreads();
    float* red = (float*)smem;                                       // [NT][CPC][2] floats (the box is dead)
#pragma unroll
    for (int j = 0; j < CPC; ++j) { red[(tid * CPC + j) * 2] = s1[j]; red[(tid * CPC + j) * 2 + 1] = s2[j]; }
    __syncthreads();
    // two levels: thread (channel ch, eighth e8) adds the voxel lanes e8, e8 + 8, ..; 32 threads add the eight sums
    {
      const int ch = tid & (GR_C - 1), e8 = tid >> 5, c2 = ch / CPC, j = ch % CPC;
      float a1 = 0.f, a2 = 0.f;
      for (int qv = e8; qv < VLC; qv += NT / GR_C) { a1 += red[((qv * GCH + c2) * CPC + j) * 2]; a2 += red[((qv * GCH + c2) * CPC + j) * 2 + 1]; }
      __syncthreads();
      red[(e8 * GR_C + ch) * 2] = a1; red[(e8 * GR_C + ch) * 2 + 1] = a2;
      __syncthreads();
    }
    if (tid < GR_C && cg0 + tid < Cl) {
      const int c2 = tid / CPC, j = tid % CPC;
      float a1 = 0.f, a2 = 0.f;
      for (int qv = 0; qv < NT / GR_C; ++qv) { a1 += red[(qv * GR_C + tid) * 2]; a2 += red[(qv * GR_C + tid) * 2 + 1]; }
      float shc[CPC];
      Elem<T>::unpack(ld_chunk<T>(low, (size_t)n * Dl * Hl * Wl * Cl + cg0 + c2 * CPC), shc);
      const size_t o = (((size_t)n * P + blockIdx.x) * Cl + cg0 + tid) * 3;
      partials[o] = shc[j]; partials[o + 1] = a1; partials[o + 2] = a2;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(NT) k_slice_copy(const void* __restrict__ src, int64_t src_stride, int c_off,
                                                   void* __restrict__ dst, int C, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int cc = (int)(i % cch);
    int64_t r = i / cch;
    st_chunk<T>(dst, (size_t)r * C + (size_t)cc * CPC,
                ld_chunk<T>(src, (size_t)r * src_stride + c_off + (size_t)cc * CPC));
  }
}


// ---- trilinear(align_corners=True) resize of float32 NCDHW planes ---------------------------------
// MedFormer's auxiliary head: F.interpolate(aux_out, size=x.shape[-3:], 'trilinear', align_corners=True)
// (/root/reference/model/dim3/medformer.py:91).  One work item = one output (fwd) / input (bwd) element.
__global__ void __launch_bounds__(NT) k_trilinear_planes_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                             int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                             int64_t total) {
  const float sd = lin_scale(Di, Do), sh = lin_scale(Hi, Ho), sw = lin_scale(Wi, Wo);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int64_t q = i;
    int w = (int)(q % Wo); q /= Wo;
    int h = (int)(q % Ho); q /= Ho;
    int d = (int)(q % Do);
    int64_t pl = q / Do;
    Lin ld = lin_src(d, sd, Di), lh = lin_src(h, sh, Hi), lw = lin_src(w, sw, Wi);
    const float* p = x + (size_t)pl * Di * Hi * Wi;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int dd = a ? ld.i1 : ld.i0;
      float wa = a ? ld.l1 : ld.l0, pa = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int hh = b ? lh.i1 : lh.i0;
        float wb = b ? lh.l1 : lh.l0;
        const float* r = p + ((size_t)dd * Hi + hh) * Wi;
        pa += wb * (lw.l0 * r[lw.i0] + lw.l1 * r[lw.i1]);
      }
      acc += wa * pa;
    }
    y[i] = acc;
  }
}

// Round 6: the same resize for Wo % 4 == 0, four consecutive outputs of a row per thread: one (plane, d) slab per blockIdx.y (no
// 64-bit index decomposition per element — the flat kernel does three per 4-byte store), the d / h interpolation set up once per
// thread, a 16-byte store.  The arithmetic per output is the flat kernel's expression: bit-identical results.  MedFormer's 16-plane
// 64^3 -> 128^3 auxiliary head: 204 us -> (profiles/r06_w_*).
__global__ void __launch_bounds__(NT) k_trilinear_planes_fwd4(const float* __restrict__ x, float* __restrict__ y,
                                                              int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  const float sd = lin_scale(Di, Do), sh = lin_scale(Hi, Ho), sw = lin_scale(Wi, Wo);
  const unsigned w4n = (unsigned)Wo / 4u, items = (unsigned)Ho * w4n;
  const unsigned idx = blockIdx.x * NT + threadIdx.x;
  if (idx >= items) return;
  const unsigned h = idx / w4n, w0 = (idx - h * w4n) * 4u;
  const unsigned pl = blockIdx.y / (unsigned)Do, d = blockIdx.y - pl * (unsigned)Do;
  const Lin ld = lin_src((int)d, sd, Di), lh = lin_src((int)h, sh, Hi);
  const float* p = x + (size_t)pl * Di * Hi * Wi;
  float out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const Lin lw = lin_src((int)w0 + j, sw, Wi);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int dd = a ? ld.i1 : ld.i0;
      float wa = a ? ld.l1 : ld.l0, pa = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int hh = b ? lh.i1 : lh.i0;
        float wb = b ? lh.l1 : lh.l0;
        const float* r = p + ((size_t)dd * Hi + hh) * Wi;
        pa += wb * (lw.l0 * r[lw.i0] + lw.l1 * r[lw.i1]);
      }
      acc += wa * pa;
    }
    out[j] = acc;
  }
  *(f32x4*)(y + (((size_t)pl * Do + d) * Ho + h) * Wo + w0) = f32x4{out[0], out[1], out[2], out[3]};
}

__global__ void __launch_bounds__(NT) k_trilinear_planes_bwd(const float* __restrict__ dy, float* __restrict__ dx,
                                                             int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                             int64_t total) {
  const float sd = lin_scale(Di, Do), sh = lin_scale(Hi, Ho), sw = lin_scale(Wi, Wo);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int64_t q = i;
    int wl = (int)(q % Wi); q /= Wi;
    int hl = (int)(q % Hi); q /= Hi;
    int dl = (int)(q % Di);
    int64_t pl = q / Di;
    int d0, d1, h0, h1, w0, w1;
    dst_range(dl, sd, Do, d0, d1);
    dst_range(hl, sh, Ho, h0, h1);
    dst_range(wl, sw, Wo, w0, w1);
    const float* p = dy + (size_t)pl * Do * Ho * Wo;
    float acc = 0.f;
    for (int d = d0; d <= d1; ++d) {
      float wd = lin_weight_to(d, sd, Di, dl);
      if (wd == 0.f) continue;
      for (int h = h0; h <= h1; ++h) {
        float wh = lin_weight_to(h, sh, Hi, hl);
        if (wh == 0.f) continue;
        const float* r = p + ((size_t)d * Ho + h) * Wo;
        float rowacc = 0.f;
        for (int w = w0; w <= w1; ++w) {
          float ww = lin_weight_to(w, sw, Wi, wl);
          if (ww != 0.f) rowacc += ww * r[w];
        }
        acc += wd * wh * rowacc;
      }
    }
    dx[i] = acc;
  }
}

static inline int grid_for(int64_t items) {
  int64_t b = (items + NT - 1) / NT;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
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

static int check_c(int dtype, int C, const char* what) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(C > 0 && C % cpc == 0, CBIM_EUNSUPPORTED, "%s channel count %d is not a multiple of %d", what, C, cpc);
  return 0;
}

extern "C" int cbim_maxpool3d_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int D, int H,
                                  int W, int C, int sD, int sH, int sW, void* stream) {
  if (int e = check_c(dtype, C, "maxpool")) return e;
  CBIM_CHECK(sD >= 1 && sH >= 1 && sW >= 1 && sD * sH * sW <= 255, CBIM_EUNSUPPORTED, "pool window too large");
  int Do = D / sD, Ho = H / sH, Wo = W / sW;
  CBIM_CHECK(Do >= 1 && Ho >= 1 && Wo >= 1, CBIM_EINVAL, "pool output empty");
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int64_t total = (int64_t)N * Do * Ho * Wo * (C / cpc);
  DISPATCH_T(dtype, k_maxpool_fwd, dim3(grid_for(total)), (hipStream_t)stream, x, y, idx, D, H, W, C, sD, sH,
             sW, Do, Ho, Wo, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_maxpool3d_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int D,
                                  int H, int W, int C, int sD, int sH, int sW, void* stream) {
  if (int e = check_c(dtype, C, "maxpool")) return e;
  int Do = D / sD, Ho = H / sH, Wo = W / sW;
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int64_t total = (int64_t)N * D * H * W * (C / cpc);
  const int64_t per_plane = (int64_t)H * W * (C / cpc);
  const int64_t planes = (int64_t)N * D;
  CBIM_CHECK(per_plane < ((int64_t)1 << 31) && planes < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "maxpool_bwd: plane of %lld chunks x %lld planes", (long long)per_plane, (long long)planes);
  (void)total;
  int64_t bx = (per_plane + NT - 1) / NT;
  if (bx > 64) bx = 64;
  DISPATCH_T(dtype, k_maxpool_bwd, dim3((unsigned)bx, (unsigned)(planes < 65535 ? planes : 65535)), (hipStream_t)stream, dy, idx, dx, D, H, W, C, sD,
             sH, sW, Do, Ho, Wo, planes);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_upcat_fwd(int dtype, const void* low, const void* skip, void* out, int N, int Dl, int Hl,
                              int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream) {
  if (int e = check_c(dtype, Cl, "upcat low")) return e;
  if (Cs > 0) if (int e = check_c(dtype, Cs, "upcat skip")) return e;
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int64_t total = (int64_t)N * D * H * W * ((Cs + Cl) / cpc);
  DISPATCH_T(dtype, k_upcat_fwd, dim3(grid_for(total)), (hipStream_t)stream, low, skip, out, Dl, Hl, Wl, Cl,
             D, H, W, Cs, skip_first, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_stats_parts(int64_t S, int C);
extern "C" int cbim_stats_finalize(const float* partials, int N, int P, int C, double count, float eps, int mode,
                                   float* out, void* stream);

extern "C" int cbim_upcat_fwd_stats(int dtype, const void* low, const void* skip, void* out, int N, int Dl, int Hl,
                                    int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, float eps,
                                    float* partials, int P, float* stats, void* stream) {
  if (int e = check_c(dtype, Cl, "upcat low")) return e;
  if (Cs > 0) if (int e = check_c(dtype, Cs, "upcat skip")) return e;
  const int cpc = dtype == CBIM_BF16 ? 8 : 4, Ct = Cs + Cl;
  const int64_t S = (int64_t)D * H * W;
  CBIM_CHECK(partials && stats && P == cbim_stats_parts(S, Ct), CBIM_EINVAL, "upcat_fwd_stats: partials must have cbim_stats_parts(S, Cs+Cl) records");
  CBIM_CHECK(Ct / cpc <= NT && S < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "upcat_fwd_stats: %d channels / %lld voxels unsupported", Ct, (long long)S);
  dim3 grid((unsigned)P, (unsigned)N);
  DISPATCH_T(dtype, k_upcat_fwd_stats, grid, (hipStream_t)stream, low, skip, out, Dl, Hl, Wl, Cl, D, H, W, Cs, skip_first, P,
             partials);
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  return cbim_stats_finalize(partials, N, P, Ct, (double)S, eps, 0, stats, stream);
}

extern "C" int cbim_upcat_bwd(int dtype, const void* dout, void* dlow, void* dskip, int N, int Dl, int Hl,
                              int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream) {
  if (int e = check_c(dtype, Cl, "upcat low")) return e;
  if (Cs > 0) if (int e = check_c(dtype, Cs, "upcat skip")) return e;
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int Ct = Cs + Cl;
  int low_lo = skip_first ? Cs : 0, skip_lo = skip_first ? 0 : Cl;
  hipStream_t st = (hipStream_t)stream;
  int64_t total = (int64_t)N * Dl * Hl * Wl * (Cl / cpc);
  if (dlow)   // NULL: the caller runs the separable adjoint (cbim_lin_adjoint_axis) itself
    DISPATCH_T(dtype, k_upcat_bwd_low, dim3(grid_for(total)), st, dout, dlow, Dl, Hl, Wl, Cl, D, H, W, Ct,
               low_lo, total);
  if (Cs > 0 && dskip) {
    int64_t t2 = (int64_t)N * D * H * W * (Cs / cpc);
    DISPATCH_T(dtype, k_slice_copy, dim3(grid_for(t2)), st, dout, (int64_t)Ct, skip_lo, dskip, Cs, t2);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

/* statistics (eps) of the virtual up-sampled tensor up(low) [N, D, H, W, Cl] — nothing is written but the records */
extern "C" int cbim_up_stats(int dtype, const void* low, int N, int Dl, int Hl, int Wl, int Cl, int D, int H, int W, float eps,
                             float* partials, int P, float* stats, void* stream) {
  if (int e = check_c(dtype, Cl, "up_stats low")) return e;
  const int cpc = dtype == CBIM_BF16 ? 8 : 4;
  const int64_t S = (int64_t)D * H * W;
  CBIM_CHECK(partials && stats && P == cbim_stats_parts(S, Cl), CBIM_EINVAL, "up_stats: partials must have cbim_stats_parts(S, Cl) records");
  CBIM_CHECK(Cl / cpc <= NT && S < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "up_stats: %d channels / %lld voxels unsupported", Cl, (long long)S);
  dim3 grid((unsigned)P, (unsigned)N);
  DISPATCH_T(dtype, k_upcat_fwd_stats, grid, (hipStream_t)stream, low, (const void*)nullptr, (void*)nullptr, Dl, Hl, Wl, Cl, D, H, W, 0, 1, P,
             partials);
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  return cbim_stats_finalize(partials, N, P, Cl, (double)S, eps, 0, stats, stream);
}

/* statistics (eps) of the virtual up-sampled tensor from the coarse grid (k_up_gram_stats): partials float [N][P][Cl][3],
   P = cbim_up_gram_parts(Dl, Hl, Wl) */
extern "C" int cbim_up_gram_parts(int Dl, int Hl, int Wl) {
  const int64_t tiles = (int64_t)((Dl + GR_D - 1) / GR_D) * ((Hl + GR_H - 1) / GR_H) * ((Wl + GR_W - 1) / GR_W);
  return (int)(tiles < 1024 ? tiles : 1024);
}
extern "C" int cbim_up_stats_gram(int dtype, const void* low, int N, int Dl, int Hl, int Wl, int Cl, int D, int H, int W, float eps,
                                  float* partials, int P, float* stats, void* stream) {
  if (int e = check_c(dtype, Cl, "up_stats_gram low")) return e;
  CBIM_CHECK(low && partials && stats && P == cbim_up_gram_parts(Dl, Hl, Wl), CBIM_EINVAL, "up_stats_gram: partials must have cbim_up_gram_parts records");
  CBIM_CHECK(D >= Dl && H >= Hl && W >= Wl && (int64_t)D * H * W < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "up_stats_gram: an up-sampling is expected");
  hipStream_t st = (hipStream_t)stream;
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute((const void*)k_up_gram_stats<bf16_tag>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute((const void*)k_up_gram_stats<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e1 == hipSuccess && e2 == hipSuccess, CBIM_ELAUNCH, "up_stats_gram: cannot raise the dynamic LDS limit");
    attr_done = true;
  }
#endif
  dim3 grid((unsigned)P, (unsigned)N, (unsigned)((Cl + GR_C - 1) / GR_C));
  if (dtype == CBIM_BF16) CBIM_LAUNCH((k_up_gram_stats<bf16_tag>), grid, dim3(NT), GR_SMEM, st, low, Dl, Hl, Wl, Cl, D, H, W, P, partials);
  else CBIM_LAUNCH((k_up_gram_stats<float>), grid, dim3(NT), GR_SMEM, st, low, Dl, Hl, Wl, Cl, D, H, W, P, partials);
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  return cbim_stats_finalize(partials, N, P, Cl, (double)D * H * W, eps, 2, stats, stream);
}

extern "C" int cbim_upcat_act_fwd(int dtype, const void* low, const void* skip, const float* stats, void* out, int N, int Dl,
                                  int Hl, int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, int act, void* stream) {
  if (int e = check_c(dtype, Cl, "upcat low")) return e;
  if (int e = check_c(dtype, Cs, "upcat skip")) return e;
  const int cpc = dtype == CBIM_BF16 ? 8 : 4, Ct = Cs + Cl;
  const int64_t S = (int64_t)D * H * W;
  CBIM_CHECK(low && skip && stats && out, CBIM_EINVAL, "null argument");
  CBIM_CHECK(Ct / cpc <= NT && S < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "upcat_act_fwd: %d channels / %lld voxels unsupported", Ct, (long long)S);
  const int P = cbim_stats_parts(S, Ct);
  dim3 grid((unsigned)P, (unsigned)N);
  DISPATCH_T(dtype, k_upcat_act_fwd, grid, (hipStream_t)stream, low, skip, stats, out, Dl, Hl, Wl, Cl, D, H, W, Cs, skip_first, P, act);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_upcat_norm_bwd(int dtype, const void* g, const void* low, const void* skip, const float* stats,
                                   const float* sums, void* dskip, void* dlow, void* dup_scratch, int N, int Dl, int Hl, int Wl,
                                   int Cl, int D, int H, int W, int Cs, int skip_first, void* stream) {
  if (int e = check_c(dtype, Cl, "upcat low")) return e;
  if (int e = check_c(dtype, Cs, "upcat skip")) return e;
  const int cpc = dtype == CBIM_BF16 ? 8 : 4, Ct = Cs + Cl;
  const int64_t S = (int64_t)D * H * W;
  CBIM_CHECK(g && low && skip && stats && sums && dskip && dup_scratch, CBIM_EINVAL, "null argument");
  CBIM_CHECK(Ct / cpc <= NT && S < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "upcat_norm_bwd: %d channels / %lld voxels unsupported", Ct, (long long)S);
  const int P = cbim_stats_parts(S, Ct);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)P, (unsigned)N);
  DISPATCH_T(dtype, k_upcat_norm_bwd, grid, st, g, low, skip, stats, sums, dskip, dup_scratch, Dl, Hl, Wl, Cl, D, H, W, Cs, skip_first, P);
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  if (!dlow) return CBIM_OK;   // the caller reduces dup_scratch with cbim_lin_adjoint_axis
  const int64_t total = (int64_t)N * Dl * Hl * Wl * (Cl / cpc);
  DISPATCH_T(dtype, k_upcat_bwd_low, dim3(grid_for(total)), st, (const void*)dup_scratch, dlow, Dl, Hl, Wl, Cl, D, H, W, Cl, 0, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_trilinear_planes_fwd(const float* x, float* y, int planes, int Di, int Hi, int Wi, int Do, int Ho,
                                         int Wo, void* stream) {
  CBIM_CHECK(planes >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && Do >= 1 && Ho >= 1 && Wo >= 1, CBIM_EINVAL,
             "trilinear: empty extent");
  int64_t total = (int64_t)planes * Do * Ho * Wo;
  if (Wo % 4 == 0 && (int64_t)planes * Do <= 65535 && (((uintptr_t)y) & 15u) == 0) {
    const unsigned items = (unsigned)Ho * (unsigned)(Wo / 4);
    CBIM_LAUNCH(k_trilinear_planes_fwd4, dim3((items + NT - 1) / NT, (unsigned)(planes * Do)), dim3(NT), 0, (hipStream_t)stream, x, y, Di, Hi,
                Wi, Do, Ho, Wo);
    return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
  }
  CBIM_LAUNCH(k_trilinear_planes_fwd, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, x, y, Di, Hi, Wi, Do, Ho,
              Wo, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_trilinear_planes_bwd(const float* dy, float* dx, int planes, int Di, int Hi, int Wi, int Do, int Ho,
                                         int Wo, void* stream) {
  CBIM_CHECK(planes >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && Do >= 1 && Ho >= 1 && Wo >= 1, CBIM_EINVAL,
             "trilinear: empty extent");
  int64_t total = (int64_t)planes * Di * Hi * Wi;
  CBIM_LAUNCH(k_trilinear_planes_bwd, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, dy, dx, Di, Hi, Wi, Do,
              Ho, Wo, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

/* one axis of the transposed trilinear interpolation: src [outer][F][src_row elements, of which `inner` from c_off are used]
   -> dst [outer][L][inner] dense.  vec = elements per work item: 0 = one 16-byte chunk of dtype, 1 = scalar (float32 only). */
extern "C" int cbim_lin_adjoint_axis(int dtype, int vec, const void* src, int64_t src_row, int64_t c_off, void* dst, int64_t outer,
                                     int F, int L, int64_t inner, void* stream) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(src && dst && outer >= 1 && F >= 1 && L >= 1 && inner >= 1, CBIM_EINVAL, "lin_adjoint_axis: bad extents");
  CBIM_CHECK(vec == 0 || (vec == 1 && dtype == CBIM_F32), CBIM_EINVAL, "lin_adjoint_axis: scalar items are float32 only");
  const int V = vec == 1 ? 1 : (dtype == CBIM_BF16 ? 8 : 4);
  CBIM_CHECK(inner % V == 0 && src_row % V == 0 && c_off % V == 0 && c_off + inner <= src_row, CBIM_EUNSUPPORTED,
             "lin_adjoint_axis: inner %lld / row %lld / offset %lld not multiples of %d", (long long)inner, (long long)src_row,
             (long long)c_off, V);
  const int64_t items = inner / V, total = outer * L * items;
  hipStream_t st = (hipStream_t)stream;
  if (vec == 1)
    CBIM_LAUNCH((k_lin_adjoint_axis<float, 1>), dim3(grid_for(total)), dim3(NT), 0, st, src, dst, F, L, items, src_row, c_off, total);
  else if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_lin_adjoint_axis<bf16_tag, 8>), dim3(grid_for(total)), dim3(NT), 0, st, src, dst, F, L, items, src_row, c_off, total);
  else
    CBIM_LAUNCH((k_lin_adjoint_axis<float, 4>), dim3(grid_for(total)), dim3(NT), 0, st, src, dst, F, L, items, src_row, c_off, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// ---- attention gate: y = x * psi (AttentionBlock.forward's `x * psi`, /root/reference/model/dim3/
// attention_unet_utils.py:35; psi is one value per voxel) ---------------------------------------------------
namespace cbim {
template <typename T>
__global__ void __launch_bounds__(NT) k_gate_fwd(const void* __restrict__ x, const float* __restrict__ psi,
                                                 void* __restrict__ y, int C, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int64_t row = i / cch;
    float f[CPC];
    Elem<T>::unpack(ld_chunk<T>(x, (size_t)i * CPC), f);
    const float s = psi[row];
#pragma unroll
    for (int j = 0; j < CPC; ++j) f[j] *= s;
    st_chunk<T>(y, (size_t)i * CPC, Elem<T>::pack(f));
  }
}
// dx = dy * psi; dpsi[row] = sum_c dy[row,c] * x[row,c]   (one thread per voxel row)
template <typename T>
__global__ void __launch_bounds__(NT) k_gate_bwd(const void* __restrict__ dy, const void* __restrict__ x,
                                                 const float* __restrict__ psi, void* __restrict__ dx,
                                                 float* __restrict__ dpsi, int C, int64_t rows) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
    const float s = psi[r];
    float acc = 0.f;
    for (int cc = 0; cc < cch; ++cc) {
      float g[CPC], f[CPC];
      Elem<T>::unpack(ld_chunk<T>(dy, ((size_t)r * cch + cc) * CPC), g);
      Elem<T>::unpack(ld_chunk<T>(x, ((size_t)r * cch + cc) * CPC), f);
#pragma unroll
      for (int j = 0; j < CPC; ++j) { acc += g[j] * f[j]; g[j] *= s; }
      st_chunk<T>(dx, ((size_t)r * cch + cc) * CPC, Elem<T>::pack(g));
    }
    dpsi[r] = acc;
  }
}
}  // namespace cbim

extern "C" int cbim_gate_fwd(int dtype, const void* x, const float* psi, void* y, int64_t rows, int C, void* stream) {
  if (int e = check_c(dtype, C, "gate")) return e;
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int64_t total = rows * (C / cpc);
  DISPATCH_T(dtype, k_gate_fwd, dim3(grid_for(total)), (hipStream_t)stream, x, psi, y, C, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_gate_bwd(int dtype, const void* dy, const void* x, const float* psi, void* dx, float* dpsi,
                             int64_t rows, int C, void* stream) {
  if (int e = check_c(dtype, C, "gate")) return e;
  DISPATCH_T(dtype, k_gate_bwd, dim3(grid_for(rows)), (hipStream_t)stream, dy, x, psi, dx, dpsi, C, rows);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(pool_up)
