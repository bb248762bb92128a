// norm_kernels.hip — InstanceNorm3d statistics / normalise+activation / backward pieces.
//
// HBM-bound streaming kernels over channels-last [voxel][C] tensors.  Work item = one 16-byte
// channel chunk (8 bf16 / 4 f32); a thread keeps a FIXED channel chunk and strides over voxels,
// so per-channel sums stay in registers; lanes of a wave read consecutive chunks (1 KiB per
// wave instruction when C*esize >= 1 KiB, whole rows otherwise).
//
// Reference semantics: nn.InstanceNorm3d(C, eps=1e-4), affine=False, biased variance
// (/root/reference/model/dim3/conv_layers.py:40-42, model/dim3/utils.py:15-21);
// backward dx = rstd*(dy - mean(dy) - xh*mean(dy*xh)).
#include "cbim_common.h"

namespace cbim {

static constexpr int NT = 256;

struct RowMap {  // thread -> (channel chunk, voxel lane)
  int cch;       // chunks per row
  int vl_count;  // voxel lanes per block
};

// ---- partial sums: (sum u, sum u*v) per channel over a slab of voxels -----------------------------
// MODE 0: u = x, v = x                      (forward statistics)
// MODE 1: u = g', v = xh  with xh=(x-mean)*rstd, g' = masked ? g*act'(xh) : g
// AFF (MODE 1, round 5): a per-channel affine between the normalisation and the activation, z = gamma * xh + beta
//      (affine float [Cl][2]): the mask is act'(z); the sums stay those of g' and g' * xh (d beta, d gamma of BatchNorm)
template <typename T, int MODE, bool AFF = false>
__global__ void __launch_bounds__(NT) k_partial_sums(const void* __restrict__ a, int64_t a_stride,
                                                     const void* __restrict__ x, int64_t x_stride,
                                                     const float* __restrict__ stats, int64_t S, int C,
                                                     int P, int act, int masked,
                                                     float* __restrict__ partials, int Cl, int c_off,
                                                     const float* __restrict__ affine = nullptr) {
  // C = channels handled by this launch (<= NT chunks), starting at channel c_off of a Cl-channel tensor
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC;
  const int vlc = NT / cch;            // host guarantees cch <= NT
  const int t = threadIdx.x;
  const int cc = t % cch, vl = t / cch;
  const int part = blockIdx.x, n = blockIdx.y;
  const int64_t per = (S + P - 1) / P;
  const int64_t v0 = (int64_t)part * per;
  int64_t v1 = v0 + per;
  if (v1 > S) v1 = S;

  float s0[CPC], s1[CPC], mean[CPC], rstd[CPC], shift[CPC], gam[CPC], bet[CPC];
  float cnt = 0.f;
#pragma unroll
  for (int j = 0; j < CPC; ++j) { s0[j] = 0.f; s1[j] = 0.f; mean[j] = 0.f; rstd[j] = 1.f; shift[j] = 0.f; gam[j] = 1.f; bet[j] = 0.f; }
  const bool active = vl < vlc;
  if (MODE == 1 && active) {
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      mean[j] = stats[((size_t)n * Cl + c_off + cc * CPC + j) * 2 + 0];
      rstd[j] = stats[((size_t)n * Cl + c_off + cc * CPC + j) * 2 + 1];
      if (AFF) { gam[j] = affine[(size_t)(c_off + cc * CPC + j) * 2]; bet[j] = affine[(size_t)(c_off + cc * CPC + j) * 2 + 1]; }
    }
  }
  const size_t nb = (size_t)n * S;
  if (MODE == 0 && active && v0 < v1) {
    // ONE shift per channel for the whole part (its first voxel, read by every lane of the channel chunk: a cache hit):
    // shifted sums of different lanes then simply add — no per-lane Chan merge (64 lanes x 8 channels of dependent
    // divisions on 4 threads was most of this kernel on a 32-channel tensor)
    Elem<T>::unpack(ld_chunk<T>(a, (nb + v0) * a_stride + c_off + (size_t)cc * CPC), shift);
  }
  if (active) {
    auto accumulate = [&](const u32x4& ra, const u32x4& rx) {
      float fa[CPC];
      Elem<T>::unpack(ra, fa);
      if (MODE == 0) {
        cnt += 1.f;
#pragma unroll
        for (int j = 0; j < CPC; ++j) { float d = fa[j] - shift[j]; s0[j] += d; s1[j] = fmaf(d, d, s1[j]); }
      } else {
        float fx[CPC];
        Elem<T>::unpack(rx, fx);
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          float xh = (fx[j] - mean[j]) * rstd[j];
          float g = masked ? fa[j] * act_grad(AFF ? fmaf(gam[j], xh, bet[j]) : xh, act) : fa[j];
          s0[j] += g;
          s1[j] = fmaf(g, xh, s1[j]);
        }
      }
    };
    // four rows per trip: all loads of a trip are issued before any is consumed (streaming kernel, latency bound
    // otherwise); rows are accumulated in the same order as a one-row loop
    constexpr int RPT = 4;
    for (int64_t v = v0 + vl; v < v1; v += RPT * vlc) {
      u32x4 ra[RPT], rx[RPT];
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int64_t vv = v + (int64_t)u * vlc;
        if (vv < v1) {
          ra[u] = ld_chunk<T>(a, (nb + vv) * a_stride + c_off + (size_t)cc * CPC);
          if (MODE == 1) rx[u] = ld_chunk<T>(x, (nb + vv) * x_stride + c_off + (size_t)cc * CPC);
          else rx[u] = ra[u];
        }
      }
#pragma unroll
      for (int u = 0; u < RPT; ++u)
        if (v + (int64_t)u * vlc < v1) accumulate(ra[u], rx[u]);
    }
  }
  // reduce over the voxel lanes through LDS: lane sums [t][2*CPC + 1], then one thread per (channel, quantity) adds the
  // vlc lanes in fixed order (deterministic)
  constexpr int W = 2 * CPC + 1;
  __shared__ float red[NT * (2 * 8 + 1)];
#pragma unroll
  for (int j = 0; j < CPC; ++j) { red[t * W + j] = s0[j]; red[t * W + CPC + j] = s1[j]; }
  red[t * W + 2 * CPC] = cnt;
  __syncthreads();
  for (int ch = t; ch < cch * CPC; ch += NT) {          // thread = one channel of this launch (wide tensors: several)
    const int c2 = ch / CPC, j = ch % CPC;
    float a0 = 0.f, a1 = 0.f, ac = 0.f;
    for (int q = 0; q < vlc; ++q) {
      const float* r = red + (q * cch + c2) * W;
      a0 += r[j]; a1 += r[CPC + j]; ac += r[2 * CPC];
    }
    Moments acc;
    if (MODE == 0) {
      // every lane of the chunk used the same shift: the part's first voxel
      float sh[CPC];
      Elem<T>::unpack(v0 < v1 ? ld_chunk<T>(a, (nb + v0) * a_stride + c_off + (size_t)c2 * CPC) : u32x4{0u, 0u, 0u, 0u}, sh);
      acc = moments_from_shifted(ac, sh[j], a0, a1);
    } else {
      acc.n = 0.f; acc.mean = a0; acc.m2 = a1;
    }
    const size_t o = (((size_t)n * P + part) * Cl + c_off + c2 * CPC + j) * 3;
    partials[o] = acc.n;
    partials[o + 1] = acc.mean;
    partials[o + 2] = acc.m2;
  }
}

// One workgroup per (n, c): FIN_T threads stride over the P records (independent loads in flight),
// then a fixed-shape tree merge through LDS — deterministic, and no serial walk over 8192 tiles.
static constexpr int FIN_T = 128;
__global__ void __launch_bounds__(FIN_T) k_stats_finalize(const float* __restrict__ partials, int N, int P,
                                                          int C, double count, float eps, int mode,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x;  // n*C + c
  const int n = i / C, c = i % C;
  const int t = threadIdx.x;
  __shared__ double red[FIN_T * 3];
  double na = 0.0, ma = 0.0, M2 = 0.0;
  const float* base = partials + ((size_t)n * P * C + c) * 3;
  for (int p = t; p < P; p += FIN_T) {
    const float* r = base + (size_t)p * C * 3;
    double nb = (double)r[0], mb = (double)r[1], Mb = (double)r[2];
    if (mode == 0) {
      if (nb > 0.0) {
        double nn = na + nb, d = mb - ma;
        ma += d * (nb / nn);
        M2 += Mb + d * d * (na * nb / nn);
        na = nn;
      }
    } else {
      ma += mb;
      M2 += Mb;
    }
  }
  red[t * 3] = na; red[t * 3 + 1] = ma; red[t * 3 + 2] = M2;
  __syncthreads();
  for (int s = FIN_T / 2; s > 0; s >>= 1) {
    if (t < s) {
      double nb = red[(t + s) * 3], mb = red[(t + s) * 3 + 1], Mb = red[(t + s) * 3 + 2];
      double xa = red[t * 3], xm = red[t * 3 + 1], xM = red[t * 3 + 2];
      if (mode == 0) {
        double nn = xa + nb;
        if (nn > 0.0) {
          double d = mb - xm;
          xm += d * (nb / nn);
          xM += Mb + d * d * (xa * nb / nn);
        }
        xa = nn;
      } else {
        xm += mb;
        xM += Mb;
      }
      red[t * 3] = xa; red[t * 3 + 1] = xm; red[t * 3 + 2] = xM;
    }
    __syncthreads();
  }
  if (t == 0) {
    if (mode == 0) {
      double var = red[0] > 0.0 ? red[2] / red[0] : 0.0;
      out[(size_t)i * 2] = (float)red[1];
      out[(size_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    } else if (mode == 1) {
      out[(size_t)i * 2] = (float)(red[1] / count);
      out[(size_t)i * 2 + 1] = (float)(red[2] / count);
    } else {
      // mode 2: records (shift, sum, sum of squares) of x - shift (k_up_gram_stats): mean and 1 / sqrt(var + eps)
      const double m = red[1] / count;
      double var = red[2] / count - m * m;
      if (var < 0.0) var = 0.0;
      out[(size_t)i * 2] = (float)((double)base[0] + m);
      out[(size_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
}

// ---- y = act((x-mean)*rstd) -------------------------------------------------------------------------
// Streaming kernels: a thread keeps one channel chunk (its statistics live in registers) and strides
// over voxel rows, 4 rows per trip so that 4 independent 16-byte loads per tensor are in flight;
// no integer division in the loop.  grid = (row blocks, N).
static constexpr int RU = 4;
// AFF (round 5): y = act(gamma * (x-mean)*rstd + beta), affine float [Cl][2] — nn.BatchNorm3d / ContBatchNorm3d (affine norms of
//      the `norm: bn` constructor branch, /root/reference/model/dim3/utils.py:15-21; vnet.py:22-33)
template <typename T, bool AFF = false>
__global__ void __launch_bounds__(NT) k_norm_act_fwd(const void* __restrict__ x, int64_t x_stride,
                                                     const float* __restrict__ stats, void* __restrict__ y,
                                                     int64_t y_stride, int64_t S, int C, int act, int Cl,
                                                     int c_off, const float* __restrict__ affine = nullptr) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC, vlc = NT / cch;
  const int cc = threadIdx.x % cch + c_off / CPC, vl = threadIdx.x / cch;
  if (vl >= vlc) return;
  const int n = blockIdx.y;
  float mean[CPC], rstd[CPC], ga[CPC], be[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    mean[j] = stats[((size_t)n * Cl + cc * CPC + j) * 2];
    rstd[j] = stats[((size_t)n * Cl + cc * CPC + j) * 2 + 1];
    ga[j] = 1.f; be[j] = 0.f;
    if (AFF) {      // z = gamma * xh + beta with xh = (x - mean) * rstd: the CENTRED form, the very expression the backward recomputes
      ga[j] = affine[(size_t)(cc * CPC + j) * 2];   // (the folded form x * (gamma rstd) - (mean gamma rstd - beta) cancels for
      be[j] = affine[(size_t)(cc * CPC + j) * 2 + 1];   //  |mean| >> std and lets the activation mask disagree with the backward's near 0)
    }
  }
  const size_t nb = (size_t)n * S;
  const int64_t step = (int64_t)gridDim.x * vlc;
  for (int64_t v = (int64_t)blockIdx.x * vlc + vl; v < S; v += step * RU) {
    u32x4 raw[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (v + u * step < S) raw[u] = ld_chunk<T>(x, (nb + v + u * step) * x_stride + (size_t)cc * CPC);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (v + u * step < S) {
        float f[CPC];
        Elem<T>::unpack(raw[u], f);
#pragma unroll
        for (int j = 0; j < CPC; ++j) f[j] = act_fwd(AFF ? fmaf(ga[j], (f[j] - mean[j]) * rstd[j], be[j]) : (f[j] - mean[j]) * rstd[j], act);
        st_chunk<T>(y, (nb + v + u * step) * y_stride + (size_t)cc * CPC, Elem<T>::pack(f));
      }
    }
  }
}

// ---- dx = rstd*(g' - m1 - xh*m2) [+ add] --------------------------------------------------------------
// AFF: z = gamma * xh + beta between normalisation and activation: g' = g * act'(z) * gamma, and `sums` are the means of
//      gamma g' and gamma g' xh (the caller scales the batch means of cbim_norm_affine_bwd_reduce by gamma — no division)
template <typename T, bool AFF = false>
__global__ void __launch_bounds__(NT) k_norm_bwd_apply(const void* __restrict__ g, int64_t g_stride,
                                                       const void* __restrict__ x, int64_t x_stride,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ sums,
                                                       const void* __restrict__ add, int64_t add_stride,
                                                       void* __restrict__ dx, int64_t dx_stride, int64_t S,
                                                       int C, int act, int masked, int Cl, int c_off,
                                                       const float* __restrict__ affine = nullptr) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC, vlc = NT / cch;
  const int cc = threadIdx.x % cch + c_off / CPC, vl = threadIdx.x / cch;
  if (vl >= vlc) return;
  const int n = blockIdx.y;
  float mean[CPC], rstd[CPC], m1[CPC], m2[CPC], gam[CPC], bet[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    mean[j] = stats[((size_t)n * Cl + cc * CPC + j) * 2];
    rstd[j] = stats[((size_t)n * Cl + cc * CPC + j) * 2 + 1];
    m1[j] = sums[((size_t)n * Cl + cc * CPC + j) * 2];
    m2[j] = sums[((size_t)n * Cl + cc * CPC + j) * 2 + 1];
    gam[j] = AFF ? affine[(size_t)(cc * CPC + j) * 2] : 1.f;
    bet[j] = AFF ? affine[(size_t)(cc * CPC + j) * 2 + 1] : 0.f;
  }
  const size_t nb = (size_t)n * S;
  const int64_t step = (int64_t)gridDim.x * vlc;
  for (int64_t v = (int64_t)blockIdx.x * vlc + vl; v < S; v += step * RU) {
    u32x4 rg[RU], rx[RU], ra[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (v + u * step < S) {
        rg[u] = ld_chunk<T>(g, (nb + v + u * step) * g_stride + (size_t)cc * CPC);
        rx[u] = ld_chunk<T>(x, (nb + v + u * step) * x_stride + (size_t)cc * CPC);
        if (add) ra[u] = ld_chunk<T>(add, (nb + v + u * step) * add_stride + (size_t)cc * CPC);
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (v + u * step < S) {
        float fg[CPC], fx[CPC], fa[CPC];
        Elem<T>::unpack(rg[u], fg);
        Elem<T>::unpack(rx[u], fx);
        if (add) Elem<T>::unpack(ra[u], fa);
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          float xh = (fx[j] - mean[j]) * rstd[j];
          float gg = masked ? fg[j] * act_grad(AFF ? fmaf(gam[j], xh, bet[j]) : xh, act) : fg[j];
          if (AFF) gg *= gam[j];
          float d = rstd[j] * (gg - m1[j] - xh * m2[j]);
          fg[j] = add ? d + fa[j] : d;
        }
        st_chunk<T>(dx, (nb + v + u * step) * dx_stride + (size_t)cc * CPC, Elem<T>::pack(fg));
      }
    }
  }
}


// ---- post-norm residual block tail (monai UnetResBlock): y = act(IN(a) + (IN(b) | b)) ------------------
// Used by SwinUNETR's UnetrBasicBlock / UnetrUpBlock (swin_unetr.py:129-226): out = lrelu(norm2(conv2(..)) + residual),
// residual = norm3(conv3(x)) when the channel count changes, else x.  One streaming pass forward; backward =
// one reduction pass (the InstanceNorm-backward means of both branches, g = dy*act'(pre) with pre recomputed)
// + one apply pass.
template <typename T>
__device__ __forceinline__ void resnorm_pre(const float* fa, const float* fb, const float* sa, const float* sb, float* xa,
                                            float* xb, float* pre) {
  constexpr int CPC = Elem<T>::CPC;
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    xa[j] = (fa[j] - sa[2 * j]) * sa[2 * j + 1];
    xb[j] = sb ? (fb[j] - sb[2 * j]) * sb[2 * j + 1] : fb[j];
    pre[j] = xa[j] + xb[j];
  }
}

// (round 6: fixed channel chunk per thread, the chunk's statistics in registers, RU rows in flight — the first form walked a flat
//  index: two 64-bit divisions and 32-64 statistics loads per 16-byte chunk; k_resnorm_apply ran at 3.2 TB/s, 313 us at 128^3 x 48)
template <typename T>
__global__ void __launch_bounds__(NT) k_resnorm_fwd(const void* __restrict__ a, int64_t as, const float* __restrict__ sa,
                                                    const void* __restrict__ b, int64_t bs, const float* __restrict__ sb,
                                                    void* __restrict__ y, int64_t ys, int64_t S, int C, int act, int Cl, int c_off) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC, vlc = NT / cch;
  const int cc = threadIdx.x % cch + c_off / CPC, vl = threadIdx.x / cch;
  if (vl >= vlc) return;
  const int n = blockIdx.y;
  float ca[2 * CPC], cb[2 * CPC];
#pragma unroll
  for (int j = 0; j < 2 * CPC; ++j) {
    ca[j] = sa[((size_t)n * Cl + cc * CPC) * 2 + j];
    cb[j] = sb ? sb[((size_t)n * Cl + cc * CPC) * 2 + j] : 0.f;
  }
  const size_t nb = (size_t)n * S;
  const int64_t step = (int64_t)gridDim.x * vlc;
  for (int64_t v = (int64_t)blockIdx.x * vlc + vl; v < S; v += step * RU) {
    u32x4 ra[RU], rb[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (v + u * step < S) {
        ra[u] = ld_chunk<T>(a, (nb + v + u * step) * as + (size_t)cc * CPC);
        rb[u] = ld_chunk<T>(b, (nb + v + u * step) * bs + (size_t)cc * CPC);
      }
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (v + u * step < S) {
        float fa[CPC], fb[CPC], xa[CPC], xb[CPC], pre[CPC];
        Elem<T>::unpack(ra[u], fa);
        Elem<T>::unpack(rb[u], fb);
        resnorm_pre<T>(fa, fb, ca, sb ? cb : nullptr, xa, xb, pre);
#pragma unroll
        for (int j = 0; j < CPC; ++j) pre[j] = act_fwd(pre[j], act);
        st_chunk<T>(y, (nb + v + u * step) * ys + (size_t)cc * CPC, Elem<T>::pack(pre));
      }
  }
}

// partial sums per slab: pa = (0, sum g, sum g*xa), pb = (0, sum g, sum g*xb)
template <typename T>
__global__ void __launch_bounds__(NT) k_resnorm_partial(const void* __restrict__ dy, int64_t dys,
                                                        const void* __restrict__ a, int64_t as,
                                                        const float* __restrict__ sa, const void* __restrict__ b,
                                                        int64_t bs, const float* __restrict__ sb, int64_t S, int C,
                                                        int P, int act, float* __restrict__ pa, float* __restrict__ pb,
                                                        int Cl, int c_off) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC, vlc = NT / cch;
  const int t = threadIdx.x, cc = t % cch, vl = t / cch;
  const int part = blockIdx.x, n = blockIdx.y;
  const int64_t per = (S + P - 1) / P, v0 = (int64_t)part * per;
  int64_t v1 = v0 + per;
  if (v1 > S) v1 = S;
  const bool active = vl < vlc;
  const int c0 = c_off + cc * CPC;
  float s0[CPC], s1[CPC], s2[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  if (active) {
    const float* sap = sa + ((size_t)n * Cl + c0) * 2;
    const float* sbp = sb ? sb + ((size_t)n * Cl + c0) * 2 : nullptr;
    const size_t nb = (size_t)n * S;
    for (int64_t v = v0 + vl; v < v1; v += vlc) {
      float fa[CPC], fb[CPC], fg[CPC], xa[CPC], xb[CPC], pre[CPC];
      Elem<T>::unpack(ld_chunk<T>(a, (nb + v) * as + c0), fa);
      Elem<T>::unpack(ld_chunk<T>(b, (nb + v) * bs + c0), fb);
      Elem<T>::unpack(ld_chunk<T>(dy, (nb + v) * dys + c0), fg);
      resnorm_pre<T>(fa, fb, sap, sbp, xa, xb, pre);
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        float g = fg[j] * act_grad(pre[j], act);
        s0[j] += g; s1[j] += g * xa[j]; s2[j] += g * xb[j];
      }
    }
  }
  __shared__ float red[NT * 3 * 8];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    red[(t * CPC + j) * 3 + 0] = s0[j]; red[(t * CPC + j) * 3 + 1] = s1[j]; red[(t * CPC + j) * 3 + 2] = s2[j];
  }
  __syncthreads();
  if (vl == 0 && active) {
    for (int j = 0; j < CPC; ++j) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int q = 0; q < vlc; ++q) {
        const float* r = red + ((q * cch + cc) * CPC + j) * 3;
        a0 += r[0]; a1 += r[1]; a2 += r[2];
      }
      size_t o = (((size_t)n * P + part) * Cl + c0 + j) * 3;
      pa[o] = 0.f; pa[o + 1] = a0; pa[o + 2] = a1;
      if (pb) { pb[o] = 0.f; pb[o + 1] = a0; pb[o + 2] = a2; }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(NT) k_resnorm_apply(const void* __restrict__ dy, int64_t dys, const void* __restrict__ a,
                                                      int64_t as, const float* __restrict__ sa,
                                                      const float* __restrict__ ma, const void* __restrict__ b, int64_t bs,
                                                      const float* __restrict__ sb, const float* __restrict__ mb,
                                                      void* __restrict__ da, void* __restrict__ db, int64_t S, int C,
                                                      int act, int Cl, int c_off) {
  constexpr int CPC = Elem<T>::CPC, RA = 2;
  const int cch = C / CPC, vlc = NT / cch;
  const int cc = threadIdx.x % cch + c_off / CPC, vl = threadIdx.x / cch;
  if (vl >= vlc) return;
  const int n = blockIdx.y;
  float ca[2 * CPC], cb[2 * CPC], qa[2 * CPC], qb[2 * CPC];      // (mean, rstd) and the two backward means, per channel of the chunk
#pragma unroll
  for (int j = 0; j < 2 * CPC; ++j) {
    const size_t o = ((size_t)n * Cl + cc * CPC) * 2 + j;
    ca[j] = sa[o]; qa[j] = ma[o];
    cb[j] = sb ? sb[o] : 0.f; qb[j] = sb ? mb[o] : 0.f;
  }
  const size_t nb = (size_t)n * S;
  const int64_t step = (int64_t)gridDim.x * vlc;
  for (int64_t v = (int64_t)blockIdx.x * vlc + vl; v < S; v += step * RA) {
    u32x4 ra[RA], rb[RA], rg[RA];
#pragma unroll
    for (int u = 0; u < RA; ++u)
      if (v + u * step < S) {
        ra[u] = ld_chunk<T>(a, (nb + v + u * step) * as + (size_t)cc * CPC);
        rb[u] = ld_chunk<T>(b, (nb + v + u * step) * bs + (size_t)cc * CPC);
        rg[u] = ld_chunk<T>(dy, (nb + v + u * step) * dys + (size_t)cc * CPC);
      }
#pragma unroll
    for (int u = 0; u < RA; ++u)
      if (v + u * step < S) {
        float fa[CPC], fb[CPC], fg[CPC], xa[CPC], xb[CPC], pre[CPC], oa[CPC], ob[CPC];
        Elem<T>::unpack(ra[u], fa);
        Elem<T>::unpack(rb[u], fb);
        Elem<T>::unpack(rg[u], fg);
        resnorm_pre<T>(fa, fb, ca, sb ? cb : nullptr, xa, xb, pre);
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          float g = fg[j] * act_grad(pre[j], act);
          oa[j] = ca[2 * j + 1] * (g - qa[2 * j] - xa[j] * qa[2 * j + 1]);
          ob[j] = sb ? cb[2 * j + 1] * (g - qb[2 * j] - xb[j] * qb[2 * j + 1]) : g;
        }
        const size_t o = (nb + v + u * step) * (size_t)Cl + (size_t)cc * CPC;
        st_chunk<T>(da, o, Elem<T>::pack(oa));
        if (db) st_chunk<T>(db, o, Elem<T>::pack(ob));
      }
  }
}

// ---- layout helpers ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) k_ncdhw_to_ndhwc(const float* __restrict__ x, void* __restrict__ y,
                                                       int C, int64_t S, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int c = (int)(i % C);
    int64_t row = i / C;  // n*S+v
    int64_t n = row / S, v = row % S;
    Elem<T>::store1(y, (size_t)i, x[((size_t)n * C + c) * S + v]);
  }
}
template <typename T>
__global__ void __launch_bounds__(NT) k_ndhwc_to_ncdhw(const void* __restrict__ x, float* __restrict__ y,
                                                       int C, int64_t S, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int64_t v = i % S;
    int64_t nc = i / S;
    int64_t n = nc / C;
    int c = (int)(nc % C);
    y[i] = Elem<T>::load1(x, ((size_t)n * S + v) * C + c);
  }
}

// row-block count of the streaming norm kernels: ~8 workgroups per CU, each thread >= RU rows
static inline unsigned row_blocks(int dtype, int64_t S, int C) {
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  int vlc = NT / (C / cpc);
  int64_t b = (S + (int64_t)vlc * RU - 1) / ((int64_t)vlc * RU);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static inline int grid_for(int64_t items) {
  int64_t b = (items + NT - 1) / NT;
  if (b > 256 * 16) b = 256 * 16;  // ~16 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace cbim

using namespace cbim;

static int check_c(int dtype, int C) {
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(C > 0 && C % cpc == 0, CBIM_EUNSUPPORTED, "channel count %d is not a multiple of %d", C, cpc);
  return 0;
}

// the kernels keep one channel chunk per thread: wider tensors are processed in groups of <= NT chunks
#define FOR_CHANNEL_GROUPS(dtype, C)                                             \
  for (int c_off = 0, cg_max = NT * ((dtype) == CBIM_BF16 ? 8 : 4), Cg = 0;     \
       c_off < (C) && ((Cg = (C) - c_off > cg_max ? cg_max : (C) - c_off), true); c_off += cg_max)

extern "C" int cbim_warm_igemm(void* stream);
extern "C" int cbim_warm_wgrad(void* stream);
extern "C" int cbim_warm_norm(void* stream);
extern "C" int cbim_warm_pool_up(void* stream);
extern "C" int cbim_warm_stem_head(void* stream);
extern "C" int cbim_warm_loss(void* stream);
extern "C" int cbim_warm_augment(void* stream);
extern "C" int cbim_warm_medformer(void* stream);
extern "C" int cbim_warm_swin(void* stream);
extern "C" int cbim_warm_optim(void* stream);
extern "C" int cbim_warm_inference(void* stream);
extern "C" int cbim_warm_attn_mfma(void* stream);
extern "C" int cbim_warm_attn_wide(void* stream);
extern "C" int cbim_warm_r32(void* stream);
extern "C" int cbim_warm_swin_mfma(void* stream);
extern "C" int cbim_warm_wgrad_r32(void* stream);
extern "C" int cbim_warm_up_tile(void* stream);
extern "C" int cbim_warm_layernorm(void* stream);
extern "C" int cbim_warm_rw(void* stream);
extern "C" int cbim_warm_pw(void* stream);
extern "C" int cbim_warm_map(void* stream);
extern "C" int cbim_warm_awg(void* stream);

// One successful no-op launch from every code object of the library (see CBIM_DEFINE_WARM); the binding calls
// this once per process before the first real launch.
extern "C" int cbim_runtime_warmup(void* stream) {
  if (int e = cbim_warm_igemm(stream)) return e;
  if (int e = cbim_warm_wgrad(stream)) return e;
  if (int e = cbim_warm_norm(stream)) return e;
  if (int e = cbim_warm_pool_up(stream)) return e;
  if (int e = cbim_warm_stem_head(stream)) return e;
  if (int e = cbim_warm_loss(stream)) return e;
  if (int e = cbim_warm_augment(stream)) return e;
  if (int e = cbim_warm_medformer(stream)) return e;
  if (int e = cbim_warm_swin(stream)) return e;
  if (int e = cbim_warm_optim(stream)) return e;
  if (int e = cbim_warm_inference(stream)) return e;
  if (int e = cbim_warm_attn_mfma(stream)) return e;
  if (int e = cbim_warm_attn_wide(stream)) return e;
  if (int e = cbim_warm_r32(stream)) return e;
  if (int e = cbim_warm_swin_mfma(stream)) return e;
  if (int e = cbim_warm_wgrad_r32(stream)) return e;
  if (int e = cbim_warm_up_tile(stream)) return e;
  if (int e = cbim_warm_layernorm(stream)) return e;
  if (int e = cbim_warm_rw(stream)) return e;
  if (int e = cbim_warm_pw(stream)) return e;
  if (int e = cbim_warm_map(stream)) return e;
  if (int e = cbim_warm_awg(stream)) return e;
  return CBIM_OK;
}

extern "C" int cbim_stats_parts(int64_t S, int C) {
  // one workgroup per part.  A workgroup has 256 / (C/8) voxel lanes: a wide tensor (C = 1024 -> 2 lanes) with
  // 256-voxel parts ran 128 dependent trips in 128 workgroups — 253 GB/s.  Parts are sized for ~16 trips per
  // lane and at least ~1024 workgroups where the volume allows, at most 4096 parts (12 bytes per part and channel)
  if (C < 512) {   // many voxel lanes per workgroup: 64-voxel parts, up to 1024 of them
    int64_t q = (S + 63) / 64;
    return (int)(q < 1 ? 1 : (q > 1024 ? 1024 : q));
  }
  int64_t lanes = 256 / (C / 8 > 0 ? (C / 8 < 256 ? C / 8 : 256) : 1);
  if (lanes < 1) lanes = 1;
  int64_t per = lanes * 16;
  if (per < 16) per = 16;
  if (per > 256) per = 256;
  while (per > 16 && (S + per - 1) / per < 1024) per /= 2;
  int64_t p = (S + per - 1) / per;
  if (p < 1) p = 1;
  if (p > 4096) p = 4096;
  return (int)p;
}

extern "C" int cbim_instnorm_stats(int dtype, const void* x, int64_t x_stride, int N, int64_t S, int C,
                                   float eps, float* partials, int P, float* stats, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(P >= 1 && N >= 1 && S >= 1, CBIM_EINVAL, "bad sizes");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(P, N);
  FOR_CHANNEL_GROUPS(dtype, C) {
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_partial_sums<bf16_tag, 0>), grid, dim3(NT), 0, st, x, x_stride, x, x_stride,
                  (const float*)nullptr, S, Cg, P, 0, 0, partials, C, c_off, (const float*)nullptr);
    else
      CBIM_LAUNCH((k_partial_sums<float, 0>), grid, dim3(NT), 0, st, x, x_stride, x, x_stride,
                  (const float*)nullptr, S, Cg, P, 0, 0, partials, C, c_off, (const float*)nullptr);
  }
  return cbim_stats_finalize(partials, N, P, C, (double)S, eps, 0, stats, stream);
}

extern "C" int cbim_stats_finalize(const float* partials, int N, int P, int C, double count, float eps,
                                   int mode, float* out, void* stream) {
  CBIM_CHECK(N >= 1 && P >= 1 && C >= 1, CBIM_EINVAL, "bad sizes");
  CBIM_LAUNCH(k_stats_finalize, dim3(N * C), dim3(FIN_T), 0, (hipStream_t)stream, partials, N,
              P, C, count, eps, mode, out);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_norm_act_fwd(int dtype, const void* x, int64_t x_stride, const float* stats, void* y,
                                 int64_t y_stride, int N, int64_t S, int C, int act, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  hipStream_t st = (hipStream_t)stream;
  FOR_CHANNEL_GROUPS(dtype, C) {
    dim3 grid(row_blocks(dtype, S, Cg), N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_norm_act_fwd<bf16_tag>), grid, dim3(NT), 0, st, x, x_stride, stats, y, y_stride, S, Cg, act, C,
                  c_off, (const float*)nullptr);
    else
      CBIM_LAUNCH((k_norm_act_fwd<float>), grid, dim3(NT), 0, st, x, x_stride, stats, y, y_stride, S, Cg, act, C,
                  c_off, (const float*)nullptr);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_norm_bwd_reduce(int dtype, const void* g, int64_t g_stride, const void* x,
                                    int64_t x_stride, const float* stats, int N, int64_t S, int C, int act,
                                    int masked, float* partials, int P, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(P, N);
  FOR_CHANNEL_GROUPS(dtype, C) {
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_partial_sums<bf16_tag, 1>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, S, Cg,
                  P, act, masked, partials, C, c_off, (const float*)nullptr);
    else
      CBIM_LAUNCH((k_partial_sums<float, 1>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, S, Cg, P,
                  act, masked, partials, C, c_off, (const float*)nullptr);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_norm_bwd_apply(int dtype, const void* g, int64_t g_stride, const void* x,
                                   int64_t x_stride, const float* stats, const float* sums, const void* add,
                                   int64_t add_stride, void* dx, int64_t dx_stride, int N, int64_t S, int C,
                                   int act, int masked, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  hipStream_t st = (hipStream_t)stream;
  FOR_CHANNEL_GROUPS(dtype, C) {
    dim3 grid(row_blocks(dtype, S, Cg), N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_norm_bwd_apply<bf16_tag>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, sums, add,
                  add_stride, dx, dx_stride, S, Cg, act, masked, C, c_off, (const float*)nullptr);
    else
      CBIM_LAUNCH((k_norm_bwd_apply<float>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, sums, add,
                  add_stride, dx, dx_stride, S, Cg, act, masked, C, c_off, (const float*)nullptr);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}


// ---- the same three passes with a per-channel affine (gamma, beta) between normalisation and activation (round 5) ---------
extern "C" int cbim_norm_affine_act_fwd(int dtype, const void* x, int64_t x_stride, const float* stats, const float* affine,
                                        void* y, int64_t y_stride, int N, int64_t S, int C, int act, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(affine, CBIM_EINVAL, "norm_affine_act_fwd: null affine");
  hipStream_t st = (hipStream_t)stream;
  FOR_CHANNEL_GROUPS(dtype, C) {
    dim3 grid(row_blocks(dtype, S, Cg), N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_norm_act_fwd<bf16_tag, true>), grid, dim3(NT), 0, st, x, x_stride, stats, y, y_stride, S, Cg, act, C, c_off, affine);
    else
      CBIM_LAUNCH((k_norm_act_fwd<float, true>), grid, dim3(NT), 0, st, x, x_stride, stats, y, y_stride, S, Cg, act, C, c_off, affine);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_norm_affine_bwd_reduce(int dtype, const void* g, int64_t g_stride, const void* x, int64_t x_stride,
                                           const float* stats, const float* affine, int N, int64_t S, int C, int act, int masked,
                                           float* partials, int P, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(affine, CBIM_EINVAL, "norm_affine_bwd_reduce: null affine");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(P, N);
  FOR_CHANNEL_GROUPS(dtype, C) {
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_partial_sums<bf16_tag, 1, true>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, S, Cg, P, act, masked,
                  partials, C, c_off, affine);
    else
      CBIM_LAUNCH((k_partial_sums<float, 1, true>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, S, Cg, P, act, masked,
                  partials, C, c_off, affine);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// ---- BatchNorm: the per-channel arithmetic between the per-image statistics and the streaming kernels (round 6) --------------------
namespace cbim {
__global__ void __launch_bounds__(256) k_bn_finish_fwd(const float* __restrict__ st, int N, int C, double S, float eps, float momentum,
                                                       float* __restrict__ rm, float* __restrict__ rv, int use_batch,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats_out, float* __restrict__ affine_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double mean_b, var_b;
  if (use_batch) {
    double sm = 0.0, sq = 0.0;
    for (int n = 0; n < N; ++n) {
      const double m = (double)st[((size_t)n * C + c) * 2], r = (double)st[((size_t)n * C + c) * 2 + 1];
      const double v = 1.0 / (r * r) - (double)eps;          // the per-image biased variance back from rstd
      sm += m;
      sq += v + m * m;
    }
    mean_b = sm / N;
    var_b = sq / N - mean_b * mean_b;                          // biased variance of the batch (equal voxel counts per image)
    if (var_b < 0.0) var_b = 0.0;
    if (rm && rv) {
      const double n = (double)N * S;
      rm[c] = rm[c] * (1.f - momentum) + momentum * (float)mean_b;
      rv[c] = rv[c] * (1.f - momentum) + momentum * (float)(var_b * (n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0)));
    }
  } else {
    mean_b = (double)rm[c];
    var_b = (double)rv[c];
  }
  const float mb = (float)mean_b, rb = (float)(1.0 / sqrt(var_b + (double)eps));
  for (int n = 0; n < N; ++n) {
    stats_out[((size_t)n * C + c) * 2] = mb;
    stats_out[((size_t)n * C + c) * 2 + 1] = rb;
  }
  affine_out[(size_t)c * 2] = gamma ? gamma[c] : 1.f;
  affine_out[(size_t)c * 2 + 1] = beta ? beta[c] : 0.f;
}

__global__ void __launch_bounds__(256) k_bn_finish_bwd(const float* __restrict__ sums, int N, int C, double cnt,
                                                       const float* __restrict__ affine, int use_batch, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, float* __restrict__ sums_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int n = 0; n < N; ++n) {
    a += (double)sums[((size_t)n * C + c) * 2];
    b += (double)sums[((size_t)n * C + c) * 2 + 1];
  }
  a /= N;
  b /= N;
  if (dgamma) dgamma[c] = (float)(b * cnt);
  if (dbeta) dbeta[c] = (float)(a * cnt);
  const double g = (double)affine[(size_t)c * 2];
  const float s0 = use_batch ? (float)(g * a) : 0.f, s1 = use_batch ? (float)(g * b) : 0.f;
  for (int n = 0; n < N; ++n) {
    sums_out[((size_t)n * C + c) * 2] = s0;
    sums_out[((size_t)n * C + c) * 2 + 1] = s1;
  }
}
}  // namespace cbim

extern "C" int cbim_bn_finish_fwd(const float* st, int N, int C, double S, float eps, float momentum, float* running_mean,
                                  float* running_var, int use_batch, const float* gamma, const float* beta, float* stats_out,
                                  float* affine_out, void* stream) {
  CBIM_CHECK(stats_out && affine_out && N >= 1 && C >= 1 && S >= 1.0, CBIM_EINVAL, "bn_finish_fwd: null operand / bad sizes");
  CBIM_CHECK(use_batch ? st != nullptr : (running_mean && running_var), CBIM_EINVAL, "bn_finish_fwd: the statistics this mode reads are missing");
  CBIM_CHECK((running_mean == nullptr) == (running_var == nullptr), CBIM_EINVAL, "bn_finish_fwd: running mean and variance go together");
  CBIM_LAUNCH(cbim::k_bn_finish_fwd, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, st, N, C, S, eps, momentum, running_mean,
              running_var, use_batch, gamma, beta, stats_out, affine_out);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_bn_finish_bwd(const float* sums, int N, int C, double cnt, const float* affine, int use_batch, float* dgamma,
                                  float* dbeta, float* sums_out, void* stream) {
  CBIM_CHECK(sums && affine && sums_out && N >= 1 && C >= 1, CBIM_EINVAL, "bn_finish_bwd: null operand / bad sizes");
  CBIM_LAUNCH(cbim::k_bn_finish_bwd, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, N, C, cnt, affine, use_batch, dgamma,
              dbeta, sums_out);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_norm_affine_bwd_apply(int dtype, const void* g, int64_t g_stride, const void* x, int64_t x_stride,
                                          const float* stats, const float* affine, const float* sums, void* dx, int64_t dx_stride,
                                          int N, int64_t S, int C, int act, int masked, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(affine, CBIM_EINVAL, "norm_affine_bwd_apply: null affine");
  hipStream_t st = (hipStream_t)stream;
  FOR_CHANNEL_GROUPS(dtype, C) {
    dim3 grid(row_blocks(dtype, S, Cg), N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_norm_bwd_apply<bf16_tag, true>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, sums, (const void*)nullptr,
                  (int64_t)0, dx, dx_stride, S, Cg, act, masked, C, c_off, affine);
    else
      CBIM_LAUNCH((k_norm_bwd_apply<float, true>), grid, dim3(NT), 0, st, g, g_stride, x, x_stride, stats, sums, (const void*)nullptr,
                  (int64_t)0, dx, dx_stride, S, Cg, act, masked, C, c_off, affine);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}


extern "C" int cbim_resnorm_fwd(int dtype, const void* a, int64_t a_stride, const float* stats_a, const void* b,
                                int64_t b_stride, const float* stats_b, void* y, int64_t y_stride, int N, int64_t S, int C,
                                int act, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(a && b && y && stats_a, CBIM_EINVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  FOR_CHANNEL_GROUPS(dtype, C) {
    dim3 grid(row_blocks(dtype, S, Cg), N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_resnorm_fwd<bf16_tag>), grid, dim3(NT), 0, st, a, a_stride, stats_a, b, b_stride, stats_b, y, y_stride, S, Cg, act, C,
                  c_off);
    else
      CBIM_LAUNCH((k_resnorm_fwd<float>), grid, dim3(NT), 0, st, a, a_stride, stats_a, b, b_stride, stats_b, y, y_stride, S, Cg, act, C,
                  c_off);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_resnorm_bwd_reduce(int dtype, const void* dy, int64_t dy_stride, const void* a, int64_t a_stride,
                                       const float* stats_a, const void* b, int64_t b_stride, const float* stats_b, int N,
                                       int64_t S, int C, int act, float* partials_a, float* partials_b, int P,
                                       void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(dy && a && b && stats_a && partials_a && (!stats_b || partials_b), CBIM_EINVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(P, N);
  FOR_CHANNEL_GROUPS(dtype, C) {
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_resnorm_partial<bf16_tag>), grid, dim3(NT), 0, st, dy, dy_stride, a, a_stride, stats_a, b, b_stride,
                  stats_b, S, Cg, P, act, partials_a, stats_b ? partials_b : (float*)nullptr, C, c_off);
    else
      CBIM_LAUNCH((k_resnorm_partial<float>), grid, dim3(NT), 0, st, dy, dy_stride, a, a_stride, stats_a, b, b_stride,
                  stats_b, S, Cg, P, act, partials_a, stats_b ? partials_b : (float*)nullptr, C, c_off);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_resnorm_bwd_apply(int dtype, const void* dy, int64_t dy_stride, const void* a, int64_t a_stride,
                                      const float* stats_a, const float* sums_a, const void* b, int64_t b_stride,
                                      const float* stats_b, const float* sums_b, void* da, void* db, int N, int64_t S,
                                      int C, int act, void* stream) {
  if (int e = check_c(dtype, C)) return e;
  CBIM_CHECK(dy && a && b && stats_a && sums_a && da && (!stats_b || sums_b), CBIM_EINVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  FOR_CHANNEL_GROUPS(dtype, C) {
    dim3 grid(row_blocks(dtype, S, Cg), N);
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_resnorm_apply<bf16_tag>), grid, dim3(NT), 0, st, dy, dy_stride, a, a_stride, stats_a, sums_a, b, b_stride, stats_b,
                  sums_b, da, db, S, Cg, act, C, c_off);
    else
      CBIM_LAUNCH((k_resnorm_apply<float>), grid, dim3(NT), 0, st, dy, dy_stride, a, a_stride, stats_a, sums_a, b, b_stride, stats_b,
                  sums_b, da, db, S, Cg, act, C, c_off);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_ncdhw_to_ndhwc(int dtype_out, const float* x, void* y, int N, int C, int64_t S,
                                   void* stream) {
  int64_t total = (int64_t)N * C * S;
  hipStream_t st = (hipStream_t)stream;
  if (dtype_out == CBIM_BF16)
    CBIM_LAUNCH((k_ncdhw_to_ndhwc<bf16_tag>), dim3(grid_for(total)), dim3(NT), 0, st, x, y, C, S, total);
  else
    CBIM_LAUNCH((k_ncdhw_to_ndhwc<float>), dim3(grid_for(total)), dim3(NT), 0, st, x, y, C, S, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_ndhwc_to_ncdhw(int dtype_in, const void* x, float* y, int N, int C, int64_t S,
                                   void* stream) {
  int64_t total = (int64_t)N * C * S;
  hipStream_t st = (hipStream_t)stream;
  if (dtype_in == CBIM_BF16)
    CBIM_LAUNCH((k_ndhwc_to_ncdhw<bf16_tag>), dim3(grid_for(total)), dim3(NT), 0, st, x, y, C, S, total);
  else
    CBIM_LAUNCH((k_ndhwc_to_ncdhw<float>), dim3(grid_for(total)), dim3(NT), 0, st, x, y, C, S, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// ---- per-(n, c) statistics arithmetic of the MedFormer blocks (a few hundred numbers; fp64 inside) ------------------------
// Each of these was 7-14 tiny ATen launches per block (cast to double, pow, sub, clamp, ..., stack): ~480 launches per step.
namespace cbim {
// MODE 0: (mean, rstd[eps_a]) -> (mean, rstd[eps_b])                      (InstanceNorm3d eps 1e-4 vs the 1e-5 default,
//         medformer_utils.py:112,158 vs conv_layers.py:40)
// MODE 1: SE gate folded into the normalisation, IN(x * s) = (x - mean) * s * rsqrt(var s^2 + eps): out = (mean, s * sqrt(rz2)),
//         rz2 = 1 / (var s^2 + eps) kept (double) for the backward          (conv_layers.py:159-175,223-236)
// MODE 2: gradient of the gate, ds = S * eps * m2 * rz2 / s  (m2 = the second InstanceNorm-backward mean of the dgrad epilogue)
template <int MODE>
__global__ void __launch_bounds__(256) k_stats_remap(const float* __restrict__ stats, const float* __restrict__ se,
                                                     const float* __restrict__ sums, double* __restrict__ rz2, float eps_a,
                                                     float eps_b, double S, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (MODE == 2) {
    out[i] = (float)(S * (double)eps_a * (double)sums[2 * i + 1] * rz2[i] / (double)se[i]);
    return;
  }
  const double r = (double)stats[2 * i + 1];
  double var = 1.0 / (r * r) - (double)eps_a;
  if (var < 0.0) var = 0.0;
  out[2 * i] = stats[2 * i];
  if (MODE == 0) out[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps_b));
  else {
    const double sd = (double)se[i];
    const double z = 1.0 / (var * sd * sd + (double)eps_a);
    rz2[i] = z;
    out[2 * i + 1] = (float)(sd * sqrt(z));
  }
}
}  // namespace cbim

extern "C" int cbim_stats_restat(const float* stats, float eps_from, float eps_to, float* out, int n, void* stream) {
  CBIM_CHECK(stats && out && n > 0, CBIM_EINVAL, "null argument");
  CBIM_LAUNCH((cbim::k_stats_remap<0>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, (const float*)nullptr,
              (const float*)nullptr, (double*)nullptr, eps_from, eps_to, 0.0, out, n);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}
extern "C" int cbim_se_fold_fwd(const float* stats, const float* se, float eps, float* out_stats, double* rz2, int n, void* stream) {
  CBIM_CHECK(stats && se && out_stats && rz2 && n > 0, CBIM_EINVAL, "null argument");
  CBIM_LAUNCH((cbim::k_stats_remap<1>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, se, (const float*)nullptr, rz2,
              eps, 0.f, 0.0, out_stats, n);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}
extern "C" int cbim_se_fold_bwd(const float* sums, const float* se, const double* rz2, float eps, double S, float* ds, int n,
                                void* stream) {
  CBIM_CHECK(sums && se && rz2 && ds && n > 0, CBIM_EINVAL, "null argument");
  CBIM_LAUNCH((cbim::k_stats_remap<2>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, se, sums,
              (double*)rz2, eps, 0.f, S, ds, n);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(norm)
