// loss_kernels.hip — fused weighted cross-entropy + soft Dice (Tversky, adaptive alpha) loss.
//
// Reference: nn.CrossEntropyLoss(weight=w) + DiceLoss()  (/root/reference/train.py:80-81,212;
// /root/reference/training/losses.py:18-58).  The reference materialises ~10 full-size fp32
// temporaries (softmax, one-hot, TP/FP/FN maps ...); here the logits are read ONCE in the forward
// (per-class sums of P*M, P and M plus the CE numerator/denominator) and once in the backward.
//
// With TP_c = sum P_c*M_c, SP_c = sum P_c, CNT_c = sum M_c over batch and space:
//   FP = SP-TP, FN = CNT-TP, alpha = clamp(FP/(FP+FN+eps), .2, .8) (NOT detached, losses.py:38-40),
//   den = TP + alpha*FP + (1-alpha)*FN = alpha*SP + (1-alpha)*CNT, dice = TP/(den+eps),
//   L = mean_c(1-dice_c);  d L/d P_c(voxel) = dL/dTP_c * M_c + dL/dSP_c.
// HBM-bound: logits are NCDHW fp32, one thread per voxel, coalesced plane reads.
#include "cbim_common.h"

namespace cbim {

static constexpr int NT = 256;
static constexpr int CMAX = 32;
static constexpr float SMOOTH = 1e-5f;

// V consecutive voxels per thread: every class plane is read with one V*4-byte load per lane (16 B for V = 4, a
// wave instruction then covers 1 KiB of one plane), labels as V int64.  CT = compile-time class bound (C <= CT) so
// the per-class arrays live in registers without predicated dead work (C = 16 ran the CT = 32 loops before).
template <int V> struct VecF;
template <> struct VecF<4> { typedef f32x4 type; };
template <> struct VecF<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecF<1> { typedef float type; };
template <int V> __device__ __forceinline__ void ldv(const float* p, float* o) {
  typename VecF<V>::type q = *(const typename VecF<V>::type*)p;
  if constexpr (V == 1) o[0] = q;
  else {
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = q[i];
  }
}
template <int V> __device__ __forceinline__ void stv(float* p, const float* o) {
  typename VecF<V>::type q;
  if constexpr (V == 1) q = o[0];
  else {
#pragma unroll
    for (int i = 0; i < V; ++i) q[i] = o[i];
  }
  *(typename VecF<V>::type*)p = q;
}

// partial layout per workgroup: [3*C + 3] = TP[C], SP[C], CNT[C], ce_num, ce_den, #labels outside [0, C)
// A label outside [0, C) (the reference raises in scatter_ / CrossEntropyLoss) is never used as an index: the voxel
// is left out of CE, TP and CNT and counted, the count comes back in out[3].
template <int CT, int V>
__global__ void __launch_bounds__(NT) k_dice_ce_fwd(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                    const float* __restrict__ wgt, int C, int64_t S,
                                                    int64_t groups, float* __restrict__ partials) {
  float tp[CT], sp[CT], cnt[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) { tp[c] = 0.f; sp[c] = 0.f; cnt[c] = 0.f; }
  float ce_num = 0.f, ce_den = 0.f, bad = 0.f;
  for (int64_t g = (int64_t)blockIdx.x * NT + threadIdx.x; g < groups; g += (int64_t)gridDim.x * NT) {
    const int64_t r = g * V;
    const int64_t n = r / S, v = r % S;
    const float* zp = z + (size_t)n * C * S + v;
    int lab[V];
#pragma unroll
    for (int i = 0; i < V; ++i) lab[i] = (int)y[r + i];
    float l[CT][V];
    float mx[V], zy[V], se[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { mx[i] = -INFINITY; zy[i] = 0.f; se[i] = 0.f; }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) ldv<V>(zp + (size_t)c * S, l[c]);
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int i = 0; i < V; ++i) { mx[i] = fmaxf(mx[i], l[c][i]); if (c == lab[i]) zy[i] = l[c][i]; }
      }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int i = 0; i < V; ++i) { l[c][i] = __expf(l[c][i] - mx[i]); se[i] += l[c][i]; }
      }
    float inv[V];
#pragma unroll
    for (int i = 0; i < V; ++i) inv[i] = 1.f / se[i];
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          float pc = l[c][i] * inv[i];
          sp[c] += pc;
          if (c == lab[i]) { tp[c] += pc; cnt[c] += 1.f; }
        }
      }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const bool ok = (unsigned)lab[i] < (unsigned)C;
      float wy = ok ? (wgt ? wgt[lab[i]] : 1.f) : 0.f;
      ce_num += wy * (__logf(se[i]) + mx[i] - zy[i]);
      ce_den += wy;
      bad += ok ? 0.f : 1.f;
    }
  }
  // workgroup reduction (fixed order): wave shuffles then 4 waves through LDS
  __shared__ float red[4][3 * CT + 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if (c < C) {
      float a = wave_sum(tp[c]), b = wave_sum(sp[c]), d = wave_sum(cnt[c]);
      if (lane == 0) { red[wave][c] = a; red[wave][CT + c] = b; red[wave][2 * CT + c] = d; }
    }
  }
  {
    float a = wave_sum(ce_num), b = wave_sum(ce_den), e = wave_sum(bad);
    if (lane == 0) { red[wave][3 * CT] = a; red[wave][3 * CT + 1] = b; red[wave][3 * CT + 2] = e; }
  }
  __syncthreads();
  float* out = partials + (size_t)blockIdx.x * (3 * C + 3);
  for (int i = threadIdx.x; i < 3 * C + 3; i += NT) {
    int src = i < 3 * C ? (i / C) * CT + (i % C) : 3 * CT + (i - 3 * C);
    out[i] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
  }
}

// single workgroup: reduce partials in fp64, evaluate the loss and its class-level derivatives.
__global__ void __launch_bounds__(NT) k_dice_ce_finalize(const float* __restrict__ partials, int nblk, int C,
                                                         float* __restrict__ out, float* __restrict__ coef) {
  __shared__ double tot[3 * CMAX + 3];
  __shared__ double part[4][3 * CMAX + 3];
  __shared__ double dice_term[CMAX];
  const int nv = 3 * C + 3;
  {   // 4 thread groups stride over the workgroup partials (independent loads in flight), fixed-order merge
    const int li = threadIdx.x & 63, pg = threadIdx.x >> 6;
    for (int i = li; i < nv; i += 64) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // 4 loads in flight per trip
      int b = pg;
      for (; b + 12 < nblk; b += 16) {
        a0 += (double)partials[(size_t)b * nv + i];
        a1 += (double)partials[(size_t)(b + 4) * nv + i];
        a2 += (double)partials[(size_t)(b + 8) * nv + i];
        a3 += (double)partials[(size_t)(b + 12) * nv + i];
      }
      for (; b < nblk; b += 4) a0 += (double)partials[(size_t)b * nv + i];
      part[pg][i] = (a0 + a1) + (a2 + a3);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nv; i += NT) tot[i] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
  __syncthreads();
  if ((int)threadIdx.x < C) {
    int c = threadIdx.x;
    double TP = tot[c], SP = tot[C + c], CNT = tot[2 * C + c];
    double eps = (double)SMOOTH;
    double FP = SP - TP, FN = CNT - TP;
    double T = FP + FN + eps;
    double r = FP / T;
    bool open = (r >= 0.2 && r <= 0.8);
    double alpha = r < 0.2 ? 0.2 : (r > 0.8 ? 0.8 : r);
    double den = TP + alpha * FP + (1.0 - alpha) * FN;
    double dice = TP / (den + eps);
    dice_term[c] = 1.0 - dice;
    // d alpha / d TP, d alpha / d SP (only inside the clamp)
    double da_dTP = open ? (FP - FN - eps) / (T * T) : 0.0;
    double da_dSP = open ? (FN + eps) / (T * T) : 0.0;
    double dden_dTP = (SP - CNT) * da_dTP;            // the explicit TP terms cancel
    double dden_dSP = alpha + (SP - CNT) * da_dSP;
    double q = TP / ((den + eps) * (den + eps));
    double ddice_dTP = 1.0 / (den + eps) - q * dden_dTP;
    double ddice_dSP = -q * dden_dSP;
    coef[c] = (float)(-ddice_dTP / C);
    coef[C + c] = (float)(-ddice_dSP / C);
    coef[2 * C + 1 + c] = (float)(1.0 - dice);        // per-class loss term: DiceLoss(reduce=False) (losses.py:48-50)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double dl = 0.0;
    for (int c = 0; c < C; ++c) dl += dice_term[c];
    dl /= C;
    double ce = tot[3 * C] / tot[3 * C + 1];
    out[0] = (float)ce;
    out[1] = (float)dl;
    out[2] = (float)(ce + dl);
    out[3] = (float)tot[3 * C + 2];   // labels outside [0, C)
    coef[2 * C] = (float)(1.0 / tot[3 * C + 1]);
  }
}

// dlogits = g_ce * dCE/dz + g_dice * dDice/dz   (grad_out = {g_ce, g_dice} on the device)
template <int CT, int V>
__global__ void __launch_bounds__(NT) k_dice_ce_bwd(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                    const float* __restrict__ wgt, const float* __restrict__ coef,
                                                    const float* __restrict__ grad_out, float* __restrict__ dz,
                                                    int C, int64_t S, int64_t groups) {
  __shared__ float cf[2 * CMAX + 1];
  if ((int)threadIdx.x < 2 * C + 1) cf[threadIdx.x] = coef[threadIdx.x];
  __syncthreads();
  const float g_ce = grad_out[0], g_dice = grad_out[1];
  const float inv_w = cf[2 * C];
  for (int64_t g = (int64_t)blockIdx.x * NT + threadIdx.x; g < groups; g += (int64_t)gridDim.x * NT) {
    const int64_t r = g * V;
    const int64_t n = r / S, v = r % S;
    const float* zp = z + (size_t)n * C * S + v;
    float* dp = dz + (size_t)n * C * S + v;
    int lab[V];
#pragma unroll
    for (int i = 0; i < V; ++i) lab[i] = (int)y[r + i];
    float l[CT][V];
    float mx[V], se[V], dot[V], wy[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { mx[i] = -INFINITY; se[i] = 0.f; dot[i] = 0.f; }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) ldv<V>(zp + (size_t)c * S, l[c]);
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int i = 0; i < V; ++i) mx[i] = fmaxf(mx[i], l[c][i]);
      }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int i = 0; i < V; ++i) { l[c][i] = __expf(l[c][i] - mx[i]); se[i] += l[c][i]; }
      }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      se[i] = 1.f / se[i];
      const bool ok = (unsigned)lab[i] < (unsigned)C;     // see k_dice_ce_fwd: such a voxel only feeds SP
      wy[i] = ok ? (wgt ? wgt[lab[i]] : 1.f) * inv_w : 0.f;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
        const float gs = cf[C + c], gt = cf[c];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          l[c][i] *= se[i];
          dot[i] += l[c][i] * (gs + (c == lab[i] ? gt : 0.f));   // sum_k p_k G_k
        }
      }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
        const float gs = cf[C + c], gt = cf[c];
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          float G = gs + (c == lab[i] ? gt : 0.f);
          float d_dice = l[c][i] * (G - dot[i]);
          float d_ce = wy[i] * (l[c][i] - (c == lab[i] ? 1.f : 0.f));
          o[i] = g_ce * d_ce + g_dice * d_dice;
        }
        stv<V>(dp + (size_t)c * S, o);
      }
  }
}

static int loss_blocks(int64_t total) {
  int64_t b = (total + NT * 8 - 1) / (NT * 8);   // upper bound over the vector widths (groups <= total)
  // four workgroups per CU (16 waves: one 64-register wave per SIMD left the loads latency-bound, 55 us for 151 MB at
  // 1x16x128^3); the single-workgroup finalize kernel walks these records, four independent loads per thread and trip
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

// voxels per thread: 4 (C <= 16) / 2 (C <= 32) consecutive voxels when every class plane stays aligned for the
// vector load (S a multiple of V, base pointer aligned); otherwise one voxel per thread
static int loss_vec(const void* logits, int64_t S, int C) {
  const int V = C <= 16 ? 4 : 2;
  if (S % V != 0 || ((size_t)logits % (V * 4)) != 0) return 1;
  return V;
}

}  // namespace cbim

using namespace cbim;

extern "C" size_t cbim_dice_ce_workspace(int N, int C, int64_t S) {
  return (size_t)loss_blocks((int64_t)N * S) * (3 * C + 3) * sizeof(float);
}

extern "C" int cbim_dice_ce_fwd(const float* logits, const int64_t* labels, const float* weight, int N, int C,
                                int64_t S, float* out, float* coef, void* workspace, size_t ws_bytes,
                                void* stream) {
  CBIM_CHECK(logits && labels && out && coef, CBIM_EINVAL, "null argument");
  CBIM_CHECK(C >= 1 && C <= CMAX, CBIM_EUNSUPPORTED, "loss supports up to %d classes (got %d)", CMAX, C);
  size_t need = cbim_dice_ce_workspace(N, C, S);
  CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "loss workspace %zu < %zu", ws_bytes, need);
  int64_t total = (int64_t)N * S;
  hipStream_t st = (hipStream_t)stream;
  const int V = loss_vec(logits, S, C);
  const int64_t groups = total / V;
  int nb = loss_blocks(total);
  if ((int64_t)nb * NT > groups) nb = (int)((groups + NT - 1) / NT);
  if (nb < 1) nb = 1;
  float* ws = (float*)workspace;
  if (C <= 16) {
    if (V == 4) CBIM_LAUNCH((k_dice_ce_fwd<16, 4>), dim3(nb), dim3(NT), 0, st, logits, labels, weight, C, S, groups, ws);
    else CBIM_LAUNCH((k_dice_ce_fwd<16, 1>), dim3(nb), dim3(NT), 0, st, logits, labels, weight, C, S, groups, ws);
  } else {
    if (V == 2) CBIM_LAUNCH((k_dice_ce_fwd<CMAX, 2>), dim3(nb), dim3(NT), 0, st, logits, labels, weight, C, S, groups, ws);
    else CBIM_LAUNCH((k_dice_ce_fwd<CMAX, 1>), dim3(nb), dim3(NT), 0, st, logits, labels, weight, C, S, groups, ws);
  }
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  CBIM_LAUNCH(k_dice_ce_finalize, dim3(1), dim3(NT), 0, st, (const float*)workspace, nb, C, out, coef);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_dice_ce_bwd(const float* logits, const int64_t* labels, const float* weight,
                                const float* coef, const float* grad_out, float* dlogits, int N, int C,
                                int64_t S, void* stream) {
  CBIM_CHECK(logits && labels && coef && grad_out && dlogits, CBIM_EINVAL, "null argument");
  CBIM_CHECK(C >= 1 && C <= CMAX, CBIM_EUNSUPPORTED, "loss supports up to %d classes (got %d)", CMAX, C);
  int64_t total = (int64_t)N * S;
  int V = loss_vec(logits, S, C);
  if (V > 1 && ((size_t)dlogits % (V * 4)) != 0) V = 1;
  const int64_t groups = total / V;
  int64_t b = (groups + NT - 1) / NT;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)b);
  if (C <= 16) {
    if (V == 4) CBIM_LAUNCH((k_dice_ce_bwd<16, 4>), grid, dim3(NT), 0, st, logits, labels, weight, coef, grad_out, dlogits, C, S, groups);
    else CBIM_LAUNCH((k_dice_ce_bwd<16, 1>), grid, dim3(NT), 0, st, logits, labels, weight, coef, grad_out, dlogits, C, S, groups);
  } else {
    if (V == 2) CBIM_LAUNCH((k_dice_ce_bwd<CMAX, 2>), grid, dim3(NT), 0, st, logits, labels, weight, coef, grad_out, dlogits, C, S, groups);
    else CBIM_LAUNCH((k_dice_ce_bwd<CMAX, 1>), grid, dim3(NT), 0, st, logits, labels, weight, coef, grad_out, dlogits, C, S, groups);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(loss)
