// map_kernels.hip — the semantic-map side of MedFormer's BidirectionAttentionBlock (round 6).
//
// A semantic map is a float32 [B, C, M] tensor with M = m0 m1 m2 <= 128 positions (64 in the AMOS configuration).  Per block the
// reference runs on it (/root/reference/model/dim3/medformer_utils.py:63-97, 102-138):
//     mapp = norm2(map)                                  nn.InstanceNorm3d over the M positions            (:113, :127)
//     map_q, map_v = map_qv(mapp).chunk(2)               1x1x1 conv = [2 inner, C] x [C, M]                (:36, :66)
//     ... both softmaxes of the bidirectional attention (attn_mfma.hip / attn_wide.hip) ...
//     map_out = map_out(map_out)                         1x1x1 conv = [C, inner] x [inner, M]              (:40, :93)
//     mapp = map_out + map                               residual                                          (:137)
// Until round 5 these were ordinary torch ops: per block a layer_norm, two aten::mm forward and four backward, their
// transposes / slices / adds — 197 aten::mm + 73 add_ + 32 slice_backward per MedFormer step, 4.96 ms of device time on
// launch-bound 64-column GEMMs (profiles/r05_x_aten_medformer.txt).  Here each of them is ONE launch of a small fp32 GEMM
//     OUT[o][n] = sum_k A[o][k] X[k][n]  (+ R[o][n])
// whose operands are read and written in whatever layout the neighbouring kernels use (transposed A / X / OUT, operands split
// over two tensors: the attention kernels take and return [M, inner] rows for q and v separately), with the two normalisation
// steps folded in: InstanceNorm of the rows of X on load (forward) and its backward as the epilogue of the GEMM that produces
// d(mapp).  Forward 2 launches, backward 4 per block instead of ~35.
//
// Tile: 256 threads, 32 output rows x 128 columns per workgroup, K in chunks of 32 through LDS; wave w owns rows 8 w .. 8 w + 7,
// lane l the columns l and l + 64 (16 accumulators), so a row's reduction over the columns is one wave butterfly.  fp32 FMA
// chains in a fixed order: bit-reproducible.
#include "cbim_common.h"

namespace cbim {

static constexpr int MG_T = 256, MG_OT = 32, MG_NT = 128, MG_KC = 32, MG_PA = 36, MG_PX = 129;

__device__ __forceinline__ float mg_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ void __launch_bounds__(MG_T) k_map_gemm(cbim_map_gemm_desc d) {
  __shared__ float At[MG_KC * MG_PA];      // [k][o]
  __shared__ float Xs[MG_KC * MG_PX];      // [k][n]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int o0 = blockIdx.x * MG_OT, n0 = blockIdx.y * MG_NT;
  const int b0 = d.reduce_batch ? 0 : (int)blockIdx.z, b1 = d.reduce_batch ? d.batch : b0 + 1;
  float acc[8][2];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][1] = 0.f;
  const float inv_n = 1.f / (float)d.Nn;
  for (int b = b0; b < b1; ++b) {
    const float* const A = d.A + (int64_t)b * d.a_batch;
    const float* const A2 = d.A2 ? d.A2 + (int64_t)b * d.a_batch : nullptr;
    const float* const X = d.X + (int64_t)b * d.x_batch;
    const float* const X2 = d.X2 ? d.X2 + (int64_t)b * d.x_batch : nullptr;
    for (int k0 = 0; k0 < d.K; k0 += MG_KC) {
      __syncthreads();
      // ---- A tile: At[k][o] -----------------------------------------------------------------------------------------------
#pragma unroll
      for (int e = tid; e < MG_OT * MG_KC; e += MG_T) {
        int o, k;
        if (d.a_t) { k = e >> 5; o = e & 31; } else { o = e >> 5; k = e & 31; }
        float v = 0.f;
        const int go = o0 + o, gk = k0 + k;
        if (go < d.O && gk < d.K) {
          if (!d.a_t) v = A[(int64_t)go * d.lda + gk];
          else if (A2 && go >= d.a_split) v = A2[(int64_t)gk * d.lda + go - d.a_split];
          else v = A[(int64_t)gk * d.lda + go];
        }
        At[k * MG_PA + o] = v;
      }
      // ---- X tile: Xs[k][n] -----------------------------------------------------------------------------------------------
#pragma unroll 4
      for (int e = tid; e < MG_KC * MG_NT; e += MG_T) {
        int k, n;
        if (d.x_t) { n = e >> 5; k = e & 31; } else { k = e >> 7; n = e & 127; }
        float v = 0.f;
        const int gk = k0 + k, gn = n0 + n;
        if (gk < d.K && gn < d.Nn) {
          if (!d.x_t) v = X[(int64_t)gk * d.ldx + gn];
          else if (X2 && gk >= d.x_split) v = X2[(int64_t)gn * d.ldx + gk - d.x_split];
          else v = X[(int64_t)gn * d.ldx + gk];
        }
        Xs[k * MG_PX + n] = v;
      }
      __syncthreads();
      if (d.ln_mode) {
        // InstanceNorm of the rows over the n columns (all of them are in the tile: Nn <= 128), biased variance from centred
        // values; the first workgroup column keeps the normalised rows and their rstd for the backward pass
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * wave + j, gk = k0 + k;
          const bool in0 = lane < d.Nn, in1 = lane + 64 < d.Nn;
          const float v0 = Xs[k * MG_PX + lane], v1 = Xs[k * MG_PX + lane + 64];
          const float mean = mg_wave_sum((in0 ? v0 : 0.f) + (in1 ? v1 : 0.f)) * inv_n;
          const float c0 = in0 ? v0 - mean : 0.f, c1 = in1 ? v1 - mean : 0.f;
          const float var = mg_wave_sum(c0 * c0 + c1 * c1) * inv_n;
          const float rstd = 1.f / sqrtf(var + d.eps);
          Xs[k * MG_PX + lane] = c0 * rstd;
          Xs[k * MG_PX + lane + 64] = c1 * rstd;
          if (blockIdx.x == 0 && gk < d.K) {
            float* xn = d.Xn + (int64_t)b * d.x_batch + (int64_t)gk * d.ldx;
            if (in0) xn[lane] = c0 * rstd;
            if (in1) xn[lane + 64] = c1 * rstd;
            if (lane == 0) d.rstd_out[(int64_t)b * d.K + gk] = rstd;
          }
        }
        __syncthreads();
      }
      // ---- 32 k-steps: two broadcast float4 of A, two floats of X, 16 FMAs --------------------------------------------------
#pragma unroll 8
      for (int k = 0; k < MG_KC; ++k) {
        const f32x4 a0 = *(const f32x4*)(At + k * MG_PA + 8 * wave), a1 = *(const f32x4*)(At + k * MG_PA + 8 * wave + 4);
        const float x0 = Xs[k * MG_PX + lane], x1 = Xs[k * MG_PX + lane + 64];
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[j][0] = fmaf(a[j], x0, acc[j][0]);
          acc[j][1] = fmaf(a[j], x1, acc[j][1]);
        }
      }
    }
  }
  // ---- epilogue ----------------------------------------------------------------------------------------------------------------
  const int bo = d.reduce_batch ? 0 : (int)blockIdx.z;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int go = o0 + 8 * wave + j;
    if (go >= d.O) continue;                         // (wave-uniform)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gn = n0 + lane + 64 * h;
      const bool in = gn < d.Nn;
      float v = acc[j][h];
      if (d.XH) {
        // InstanceNorm backward over the row: rstd (g - mean(g) - xh mean(g xh)); both halves of the row are in this lane pair
        const float xh0 = lane < d.Nn ? d.XH[(int64_t)bo * d.o_batch + (int64_t)go * d.ldo + lane] : 0.f;
        const float xh1 = lane + 64 < d.Nn ? d.XH[(int64_t)bo * d.o_batch + (int64_t)go * d.ldo + lane + 64] : 0.f;
        const float g0 = lane < d.Nn ? acc[j][0] : 0.f, g1 = lane + 64 < d.Nn ? acc[j][1] : 0.f;
        const float m1 = mg_wave_sum(g0 + g1) * inv_n, m2 = mg_wave_sum(g0 * xh0 + g1 * xh1) * inv_n;
        v = d.rstd_in[(int64_t)bo * d.O + go] * (v - m1 - (h ? xh1 : xh0) * m2);
      }
      if (!in) continue;
      if (d.R) v += d.R[(int64_t)bo * d.r_batch + (int64_t)go * d.ldr + gn];
      if (!d.o_t) d.OUT[(int64_t)bo * d.o_batch + (int64_t)go * d.ldo + gn] = v;
      else if (d.OUT2 && go >= d.o_split) d.OUT2[(int64_t)bo * d.o_batch + (int64_t)gn * d.ldo + go - d.o_split] = v;
      else d.OUT[(int64_t)bo * d.o_batch + (int64_t)gn * d.ldo + go] = v;
    }
  }
}

}  // namespace cbim

using namespace cbim;

extern "C" int cbim_map_gemm(const cbim_map_gemm_desc* d, void* stream) {
  CBIM_CHECK(d && d->A && d->X && d->OUT, CBIM_EINVAL, "map gemm: null operand");
  CBIM_CHECK(d->O > 0 && d->K > 0 && d->Nn > 0 && d->batch > 0, CBIM_EINVAL, "map gemm: empty problem %d x %d x %d, batch %d", d->O, d->K,
             d->Nn, d->batch);
  CBIM_CHECK(!d->ln_mode || (!d->x_t && d->Nn <= MG_NT && d->Xn && d->rstd_out && !d->reduce_batch), CBIM_EUNSUPPORTED,
             "map gemm: normalise-on-load needs X as [k][n] rows of at most %d columns", MG_NT);
  CBIM_CHECK(!d->XH || (!d->o_t && d->Nn <= MG_NT && d->rstd_in && !d->reduce_batch), CBIM_EUNSUPPORTED,
             "map gemm: the InstanceNorm-backward epilogue needs OUT as [o][n] rows of at most %d columns", MG_NT);
  CBIM_CHECK(!d->R || !d->o_t, CBIM_EUNSUPPORTED, "map gemm: a residual needs OUT as [o][n]");
  CBIM_CHECK(!d->A2 || d->a_t, CBIM_EUNSUPPORTED, "map gemm: a second A tensor needs A as [k][o]");
  CBIM_CHECK(!d->X2 || d->x_t, CBIM_EUNSUPPORTED, "map gemm: a second X tensor needs X as [n][k]");
  CBIM_CHECK(!d->OUT2 || d->o_t, CBIM_EUNSUPPORTED, "map gemm: a second OUT tensor needs OUT as [n][o]");
  dim3 grid((unsigned)((d->O + MG_OT - 1) / MG_OT), (unsigned)((d->Nn + MG_NT - 1) / MG_NT), (unsigned)(d->reduce_batch ? 1 : d->batch));
  CBIM_LAUNCH(k_map_gemm, grid, dim3(MG_T), 0, (hipStream_t)stream, *d);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "map gemm launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(map)
