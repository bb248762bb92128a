// layernorm_kernels.hip — nn.LayerNorm over the channel axis of channels-last token rows (SwinUNETR trunk).
//
// Replaces the aten::native_layer_norm / native_layer_norm_backward launches behind
//   SwinTransformerBlock.norm1 / norm2            (/root/reference/model/dim3/swin_unetr.py:539,550,556,611)
//   PatchMerging.norm (8 x dim channels)          (swin_unetr.py:679,728)
//   SwinTransformer.proj_out = F.layer_norm(x, [C]) without affine parameters   (swin_unetr.py:970-983)
// The rows are short (48 ... 3072 fp32 channels) and there are up to 262 144 of them: ATen's kernels spend 49 us per launch
// on them (profiles/r03_d_swin_unetr_kernels.txt: 10.9 % of the SwinUNETR step, three launches per LayerNorm in training).
//
// Layout: x float [rows][C] (the residual stream of the trunk stays fp32 in both engine modes); y in fp32 or bf16 (the
// token Linears consume bf16 in the bf16 engine mode — the cast rides on the store); rowstats float [rows][2] = (mean, rstd).
// A row is owned by LPR = 16 / 32 / 64 lanes of one wave (float4 chunks, K per lane), so mean and the CENTRED variance are
// two butterfly reductions over registers; bandwidth bound: 1 read + 1 write forward, 2 reads + 1 write backward.
// d(gamma), d(beta): per-lane running sums over the rows a lane visits (its channel chunks never change), combined across
// the workgroup's row groups through LDS in fixed order, one record per workgroup, summed by a second kernel in block
// order — deterministic, no atomics.
#include "cbim_common.h"

namespace cbim {

static constexpr int LN_T = 256;

template <int K, bool BF16_OUT>
__global__ void __launch_bounds__(LN_T) k_layernorm_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, void* __restrict__ y,
                                                        float* __restrict__ rowstats, int64_t rows, int C, int lpr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / lpr, sub = lane / lpr, l = lane % lpr;
  const int chunks = C / 4;
  const float inv_c = 1.f / (float)C;
  f32x4 gm[K], bt[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = l + k * lpr;
    gm[k] = f32x4{1.f, 1.f, 1.f, 1.f};
    bt[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < chunks) {
      if (gamma) gm[k] = *(const f32x4*)(gamma + 4 * c);
      if (beta) bt[k] = *(const f32x4*)(beta + 4 * c);
    }
  }
  const int64_t rows_per_it = (int64_t)gridDim.x * 4 * rpw;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rpw + sub; r0 - sub < rows; r0 += rows_per_it) {
    const bool live = r0 < rows;                      // (all lanes of the wave run the shuffles)
    f32x4 v[K];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = l + k * lpr;
      v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live && c < chunks) v[k] = *(const f32x4*)(x + r0 * C + 4 * c);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    for (int m = 1; m < lpr; m <<= 1) s += __shfl_xor(s, m, 64);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = l + k * lpr;
      if (c < chunks) {
        const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    for (int m = 1; m < lpr; m <<= 1) q += __shfl_xor(q, m, 64);
    const float rstd = rsqrtf(q * inv_c + eps);
    if (!live) continue;
    if (l == 0 && rowstats) { rowstats[2 * r0] = mean; rowstats[2 * r0 + 1] = rstd; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = l + k * lpr;
      if (c >= chunks) continue;
      const float o0 = (v[k].x - mean) * rstd * gm[k].x + bt[k].x, o1 = (v[k].y - mean) * rstd * gm[k].y + bt[k].y,
                  o2 = (v[k].z - mean) * rstd * gm[k].z + bt[k].z, o3 = (v[k].w - mean) * rstd * gm[k].w + bt[k].w;
      if (BF16_OUT) *(u32x2*)((bf16_t*)y + r0 * C + 4 * c) = u32x2{pk_bf16(o0, o1), pk_bf16(o2, o3)};
      else *(f32x4*)((float*)y + r0 * C + 4 * c) = f32x4{o0, o1, o2, o3};
    }
  }
}

// dx = rstd * (g - mean(g) - xh * mean(g * xh)),  g = dy * gamma;  partial d(gamma) = sum dy * xh, d(beta) = sum dy
template <int K, bool BF16_IN>
__global__ void __launch_bounds__(LN_T) k_layernorm_bwd(const void* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ rowstats,
                                                        float* __restrict__ dx, float* __restrict__ partials, int64_t rows,
                                                        int C, int lpr, const float* __restrict__ add) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / lpr, sub = lane / lpr, l = lane % lpr;
  const int chunks = C / 4;
  const float inv_c = 1.f / (float)C;
  f32x4 gm[K], dg[K], db[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = l + k * lpr;
    gm[k] = f32x4{1.f, 1.f, 1.f, 1.f};
    if (gamma && c < chunks) gm[k] = *(const f32x4*)(gamma + 4 * c);
    dg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t rows_per_it = (int64_t)gridDim.x * 4 * rpw;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rpw + sub; r0 - sub < rows; r0 += rows_per_it) {
    const bool live = r0 < rows;
    float mean = 0.f, rstd = 0.f;
    if (live) { mean = rowstats[2 * r0]; rstd = rowstats[2 * r0 + 1]; }
    f32x4 g[K], xh[K];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = l + k * lpr;
      g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      xh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live && c < chunks) {
        f32x4 d;
        if (BF16_IN) {
          const u32x2 w = *(const u32x2*)((const bf16_t*)dy + r0 * C + 4 * c);
          d = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                    __uint_as_float(w.y & 0xffff0000u)};
        } else d = *(const f32x4*)((const float*)dy + r0 * C + 4 * c);
        const f32x4 xv = *(const f32x4*)(x + r0 * C + 4 * c);
        xh[k] = f32x4{(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
        db[k] += d;
        dg[k] += d * xh[k];
        g[k] = d * gm[k];
        a += (g[k].x + g[k].y) + (g[k].z + g[k].w);
        b += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
      }
    }
    for (int m = 1; m < lpr; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
    if (!live) continue;
    const float ma = a * inv_c, mb = b * inv_c;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = l + k * lpr;
      if (c >= chunks) continue;
      f32x4 o = f32x4{rstd * (g[k].x - ma - xh[k].x * mb), rstd * (g[k].y - ma - xh[k].y * mb),
                      rstd * (g[k].z - ma - xh[k].z * mb), rstd * (g[k].w - ma - xh[k].w * mb)};
      if (add) {      // + the gradient that reached the same rows past the norm (the residual stream: x + f(LN(x)))
        const f32x4 a4 = *(const f32x4*)(add + r0 * C + 4 * c);
        o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
      }
      *(f32x4*)(dx + r0 * C + 4 * c) = o;
    }
  }
  if (!partials) return;
  // the 4 * rpw row groups of the workgroup hold partial sums of the same channel chunks: add them in group order
  __shared__ f32x4 red[LN_T * 2];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    __syncthreads();
    red[threadIdx.x * 2] = dg[k];
    red[threadIdx.x * 2 + 1] = db[k];
    __syncthreads();
    const int c = l + k * lpr;
    if (wave == 0 && sub == 0 && c < chunks) {
      f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < 4; ++w)
        for (int s2 = 0; s2 < rpw; ++s2) {
          const int t = w * 64 + s2 * lpr + l;
          sg += red[t * 2];
          sb += red[t * 2 + 1];
        }
      float* o = partials + ((size_t)blockIdx.x * 2) * C + 4 * c;
      *(f32x4*)o = sg;
      *(f32x4*)(o + C) = sb;
    }
  }
}

// dgamma[c] = sum over blocks of partials[b][0][c], dbeta likewise.  1024 threads = 64 values x 16 block phases: phase ph
// adds blocks ph, ph + 16, ... in four interleaved chains (16 independent loads in flight), the 16 phase sums meet in LDS
// and are added in phase order.  (One thread per value walking 2048 blocks took 185 us per call.)
static constexpr int LNF_T = 1024;
__global__ void __launch_bounds__(LNF_T) k_layernorm_bwd_finish(const float* __restrict__ partials, int P, int C,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[LNF_T];
  const int li = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + li;                  // 0 .. 2C-1
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < 2 * C) {
    int b = ph;
    for (; b + 48 < P; b += 64) {
      a[0] += partials[(size_t)b * 2 * C + i];
      a[1] += partials[(size_t)(b + 16) * 2 * C + i];
      a[2] += partials[(size_t)(b + 32) * 2 * C + i];
      a[3] += partials[(size_t)(b + 48) * 2 * C + i];
    }
    for (int k = 0; b < P; b += 16, ++k) a[k] += partials[(size_t)b * 2 * C + i];
  }
  red[threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (ph == 0 && i < 2 * C) {
    float s = red[li];
#pragma unroll
    for (int q = 1; q < 16; ++q) s += red[q * 64 + li];
    if (i < C) { if (dgamma) dgamma[i] = s; }
    else if (dbeta) dbeta[i - C] = s;
  }
}

// Column sums of token rows: out[c] = sum over rows of x[row][c] (the bias gradient of the trunk's token Linears,
// swin_unetr.py:467-490,640-643: `g.float().sum(0)` was a full fp32 copy of the [tokens, C] gradient plus an ATen reduction).
// thread = (8-channel chunk, row lane); rows strided over the grid; per-workgroup records summed in block order by
// k_layernorm_bwd_finish's phase reduction.
template <bool BF16_IN>
__global__ void __launch_bounds__(LN_T) k_colsum_partial(const void* __restrict__ x, int64_t rows, int C, float* __restrict__ partials) {
  constexpr int CP = BF16_IN ? 8 : 4;                 // channels per 16-byte chunk
  const int cch = C / CP, rl = LN_T / cch;
  const int cc = threadIdx.x % cch, rr = threadIdx.x / cch;
  float a[CP];
#pragma unroll
  for (int j = 0; j < CP; ++j) a[j] = 0.f;
  if (rr < rl) {
    for (int64_t r = (int64_t)blockIdx.x * rl + rr; r < rows; r += (int64_t)gridDim.x * rl) {
      const u32x4 v = *(const u32x4*)((const char*)x + (r * C + (int64_t)cc * CP) * (BF16_IN ? 2 : 4));
      float f[CP];
      if (BF16_IN) Elem<bf16_tag>::unpack(v, f);
      else { f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w); }
#pragma unroll
      for (int j = 0; j < CP; ++j) a[j] += f[j];
    }
  }
  __shared__ float red[LN_T * 8];
#pragma unroll
  for (int j = 0; j < CP; ++j) red[threadIdx.x * CP + j] = a[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += LN_T) {
    const int c2 = c / CP, j = c % CP;
    float s = 0.f;
    for (int q = 0; q < rl; ++q) s += red[(q * cch + c2) * CP + j];
    partials[(size_t)blockIdx.x * C + c] = s;
  }
}

}  // namespace cbim

using namespace cbim;

static int ln_lpr(int C) {
  const int chunks = C / 4;
  int l = 16;
  while (l < 64 && l < chunks) l <<= 1;
  return l;
}
static int ln_blocks(int64_t rows, int lpr) {
  const int64_t per = 4 * (64 / lpr);
  int64_t b = (rows + per - 1) / per;
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}

extern "C" size_t cbim_layernorm_bwd_workspace(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C % 4) return 0;
  return (size_t)ln_blocks(rows, ln_lpr(C)) * 2 * C * sizeof(float);
}

#define LN_DISPATCH_K(KERNEL, FLAG, ...)                                                        \
  do {                                                                                          \
    if (kk <= 1) CBIM_LAUNCH((KERNEL<1, FLAG>), grid, dim3(LN_T), 0, st, __VA_ARGS__);          \
    else if (kk <= 2) CBIM_LAUNCH((KERNEL<2, FLAG>), grid, dim3(LN_T), 0, st, __VA_ARGS__);     \
    else if (kk <= 3) CBIM_LAUNCH((KERNEL<3, FLAG>), grid, dim3(LN_T), 0, st, __VA_ARGS__);     \
    else if (kk <= 6) CBIM_LAUNCH((KERNEL<6, FLAG>), grid, dim3(LN_T), 0, st, __VA_ARGS__);     \
    else CBIM_LAUNCH((KERNEL<12, FLAG>), grid, dim3(LN_T), 0, st, __VA_ARGS__);                 \
  } while (0)

extern "C" int cbim_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int out_dtype, void* y,
                                  float* rowstats, int64_t rows, int C, void* stream) {
  CBIM_CHECK(x && y && rows > 0, CBIM_EINVAL, "null argument");
  CBIM_CHECK(out_dtype == CBIM_F32 || out_dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", out_dtype);
  CBIM_CHECK(C >= 4 && C % 4 == 0 && C <= 3072, CBIM_EUNSUPPORTED, "layernorm: %d channels (a multiple of 4 up to 3072)", C);
  const int lpr = ln_lpr(C), kk = (C / 4 + lpr - 1) / lpr;
  dim3 grid((unsigned)ln_blocks(rows, lpr));
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == CBIM_BF16) LN_DISPATCH_K(k_layernorm_fwd, true, x, gamma, beta, eps, y, rowstats, rows, C, lpr);
  else LN_DISPATCH_K(k_layernorm_fwd, false, x, gamma, beta, eps, y, rowstats, rows, C, lpr);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_layernorm_bwd(int dy_dtype, const void* dy, const float* x, const float* gamma, const float* rowstats,
                                  const float* add, float* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                                  int64_t rows, int C, void* stream) {
  CBIM_CHECK(dy && x && rowstats && dx && rows > 0, CBIM_EINVAL, "null argument");
  CBIM_CHECK(dy_dtype == CBIM_F32 || dy_dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dy_dtype);
  CBIM_CHECK(C >= 4 && C % 4 == 0 && C <= 3072, CBIM_EUNSUPPORTED, "layernorm: %d channels (a multiple of 4 up to 3072)", C);
  const bool want_p = dgamma || dbeta;
  CBIM_CHECK(!want_p || (workspace && ws_bytes >= cbim_layernorm_bwd_workspace(rows, C)), CBIM_EWORKSPACE,
             "layernorm_bwd workspace %zu < %zu", ws_bytes, cbim_layernorm_bwd_workspace(rows, C));
  const int lpr = ln_lpr(C), kk = (C / 4 + lpr - 1) / lpr;
  const int P = ln_blocks(rows, lpr);
  dim3 grid((unsigned)P);
  hipStream_t st = (hipStream_t)stream;
  float* part = want_p ? (float*)workspace : nullptr;
  if (dy_dtype == CBIM_BF16) LN_DISPATCH_K(k_layernorm_bwd, true, dy, x, gamma, rowstats, dx, part, rows, C, lpr, add);
  else LN_DISPATCH_K(k_layernorm_bwd, false, dy, x, gamma, rowstats, dx, part, rows, C, lpr, add);
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  if (want_p) {
    CBIM_LAUNCH(k_layernorm_bwd_finish, dim3((unsigned)((2 * C + 63) / 64)), dim3(LNF_T), 0, st, (const float*)part, P, C, dgamma, dbeta);
    if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  }
  return CBIM_OK;
}

extern "C" size_t cbim_colsum_workspace(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)1024 * C * sizeof(float);
}

extern "C" int cbim_colsum(int dtype, const void* x, int64_t rows, int C, float* out, void* workspace, size_t ws_bytes, void* stream) {
  CBIM_CHECK(x && out && rows > 0, CBIM_EINVAL, "null argument");
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  const int cp = dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(C % cp == 0 && C / cp <= LN_T && C % 2 == 0, CBIM_EUNSUPPORTED, "colsum: %d channels (a multiple of %d up to %d)", C, cp, LN_T * cp);
  CBIM_CHECK(workspace && ws_bytes >= cbim_colsum_workspace(rows, C), CBIM_EWORKSPACE, "colsum workspace too small");
  const int rl = LN_T / (C / cp);
  int64_t P = (rows + (int64_t)rl * 16 - 1) / ((int64_t)rl * 16);       // ~16 rows per thread
  if (P > 1024) P = 1024;
  if (P < 1) P = 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CBIM_BF16) CBIM_LAUNCH((k_colsum_partial<true>), dim3((unsigned)P), dim3(LN_T), 0, st, x, rows, C, (float*)workspace);
  else CBIM_LAUNCH((k_colsum_partial<false>), dim3((unsigned)P), dim3(LN_T), 0, st, x, rows, C, (float*)workspace);
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  // records [P][C] summed in block order: the finish kernel takes [P][2 C'] with C' = C / 2 and two output halves
  CBIM_LAUNCH(k_layernorm_bwd_finish, dim3((unsigned)((C + 63) / 64)), dim3(LNF_T), 0, st, (const float*)workspace, (int)P, C / 2, out, out + C / 2);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(layernorm)
