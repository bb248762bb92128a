// attn_gemm_kernels.hip — the softmax side of MedFormer's BidirectionAttention for ONE WIDE HEAD (round 6): what is left between
// the GEMMs when the core of /root/reference/model/dim3/medformer_utils.py:63-97 is written as matrix products on the engine's row-GEMM
// kernels (conv_pw.hip, token mode).  config/lits/medformer_3d.yaml builds every block with num_heads = 1, so d_head is the channel
// count (128 / 256 / 320 at 32^3 / 16^3 / 8^3); the register-resident MFMA kernel (attn_mfma.hip) is built for d_head 32 and the
// vector-ALU kernel that took these shapes (attn_wide.hip: thread = voxel, fp32 dot products against LDS broadcasts) spends 460 us
// forward / 880 us backward per block on < 2 GFLOP (profiles/r06_lits_medformer_kernels.txt: 26 ms of the 50 ms step).
//
// With S = scale Q MQ^T  ([L voxels, M codes], fp32 from the row GEMM), per block:
//     P = softmax over the codes of a voxel (rows of S)          feat_out = P MV                     (:80, :85)
//     C = softmax over the voxels of a code (columns of S)       map_out  = C^T FV                   (:82, :90)
// and backward, with dP = dfeat_out MV^T, dC = FV dmap_out^T (row GEMMs):
//     dS = scale ( P o (dP - rowsum(dP o P))  +  C o (dC - colsum(dC o C)) ),   colsum(dC o C)[m] = < dmap_out[m], map_out[m] >
//     dQ = dS MQ,  dMQ = dS^T Q,  dFV = C dmap_out,  dMV = P^T dfeat_out        (row / weight-gradient GEMMs)
// This file: rows of S -> P (bf16) + per-workgroup column records; records -> log-sum-exp per code; S -> C (bf16); the dS pass.
// P, C and dS are bf16 because they are MFMA operands next (attn_mfma.hip rounds the same quantities the same way); all sums
// are fp32 in a fixed order (records merged in workgroup order): bit-reproducible.
#include "cbim_common.h"

namespace cbim {

static constexpr int AG_T = 256;          // threads per workgroup
static constexpr int AG_ROWS = 256;       // rows of S per workgroup = one column record (include/cbim_hip.h: CBIM_AWG_ROWS)

__device__ __forceinline__ void ag_merge(float& m, float& s, float m2, float s2) {     // online-softmax pair (max, sum exp(. - max))
  const float mm = fmaxf(m, m2);
  const float a = m == mm ? 1.f : expf(m - mm), b = m2 == mm ? 1.f : expf(m2 - mm);   // (-inf - -inf never formed)
  s = s * a + s2 * b;
  m = mm;
}

// M codes = LPR lanes x 8; a wave covers 64 / LPR rows per pass
template <int LPR>
__global__ void __launch_bounds__(AG_T) k_awg_rows(const float* __restrict__ S, int64_t L, float scale, bf16_t* __restrict__ P,
                                                   float* __restrict__ rec) {
  constexpr int M = LPR * 8, RW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane % LPR, lr = lane / LPR;
  const int64_t r0 = (int64_t)blockIdx.x * AG_ROWS;
  float cm[8], cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cm[j] = -INFINITY; cs[j] = 0.f; }
  for (int it = 0; it < AG_ROWS / (4 * RW); ++it) {
    const int64_t row = r0 + (int64_t)(it * 4 + wave) * RW + lr;
    const bool ok = row < L;
    float s[8];
    if (ok) {
      const f32x4 a = *(const f32x4*)(S + row * M + lc * 8), b = *(const f32x4*)(S + row * M + lc * 8 + 4);
      s[0] = a.x * scale; s[1] = a.y * scale; s[2] = a.z * scale; s[3] = a.w * scale;
      s[4] = b.x * scale; s[5] = b.y * scale; s[6] = b.z * scale; s[7] = b.w * scale;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = -INFINITY;
    }
    float rm = s[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) rm = fmaxf(rm, s[j]);
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) rm = fmaxf(rm, __shfl_xor(rm, o, 64));
    float e[8], rs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { e[j] = ok ? expf(s[j] - rm) : 0.f; rs += e[j]; }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) rs += __shfl_xor(rs, o, 64);
    if (ok) {
      const float inv = 1.f / rs;
      u32x4 v;
      v.x = pk_bf16(e[0] * inv, e[1] * inv); v.y = pk_bf16(e[2] * inv, e[3] * inv);
      v.z = pk_bf16(e[4] * inv, e[5] * inv); v.w = pk_bf16(e[6] * inv, e[7] * inv);
      *(u32x4*)(P + row * M + lc * 8) = v;
#pragma unroll
      for (int j = 0; j < 8; ++j) ag_merge(cm[j], cs[j], s[j], 1.f);
    }
  }
  // the rows of this wave (lanes with the same code chunk), then the four waves through LDS in wave order
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m2 = __shfl_xor(cm[j], o, 64), s2 = __shfl_xor(cs[j], o, 64);
      ag_merge(cm[j], cs[j], m2, s2);      // (symmetric in its two pairs — fp multiply / add commute —: both partners hold the same result)
    }
  }
  __shared__ float red[4][M][2];
  if (lr == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave][lc * 8 + j][0] = cm[j]; red[wave][lc * 8 + j][1] = cs[j]; }
  }
  __syncthreads();
  if (threadIdx.x < M) {
    float m = red[0][threadIdx.x][0], s = red[0][threadIdx.x][1];
    for (int w = 1; w < 4; ++w) ag_merge(m, s, red[w][threadIdx.x][0], red[w][threadIdx.x][1]);
    rec[((size_t)blockIdx.x * M + threadIdx.x) * 2] = m;
    rec[((size_t)blockIdx.x * M + threadIdx.x) * 2 + 1] = s;
  }
}

// records [nrec][M][2] -> lse[m] = max + log(sum): 256 / M threads per code, each over every (256 / M)-th record in order, their
// partial pairs merged in thread order (fixed order: bit-reproducible)
__global__ void __launch_bounds__(AG_T) k_awg_merge(const float* __restrict__ rec, int nrec, int M, float* __restrict__ lse) {
  const int m_ = threadIdx.x % M, g = threadIdx.x / M, G = AG_T / M;
  float m = -INFINITY, s = 0.f;
  for (int r = g; r < nrec; r += G) ag_merge(m, s, rec[((size_t)r * M + m_) * 2], rec[((size_t)r * M + m_) * 2 + 1]);
  __shared__ float red[AG_T][2];
  red[threadIdx.x][0] = m; red[threadIdx.x][1] = s;
  __syncthreads();
  if (g == 0) {
    for (int k = 1; k < G; ++k) ag_merge(m, s, red[k * M + m_][0], red[k * M + m_][1]);
    lse[m_] = m + logf(s);
  }
}

// C[r][m] = exp(scale S[r][m] - lse[m]) -> bf16; thread = 8 codes of a row
__global__ void __launch_bounds__(AG_T) k_awg_cols(const float* __restrict__ S, int64_t L, int M, float scale,
                                                   const float* __restrict__ lse, bf16_t* __restrict__ Cc) {
  const int cpr = M / 8;
  const int64_t i = (int64_t)blockIdx.x * AG_T + threadIdx.x;
  if (i >= L * cpr) return;
  const int64_t row = i / cpr;
  const int lc = (int)(i % cpr);
  const f32x4 a = *(const f32x4*)(S + row * M + lc * 8), b = *(const f32x4*)(S + row * M + lc * 8 + 4);
  const f32x4 la = *(const f32x4*)(lse + lc * 8), lb = *(const f32x4*)(lse + lc * 8 + 4);
  u32x4 v;
  v.x = pk_bf16(expf(a.x * scale - la.x), expf(a.y * scale - la.y));
  v.y = pk_bf16(expf(a.z * scale - la.z), expf(a.w * scale - la.w));
  v.z = pk_bf16(expf(b.x * scale - lb.x), expf(b.y * scale - lb.y));
  v.w = pk_bf16(expf(b.z * scale - lb.z), expf(b.w * scale - lb.w));
  *(u32x4*)(Cc + row * M + lc * 8) = v;
}

// colsum[m] = < dmo[m][:], mo[m][:] > over D (= sum over the voxels of dC o C): one wave per code
__global__ void __launch_bounds__(64) k_awg_rowdot(const float* __restrict__ a, const float* __restrict__ b, int D,
                                                   float* __restrict__ out) {
  const int m_ = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s = fmaf(a[(size_t)m_ * D + d], b[(size_t)m_ * D + d], s);
  s = wave_sum(s);
  if (lane == 0) out[m_] = s;
}

// dS = scale (P o (dP - rowsum(dP o P)) + C o (dC - colsum)) -> bf16
template <int LPR>
__global__ void __launch_bounds__(AG_T) k_awg_ds(const float* __restrict__ dP, const bf16_t* __restrict__ P,
                                                 const float* __restrict__ dC, const bf16_t* __restrict__ Cc,
                                                 const float* __restrict__ colsum, int64_t L, float scale, bf16_t* __restrict__ dS) {
  constexpr int M = LPR * 8, RW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane % LPR, lr = lane / LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RW + lr;
  const bool ok = row < L;
  float p[8], c[8], gp[8], gc[8];
  if (ok) {
    const u32x4 pv = *(const u32x4*)(P + row * M + lc * 8), cv = *(const u32x4*)(Cc + row * M + lc * 8);
    const unsigned pw[4] = {pv.x, pv.y, pv.z, pv.w}, cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      p[2 * j] = __uint_as_float(pw[j] << 16); p[2 * j + 1] = __uint_as_float(pw[j] & 0xffff0000u);
      c[2 * j] = __uint_as_float(cw[j] << 16); c[2 * j + 1] = __uint_as_float(cw[j] & 0xffff0000u);
    }
    const f32x4 a = *(const f32x4*)(dP + row * M + lc * 8), b = *(const f32x4*)(dP + row * M + lc * 8 + 4);
    const f32x4 e = *(const f32x4*)(dC + row * M + lc * 8), f = *(const f32x4*)(dC + row * M + lc * 8 + 4);
    gp[0] = a.x; gp[1] = a.y; gp[2] = a.z; gp[3] = a.w; gp[4] = b.x; gp[5] = b.y; gp[6] = b.z; gp[7] = b.w;
    gc[0] = e.x; gc[1] = e.y; gc[2] = e.z; gc[3] = e.w; gc[4] = f.x; gc[5] = f.y; gc[6] = f.z; gc[7] = f.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = c[j] = gp[j] = gc[j] = 0.f;
  }
  float rs = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) rs = fmaf(gp[j], p[j], rs);
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) rs += __shfl_xor(rs, o, 64);
  if (ok) {
    float o_[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o_[j] = scale * (p[j] * (gp[j] - rs) + c[j] * (gc[j] - colsum[lc * 8 + j]));
    u32x4 v;
    v.x = pk_bf16(o_[0], o_[1]); v.y = pk_bf16(o_[2], o_[3]); v.z = pk_bf16(o_[4], o_[5]); v.w = pk_bf16(o_[6], o_[7]);
    *(u32x4*)(dS + row * M + lc * 8) = v;
  }
}

static bool awg_codes_ok(int M) { return M == 32 || M == 64 || M == 128; }

}  // namespace cbim

using namespace cbim;

// P = softmax over the M codes of every row of scale * S (bf16), rec[ceil(L / CBIM_AWG_ROWS)][M][2] = column records
extern "C" int cbim_awg_rows(const float* S, int64_t L, int M, float scale, void* P, float* rec, void* stream) {
  CBIM_CHECK(S && P && rec && L >= 1, CBIM_EINVAL, "awg_rows: null operand / no rows");
  CBIM_CHECK(awg_codes_ok(M), CBIM_EUNSUPPORTED, "awg_rows: %d codes (32, 64 or 128)", M);
  const dim3 grid((unsigned)((L + AG_ROWS - 1) / AG_ROWS));
  hipStream_t st = (hipStream_t)stream;
  if (M == 32) CBIM_LAUNCH((k_awg_rows<4>), grid, dim3(AG_T), 0, st, S, L, scale, (bf16_t*)P, rec);
  else if (M == 64) CBIM_LAUNCH((k_awg_rows<8>), grid, dim3(AG_T), 0, st, S, L, scale, (bf16_t*)P, rec);
  else CBIM_LAUNCH((k_awg_rows<16>), grid, dim3(AG_T), 0, st, S, L, scale, (bf16_t*)P, rec);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// lse[m] from the records, then C = exp(scale S - lse) (bf16): the softmax over the voxels of every code
extern "C" int cbim_awg_cols(const float* S, int64_t L, int M, float scale, const float* rec, float* lse, void* Cc, void* stream) {
  CBIM_CHECK(S && rec && lse && Cc && L >= 1, CBIM_EINVAL, "awg_cols: null operand / no rows");
  CBIM_CHECK(awg_codes_ok(M), CBIM_EUNSUPPORTED, "awg_cols: %d codes (32, 64 or 128)", M);
  hipStream_t st = (hipStream_t)stream;
  const int nrec = (int)((L + AG_ROWS - 1) / AG_ROWS);
  CBIM_LAUNCH(k_awg_merge, dim3(1), dim3(AG_T), 0, st, rec, nrec, M, lse);
  const int64_t items = L * (M / 8);
  CBIM_LAUNCH(k_awg_cols, dim3((unsigned)((items + AG_T - 1) / AG_T)), dim3(AG_T), 0, st, S, L, M, scale, (const float*)lse, (bf16_t*)Cc);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// dS (bf16) from dP, P, dC, C and the map-side pair (dmo, mo) [M, D]; colsum_ws: M floats of scratch
extern "C" int cbim_awg_ds(const float* dP, const void* P, const float* dC, const void* Cc, const float* dmo, const float* mo, int D,
                           int64_t L, int M, float scale, float* colsum_ws, void* dS, void* stream) {
  CBIM_CHECK(dP && P && dC && Cc && dmo && mo && colsum_ws && dS && L >= 1 && D >= 1, CBIM_EINVAL, "awg_ds: null operand / bad sizes");
  CBIM_CHECK(awg_codes_ok(M), CBIM_EUNSUPPORTED, "awg_ds: %d codes (32, 64 or 128)", M);
  hipStream_t st = (hipStream_t)stream;
  CBIM_LAUNCH(k_awg_rowdot, dim3(M), dim3(64), 0, st, dmo, mo, D, colsum_ws);
  const int rows_per_wg = 4 * (64 / (M / 8));
  const dim3 grid((unsigned)((L + rows_per_wg - 1) / rows_per_wg));
  if (M == 32) CBIM_LAUNCH((k_awg_ds<4>), grid, dim3(AG_T), 0, st, dP, (const bf16_t*)P, dC, (const bf16_t*)Cc, (const float*)colsum_ws, L, scale, (bf16_t*)dS);
  else if (M == 64) CBIM_LAUNCH((k_awg_ds<8>), grid, dim3(AG_T), 0, st, dP, (const bf16_t*)P, dC, (const bf16_t*)Cc, (const float*)colsum_ws, L, scale, (bf16_t*)dS);
  else CBIM_LAUNCH((k_awg_ds<16>), grid, dim3(AG_T), 0, st, dP, (const bf16_t*)P, dC, (const bf16_t*)Cc, (const float*)colsum_ws, L, scale, (bf16_t*)dS);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(awg)
