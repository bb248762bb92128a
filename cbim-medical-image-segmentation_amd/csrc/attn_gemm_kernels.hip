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

// Layout: S float32 [L voxels][H heads][M codes] (what the row GEMM writes when its weight is the block matrix of all heads), seen here as
// R = L H rows of M codes.  A row is held by LPR = 4 | 8 | 16 lanes x 8 codes (the first LPR with 8 LPR >= M; lanes past M idle), a wave
// covers RW = 64 / LPR rows per pass; H <= RW and H | RW, so the head of a lane's rows (row % H = (lane / LPR) % H) never changes.
template <int LPR>
__global__ void __launch_bounds__(AG_T) k_awg_rows(const float* __restrict__ S, int64_t R, int M, int H, float scale,
                                                   bf16_t* __restrict__ P, float* __restrict__ rec) {
  constexpr int RW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane % LPR, lr = lane / LPR;
  const bool act = lc * 8 < M;
  const int64_t r0 = (int64_t)blockIdx.x * AG_ROWS;
  float cm[8], cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cm[j] = -INFINITY; cs[j] = 0.f; }
  for (int it = 0; it < AG_ROWS / (4 * RW); ++it) {
    const int64_t row = r0 + (int64_t)(it * 4 + wave) * RW + lr;
    const bool ok = row < R && act;
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
  // the rows of this wave that belong to the same head (lanes with the same code chunk and the same lr % H), then the four waves
  // through LDS in wave order
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    if (o < LPR * H) continue;                 // (lr bits below log2 H tell the heads apart)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m2 = __shfl_xor(cm[j], o, 64), s2 = __shfl_xor(cs[j], o, 64);
      ag_merge(cm[j], cs[j], m2, s2);          // (symmetric in its two pairs — fp multiply / add commute —: both partners hold the same result)
    }
  }
  __shared__ float red[4][RW][LPR * 8][2];     // [wave][head (< H <= RW)][code]
  if (lr < H && act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave][lr][lc * 8 + j][0] = cm[j]; red[wave][lr][lc * 8 + j][1] = cs[j]; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < H * M; t += AG_T) {
    const int hc = t / M, code = t % M;
    float m = red[0][hc][code][0], s = red[0][hc][code][1];
    for (int w = 1; w < 4; ++w) ag_merge(m, s, red[w][hc][code][0], red[w][hc][code][1]);
    rec[((size_t)blockIdx.x * H * M + t) * 2] = m;
    rec[((size_t)blockIdx.x * H * M + t) * 2 + 1] = s;
  }
}

// records [nrec][HM][2] -> lse[c] = max + log(sum) per column c = head * M + code: 64 columns per workgroup, 4 threads per column
// each over every 4th record in order, their partial pairs merged in thread order (fixed order: bit-reproducible)
__global__ void __launch_bounds__(AG_T) k_awg_merge(const float* __restrict__ rec, int nrec, int HM, float* __restrict__ lse) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float m = -INFINITY, s = 0.f;
  if (col < HM)
    for (int r = g; r < nrec; r += 4) ag_merge(m, s, rec[((size_t)r * HM + col) * 2], rec[((size_t)r * HM + col) * 2 + 1]);
  __shared__ float red[AG_T][2];
  red[threadIdx.x][0] = m; red[threadIdx.x][1] = s;
  __syncthreads();
  if (g == 0 && col < HM) {
    for (int k = 1; k < 4; ++k) ag_merge(m, s, red[k * 64 + threadIdx.x][0], red[k * 64 + threadIdx.x][1]);
    lse[col] = m + logf(s);
  }
}

// C[r][m] = exp(scale S[r][m] - lse[r % H][m]) -> bf16; thread = 8 codes of a row
__global__ void __launch_bounds__(AG_T) k_awg_cols(const float* __restrict__ S, int64_t R, int M, int H, float scale,
                                                   const float* __restrict__ lse, bf16_t* __restrict__ Cc) {
  const int cpr = M / 8;
  const int64_t i = (int64_t)blockIdx.x * AG_T + threadIdx.x;
  if (i >= R * cpr) return;
  const int64_t row = i / cpr;
  const int lc = (int)(i % cpr);
  const float* ls = lse + (size_t)(row % H) * M + lc * 8;
  const f32x4 a = *(const f32x4*)(S + row * M + lc * 8), b = *(const f32x4*)(S + row * M + lc * 8 + 4);
  const f32x4 la = *(const f32x4*)ls, lb = *(const f32x4*)(ls + 4);
  u32x4 v;
  v.x = pk_bf16(expf(a.x * scale - la.x), expf(a.y * scale - la.y));
  v.y = pk_bf16(expf(a.z * scale - la.z), expf(a.w * scale - la.w));
  v.z = pk_bf16(expf(b.x * scale - lb.x), expf(b.y * scale - lb.y));
  v.w = pk_bf16(expf(b.z * scale - lb.z), expf(b.w * scale - lb.w));
  *(u32x4*)(Cc + row * M + lc * 8) = v;
}

// colsum[h][m] = sum over the head's channels c = d H + h of dmo[m][c] mo[m][c] (= sum over the voxels of dC o C): one wave per (h, m)
__global__ void __launch_bounds__(64) k_awg_rowdot(const float* __restrict__ a, const float* __restrict__ b, int inner, int M, int H,
                                                   float* __restrict__ out) {
  const int h = blockIdx.x / M, m_ = blockIdx.x % M, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane * H + h; c < inner; c += 64 * H) s = fmaf(a[(size_t)m_ * inner + c], b[(size_t)m_ * inner + c], s);
  s = wave_sum(s);
  if (lane == 0) out[blockIdx.x] = s;
}

// dS = scale (P o (dP - rowsum(dP o P)) + C o (dC - colsum[row % H])) -> bf16
template <int LPR>
__global__ void __launch_bounds__(AG_T) k_awg_ds(const float* __restrict__ dP, const bf16_t* __restrict__ P,
                                                 const float* __restrict__ dC, const bf16_t* __restrict__ Cc,
                                                 const float* __restrict__ colsum, int64_t R, int M, int H, float scale,
                                                 bf16_t* __restrict__ dS) {
  constexpr int RW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane % LPR, lr = lane / LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RW + lr;
  const bool ok = row < R && lc * 8 < M;
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
    const float* cs = colsum + (size_t)(row % H) * M + lc * 8;
    float o_[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o_[j] = scale * (p[j] * (gp[j] - rs) + c[j] * (gc[j] - cs[j]));
    u32x4 v;
    v.x = pk_bf16(o_[0], o_[1]); v.y = pk_bf16(o_[2], o_[3]); v.z = pk_bf16(o_[4], o_[5]); v.w = pk_bf16(o_[6], o_[7]);
    *(u32x4*)(dS + row * M + lc * 8) = v;
  }
}

static int awg_lpr(int M) { return M <= 32 ? 4 : M <= 64 ? 8 : 16; }
static bool awg_shape_ok(int M, int H) {
  if (M < 8 || M > 128 || M % 8) return false;
  const int rw = 64 / awg_lpr(M);
  return (H == 1 || H == 2 || H == 4 || H == 8) && H <= rw;
}

}  // namespace cbim

using namespace cbim;

// P = softmax over the M codes of every (voxel, head) row of scale * S (bf16), rec[ceil(L H / CBIM_AWG_ROWS)][H][M][2] = column records
extern "C" int cbim_awg_rows(const float* S, int64_t L, int heads, int M, float scale, void* P, float* rec, void* stream) {
  CBIM_CHECK(S && P && rec && L >= 1, CBIM_EINVAL, "awg_rows: null operand / no rows");
  CBIM_CHECK(awg_shape_ok(M, heads), CBIM_EUNSUPPORTED, "awg_rows: %d codes x %d heads (codes 8..128 in multiples of 8; 1 | 2 | 4 | 8 heads, at most 64 / lanes-per-row)", M, heads);
  const int64_t R = L * heads;
  const dim3 grid((unsigned)((R + AG_ROWS - 1) / AG_ROWS));
  hipStream_t st = (hipStream_t)stream;
  switch (awg_lpr(M)) {
    case 4: CBIM_LAUNCH((k_awg_rows<4>), grid, dim3(AG_T), 0, st, S, R, M, heads, scale, (bf16_t*)P, rec); break;
    case 8: CBIM_LAUNCH((k_awg_rows<8>), grid, dim3(AG_T), 0, st, S, R, M, heads, scale, (bf16_t*)P, rec); break;
    default: CBIM_LAUNCH((k_awg_rows<16>), grid, dim3(AG_T), 0, st, S, R, M, heads, scale, (bf16_t*)P, rec); break;
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// lse[h][m] from the records, then C = exp(scale S - lse) (bf16): the softmax over the voxels of every (head, code)
extern "C" int cbim_awg_cols(const float* S, int64_t L, int heads, int M, float scale, const float* rec, float* lse, void* Cc, void* stream) {
  CBIM_CHECK(S && rec && lse && Cc && L >= 1, CBIM_EINVAL, "awg_cols: null operand / no rows");
  CBIM_CHECK(awg_shape_ok(M, heads), CBIM_EUNSUPPORTED, "awg_cols: %d codes x %d heads", M, heads);
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = L * heads;
  const int nrec = (int)((R + AG_ROWS - 1) / AG_ROWS), HM = heads * M;
  CBIM_LAUNCH(k_awg_merge, dim3((HM + 63) / 64), dim3(AG_T), 0, st, rec, nrec, HM, lse);
  const int64_t items = R * (M / 8);
  CBIM_LAUNCH(k_awg_cols, dim3((unsigned)((items + AG_T - 1) / AG_T)), dim3(AG_T), 0, st, S, R, M, heads, scale, (const float*)lse, (bf16_t*)Cc);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// dS (bf16) from dP, P, dC, C [L][H][M] and the map-side pair (dmo, mo) [M][inner], channel c = d H + h; colsum_ws: H M floats
extern "C" int cbim_awg_ds(const float* dP, const void* P, const float* dC, const void* Cc, const float* dmo, const float* mo, int inner,
                           int64_t L, int heads, int M, float scale, float* colsum_ws, void* dS, void* stream) {
  CBIM_CHECK(dP && P && dC && Cc && dmo && mo && colsum_ws && dS && L >= 1 && inner >= heads, CBIM_EINVAL, "awg_ds: null operand / bad sizes");
  CBIM_CHECK(awg_shape_ok(M, heads) && inner % heads == 0, CBIM_EUNSUPPORTED, "awg_ds: %d codes x %d heads, %d channels", M, heads, inner);
  hipStream_t st = (hipStream_t)stream;
  CBIM_LAUNCH(k_awg_rowdot, dim3(heads * M), dim3(64), 0, st, dmo, mo, inner, M, heads, colsum_ws);
  const int64_t R = L * heads;
  const int lpr = awg_lpr(M), rows_per_wg = 4 * (64 / lpr);
  const dim3 grid((unsigned)((R + rows_per_wg - 1) / rows_per_wg));
  switch (lpr) {
    case 4: CBIM_LAUNCH((k_awg_ds<4>), grid, dim3(AG_T), 0, st, dP, (const bf16_t*)P, dC, (const bf16_t*)Cc, (const float*)colsum_ws, R, M, heads, scale, (bf16_t*)dS); break;
    case 8: CBIM_LAUNCH((k_awg_ds<8>), grid, dim3(AG_T), 0, st, dP, (const bf16_t*)P, dC, (const bf16_t*)Cc, (const float*)colsum_ws, R, M, heads, scale, (bf16_t*)dS); break;
    default: CBIM_LAUNCH((k_awg_ds<16>), grid, dim3(AG_T), 0, st, dP, (const bf16_t*)P, dC, (const bf16_t*)Cc, (const float*)colsum_ws, R, M, heads, scale, (bf16_t*)dS); break;
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(awg)
