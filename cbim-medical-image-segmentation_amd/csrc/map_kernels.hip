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
// Tile: 256 threads, 16 output rows x 64 (128) columns per workgroup, K in chunks of 64 through LDS.  Inside a chunk the four
// waves split the K steps (16 each) and every lane keeps a 4 x 4 (4 x 8) register tile of the WHOLE workgroup tile: a k-step is two
// 16-byte LDS reads for 16 (32) FMAs; the four partial tiles meet in LDS at the end and are added in wave order.  These GEMMs are
// parallelism- and latency-starved (13 MFLOP over 20-120 workgroups), and the round-6 measurements say where the time goes:
//   v1  operands staged by scalar loads and waited for, wave = 8 rows x all columns                         73 us per launch
//   v2  register prefetch of the next chunk, 16-byte loads, 16-row tiles                                    22 us
//   v3  the A rows by wave-uniform scalar loads (s_load_dwordx8/16 straight from global memory)             51 us  (rejected)
//   v4  v2 + the row normalisation by quad DPP sums instead of twelve ds_bpermute per row                   18 us  (G1 58 -> 25)
//   v5  v4 + four chunks in flight                                                                          18 us  (not the loads)
//   v6  register tiles + K split over the waves: v2-v5 read A as a 64-lane broadcast ds_read_b128 per four FMAs — the LDS
//       pipe, not the FMA pipe or the loads, set the 2-5 us per chunk                                        10 us  (this)
//   v7  v6 + four chunks in flight                                                                          11 us  (rejected)
// What is left is ~4.5 us of launch-to-first-result per kernel (an empty-ish kernel such as k_stats_finalize measures 5 us in the
// same trace) and ~1.2 us per chunk of barrier / staging / dependent-load chain at one workgroup per CU.
// (profiles/r06_h .. r06_n_medformer_kernels.txt; the aten::mm launches it replaces take ~11 us each.)  fp32 FMA chains in a fixed
// order: bit-reproducible.
#include "cbim_common.h"

namespace cbim {

static constexpr int MG_T = 256, MG_OT = 16, MG_KC = 64, MG_PA = 20;

// butterfly partner inside a quad / a row of 16 lanes by DPP (no LDS round trip; __shfl_xor is a ds_bpermute: ~100 cycles each —
// twelve dependent ones per map row made the normalisation 10 us per chunk), across rows by ds_bpermute
template <int MSK>
__device__ __forceinline__ float mg_bfly(float v) {
#ifdef CBIM_EMU
  return __shfl_xor(v, MSK, 64);
#else
  if (MSK >= 16) return __shfl_xor(v, MSK, 64);
  constexpr int ctrl = MSK == 1 ? 0xB1 : MSK == 2 ? 0x4E : MSK == 4 ? 0x141 : 0x140;   // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false));
#endif
}
__device__ __forceinline__ float mg_wave_sum(float v) {
  v += mg_bfly<1>(v); v += mg_bfly<2>(v); v += mg_bfly<4>(v); v += mg_bfly<8>(v); v += mg_bfly<16>(v); v += mg_bfly<32>(v);
  return v;
}
__device__ __forceinline__ float mg_quad_sum(float v) {
  v += mg_bfly<1>(v); v += mg_bfly<2>(v);
  return v;
}

// the integers of cbim_map_gemm_desc (the pointers travel as __restrict__ kernel arguments)
struct MgDims {
  int64_t lda, a_batch, ldx, x_batch, ldo, o_batch, ldr, r_batch;
  int a_t, a_split, x_t, x_split, o_t, o_split;
  int O, K, Nn, batch, reduce_batch, ln_mode;
  float eps;
};

// NC: 64-column blocks per tile (1: n-tile 64, 2: n-tile 128); VEC: every operand row is a whole number of aligned 16-byte groups
template <int NC, bool VEC>
__global__ void __launch_bounds__(MG_T) k_map_gemm(const float* __restrict__ gA, const float* __restrict__ gA2, const float* __restrict__ gX,
                                                   const float* __restrict__ gX2, float* __restrict__ gOUT, float* __restrict__ gOUT2,
                                                   const float* __restrict__ gR, float* __restrict__ gXn, float* __restrict__ g_rstd_out,
                                                   const float* __restrict__ gXH, const float* __restrict__ g_rstd_in, MgDims d) {
  constexpr int NT = 64 * NC, XR = 16 * NC, PX = NT + 4;   // columns per tile; X values per thread and chunk; pitch of Xs
  __shared__ __attribute__((aligned(16))) float At[MG_KC * MG_PA];   // [k][o]
  __shared__ __attribute__((aligned(16))) float Xs[MG_KC * PX];      // [k][n]; at the end: the waves' partial tiles [4][16][NT]
  static_assert(4 * MG_OT * NT <= MG_KC * PX, "partial tiles fit the X tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ty = lane >> 4, tx = lane & 15;            // register tile: rows 4 ty .. + 3, columns 4 tx .. + 3 (+ 64 h)
  const int o0 = blockIdx.x * MG_OT, n0 = blockIdx.y * NT;
  const int b0 = d.reduce_batch ? 0 : (int)blockIdx.z, b1 = d.reduce_batch ? d.batch : b0 + 1;
  f32x4 acc[4][NC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int h = 0; h < NC; ++h) acc[j][h] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv_n = 1.f / (float)d.Nn;
  const int nchunk = (d.K + MG_KC - 1) / MG_KC, nunits = (b1 - b0) * nchunk;
  float ra[4], rx[XR];
  // ---- the operand values of unit u = (image, K chunk) this thread stages ------------------------------------------------------
  // A: 16 o x 64 k = 256 groups of 4 (along k for a_t = 0, along o for a_t = 1: consecutive addresses either way)
  // X: 64 k x NT n = XR / 4 groups of 4 per thread — along n for x_t = 0; along k for x_t = 1, where the lanes of a wave take 64
  //    consecutive n so that their LDS stores are conflict free
  auto fetch = [&](int u) {
    const int b = b0 + u / nchunk, k0 = (u % nchunk) * MG_KC;
    const float* const X = gX + (int64_t)b * d.x_batch;
    const float* const X2 = gX2 ? gX2 + (int64_t)b * d.x_batch : nullptr;
    {
      const float* const A = gA + (int64_t)b * d.a_batch;
      const float* const A2 = gA2 ? gA2 + (int64_t)b * d.a_batch : nullptr;
      int o, k;
      if (d.a_t) { k = tid >> 2; o = (tid & 3) * 4; } else { o = tid >> 4; k = (tid & 15) * 4; }
      const int go = o0 + o, gk = k0 + k;
      const float* src;
      if (!d.a_t) src = A + (int64_t)go * d.lda + gk;
      else if (A2 && go >= d.a_split) src = A2 + (int64_t)gk * d.lda + go - d.a_split;   // (splits are multiples of 4: whole groups)
      else src = A + (int64_t)gk * d.lda + go;
      const int lim = d.a_t ? d.O - go : d.K - gk;          // valid elements of the group
      const bool row_ok = d.a_t ? gk < d.K : go < d.O;
      if (VEC) {
        const f32x4 v = (row_ok && lim > 0) ? *(const f32x4*)src : f32x4{0.f, 0.f, 0.f, 0.f};
        ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = (row_ok && i < lim) ? src[i] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < XR / 4; ++i) {
      const int e = tid + MG_T * i;
      int k, n;
      if (d.x_t) { n = e % NT; k = (e / NT) * 4; } else { k = e / (NT / 4); n = (e % (NT / 4)) * 4; }
      const int gk = k0 + k, gn = n0 + n;
      const float* src;
      if (!d.x_t) src = X + (int64_t)gk * d.ldx + gn;
      else if (X2 && gk >= d.x_split) src = X2 + (int64_t)gn * d.ldx + gk - d.x_split;
      else src = X + (int64_t)gn * d.ldx + gk;
      const int lim = d.x_t ? d.K - gk : d.Nn - gn;
      const bool row_ok = d.x_t ? gn < d.Nn : gk < d.K;
      if (VEC) {
        const f32x4 v = (row_ok && lim > 0) ? *(const f32x4*)src : f32x4{0.f, 0.f, 0.f, 0.f};
        rx[4 * i] = v.x; rx[4 * i + 1] = v.y; rx[4 * i + 2] = v.z; rx[4 * i + 3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) rx[4 * i + j] = (row_ok && j < lim) ? src[j] : 0.f;
      }
    }
  };
  auto stage = [&]() {
    {
      int o, k;
      if (d.a_t) { k = tid >> 2; o = (tid & 3) * 4; *(f32x4*)(At + k * MG_PA + o) = f32x4{ra[0], ra[1], ra[2], ra[3]}; }
      else {
        o = tid >> 4; k = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) At[(k + i) * MG_PA + o] = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < XR / 4; ++i) {
      const int e = tid + MG_T * i;
      if (d.x_t) {
        const int n = e % NT, k = (e / NT) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) Xs[(k + j) * PX + n] = rx[4 * i + j];
      } else {
        const int k = e / (NT / 4), n = (e % (NT / 4)) * 4;
        *(f32x4*)(Xs + k * PX + n) = f32x4{rx[4 * i], rx[4 * i + 1], rx[4 * i + 2], rx[4 * i + 3]};
      }
    }
  };
  if (nunits > 0) fetch(0);
  for (int u = 0; u < nunits; ++u) {
    const int b = b0 + u / nchunk, k0 = (u % nchunk) * MG_KC;
    __syncthreads();                                   // the previous unit's products are done with the tiles
    stage();
    __syncthreads();
    if (d.ln_mode) {
      // InstanceNorm of the rows over the n columns (all of them are in the tile: Nn <= 128), biased variance from centred
      // values; the first workgroup column keeps the normalised rows and their rstd for the backward pass.  A row = one quad:
      // thread t owns the columns (t & 3) + 4 i of row t >> 2 (conflict-free LDS walks), two quad butterflies per pass
      constexpr int CPT = NT / 4;
      const int k = tid >> 2, gk = k0 + k, c0 = tid & 3;
      float v[CPT];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        v[i] = c0 + 4 * i < d.Nn ? Xs[k * PX + c0 + 4 * i] : 0.f;
        sum += v[i];
      }
      const float mean = mg_quad_sum(sum) * inv_n;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        v[i] = c0 + 4 * i < d.Nn ? v[i] - mean : 0.f;
        sq += v[i] * v[i];
      }
      const float rstd = 1.f / sqrtf(mg_quad_sum(sq) * inv_n + d.eps);
      const bool keep = blockIdx.x == 0 && gk < d.K;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        Xs[k * PX + c0 + 4 * i] = v[i] * rstd;
        if (keep && c0 + 4 * i < d.Nn) gXn[(int64_t)b * d.x_batch + (int64_t)gk * d.ldx + c0 + 4 * i] = v[i] * rstd;
      }
      if (keep && c0 == 0) g_rstd_out[(int64_t)b * d.K + gk] = rstd;
      __syncthreads();
    }
    if (u + 1 < nunits) fetch(u + 1);                  // in flight while this unit is multiplied out
    // ---- this wave's 16 k-steps of the chunk: two (NC + 1) 16-byte reads, 16 NC FMAs each ----------------------------------------
#pragma unroll
    for (int i = 0; i < MG_KC / 4; ++i) {
      const int k = (MG_KC / 4) * wave + i;
      const f32x4 a4 = *(const f32x4*)(At + k * MG_PA + 4 * ty);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int h = 0; h < NC; ++h) {
        const f32x4 x4 = *(const f32x4*)(Xs + k * PX + 4 * tx + 64 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j][h].x = fmaf(a[j], x4.x, acc[j][h].x);
          acc[j][h].y = fmaf(a[j], x4.y, acc[j][h].y);
          acc[j][h].z = fmaf(a[j], x4.z, acc[j][h].z);
          acc[j][h].w = fmaf(a[j], x4.w, acc[j][h].w);
        }
      }
    }
  }
  // ---- the four waves' partial tiles meet in LDS: P[wave][row][col] ---------------------------------------------------------------
  __syncthreads();
  float* const P = Xs;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int h = 0; h < NC; ++h) *(f32x4*)(P + (wave * MG_OT + 4 * ty + j) * NT + 4 * tx + 64 * h) = acc[j][h];
  __syncthreads();
  // ---- epilogue: wave w finishes rows 4 w .. 4 w + 3, lane l the columns l (+ 64) -------------------------------------------------
  const int bo = d.reduce_batch ? 0 : (int)blockIdx.z;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 4 * wave + j, go = o0 + row;
    if (go >= d.O) continue;                         // (wave-uniform)
    float v[NC];
#pragma unroll
    for (int h = 0; h < NC; ++h) {
      const float* q = P + row * NT + lane + 64 * h;
      v[h] = ((q[0] + q[MG_OT * NT]) + q[2 * MG_OT * NT]) + q[3 * MG_OT * NT];     // wave order
    }
    if (gXH) {
      // InstanceNorm backward over the row: rstd (g - mean(g) - xh mean(g xh)); the whole row lives in this wave
      float xh[NC];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int h = 0; h < NC; ++h) {
        const bool in = lane + 64 * h < d.Nn;
        xh[h] = in ? gXH[(int64_t)bo * d.o_batch + (int64_t)go * d.ldo + lane + 64 * h] : 0.f;
        s1 += in ? v[h] : 0.f;
        s2 += in ? v[h] * xh[h] : 0.f;
      }
      const float m1 = mg_wave_sum(s1) * inv_n, m2 = mg_wave_sum(s2) * inv_n;
      const float rs = g_rstd_in[(int64_t)bo * d.O + go];
#pragma unroll
      for (int h = 0; h < NC; ++h) v[h] = rs * (v[h] - m1 - xh[h] * m2);
    }
#pragma unroll
    for (int h = 0; h < NC; ++h) {
      const int gn = n0 + lane + 64 * h;
      if (gn >= d.Nn) continue;
      float r = v[h];
      if (gR) r += gR[(int64_t)bo * d.r_batch + (int64_t)go * d.ldr + gn];
      if (!d.o_t) gOUT[(int64_t)bo * d.o_batch + (int64_t)go * d.ldo + gn] = r;
      else if (gOUT2 && go >= d.o_split) gOUT2[(int64_t)bo * d.o_batch + (int64_t)gn * d.ldo + go - d.o_split] = r;
      else gOUT[(int64_t)bo * d.o_batch + (int64_t)gn * d.ldo + go] = r;
    }
  }
}

}  // namespace cbim

using namespace cbim;

static bool mg_al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <int NC, bool VEC>
static void mg_launch(const cbim_map_gemm_desc* d, const MgDims& m, dim3 grid, hipStream_t st) {
  CBIM_LAUNCH((k_map_gemm<NC, VEC>), grid, dim3(MG_T), 0, st, d->A, d->A2, d->X, d->X2, d->OUT, d->OUT2, d->R, d->Xn, d->rstd_out, d->XH,
              d->rstd_in, m);
}

extern "C" int cbim_map_gemm(const cbim_map_gemm_desc* d, void* stream) {
  CBIM_CHECK(d && d->A && d->X && d->OUT, CBIM_EINVAL, "map gemm: null operand");
  CBIM_CHECK(d->O > 0 && d->K > 0 && d->Nn > 0 && d->batch > 0, CBIM_EINVAL, "map gemm: empty problem %d x %d x %d, batch %d", d->O, d->K,
             d->Nn, d->batch);
  CBIM_CHECK(!d->ln_mode || (!d->x_t && d->Nn <= 128 && d->Xn && d->rstd_out && !d->reduce_batch), CBIM_EUNSUPPORTED,
             "map gemm: normalise-on-load needs X as [k][n] rows of at most 128 columns");
  CBIM_CHECK(!d->XH || (!d->o_t && d->Nn <= 128 && d->rstd_in && !d->reduce_batch), CBIM_EUNSUPPORTED,
             "map gemm: the InstanceNorm-backward epilogue needs OUT as [o][n] rows of at most 128 columns");
  CBIM_CHECK(!d->R || !d->o_t, CBIM_EUNSUPPORTED, "map gemm: a residual needs OUT as [o][n]");
  CBIM_CHECK(!d->A2 || d->a_t, CBIM_EUNSUPPORTED, "map gemm: a second A tensor needs A as [k][o]");
  CBIM_CHECK(!d->X2 || d->x_t, CBIM_EUNSUPPORTED, "map gemm: a second X tensor needs X as [n][k]");
  CBIM_CHECK(!d->OUT2 || d->o_t, CBIM_EUNSUPPORTED, "map gemm: a second OUT tensor needs OUT as [n][o]");
  CBIM_CHECK((!d->A2 || d->a_split % 4 == 0) && (!d->X2 || d->x_split % 4 == 0), CBIM_EUNSUPPORTED,
             "map gemm: operand splits (%d, %d) must be multiples of 4", d->a_split, d->x_split);
  MgDims m;
  m.lda = d->lda; m.a_batch = d->a_batch; m.ldx = d->ldx; m.x_batch = d->x_batch; m.ldo = d->ldo; m.o_batch = d->o_batch;
  m.ldr = d->ldr; m.r_batch = d->r_batch; m.a_t = d->a_t; m.a_split = d->a_split; m.x_t = d->x_t; m.x_split = d->x_split;
  m.o_t = d->o_t; m.o_split = d->o_split; m.O = d->O; m.K = d->K; m.Nn = d->Nn; m.batch = d->batch; m.reduce_batch = d->reduce_batch;
  m.ln_mode = d->ln_mode; m.eps = d->eps;
  // 64-column tiles when the rows fit (or the normalisation steps do not need a whole row in the tile: more workgroups);
  // 128 otherwise
  const bool whole_row = d->ln_mode || d->XH;
  const int nc = (d->Nn <= 64 || !whole_row) ? 1 : 2;
  const int NT = 64 * nc;
  // 16-byte loads: every group of 4 an operand row is read in is whole and aligned
  const bool a_vec = d->lda % 4 == 0 && d->a_batch % 4 == 0 && mg_al16(d->A) && (!d->A2 || mg_al16(d->A2)) &&
                     (d->a_t ? d->O % 4 == 0 : d->K % 4 == 0);
  const bool x_vec = d->ldx % 4 == 0 && d->x_batch % 4 == 0 && mg_al16(d->X) && (!d->X2 || mg_al16(d->X2)) &&
                     (d->x_t ? d->K % 4 == 0 : d->Nn % 4 == 0);
  const bool vec = a_vec && x_vec;
  dim3 grid((unsigned)((d->O + MG_OT - 1) / MG_OT), (unsigned)((d->Nn + NT - 1) / NT), (unsigned)(d->reduce_batch ? 1 : d->batch));
  hipStream_t st = (hipStream_t)stream;
  if (nc == 1) { if (vec) mg_launch<1, true>(d, m, grid, st); else mg_launch<1, false>(d, m, grid, st); }
  else { if (vec) mg_launch<2, true>(d, m, grid, st); else mg_launch<2, false>(d, m, grid, st); }
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "map gemm launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}


// ---- SEBlock excitation on the [N, C] channel means (round 6) ------------------------------------------------------------------
//   gate = sigmoid(W2 relu(W1 m + b1) + b2),  W1 [H, C], W2 [C, H]        (/root/reference/model/dim3/conv_layers.py:159-175: squeeze ->
//   1x1x1 conv -> ReLU -> 1x1x1 conv -> Sigmoid; one per MBConv of MedFormer's 16 BidirectionAttentionBlocks)
// As torch ops this was two aten::linear, relu, sigmoid forward and four aten::mm, sigmoid_backward, threshold_backward and their
// glue backward: ~13 launches per block on vectors of 64 ... 1280 numbers.  Here: two launches forward (the second applies ReLU to
// its input on load and the sigmoid to its output), three backward (W2^T of the sigmoid-weighted gradient with the ReLU mask in the
// epilogue; both outer products and bias gradients; W1^T for the gradient of the means).  Fixed summation order.
namespace cbim {

// y[n][o] = epi(sum_k W[o][k] * pro(x[n][k]) + b[o]); one wave per output row, lanes stride the row (coalesced)
template <bool RELU_IN, bool SIGMOID_OUT>
__global__ void __launch_bounds__(256) k_se_rows(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ x,
                                                 float* __restrict__ y, int N, int O, int K) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  const float* w = W + (size_t)o * K;
  for (int n = 0; n < N; ++n) {
    float acc = 0.f;
    if ((K & 3) == 0) {          // rows of whole 16-byte groups (the weights are [O][K] contiguous: every row then starts aligned)
      const float* xr = x + (size_t)n * K;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 wv = *(const f32x4*)(w + k);
        f32x4 xv = *(const f32x4*)(xr + k);
        if (RELU_IN) { xv.x = xv.x > 0.f ? xv.x : 0.f; xv.y = xv.y > 0.f ? xv.y : 0.f; xv.z = xv.z > 0.f ? xv.z : 0.f; xv.w = xv.w > 0.f ? xv.w : 0.f; }
        a0 = fmaf(wv.x, xv.x, a0); a1 = fmaf(wv.y, xv.y, a1); a2 = fmaf(wv.z, xv.z, a2); a3 = fmaf(wv.w, xv.w, a3);
      }
      acc = (a0 + a1) + (a2 + a3);
    } else {
      for (int k = lane; k < K; k += 64) {
        float v = x[(size_t)n * K + k];
        if (RELU_IN) v = v > 0.f ? v : 0.f;
        acc = fmaf(w[k], v, acc);
      }
    }
    acc = mg_wave_sum(acc);
    if (lane == 0) {
      float r = acc + (b ? b[o] : 0.f);
      if (SIGMOID_OUT) r = 1.f / (1.f + expf(-r));
      y[(size_t)n * O + o] = r;
    }
  }
}

// the input vector of the transposed products and of the outer products: MODE 0: v[o] = u[o]; MODE 1: v[o] = dg[o] g[o] (1 - g[o])
template <int MODE>
__device__ __forceinline__ float se_vec(const float* __restrict__ u, const float* __restrict__ g, size_t i) {
  if (MODE == 0) return u[i];
  const float gg = g[i];
  return u[i] * gg * (1.f - gg);
}

// y[n][k] = mask(sum_o W[o][k] * v[n][o]): 16 columns per workgroup (a quarter wave reads 64 contiguous bytes of a row), 64 row
// groups stride the rows and meet in LDS in group order; RELU_MASK: y *= [z[n][k] > 0].  (64 columns per workgroup left the
// 1280 x 320 product of the widest MBConv on five workgroups: 33 us.)
template <int MODE, bool RELU_MASK>
__global__ void __launch_bounds__(1024) k_se_cols(const float* __restrict__ W, const float* __restrict__ u, const float* __restrict__ g,
                                                  const float* __restrict__ z, float* __restrict__ y, int O, int K) {
  __shared__ float red[64][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4, k = blockIdx.x * 16 + c, n = blockIdx.y;
  float acc = 0.f;
  if (k < K)
    for (int o = rg; o < O; o += 64) acc = fmaf(W[(size_t)o * K + k], se_vec<MODE>(u, g, (size_t)n * O + o), acc);
  red[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && k < K) {
    float r = 0.f;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) r += red[q][c];
    if (RELU_MASK) r = z[(size_t)n * K + k] > 0.f ? r : 0.f;
    y[(size_t)n * K + k] = r;
  }
}

// dW[o][k] = sum_n v[n][o] * pro(x[n][k]), db[o] = sum_n v[n][o]; blockIdx.y = 0: (W2: v from (dg, g), x = relu(z1)),
// 1: (W1: v = dz1, x = m) — both outer products of the backward in one launch
__global__ void __launch_bounds__(256) k_se_outer(const float* __restrict__ dg, const float* __restrict__ g, const float* __restrict__ z1,
                                                  const float* __restrict__ dz1, const float* __restrict__ m, float* __restrict__ dW2,
                                                  float* __restrict__ db2, float* __restrict__ dW1, float* __restrict__ db1, int N, int C,
                                                  int H) {
  const bool second = blockIdx.y == 1;
  const int O = second ? H : C, K = second ? C : H;
  const int o = blockIdx.x;
  if (o >= O) return;
  float* dW = second ? dW1 : dW2;
  float* db = second ? db1 : db2;
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
      const float v = second ? dz1[(size_t)n * H + o] : se_vec<1>(dg, g, (size_t)n * C + o);
      float xv = second ? m[(size_t)n * C + k] : z1[(size_t)n * H + k];
      if (!second) xv = xv > 0.f ? xv : 0.f;
      acc = fmaf(v, xv, acc);
    }
    dW[(size_t)o * K + k] = acc;
  }
  if (threadIdx.x == 0 && db) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += second ? dz1[(size_t)n * H + o] : se_vec<1>(dg, g, (size_t)n * C + o);
    db[o] = acc;
  }
}

}  // namespace cbim

extern "C" int cbim_se_gate_fwd(const float* mean, const float* W1, const float* b1, const float* W2, const float* b2, float* z1,
                                float* gate, int N, int C, int H, void* stream) {
  CBIM_CHECK(mean && W1 && W2 && z1 && gate && N > 0 && C > 0 && H > 0, CBIM_EINVAL, "se gate: bad argument");
  hipStream_t st = (hipStream_t)stream;
  CBIM_LAUNCH((k_se_rows<false, false>), dim3((unsigned)((H + 3) / 4)), dim3(256), 0, st, W1, b1, mean, z1, N, H, C);
  CBIM_LAUNCH((k_se_rows<true, true>), dim3((unsigned)((C + 3) / 4)), dim3(256), 0, st, W2, b2, (const float*)z1, gate, N, C, H);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "se gate fwd launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

extern "C" int cbim_se_gate_bwd(const float* dgate, const float* gate, const float* z1, const float* mean, const float* W1,
                                const float* W2, float* dz1_ws, float* dW1, float* db1, float* dW2, float* db2, float* dmean, int N,
                                int C, int H, void* stream) {
  CBIM_CHECK(dgate && gate && z1 && mean && W1 && W2 && dz1_ws && dW1 && dW2 && N > 0 && C > 0 && H > 0, CBIM_EINVAL, "se gate bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // dz1 = (W2^T (dgate g (1 - g))) * [z1 > 0]
  CBIM_LAUNCH((k_se_cols<1, true>), dim3((unsigned)((H + 15) / 16), (unsigned)N), dim3(1024), 0, st, W2, dgate, gate, z1, dz1_ws, C, H);
  CBIM_LAUNCH(k_se_outer, dim3((unsigned)(C > H ? C : H), 2), dim3(256), 0, st, dgate, gate, z1, (const float*)dz1_ws, mean, dW2, db2, dW1, db1, N,
              C, H);
  if (dmean)
    CBIM_LAUNCH((k_se_cols<0, false>), dim3((unsigned)((C + 15) / 16), (unsigned)N), dim3(1024), 0, st, W1, (const float*)dz1_ws, (const float*)nullptr,
                (const float*)nullptr, dmean, H, C);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "se gate bwd launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(map)
