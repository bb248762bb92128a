// attn_wide.hip — BidirectionAttention core and SemanticMapGeneration backward for the head / map sizes the
// register-resident kernels of medformer_kernels.hip (d_head 8|16|32, <= 64 codes) and attn_mfma.hip
// (d_head 32, 64 codes) do not take: any d_head, up to 128 map codes.  Shipped configurations that need it:
// config/acdc/medformer_3d.yaml (map_size [2,6,6] = 72 codes, d_head 64 and 80) and config/lits/medformer_3d.yaml
// (num_heads all 1 -> d_head 64 ... 320).  Same math, same layouts and the same per-block record formats as
// k_attn_fwd / k_attn_bwd (/root/reference/model/dim3/medformer_utils.py:63-97), so k_attn_bwd_reduce and the
// Python side are shared.
//
// One wave (64 voxels) per workgroup, thread = voxel.  The [voxel][code] logit matrix lives in LDS (each thread owns
// a row, odd pitch >= M: conflict-free both for "own row" walks and for column walks), the head dimension is
// streamed in chunks of 32: the map-side chunk [code][32] is staged once per workgroup and read as 16-byte
// broadcasts, the voxel-side chunk sits in registers.  Products that sum over the voxels of the block (column-softmax
// records, dq_m / dv_m partials) stage the voxel-side chunk in LDS and give each thread one or two codes.
// Every sum has a fixed order: results are bit-reproducible.
#include "cbim_common.h"

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

namespace cbim {

static constexpr int WT = 64;        // voxels (= threads) per workgroup
static constexpr int MWC = 128;      // max map codes
static constexpr int DC = 32;        // head-dimension chunk
// pitch of the [voxel][code] matrices: the first odd number >= M (conflict-free own-row and column walks).  LDS is
// sized per launch from M, so 64 codes leave room for three workgroups per CU (one was resident with a fixed 129).
__host__ __device__ __forceinline__ int code_pitch(int M) { return M | 1; }
static constexpr int XQ = DC + 4;    // pitch of the staged voxel-side chunk (16-byte aligned rows)

// map-side chunk: Wc[j][k] = src[j*inner + (d0+k)*heads] for j < M, k < DC (zero past dh)
__device__ __forceinline__ void stage_codes(float* Wc, const float* __restrict__ src, int inner, int heads, int dh,
                                            int M, int d0, int t) {
  for (int i = t; i < M * DC; i += WT) {
    const int j = i / DC, k = i % DC;
    Wc[i] = d0 + k < dh ? src[(size_t)j * inner + (size_t)(d0 + k) * heads] : 0.f;
  }
}

// voxel-side chunk of this thread's row into registers
template <typename T>
__device__ __forceinline__ void load_row_chunk(float* x, const void* __restrict__ p, size_t base, int heads, int dh,
                                               int d0, bool valid) {
#pragma unroll
  for (int k = 0; k < DC; ++k) x[k] = (valid && d0 + k < dh) ? Elem<T>::load1(p, base + (size_t)(d0 + k) * heads) : 0.f;
}

// The library is built with -ffp-contract=off (bit-identical results between builds), so a*b+c written with
// operators costs a multiply AND an add, and `s += ...` over a chunk is one 64-deep dependent chain; the helpers below
// use fmaf explicitly and keep U independent sums (U codes at a time) so that the LDS reads of U rows are in flight
// together.
template <int U>
__device__ __forceinline__ void dot_rows(const float* x, const float* w, float* s) {   // s[u] = <x, w[u][:]>
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < DC / 4; ++k4) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4 v = ((const f32x4*)(w + u * DC))[k4];
      s[u] = fmaf(x[4 * k4], v.x, s[u]);
      s[u] = fmaf(x[4 * k4 + 1], v.y, s[u]);
      s[u] = fmaf(x[4 * k4 + 2], v.z, s[u]);
      s[u] = fmaf(x[4 * k4 + 3], v.w, s[u]);
    }
  }
}

__device__ __forceinline__ float dot_chunk(const float* x, const float* w) {
  float s;
  dot_rows<1>(x, w, &s);
  return s;
}

__device__ __forceinline__ void axpy_chunk(float* o, float e, const float* w) {
  const f32x4* w4 = (const f32x4*)w;
#pragma unroll
  for (int k4 = 0; k4 < DC / 4; ++k4) {
    const f32x4 v = w4[k4];
    o[4 * k4] = fmaf(e, v.x, o[4 * k4]);
    o[4 * k4 + 1] = fmaf(e, v.y, o[4 * k4 + 1]);
    o[4 * k4 + 2] = fmaf(e, v.z, o[4 * k4 + 2]);
    o[4 * k4 + 3] = fmaf(e, v.w, o[4 * k4 + 3]);
  }
}

static constexpr int JU = 4;   // codes per trip of the per-voxel loops

// logits of this block: Et[j] = scale * sum_d q[d] mq[j][d]  (own row of E)
template <typename T>
__device__ __forceinline__ void block_logits(float* Et, float* Wc, const void* __restrict__ qv, size_t qbase,
                                             const float* __restrict__ mqh, int inner, int heads, int dh, int M,
                                             float scale, bool valid, int t) {
  for (int d0 = 0; d0 < dh; d0 += DC) {
    __syncthreads();
    stage_codes(Wc, mqh, inner, heads, dh, M, d0, t);
    float q[DC];
    load_row_chunk<T>(q, qv, qbase, heads, dh, d0, valid);
    __syncthreads();
    int j = 0;
    for (; j + JU <= M; j += JU) {
      float s[JU];
      dot_rows<JU>(q, Wc + j * DC, s);
#pragma unroll
      for (int u = 0; u < JU; ++u) Et[j + u] = d0 == 0 ? s[u] : Et[j + u] + s[u];
    }
    for (; j < M; ++j) {
      const float s = dot_chunk(q, Wc + j * DC);
      Et[j] = d0 == 0 ? s : Et[j] + s;
    }
  }
  for (int j = 0; j < M; ++j) Et[j] *= scale;
}

// partial[j][d0..] = sum over the block's voxels r of Mx[r][j] * X[r][k] for this thread's codes j = t, t + 64
__device__ __forceinline__ void code_times_voxels(const float* Mx, int EP, const float* X, float* dst, int64_t dst_pitch,
                                                  int dh, int M, int d0, int t, float mul) {
  for (int j = t; j < M; j += WT) {
    float acc[DC];
#pragma unroll
    for (int k = 0; k < DC; ++k) acc[k] = 0.f;
    for (int r = 0; r < WT; ++r) axpy_chunk(acc, Mx[r * EP + j], X + r * XQ);
#pragma unroll
    for (int k = 0; k < DC; ++k)
      if (d0 + k < dh) dst[(size_t)j * dst_pitch + d0 + k] = acc[k] * mul;
  }
}

// part record per (n, h, block, j): [colmax, colsum, acc[dh]]  (the format k_attn_fwd writes)
template <typename T>
__global__ void __launch_bounds__(WT) k_attnw_fwd(const void* __restrict__ qv, int64_t rs, const float* __restrict__ mq,
                                                  const float* __restrict__ mv, void* __restrict__ fo,
                                                  float* __restrict__ part, int L, int heads, int dh, int M, float scale,
                                                  int nblk) {
  CBIM_DYN_SMEM(smem);
  const int EP = code_pitch(M);
  float* E = (float*)smem;       // [WT][EP]
  float* X = E + WT * EP;        // [WT][XQ]
  float* Wc = X + WT * XQ;       // [M][DC]
  float* colm = Wc + M * DC;     // [M]
  const int t = threadIdx.x, blk = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int inner = heads * dh;
  const int l = blk * WT + t;
  const bool valid = l < L;
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs;
  const float* mqh = mq + (size_t)n * M * inner + h;
  const float* mvh = mv + (size_t)n * M * inner + h;
  float* Et = E + t * EP;
  block_logits<T>(Et, Wc, qv, row + h, mqh, inner, heads, dh, M, scale, valid, t);
  // feature side: softmax over the M codes of this voxel, times the map values
  float mx = -INFINITY;
  for (int j = 0; j < M; ++j) mx = fmaxf(mx, Et[j]);
  float sum = 0.f;
  for (int j = 0; j < M; ++j) sum += expf(Et[j] - mx);
  const float inv = 1.f / sum;
  const size_t orow = ((size_t)n * L + (valid ? l : 0)) * inner + h;
  for (int d0 = 0; d0 < dh; d0 += DC) {
    __syncthreads();
    stage_codes(Wc, mvh, inner, heads, dh, M, d0, t);
    __syncthreads();
    float o[DC];
#pragma unroll
    for (int k = 0; k < DC; ++k) o[k] = 0.f;
    for (int j = 0; j < M; ++j) axpy_chunk(o, expf(Et[j] - mx), Wc + j * DC);
    if (valid) {
#pragma unroll
      for (int k = 0; k < DC; ++k)
        if (d0 + k < dh) Elem<T>::store1(fo, orow + (size_t)(d0 + k) * heads, o[k] * inv);
    }
  }
  // map side: online-softmax record of this block's voxels per code
  if (!valid)
    for (int j = 0; j < M; ++j) Et[j] = -INFINITY;
  __syncthreads();
  for (int j = t; j < M; j += WT) {
    float m = -INFINITY;
    for (int r = 0; r < WT; ++r) m = fmaxf(m, E[r * EP + j]);
    colm[j] = m;
  }
  __syncthreads();
  for (int j = 0; j < M; ++j) Et[j] = valid ? expf(Et[j] - colm[j]) : 0.f;
  __syncthreads();
  float* pb = part + (((size_t)n * heads + h) * nblk + blk) * M * (dh + 2);
  for (int j = t; j < M; j += WT) {
    float s = 0.f;
    for (int r = 0; r < WT; ++r) s += E[r * EP + j];
    pb[(size_t)j * (dh + 2)] = colm[j];
    pb[(size_t)j * (dh + 2) + 1] = s;
  }
  for (int d0 = 0; d0 < dh; d0 += DC) {
    __syncthreads();
    float v[DC];
    load_row_chunk<T>(v, qv, row + inner + h, heads, dh, d0, valid);
#pragma unroll
    for (int k = 0; k < DC; ++k) X[t * XQ + k] = v[k];
    __syncthreads();
    code_times_voxels(E, EP, X, pb + 2, dh + 2, dh, M, d0, t, 1.f);
  }
}

// merge the per-block records: map_out[n][j][d*heads+h], colstat[n][h][j] = (max, sum); one wave per (n, h, j),
// records walked in launch order
__global__ void __launch_bounds__(WT) k_attnw_merge(const float* __restrict__ part, float* __restrict__ map_out,
                                                    float* __restrict__ colstat, int heads, int M, int dh, int nblk) {
  const int j = blockIdx.x % M, h = (blockIdx.x / M) % heads, n = blockIdx.x / (M * heads);
  const int inner = heads * dh, t = threadIdx.x;
  const size_t rec = (size_t)dh + 2;
  const float* base = part + (((size_t)n * heads + h) * nblk * M + j) * rec;
  float mx = -INFINITY;
  for (int b = 0; b < nblk; ++b) mx = fmaxf(mx, base[(size_t)b * M * rec]);
  float S = 0.f;
  for (int b = 0; b < nblk; ++b) S += base[(size_t)b * M * rec + 1] * expf(base[(size_t)b * M * rec] - mx);
  for (int d = t; d < dh; d += WT) {
    float A = 0.f;
    for (int b = 0; b < nblk; ++b) {
      const float* p = base + (size_t)b * M * rec;
      A += p[2 + d] * expf(p[0] - mx);
    }
    map_out[((size_t)n * M + j) * inner + (size_t)d * heads + h] = A / S;
  }
  if (t == 0) {
    colstat[(((size_t)n * heads + h) * M + j) * 2] = mx;
    colstat[(((size_t)n * heads + h) * M + j) * 2 + 1] = S;
  }
}

// backward: part record per (n, h, block): [2][M][dh] = (dmv partial, dmq partial)  (k_attn_bwd's format)
template <typename T>
__global__ void __launch_bounds__(WT) k_attnw_bwd(const void* __restrict__ qv, int64_t rs, const float* __restrict__ mq,
                                                  const float* __restrict__ mv, const float* __restrict__ colstat,
                                                  const float* __restrict__ map_out, const void* __restrict__ dfo,
                                                  const float* __restrict__ dmo, void* __restrict__ dqv,
                                                  float* __restrict__ part, int L, int heads, int dh, int M, float scale,
                                                  int nblk) {
  CBIM_DYN_SMEM(smem);
  const int EP = code_pitch(M);
  float* A = (float*)smem;      // [WT][EP] logits, later P1
  float* B = A + WT * EP;       // [WT][EP] g.mv_j, later dA
  float* X = B + WT * EP;       // [WT][XQ]
  float* Wc = X + WT * XQ;      // [M][DC]
  float* cj = Wc + M * DC;      // <map_out_j, dmap_out_j>
  float* cM = cj + M;           // column max
  float* cIS = cM + M;          // 1 / column sum
  const int t = threadIdx.x, blk = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int inner = heads * dh;
  const int l = blk * WT + t;
  const bool valid = l < L;
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs;
  const size_t grow = ((size_t)n * L + (valid ? l : 0)) * inner + h;
  const size_t drow = ((size_t)n * L + (valid ? l : 0)) * (2 * (size_t)inner) + h;
  const float* mqh = mq + (size_t)n * M * inner + h;
  const float* mvh = mv + (size_t)n * M * inner + h;
  const float* dmh = dmo + (size_t)n * M * inner + h;
  const float* moh = map_out + (size_t)n * M * inner + h;
  for (int j = t; j < M; j += WT) {
    float c = 0.f;
    for (int d = 0; d < dh; ++d) c += moh[(size_t)j * inner + (size_t)d * heads] * dmh[(size_t)j * inner + (size_t)d * heads];
    cj[j] = c;
    cM[j] = colstat[(((size_t)n * heads + h) * M + j) * 2];
    cIS[j] = 1.f / colstat[(((size_t)n * heads + h) * M + j) * 2 + 1];
  }
  float* At = A + t * EP;
  float* Bt = B + t * EP;
  block_logits<T>(At, Wc, qv, row + h, mqh, inner, heads, dh, M, scale, valid, t);   // cj/cM/cIS are published by its barriers
  float mx = -INFINITY;
  for (int j = 0; j < M; ++j) mx = fmaxf(mx, At[j]);
  float sum = 0.f;
  for (int j = 0; j < M; ++j) sum += expf(At[j] - mx);
  const float inv = 1.f / sum;
  // feature side: B = g.mv_j, then dA1 = P1 (g.mv_j - sum_j' P1 g.mv_j')
  for (int d0 = 0; d0 < dh; d0 += DC) {
    __syncthreads();
    stage_codes(Wc, mvh, inner, heads, dh, M, d0, t);
    float g[DC];
    load_row_chunk<T>(g, dfo, grow, heads, dh, d0, valid);
    __syncthreads();
    int j = 0;
    for (; j + JU <= M; j += JU) {
      float s[JU];
      dot_rows<JU>(g, Wc + j * DC, s);
#pragma unroll
      for (int u = 0; u < JU; ++u) Bt[j + u] = d0 == 0 ? s[u] : Bt[j + u] + s[u];
    }
    for (; j < M; ++j) {
      const float s = dot_chunk(g, Wc + j * DC);
      Bt[j] = d0 == 0 ? s : Bt[j] + s;
    }
  }
  float rr = 0.f;
  for (int j = 0; j < M; ++j) rr += expf(At[j] - mx) * inv * Bt[j];
  for (int j = 0; j < M; ++j) Bt[j] = expf(At[j] - mx) * inv * (Bt[j] - rr);
  // map side: P2 = exp(a - colmax)/colsum; dA2 = P2 (v.dmo_j - <map_out_j, dmo_j>); dv = sum_j P2 dmo_j
  for (int d0 = 0; d0 < dh; d0 += DC) {
    __syncthreads();
    stage_codes(Wc, dmh, inner, heads, dh, M, d0, t);
    float v[DC], dv[DC];
    load_row_chunk<T>(v, qv, row + inner + h, heads, dh, d0, valid);
#pragma unroll
    for (int k = 0; k < DC; ++k) dv[k] = 0.f;
    __syncthreads();
    int j = 0;
    for (; j + JU <= M; j += JU) {
      float s[JU], p2[JU];
      dot_rows<JU>(v, Wc + j * DC, s);
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        p2[u] = expf(At[j + u] - cM[j + u]) * cIS[j + u];
        Bt[j + u] += p2[u] * (d0 == 0 ? s[u] - cj[j + u] : s[u]);
      }
#pragma unroll
      for (int u = 0; u < JU; ++u) axpy_chunk(dv, p2[u], Wc + (j + u) * DC);
    }
    for (; j < M; ++j) {
      const float p2 = expf(At[j] - cM[j]) * cIS[j];
      const float s = dot_chunk(v, Wc + j * DC);
      Bt[j] += p2 * (d0 == 0 ? s - cj[j] : s);
      axpy_chunk(dv, p2, Wc + j * DC);
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < DC; ++k)
        if (d0 + k < dh) Elem<T>::store1(dqv, drow + inner + (size_t)(d0 + k) * heads, dv[k]);
    }
  }
  // dq = scale * sum_j dA_j mq_j
  for (int d0 = 0; d0 < dh; d0 += DC) {
    __syncthreads();
    stage_codes(Wc, mqh, inner, heads, dh, M, d0, t);
    float dq[DC];
#pragma unroll
    for (int k = 0; k < DC; ++k) dq[k] = 0.f;
    __syncthreads();
    for (int j = 0; j < M; ++j) axpy_chunk(dq, Bt[j], Wc + j * DC);
    if (valid) {
#pragma unroll
      for (int k = 0; k < DC; ++k)
        if (d0 + k < dh) Elem<T>::store1(dqv, drow + (size_t)(d0 + k) * heads, dq[k] * scale);
    }
  }
  // map-side gradients: dmv partial = sum_vox P1 g, dmq partial = scale * sum_vox dA q  (rows past L contribute 0)
  for (int j = 0; j < M; ++j) {
    At[j] = valid ? expf(At[j] - mx) * inv : 0.f;
    if (!valid) Bt[j] = 0.f;
  }
  float* pbase = part + (((size_t)n * heads + h) * nblk + blk) * 2 * (size_t)M * dh;
  for (int stage = 0; stage < 2; ++stage)
    for (int d0 = 0; d0 < dh; d0 += DC) {
      __syncthreads();
      float x[DC];
      if (stage == 0) load_row_chunk<T>(x, dfo, grow, heads, dh, d0, valid);
      else load_row_chunk<T>(x, qv, row + h, heads, dh, d0, valid);
#pragma unroll
      for (int k = 0; k < DC; ++k) X[t * XQ + k] = x[k];
      __syncthreads();
      code_times_voxels(stage == 0 ? A : B, EP, X, pbase + (size_t)stage * M * dh, dh, dh, M, d0, t, stage == 0 ? 1.f : scale);
    }
}

// SemanticMapGeneration backward (medformer_utils.py:218-228) for more than 64 codes — same contract as k_mappool_bwd:
// dfeat[l,c] = sum_j P[l,j] dmap[c,j];  dlogit[l,j] = P[l,j] (sum_c feat[l,c] dmap[c,j] - <map_j, dmap_j>)
template <typename T>
__global__ void __launch_bounds__(WT) k_mappoolw_bwd(const void* __restrict__ fw, int64_t rs, const float* __restrict__ map,
                                                     const float* __restrict__ colstat, const float* __restrict__ dmap,
                                                     void* __restrict__ dfw, int64_t drs, int L, int C, int M) {
  CBIM_DYN_SMEM(smem);
  const int EP = code_pitch(M);
  float* P = (float*)smem;      // [WT][EP]
  float* TT = P + WT * EP;      // [WT][EP]
  float* Wc = TT + WT * EP;     // [M][DC]: dmap[c0+k][j]
  float* cj = Wc + M * DC;
  const int t = threadIdx.x, n = blockIdx.z;
  const int l = blockIdx.x * WT + t;
  const bool valid = l < L;
  for (int j = t; j < M; j += WT) {
    float c = 0.f;
    for (int k = 0; k < C; ++k) c += map[((size_t)n * C + k) * M + j] * dmap[((size_t)n * C + k) * M + j];
    cj[j] = c;
  }
  const size_t row = ((size_t)n * L + (valid ? l : 0)) * rs, drow = ((size_t)n * L + (valid ? l : 0)) * drs;
  float* Pt = P + t * EP;
  float* Tt = TT + t * EP;
  for (int j = 0; j < M; ++j) {
    Pt[j] = valid ? expf(Elem<T>::load1(fw, row + C + j) - colstat[((size_t)n * M + j) * 2]) / colstat[((size_t)n * M + j) * 2 + 1]
                  : 0.f;
    Tt[j] = 0.f;
  }
  for (int c0 = 0; c0 < C; c0 += DC) {
    __syncthreads();
    for (int i = t; i < M * DC; i += WT) {
      const int j = i / DC, k = i % DC;
      Wc[i] = c0 + k < C ? dmap[((size_t)n * C + c0 + k) * M + j] : 0.f;
    }
    float f[DC], g[DC];
#pragma unroll
    for (int k = 0; k < DC; ++k) {
      f[k] = (valid && c0 + k < C) ? Elem<T>::load1(fw, row + c0 + k) : 0.f;
      g[k] = 0.f;
    }
    __syncthreads();
    int j = 0;
    for (; j + JU <= M; j += JU) {
      float s[JU];
      dot_rows<JU>(f, Wc + j * DC, s);
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        Tt[j + u] += s[u];
        axpy_chunk(g, Pt[j + u], Wc + (j + u) * DC);
      }
    }
    for (; j < M; ++j) {
      Tt[j] += dot_chunk(f, Wc + j * DC);
      axpy_chunk(g, Pt[j], Wc + j * DC);
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < DC; ++k)
        if (c0 + k < C) Elem<T>::store1(dfw, drow + c0 + k, g[k]);
    }
  }
  __syncthreads();   // cj
  if (valid)
    for (int j = 0; j < M; ++j) Elem<T>::store1(dfw, drow + C + j, Pt[j] * (Tt[j] - cj[j]));
}

}  // namespace cbim

using namespace cbim;

static size_t fwd_smem(int M) { return (size_t)(WT * code_pitch(M) + WT * XQ + M * DC + M) * sizeof(float); }
static size_t bwd_smem(int M) { return (size_t)(2 * WT * code_pitch(M) + WT * XQ + M * DC + 3 * M) * sizeof(float); }
static size_t pool_smem(int M) { return (size_t)(2 * WT * code_pitch(M) + M * DC + M) * sizeof(float); }

// kernels that ask for more than 64 KiB of LDS need the attribute once per process
template <typename K>
static int allow_lds(K kernel, size_t bytes, const char* what) {
#ifndef CBIM_EMU
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
#endif
  return CBIM_OK;
}

static int wide_launch_ok(const char* what) {
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "%s launch: %s", what, hipGetErrorString(e));
  return CBIM_OK;
}

extern "C" int cbim_attn_wide_max_codes(void) { return MWC; }
extern "C" int cbim_attn_wide_records(int L) { return (L + WT - 1) / WT; }

// launched by cbim_bidir_attn_fwd (medformer_kernels.hip) when d_head is not 8|16|32 or there are more than 64 codes;
// writes feat_out, map_out and colstat
extern "C" int cbim_attn_fwd_wide_launch(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                         void* feat_out, float* map_out, float* colstat, float* part, int N, int L,
                                         int heads, int dh, int M, float scale, void* stream) {
  static bool attr_done = false;
  if (!attr_done) {
    if (int e = allow_lds(k_attnw_fwd<bf16_tag>, fwd_smem(MWC), "bidir_attn_fwd (wide)")) return e;
    if (int e = allow_lds(k_attnw_fwd<float>, fwd_smem(MWC), "bidir_attn_fwd (wide)")) return e;
    attr_done = true;
  }
  const int nblk = (L + WT - 1) / WT;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nblk, heads, N);
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_attnw_fwd<bf16_tag>), grid, dim3(WT), fwd_smem(M), st, qv, qv_stride, mq, mv, feat_out, part, L, heads, dh, M,
                scale, nblk);
  else
    CBIM_LAUNCH((k_attnw_fwd<float>), grid, dim3(WT), fwd_smem(M), st, qv, qv_stride, mq, mv, feat_out, part, L, heads, dh, M,
                scale, nblk);
  if (int e = wide_launch_ok("bidir_attn_fwd (wide)")) return e;
  CBIM_LAUNCH(k_attnw_merge, dim3(N * heads * M), dim3(WT), 0, st, (const float*)part, map_out, colstat, heads, M, dh, nblk);
  return wide_launch_ok("bidir_attn_merge (wide)");
}

// writes d_qv and the per-block (dmv, dmq) partials; the caller reduces them with k_attn_bwd_reduce over
// cbim_attn_wide_records(L) records
extern "C" int cbim_attn_bwd_wide_launch(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                         const float* colstat, const float* map_out, const void* d_feat_out,
                                         const float* d_map_out, void* d_qv, float* part, int N, int L, int heads, int dh,
                                         int M, float scale, void* stream) {
  static bool attr_done = false;
  if (!attr_done) {
    if (int e = allow_lds(k_attnw_bwd<bf16_tag>, bwd_smem(MWC), "bidir_attn_bwd (wide)")) return e;
    if (int e = allow_lds(k_attnw_bwd<float>, bwd_smem(MWC), "bidir_attn_bwd (wide)")) return e;
    attr_done = true;
  }
  const int nblk = (L + WT - 1) / WT;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nblk, heads, N);
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_attnw_bwd<bf16_tag>), grid, dim3(WT), bwd_smem(M), st, qv, qv_stride, mq, mv, colstat, map_out, d_feat_out,
                d_map_out, d_qv, part, L, heads, dh, M, scale, nblk);
  else
    CBIM_LAUNCH((k_attnw_bwd<float>), grid, dim3(WT), bwd_smem(M), st, qv, qv_stride, mq, mv, colstat, map_out, d_feat_out,
                d_map_out, d_qv, part, L, heads, dh, M, scale, nblk);
  return wide_launch_ok("bidir_attn_bwd (wide)");
}

extern "C" int cbim_mappool_bwd_wide_launch(int dtype, const void* fw, int64_t fw_stride, const float* map,
                                            const float* colstat, const float* dmap, void* dfw, int64_t dfw_stride, int N,
                                            int L, int C, int M, void* stream) {
  static bool attr_done = false;
  if (!attr_done) {
    if (int e = allow_lds(k_mappoolw_bwd<bf16_tag>, pool_smem(MWC), "colsoftmax_pool_bwd (wide)")) return e;
    if (int e = allow_lds(k_mappoolw_bwd<float>, pool_smem(MWC), "colsoftmax_pool_bwd (wide)")) return e;
    attr_done = true;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((L + WT - 1) / WT, 1, N);
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_mappoolw_bwd<bf16_tag>), grid, dim3(WT), pool_smem(M), st, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride,
                L, C, M);
  else
    CBIM_LAUNCH((k_mappoolw_bwd<float>), grid, dim3(WT), pool_smem(M), st, fw, fw_stride, map, colstat, dmap, dfw, dfw_stride, L,
                C, M);
  return wide_launch_ok("colsoftmax_pool_bwd (wide)");
}

CBIM_DEFINE_WARM(attn_wide)
