// attn_mfma.hip — MedFormer BidirectionAttention core on the matrix cores (bf16, d_head 32, 64 map codes).
//
// Same math and interface as k_attn_fwd / k_attn_bwd of medformer_kernels.hip (reference:
// /root/reference/model/dim3/medformer_utils.py:63-97); those remain the fp32 / generic path.  One WAVE owns 32
// voxels of one head.  Every product is a 32x32x16 bf16 MFMA, arranged so that NO cross-lane shuffle and NO LDS is
// needed between the GEMMs:
//   * Q is gathered once per lane as "voxel l&31, d = 8*(l>>5)+j": that register content is at the same time the
//     A operand (rows = voxels) of S = Q MQ^T and the B operand (cols = voxels) of S^T = MQ Q^T.
//   * S^T lands with lane = voxel, registers = codes: the feature-side softmax over the 64 codes is a per-lane loop
//     plus ONE exchange with lane^32.  S lands with lane = code, registers = voxels: the map-side (column) maximum /
//     sum over voxels is likewise per-lane plus one exchange.
//   * A product that sums over the index held in REGISTERS takes the accumulator registers directly as an operand:
//     the k-slot (8*(l>>5)+j) of k-step t is DEFINED as register 8t+j, i.e. index (j&3)+8*(2t+(j>>2))+4*(l>>5); the
//     other operand is gathered with the same slot order, so the sum over k is merely permuted.
// Per wave and 32 voxels: 16 MFMAs forward, 40 backward, ~100 scalar gathers per lane; the map side is written as one
// online-softmax record per wave (merged by k_attn_merge) exactly like the vector-ALU kernel does per workgroup.
#include "cbim_common.h"

namespace cbim {

static constexpr int MW = 256;   // threads per workgroup = 4 waves = 128 voxels
static constexpr int MDH = 32, MCODES = 64;

__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
  v.x = pk_bf16(f[0], f[1]); v.y = pk_bf16(f[2], f[3]); v.z = pk_bf16(f[4], f[5]); v.w = pk_bf16(f[6], f[7]);
  return v;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// index held by accumulator register r of a lane in half `hf` (rows of a 32x32 C/D tile)
__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
// index addressed by k-slot j of k-step t when the operand is taken from accumulator registers 8t..8t+7
__device__ __forceinline__ int slot_row(int t, int j, int hf) { return (j & 3) + 8 * (2 * t + (j >> 2)) + 4 * hf; }

// gather a [voxel-major] fragment: lane = voxel li (row/col), k = d = 16*s + 8*half + j, from bf16 rows
__device__ __forceinline__ u32x4 gather_vox_d(const bf16_t* base, bool valid, int s, int half, int heads) {
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = valid ? bf2f(base[(size_t)(16 * s + 8 * half + j) * heads]) : 0.f;
  return pack8(f);
}

__global__ void __launch_bounds__(MW) k_attn_fwd_mfma(const void* __restrict__ qv, int64_t rs,
                                                      const float* __restrict__ mq, const float* __restrict__ mv,
                                                      void* __restrict__ fo, float* __restrict__ part, int L, int heads,
                                                      float scale, int nrec) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, n = blockIdx.z;
  const int rec = blockIdx.x * (MW / 64) + wave;
  const int v0 = rec * 32;
  if (v0 >= L) return;                       // whole wave
  const int inner = heads * MDH;
  const bf16_t* qvb = (const bf16_t*)qv;
  const int vox = v0 + li;
  const bool valid = vox < L;
  const bf16_t* qrow = qvb + ((size_t)n * L + (valid ? vox : 0)) * rs + h;
  const float* mqh = mq + (size_t)n * MCODES * inner + h;
  const float* mvh = mv + (size_t)n * MCODES * inner + h;

  u32x4 Qf[2], MQf[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Qf[s] = gather_vox_d(qrow, valid, s, half, heads);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = mqh[(size_t)(32 * mb + li) * inner + (size_t)(16 * s + 8 * half + j) * heads];
      MQf[mb][s] = pack8(f);
    }
  }
  f32x16 ST[2], S[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    ST[b] = zero16(); S[b] = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ST[b] = mma16(MQf[b][s], Qf[s], ST[b]);      // rows = codes 32b.., cols = voxels
      S[b] = mma16(Qf[s], MQf[b][s], S[b]);        // rows = voxels, cols = codes 32b..
    }
  }
  // ---- feature side: softmax over the 64 codes of voxel `li` (this lane + lane^32) -------------------------
  float mx = -INFINITY;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ST[b][r] *= scale; mx = fmaxf(mx, ST[b][r]); }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ST[b][r] = expf(ST[b][r] - mx); sum += ST[b][r]; }
  sum += __shfl_xor(sum, 32, 64);
  f32x16 OT = zero16();                            // rows = d, cols = voxels
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float pf[8], wf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pf[j] = ST[b][8 * t + j];
        wf[j] = mvh[(size_t)(32 * b + slot_row(t, j, half)) * inner + (size_t)li * heads];   // MV[code][d = li]
      }
      OT = mma16(pack8(wf), pack8(pf), OT);
    }
  if (valid) {
    const float inv = 1.f / sum;
    bf16_t* orow = (bf16_t*)fo + ((size_t)n * L + vox) * inner + h;
#pragma unroll
    for (int r = 0; r < 16; ++r) orow[(size_t)acc_row(r, half) * heads] = (bf16_t)pk_bf16(OT[r] * inv, 0.f);
  }
  // ---- map side: online-softmax record of this wave's 32 voxels per code (lane = code, registers = voxels) ----
  u32x4 VCf[2];                                    // V as B operand: col d = li, k-slot -> voxel slot_row(t, j, half)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int vv = v0 + slot_row(t, j, half);
      f[j] = vv < L ? bf2f(qvb[((size_t)n * L + vv) * rs + inner + (size_t)li * heads + h]) : 0.f;
    }
    VCf[t] = pack8(f);
  }
  float* prec = part + (((size_t)n * heads + h) * nrec + rec) * MCODES * (MDH + 2);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float cm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[b][r] = (v0 + acc_row(r, half) < L) ? S[b][r] * scale : -INFINITY;
      cm = fmaxf(cm, S[b][r]);
    }
    cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
    float cs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[b][r] = expf(S[b][r] - cm); cs += S[b][r]; }
    cs += __shfl_xor(cs, 32, 64);
    f32x16 acc = zero16();                         // rows = codes 32b.., cols = d
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float ef[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ef[j] = S[b][8 * t + j];
      acc = mma16(pack8(ef), VCf[t], acc);
    }
    if (half == 0) {
      prec[(size_t)(32 * b + li) * (MDH + 2)] = cm;
      prec[(size_t)(32 * b + li) * (MDH + 2) + 1] = cs;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) prec[(size_t)(32 * b + acc_row(r, half)) * (MDH + 2) + 2 + li] = acc[r];
  }
}


// ---- backward -------------------------------------------------------------------------------------------------
// Map-side operand fragments are identical for every wave of a (n, head): the 4 waves of a workgroup build them
// once into LDS (20 fragments of 1 KiB).  Fragment kinds:
//   R-layout  (lane = code li of block mb, k = d = 16s + 8*half + j)           : MQ, MV, DMO      [mb][s]
//   slot layout (lane = d li, k-slot j of step t = code 32mb + slot_row(t,j,half)) : MQ^T, DMO^T  [mb][t]
// part record per (n, h, wave record): [2][64][32] = (dmv partial, dmq partial)
__global__ void __launch_bounds__(MW) k_attn_bwd_mfma(const void* __restrict__ qv, int64_t rs,
                                                      const float* __restrict__ mq, const float* __restrict__ mv,
                                                      const float* __restrict__ colstat, const float* __restrict__ map_out,
                                                      const void* __restrict__ dfo, const float* __restrict__ dmo,
                                                      void* __restrict__ dqv, float* __restrict__ part, int L, int heads,
                                                      float scale, int nrec) {
  __shared__ u32x4 frag_s[20][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, n = blockIdx.z;
  const int rec = blockIdx.x * (MW / 64) + wave;
  const int v0 = rec * 32;
  const bool wave_on = v0 < L;
  const int inner = heads * MDH;
  const float* mqh = mq + (size_t)n * MCODES * inner + h;
  const float* mvh = mv + (size_t)n * MCODES * inner + h;
  const float* dmh = dmo + (size_t)n * MCODES * inner + h;
  const float* moh = map_out + (size_t)n * MCODES * inner + h;
  // fragment ids: 0-3 MQ[mb][s], 4-7 MV[mb][s], 8-11 DMO[mb][s], 12-15 MQ^T[mb][t], 16-19 DMO^T[mb][t]
  for (int f = wave; f < 20; f += MW / 64) {
    const int kind = f >> 2, mb = (f >> 1) & 1, st = f & 1;
    const float* src = kind == 0 || kind == 3 ? mqh : (kind == 1 ? mvh : dmh);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (kind < 3) v[j] = src[(size_t)(32 * mb + li) * inner + (size_t)(16 * st + 8 * half + j) * heads];
      else v[j] = src[(size_t)(32 * mb + slot_row(st, j, half)) * inner + (size_t)li * heads];
    }
    frag_s[f][lane] = pack8(v);
  }
  __syncthreads();
  if (!wave_on) return;
  const bf16_t* qvb = (const bf16_t*)qv;
  const bf16_t* gb = (const bf16_t*)dfo;
  const int vox = v0 + li;
  const bool valid = vox < L;
  const bf16_t* qrow = qvb + ((size_t)n * L + (valid ? vox : 0)) * rs + h;
  const bf16_t* grow = gb + ((size_t)n * L + (valid ? vox : 0)) * inner + h;
  u32x4 Qf[2], Vf[2], Gf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Qf[s] = gather_vox_d(qrow, valid, s, half, heads);
    Vf[s] = gather_vox_d(qrow + inner, valid, s, half, heads);
    Gf[s] = gather_vox_d(grow, valid, s, half, heads);
  }
  // per-code constants of this lane's code in block b (S layout): column max / 1/sum / c = <map_out_j, dmo_j>
  float cM[2], cIS[2], cC[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int code = 32 * b + li;
    cM[b] = colstat[(((size_t)n * heads + h) * MCODES + code) * 2];
    cIS[b] = 1.f / colstat[(((size_t)n * heads + h) * MCODES + code) * 2 + 1];
    float c = 0.f;
    for (int d = 0; d < MDH; ++d) c += moh[(size_t)code * inner + (size_t)d * heads] * dmh[(size_t)code * inner + (size_t)d * heads];
    cC[b] = c;
  }
  // ================= T phase: lane = voxel, registers = codes =================================================
  float m_v, l_v, r_v;
  {
    f32x16 ST[2], dP1[2], dP2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      ST[b] = zero16(); dP1[b] = zero16(); dP2[b] = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        ST[b] = mma16(frag_s[0 + 2 * b + s][lane], Qf[s], ST[b]);
        dP1[b] = mma16(frag_s[4 + 2 * b + s][lane], Gf[s], dP1[b]);     // g . mv_code
        dP2[b] = mma16(frag_s[8 + 2 * b + s][lane], Vf[s], dP2[b]);     // v . dmo_code
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { ST[b][r] *= scale; mx = fmaxf(mx, ST[b][r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += expf(ST[b][r] - mx);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    float rr = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) rr += expf(ST[b][r] - mx) * inv * dP1[b][r];
    rr += __shfl_xor(rr, 32, 64);
    m_v = mx; l_v = inv; r_v = rr;
    // dA^T = P1 (dP1 - r) + P2 (dP2 - c_code);  P2^T kept in dP2, dA^T in dP1
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int src = acc_row(r, half);                      // lane (either half) owning code 32b + src in the S layout
        const float cm = __shfl(cM[b], src, 64), cis = __shfl(cIS[b], src, 64), cc = __shfl(cC[b], src, 64);
        const float p1 = expf(ST[b][r] - mx) * inv;
        const float p2 = valid ? expf(ST[b][r] - cm) * cis : 0.f;
        dP1[b][r] = p1 * (dP1[b][r] - rr) + p2 * (dP2[b][r] - cc);
        dP2[b][r] = p2;
      }
    f32x16 dqT = zero16(), dvT = zero16();                        // rows = d, cols = voxels
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float a[8], p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = dP1[b][8 * t + j]; p[j] = dP2[b][8 * t + j]; }
        dqT = mma16(frag_s[12 + 2 * b + t][lane], pack8(a), dqT);   // sum_code mq[code][d] dA[vox][code]
        dvT = mma16(frag_s[16 + 2 * b + t][lane], pack8(p), dvT);   // sum_code dmo[code][d] P2[vox][code]
      }
    if (valid) {
      bf16_t* drow = (bf16_t*)dqv + ((size_t)n * L + vox) * (2 * inner) + h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        drow[(size_t)acc_row(r, half) * heads] = (bf16_t)pk_bf16(dqT[r] * scale, 0.f);
        drow[inner + (size_t)acc_row(r, half) * heads] = (bf16_t)pk_bf16(dvT[r], 0.f);
      }
    }
  }
  // ================= S phase: lane = code, registers = voxels ================================================
  u32x4 QCf[2], GCf[2];                                           // col d = li, k-slot -> voxel slot_row(t, j, half)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float fq[8], fg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int vv = v0 + slot_row(t, j, half);
      const bool ok = vv < L;
      fq[j] = ok ? bf2f(qvb[((size_t)n * L + vv) * rs + (size_t)li * heads + h]) : 0.f;
      fg[j] = ok ? bf2f(gb[((size_t)n * L + vv) * inner + (size_t)li * heads + h]) : 0.f;
    }
    QCf[t] = pack8(fq);
    GCf[t] = pack8(fg);
  }
  float* prec = part + (((size_t)n * heads + h) * nrec + rec) * 2 * MCODES * MDH;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    f32x16 S = zero16(), dP1 = zero16(), dP2 = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      S = mma16(Qf[s], frag_s[0 + 2 * b + s][lane], S);
      dP1 = mma16(Gf[s], frag_s[4 + 2 * b + s][lane], dP1);
      dP2 = mma16(Vf[s], frag_s[8 + 2 * b + s][lane], dP2);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int src = acc_row(r, half);                           // lane owning voxel v0 + src in the T layout
      const float mv_ = __shfl(m_v, src, 64), lv_ = __shfl(l_v, src, 64), rv_ = __shfl(r_v, src, 64);
      const bool ok = v0 + src < L;
      const float sc = S[r] * scale;
      const float p1 = ok ? expf(sc - mv_) * lv_ : 0.f;
      const float p2 = ok ? expf(sc - cM[b]) * cIS[b] : 0.f;
      S[r] = p1;                                                  // P1
      dP1[r] = (p1 * (dP1[r] - rv_) + p2 * (dP2[r] - cC[b])) * scale;   // dA * scale
    }
    f32x16 dmvp = zero16(), dmqp = zero16();                      // rows = codes 32b.., cols = d
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a[8], p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { p[j] = S[8 * t + j]; a[j] = dP1[8 * t + j]; }
      dmvp = mma16(pack8(p), GCf[t], dmvp);                     // sum_vox P1[vox][code] g[vox][d]
      dmqp = mma16(pack8(a), QCf[t], dmqp);                       // sum_vox dA[vox][code] q[vox][d] * scale
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int code = 32 * b + acc_row(r, half);
      prec[((size_t)0 * MCODES + code) * MDH + li] = dmvp[r];
      prec[((size_t)1 * MCODES + code) * MDH + li] = dmqp[r];
    }
  }
}

}  // namespace cbim

using namespace cbim;

extern "C" int cbim_attn_mfma_records(int L) { return (L + 31) / 32; }

// launched by cbim_bidir_attn_fwd (medformer_kernels.hip) for bf16 / d_head 32 / 64 codes
extern "C" int cbim_attn_fwd_mfma_launch(const void* qv, int64_t qv_stride, const float* mq, const float* mv, void* feat_out,
                                         float* part, int N, int L, int heads, float scale, void* stream) {
  int nrec = (L + 31) / 32;
  dim3 grid((nrec + MW / 64 - 1) / (MW / 64), heads, N);
  CBIM_LAUNCH(k_attn_fwd_mfma, grid, dim3(MW), 0, (hipStream_t)stream, qv, qv_stride, mq, mv, feat_out, part, L, heads, scale,
              nrec);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "bidir_attn_fwd (mfma) launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

extern "C" int cbim_attn_bwd_mfma_launch(const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                                         const float* colstat, const float* map_out, const void* d_feat_out,
                                         const float* d_map_out, void* d_qv, float* part, int N, int L, int heads, float scale,
                                         void* stream) {
  int nrec = (L + 31) / 32;
  dim3 grid((nrec + MW / 64 - 1) / (MW / 64), heads, N);
  CBIM_LAUNCH(k_attn_bwd_mfma, grid, dim3(MW), 0, (hipStream_t)stream, qv, qv_stride, mq, mv, colstat, map_out, d_feat_out,
              d_map_out, d_qv, part, L, heads, scale, nrec);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "bidir_attn_bwd (mfma) launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(attn_mfma)
