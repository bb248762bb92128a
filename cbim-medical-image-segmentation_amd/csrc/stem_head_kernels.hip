// stem_head_kernels.hip — the two HBM-bound convolutions at the model boundary.
//
//   stem: inconv.conv1 = nn.Conv3d(in_ch, base, k, pad k//2, bias=False)  (unet_utils.py:14,19):
//         reads the caller's NCDHW fp32 volume (in_ch is 1..4), writes channels-last T.
//         AI ~ 13 FLOP/B -> direct VALU kernel, LDS halo of the input, weights broadcast from LDS.
//   head: outc = nn.Conv3d(base, classes, 1) with bias (unet.py:47): channels-last T in, NCDHW fp32
//         logits out (the layout nn.CrossEntropyLoss / DiceLoss consume, train.py:212).
#include "cbim_common.h"
#include <stdlib.h>

namespace cbim {

static constexpr int NT = 256;
static constexpr int MAXTAPS = 27;
static constexpr int KMAX = 32;   // classes handled per pass of the head kernels

struct StemParams {
  const float* x; const float* w; void* y; const void* dy; float* ws;
  int N, Cin, Di, Hi, Wi, Cout, kD, kH, kW, pD, pH, pW, Do, Ho, Wo;
  int tiles_d, tiles_h, tiles_w, hD, hH, hW, taps;
  int strips_per_n, tiles_per_strip;
};

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// tile = 4 x 8 x 8 output voxels, one thread per voxel, 8 couts per register pass.
template <typename T>
__global__ void __launch_bounds__(NT) k_stem_fwd(StemParams p) {
  constexpr int CPC = Elem<T>::CPC;
  CBIM_DYN_SMEM(smem);
  const int hV = p.hD * p.hH * p.hW;
  float* xL = (float*)smem;                       // [Cin][hV]
  float* wL = xL + (size_t)p.Cin * hV;            // [taps*Cin][Cout]  (cout fastest -> broadcast reads)
  const int tid = threadIdx.x;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int n = bid / tiles_per_n, t = bid % tiles_per_n;
  const int od0 = (t / (p.tiles_w * p.tiles_h)) * 4, oh0 = ((t / p.tiles_w) % p.tiles_h) * 8, ow0 = (t % p.tiles_w) * 8;
  const size_t Sin = (size_t)p.Di * p.Hi * p.Wi;
  for (int i = tid; i < p.Cin * hV; i += NT) {
    int ci = i / hV, hv = i % hV;
    int hw = hv % p.hW, r = hv / p.hW, hh = r % p.hH, hd = r / p.hH;
    int id = od0 - p.pD + hd, ih = oh0 - p.pH + hh, iw = ow0 - p.pW + hw;
    float v = 0.f;
    if (id >= 0 && id < p.Di && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi)
      v = p.x[((size_t)n * p.Cin + ci) * Sin + ((size_t)id * p.Hi + ih) * p.Wi + iw];
    xL[i] = v;
  }
  for (int i = tid; i < p.taps * p.Cin * p.Cout; i += NT) {
    int co = i % p.Cout, r = i / p.Cout;       // r = tap*Cin + ci
    int ci = r % p.Cin, tap = r / p.Cin;
    wL[i] = p.w[((size_t)co * p.Cin + ci) * p.taps + tap];
  }
  __syncthreads();
  const int tw = tid & 7, th = (tid >> 3) & 7, td = tid >> 6;
  const int od = od0 + td, oh = oh0 + th, ow = ow0 + tw;
  const bool ok = od < p.Do && oh < p.Ho && ow < p.Wo;
  const int hv0 = (td * p.hH + th) * p.hW + tw;
  const size_t row = (size_t)n * p.Do * p.Ho * p.Wo + ((size_t)od * p.Ho + oh) * p.Wo + ow;
  for (int cg = 0; cg < p.Cout; cg += CPC) {
    float acc[CPC];
#pragma unroll
    for (int j = 0; j < CPC; ++j) acc[j] = 0.f;
    for (int ci = 0; ci < p.Cin; ++ci) {
      int tap = 0;
      for (int kd = 0; kd < p.kD; ++kd)
        for (int kh = 0; kh < p.kH; ++kh)
          for (int kw = 0; kw < p.kW; ++kw, ++tap) {
            float xv = xL[ci * hV + hv0 + (kd * p.hH + kh) * p.hW + kw];
            const float* wr = wL + (size_t)(tap * p.Cin + ci) * p.Cout + cg;
#pragma unroll
            for (int j = 0; j < CPC; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
          }
    }
    if (ok) st_chunk<T>(p.y, row * p.Cout + cg, Elem<T>::pack(acc));
  }
}

// ---- stem forward on the matrix cores: one input channel, <= 27 taps, bf16 rows of Cout = 32 * CP channels --------------
// out^T[co][v] = sum_tap W[co][tap] X[tap][v]: v_mfma_f32_16x16x32_bf16 with the (<= 27, zero padded) TAPS as the contraction.
// A = W rows (co permuted inside a 32-channel pair so that a lane ends up with 8 consecutive channels of one voxel: one
// 16-byte store), B = the im2col column of a voxel gathered from the LDS halo (8 ds_read_b32 per lane and 16 voxels).  Both
// operands are split into bf16 hi + lo (hi*hi + lo*hi + hi*lo: the fp32 image and the fp32 master weights keep ~2^-17), the
// matrix cores have time to spare: the kernel is bound by its 16-byte output stores.  k_stem_fwd spends 27 x 32 fp32 FMAs and
// 27 x 9 LDS reads per voxel (152 us at 128^3 x 32; the output is written in ~25 us).
template <int CP>
__global__ void __launch_bounds__(NT) k_stem_fwd_mfma(StemParams p) {
  CBIM_DYN_SMEM(smem);
  unsigned* halo = (unsigned*)smem;                       // [hD][hH][hW]: bf16 hi | bf16 lo << 16
  const int hV = p.hD * p.hH * p.hW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w, n_tiles = p.N * tiles_per_n;
  const size_t Sin = (size_t)p.Di * p.Hi * p.Wi;
  // weight fragments (hi, lo) and the halo offsets of this lane's 8 taps: once per (persistent) workgroup
  u32x4 wh[CP][2], wl[CP][2];
  int off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int tp = 8 * g + j;
    const int kw = tp % p.kW, kh = (tp / p.kW) % p.kH, kd = tp / (p.kW * p.kH);
    off[j] = tp < p.taps ? (kd * p.hH + kh) * p.hW + kw : 0;
  }
#pragma unroll
  for (int q = 0; q < CP; ++q)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int co = 32 * q + 8 * (li >> 2) + 4 * u + (li & 3);
      float fh[8], fl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tp = 8 * g + j;
        const float wv = tp < p.taps ? p.w[(size_t)co * p.taps + tp] : 0.f;
        fh[j] = Elem<bf16_tag>::round(wv);
        fl[j] = wv - fh[j];
      }
      wh[q][u] = Elem<bf16_tag>::pack(fh);
      wl[q][u] = Elem<bf16_tag>::pack(fl);
    }
  // this thread's halo items (the same positions in every tile)
  constexpr int HI = 3;                                   // 600 halo values of a 3x3x3 stem over 256 threads
  int hpos[HI];
#pragma unroll
  for (int e = 0; e < HI; ++e) {
    const int i = tid + NT * e;
    const int hw = i % p.hW, r = i / p.hW, hh = r % p.hH, hd = r / p.hH;
    hpos[e] = i < hV ? (hd | (hh << 8) | (hw << 16)) : -1;
  }
  for (int tile = (int)xcd_remap(blockIdx.x, gridDim.x); tile < n_tiles; tile += gridDim.x) {
    const int n = tile / tiles_per_n, t = tile % tiles_per_n;
    const int od0 = (t / (p.tiles_w * p.tiles_h)) * 4, oh0 = ((t / p.tiles_w) % p.tiles_h) * 8, ow0 = (t % p.tiles_w) * 8;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < HI; ++e) {
      if (hpos[e] < 0) continue;
      const int id = od0 - p.pD + (hpos[e] & 255), ih = oh0 - p.pH + ((hpos[e] >> 8) & 255), iw = ow0 - p.pW + (hpos[e] >> 16);
      float v = 0.f;
      if (id >= 0 && id < p.Di && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi) v = p.x[(size_t)n * Sin + ((size_t)id * p.Hi + ih) * p.Wi + iw];
      const unsigned hi = pk_bf16(v, 0.f) & 0xffffu;
      const unsigned lo = pk_bf16(v - __uint_as_float(hi << 16), 0.f) & 0xffffu;
      halo[tid + NT * e] = hi | (lo << 16);
    }
    __syncthreads();
    const int td = wave, od = od0 + td;
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
      const int th = 2 * r2 + (li >> 3), tw = li & 7;
      const int base = (td * p.hH + th) * p.hW + tw;
      unsigned u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = halo[base + off[j]];
      u32x4 bh, bl;
      bh.x = (u[0] & 0xffffu) | (u[1] << 16); bh.y = (u[2] & 0xffffu) | (u[3] << 16);
      bh.z = (u[4] & 0xffffu) | (u[5] << 16); bh.w = (u[6] & 0xffffu) | (u[7] << 16);
      bl.x = (u[0] >> 16) | (u[1] & 0xffff0000u); bl.y = (u[2] >> 16) | (u[3] & 0xffff0000u);
      bl.z = (u[4] >> 16) | (u[5] & 0xffff0000u); bl.w = (u[6] >> 16) | (u[7] & 0xffff0000u);
      const int oh = oh0 + th, ow = ow0 + tw;
      const bool ok = od < p.Do && oh < p.Ho && ow < p.Wo;
      const size_t row = (size_t)n * p.Do * p.Ho * p.Wo + ((size_t)od * p.Ho + oh) * p.Wo + ow;
#pragma unroll
      for (int q = 0; q < CP; ++q) {
        f32x4 d[2];
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          d[uu] = f32x4{0.f, 0.f, 0.f, 0.f};
          d[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl[q][uu]), __builtin_bit_cast(bf16x8, bh), d[uu], 0, 0, 0);
          d[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[q][uu]), __builtin_bit_cast(bf16x8, bl), d[uu], 0, 0, 0);
          d[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[q][uu]), __builtin_bit_cast(bf16x8, bh), d[uu], 0, 0, 0);
        }
        const float f[8] = {d[0][0], d[0][1], d[0][2], d[0][3], d[1][0], d[1][1], d[1][2], d[1][3]};
        if (ok) st_chunk<bf16_tag>(p.y, row * p.Cout + 32 * q + 8 * g, Elem<bf16_tag>::pack(f));
      }
    }
  }
}

// LDS transpose read / wave-level LDS fence of the matrix-core kernels below
__device__ __forceinline__ u32x2 hd_tr16_b64(const unsigned char* p) {
#ifdef CBIM_EMU
  unsigned short o[4];
  emu_ds_read_tr16_b64(p, o);
  u32x2 r;
  r.x = (unsigned)o[0] | ((unsigned)o[1] << 16);
  r.y = (unsigned)o[2] | ((unsigned)o[3] << 16);
  return r;
#else
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(u32x2, v);
#endif
}
// LDS written by some lanes of the wave, read by others: DS instructions of a wave execute in order, so only the compiler
// (and the host-side executor, whose lanes are fibers) needs a fence
__device__ __forceinline__ void hd_wave_sync() {
#ifdef CBIM_EMU
  (void)__any(0);
#else
  asm volatile("" ::: "memory");
#endif
}


// ---- stem weight gradient on the matrix cores: one input channel, bf16 dy rows of Cout = 32 * CP channels ------------------
// dw[co][tap] = sum_v dy[v][co] x[v + tap]: v_mfma_f32_16x16x32_bf16 contracting over 32 VOXELS (4 w-rows of the 4x8x8 tile).
// A = dy^T through the LDS transpose read (as k_wgrad_r32), B = the im2col column of a tap: the 8 voxels of a w-row are 8
// consecutive halo values starting at kw — one 16-byte + one 4-byte LDS read and a 16-bit funnel shift; the fp32 image is
// split into bf16 hi + lo planes (two MFMAs, ~2^-17).  Accumulators [co 32 CP][tap 32] stay in registers over the
// workgroup's strip of tiles; slab per workgroup + k_stem_wgrad_reduce as before.  k_stem_wgrad spends 27 x 8 FMAs per
// (voxel row, co) thread on the vector ALU: 196 us at 128^3 x 32 (dy is read in ~25 us).
template <int CP>
__global__ void __launch_bounds__(NT) k_stem_wgrad_mfma(StemParams p) {
  constexpr int Cout = 32 * CP, ROWB = Cout * 2;
  CBIM_DYN_SMEM(smem);
  const int rows = p.hD * p.hH;
  unsigned char* hiL = smem;                                  // [hD*hH][16] bf16
  unsigned char* loL = hiL + rows * 32;
  unsigned char* dyL = loL + rows * 32;                       // [256 voxels][Cout] bf16 (offset: a multiple of 64 B)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int n = blockIdx.x / p.strips_per_n, strip = blockIdx.x % p.strips_per_n;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int t_begin = strip * p.tiles_per_strip;
  int t_end = t_begin + p.tiles_per_strip;
  if (t_end > tiles_per_n) t_end = tiles_per_n;
  const size_t Sin = (size_t)p.Di * p.Hi * p.Wi;
  // this lane's two taps (tap tile tt: tap = li + 16 tt)
  int trow[2], tkw[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int tp = li + 16 * tt < p.taps ? li + 16 * tt : 0;
    const int kw = tp % p.kW, kh = (tp / p.kW) % p.kH, kd = tp / (p.kW * p.kH);
    trow[tt] = kd * p.hH + kh;
    tkw[tt] = kw;
  }
  f32x4 acc[CP][2][2];
#pragma unroll
  for (int q = 0; q < CP; ++q)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) acc[q][u][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = t_begin; t < t_end; ++t) {
    const int od0 = (t / (p.tiles_w * p.tiles_h)) * 4, oh0 = ((t / p.tiles_w) % p.tiles_h) * 8, ow0 = (t % p.tiles_w) * 8;
    __syncthreads();
    for (int i = tid; i < rows * 16; i += NT) {
      const int hw = i & 15, r = i >> 4, hh = r % p.hH, hd = r / p.hH;
      const int id = od0 - p.pD + hd, ih = oh0 - p.pH + hh, iw = ow0 - p.pW + hw;
      float v = 0.f;
      if (hw < p.hW && id >= 0 && id < p.Di && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi)
        v = p.x[(size_t)n * Sin + ((size_t)id * p.Hi + ih) * p.Wi + iw];
      const unsigned hi = pk_bf16(v, 0.f) & 0xffffu;
      ((bf16_t*)hiL)[i] = (bf16_t)hi;
      ((bf16_t*)loL)[i] = (bf16_t)(pk_bf16(v - __uint_as_float(hi << 16), 0.f) & 0xffffu);
    }
#pragma unroll
    for (int u = 0; u < 4 * CP; ++u) {                         // 256 voxels x 4 CP chunks of 16 bytes
      const int item = tid + NT * u, vx = item / (4 * CP), ch = item % (4 * CP);
      const int od = od0 + (vx >> 6), oh = oh0 + ((vx >> 3) & 7), ow = ow0 + (vx & 7);
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (od < p.Do && oh < p.Ho && ow < p.Wo) {
        const size_t row = (size_t)n * p.Do * p.Ho * p.Wo + ((size_t)od * p.Ho + oh) * p.Wo + ow;
        val = ld_chunk<bf16_tag>(p.dy, row * Cout + (size_t)ch * 8);
      }
      *(u32x4*)(dyL + (size_t)vx * ROWB + ch * 16) = val;
    }
    __syncthreads();
    const int td = wave;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                           // two 32-voxel steps per wave: w-rows th0 .. th0 + 3 of plane td
      const int th0 = 4 * ks;
      u32x4 bh[2], bl[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int r = (td * p.hH + th0 + g) + trow[tt];
        const int kw = tkw[tt];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          const unsigned char* src = (part ? loL : hiL) + r * 32;
          const u32x4 a = *(const u32x4*)src;
          const unsigned a4 = *(const unsigned*)(src + 16);
          const unsigned d0 = a.x, d1 = a.y, d2 = a.z, d3 = a.w;
          u32x4 o;
          o.x = kw == 0 ? d0 : kw == 1 ? (d0 >> 16) | (d1 << 16) : d1;
          o.y = kw == 0 ? d1 : kw == 1 ? (d1 >> 16) | (d2 << 16) : d2;
          o.z = kw == 0 ? d2 : kw == 1 ? (d2 >> 16) | (d3 << 16) : d3;
          o.w = kw == 0 ? d3 : kw == 1 ? (d3 >> 16) | (a4 << 16) : a4;
          if (part) bl[tt] = o; else bh[tt] = o;
        }
      }
#pragma unroll
      for (int q = 0; q < CP; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned char* a = dyL + (size_t)(td * 64 + th0 * 8 + 8 * g + (li >> 2)) * ROWB + (32 * q + 16 * u) * 2 + (li & 3) * 8;
          const u32x2 a0 = hd_tr16_b64(a), a1 = hd_tr16_b64(a + 4 * ROWB);
          const u32x4 af = u32x4{a0.x, a0.y, a1.x, a1.y};
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            acc[q][u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bl[tt]), acc[q][u][tt], 0, 0, 0);
            acc[q][u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bh[tt]), acc[q][u][tt], 0, 0, 0);
          }
        }
    }
  }
  // four waves' partial sums -> the workgroup's slab [taps][Cout], fixed order
  __syncthreads();
  float* red = (float*)smem;                                   // [4][32 taps][Cout]
#pragma unroll
  for (int q = 0; q < CP; ++q)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * 32 + li + 16 * tt) * Cout + 32 * q + 16 * u + 4 * g + i] = acc[q][u][tt][i];
  __syncthreads();
  const size_t slab = (size_t)p.taps * Cout;
  for (int o = tid; o < p.taps * Cout; o += NT) {
    const int co = o % Cout, k = o / Cout;
    const float a = (red[(0 * 32 + k) * Cout + co] + red[(1 * 32 + k) * Cout + co]) + (red[(2 * 32 + k) * Cout + co] + red[(3 * 32 + k) * Cout + co]);
    p.ws[(size_t)blockIdx.x * slab + o] = a;
  }
}

// dw[co][ci][tap] = sum_v dy[v][co] * x[v+tap-p][ci]; thread = (co lane, row group); a thread walks whole
// 8-voxel W rows so each halo value it reads from LDS feeds up to kW FMAs (LDS reads per FMA ~0.4);
// strip of tiles per workgroup; per-strip slab [Cin][taps][Cout] then a fixed-order reduce.
// CG input channels share one pass over dy (multi-modal inputs: dy is 48 bf16 channels x 128^3 = 201 MB, read once
// instead of once per input channel).
template <typename T, int CG>
__global__ void __launch_bounds__(NT) k_stem_wgrad(StemParams p) {
  CBIM_DYN_SMEM(smem);
  const int hV = p.hD * p.hH * p.hW;
  const int hVp = (hV + 3) & ~3;
  float* xL = (float*)smem;                           // [CG][hVp]
  float* red = xL + CG * hVp;                         // [NT/32 groups][MAXTAPS][32]
  const int tid = threadIdx.x, col = tid & 31, vg = tid >> 5;   // 8 row groups
  const int n = blockIdx.x / p.strips_per_n, strip = blockIdx.x % p.strips_per_n;
  const int cob = blockIdx.y;   // block of 32 couts
  const int co = cob * 32 + col;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int t_begin = strip * p.tiles_per_strip;
  int t_end = t_begin + p.tiles_per_strip;
  if (t_end > tiles_per_n) t_end = tiles_per_n;
  const size_t Sin = (size_t)p.Di * p.Hi * p.Wi;
  const size_t slab = (size_t)p.Cin * p.taps * p.Cout;
  for (int ci0 = 0; ci0 < p.Cin; ci0 += CG) {
    float acc[CG][MAXTAPS];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int k = 0; k < MAXTAPS; ++k) acc[c][k] = 0.f;
    for (int t = t_begin; t < t_end; ++t) {
      const int od0 = (t / (p.tiles_w * p.tiles_h)) * 4, oh0 = ((t / p.tiles_w) % p.tiles_h) * 8, ow0 = (t % p.tiles_w) * 8;
      __syncthreads();
      for (int hv = tid; hv < hV; hv += NT) {
        int hw = hv % p.hW, r = hv / p.hW, hh = r % p.hH, hd = r / p.hH;
        int id = od0 - p.pD + hd, ih = oh0 - p.pH + hh, iw = ow0 - p.pW + hw;
        const bool in = id >= 0 && id < p.Di && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
#pragma unroll
        for (int c = 0; c < CG; ++c)
          xL[c * hVp + hv] = (in && ci0 + c < p.Cin)
                                 ? p.x[((size_t)n * p.Cin + ci0 + c) * Sin + ((size_t)id * p.Hi + ih) * p.Wi + iw] : 0.f;
      }
      __syncthreads();
      for (int rowi = vg; rowi < 32; rowi += 8) {      // rows (td, th) of the 4x8x8 tile
        const int td = rowi >> 3, th = rowi & 7;
        const int od = od0 + td, oh = oh0 + th;
        float g[8];
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
          g[w8] = 0.f;
          if (co < p.Cout && od < p.Do && oh < p.Ho && ow0 + w8 < p.Wo) {
            size_t row = (size_t)n * p.Do * p.Ho * p.Wo + ((size_t)od * p.Ho + oh) * p.Wo + ow0 + w8;
            g[w8] = Elem<T>::load1(p.dy, row * p.Cout + co);
          }
        }
        // static 3x3x3 tap lattice (extents <= 3 per axis, guarded): accumulator indices stay compile-time
#pragma unroll
        for (int c = 0; c < CG; ++c) {
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
            if (kd < p.kD) {
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                if (kh < p.kH) {
                  const float* xr = xL + c * hVp + ((td + kd) * p.hH + th + kh) * p.hW;
                  float xv[10];
#pragma unroll
                  for (int i = 0; i < 10; ++i) xv[i] = i < 8 + p.kW - 1 ? xr[i] : 0.f;
#pragma unroll
                  for (int kw = 0; kw < 3; ++kw) {
                    if (kw < p.kW) {
                      float a = acc[c][(kd * 3 + kh) * 3 + kw];
#pragma unroll
                      for (int w8 = 0; w8 < 8; ++w8) a = fmaf(g[w8], xv[w8 + kw], a);
                      acc[c][(kd * 3 + kh) * 3 + kw] = a;
                    }
                  }
                }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < MAXTAPS; ++k) red[(vg * MAXTAPS + k) * 32 + col] = acc[c][k];
      __syncthreads();
      if (ci0 + c < p.Cin)
        for (int i = tid; i < p.taps * 32; i += NT) {
          int cc = i & 31, k = i >> 5;
          float a = 0.f;
          for (int g = 0; g < 8; ++g) a += red[(g * MAXTAPS + k) * 32 + cc];
          if (cob * 32 + cc < p.Cout)
            p.ws[(size_t)blockIdx.x * slab + ((size_t)(ci0 + c) * p.taps + k) * p.Cout + cob * 32 + cc] = a;
        }
    }
  }
}

// ---- stem weight gradient for kernels beyond 3x3x3 (VNet's 5x5x5 input convolution, vnet.py:46: 125 taps) ----------------
// thread = (tap slot 0..127, half of a 16-cout block): 8 accumulators; per output voxel of the 4x8x8 tile one halo value (its
// own tap's) and the voxel's 8 dy values (two 16-byte LDS reads shared by the 128 threads of the half) feed 8 FMAs.  Strip
// of tiles per workgroup, slab [Cin][taps][Cout] per strip, then k_stem_wgrad_reduce as for the small stems.
template <typename T>
__global__ void __launch_bounds__(NT) k_stem_wgrad_taps(StemParams p) {
  CBIM_DYN_SMEM(smem);
  const int hV = p.hD * p.hH * p.hW;
  const int hVp = (hV + 3) & ~3;
  float* xL = (float*)smem;                           // [hVp]
  float* dL = xL + hVp;                               // [256 voxels][16 couts]
  const int tid = threadIdx.x, slot = tid & 127, half = tid >> 7;
  const int n = blockIdx.x / p.strips_per_n, strip = blockIdx.x % p.strips_per_n;
  const int co0 = blockIdx.y * 16;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int t_begin = strip * p.tiles_per_strip;
  int t_end = t_begin + p.tiles_per_strip;
  if (t_end > tiles_per_n) t_end = tiles_per_n;
  const size_t Sin = (size_t)p.Di * p.Hi * p.Wi;
  const size_t slab = (size_t)p.Cin * p.taps * p.Cout;
  const int tap = slot < p.taps ? slot : 0;
  const int kw = tap % p.kW, kh = (tap / p.kW) % p.kH, kd = tap / (p.kW * p.kH);
  const int toff = (kd * p.hH + kh) * p.hW + kw;
  for (int ci = 0; ci < p.Cin; ++ci) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int t = t_begin; t < t_end; ++t) {
      const int od0 = (t / (p.tiles_w * p.tiles_h)) * 4, oh0 = ((t / p.tiles_w) % p.tiles_h) * 8, ow0 = (t % p.tiles_w) * 8;
      __syncthreads();
      for (int hv = tid; hv < hV; hv += NT) {
        int hw = hv % p.hW, r = hv / p.hW, hh = r % p.hH, hd = r / p.hH;
        int id = od0 - p.pD + hd, ih = oh0 - p.pH + hh, iw = ow0 - p.pW + hw;
        const bool in = id >= 0 && id < p.Di && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
        xL[hv] = in ? p.x[((size_t)n * p.Cin + ci) * Sin + ((size_t)id * p.Hi + ih) * p.Wi + iw] : 0.f;
      }
      {
        const int tw = tid & 7, th = (tid >> 3) & 7, td = tid >> 6;
        const int od = od0 + td, oh = oh0 + th, ow = ow0 + tw;
        const bool ok = od < p.Do && oh < p.Ho && ow < p.Wo;
        const size_t row = (size_t)n * p.Do * p.Ho * p.Wo + ((size_t)od * p.Ho + oh) * p.Wo + ow;
        for (int j = 0; j < 16; ++j)
          dL[tid * 16 + j] = (ok && co0 + j < p.Cout) ? Elem<T>::load1(p.dy, row * p.Cout + co0 + j) : 0.f;
      }
      __syncthreads();
      for (int v = 0; v < 256; ++v) {
        const int tw = v & 7, th = (v >> 3) & 7, td = v >> 6;
        const float xv = xL[(td * p.hH + th) * p.hW + tw + toff];
        const float* dv = dL + v * 16 + half * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(dv[j], xv, acc[j]);
      }
    }
    if (slot < p.taps)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = co0 + half * 8 + j;
        if (co < p.Cout) p.ws[(size_t)blockIdx.x * slab + ((size_t)ci * p.taps + slot) * p.Cout + co] = acc[j];
      }
  }
}

// 16 outputs per workgroup x 16 slab phases: thread (o, ph) adds slabs ph, ph+16, .. (two independent chains), the 16 phase
// sums are combined through LDS in fixed order.  (One thread per output walking all 1024 slabs: 4 workgroups, 80 us.)
__global__ void __launch_bounds__(NT) k_stem_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw,
                                                          int n_slabs, int Cin, int taps, int Cout) {
  __shared__ float red[16][17];
  const int total = Cin * taps * Cout;
  const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
  for (int base = blockIdx.x * 16; base < total; base += gridDim.x * 16) {
    const int i = base + o;
    float a0 = 0.f, a1 = 0.f;
    if (i < total) {
      int s = ph;
      for (; s + 16 < n_slabs; s += 32) {
        a0 += ws[(size_t)s * total + i];
        a1 += ws[(size_t)(s + 16) * total + i];
      }
      if (s < n_slabs) a0 += ws[(size_t)s * total + i];
    }
    red[ph][o] = a0 + a1;
    __syncthreads();
    if (ph == 0 && i < total) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += red[q][o];
      const int co = i % Cout, r = i / Cout;
      const int k = r % taps, ci = r / taps;
      dw[((size_t)co * Cin + ci) * taps + k] = acc;
    }
    __syncthreads();
  }
}

// ---- head ------------------------------------------------------------------------------------------------
// one thread per voxel; the voxel's channels stream through registers chunk by chunk; weights and
// bias are broadcast LDS reads; logits are written plane by plane (coalesced over voxels).
template <typename T>
__global__ void __launch_bounds__(NT) k_head_fwd(const void* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ b, float* __restrict__ logits,
                                                 int64_t S, int Cin, int K, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  CBIM_DYN_SMEM(smem);
  float* wL = (float*)smem;  // [Cin][K]  (k fastest)
  for (int i = threadIdx.x; i < Cin * K; i += NT) {
    int k = i % K, c = i / K;
    wL[i] = w[(size_t)k * Cin + c];
  }
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += KMAX) {
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < total; r += (int64_t)gridDim.x * NT) {
      float out[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) out[k] = (k0 + k < K) ? b[k0 + k] : 0.f;
      for (int cb = 0; cb < Cin; cb += 8 * CPC) {   // 8 chunk loads in flight before any is used
        u32x4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (cb + u * CPC < Cin) raw[u] = ld_chunk<T>(x, (size_t)r * Cin + cb + u * CPC);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c0 = cb + u * CPC;
          if (c0 < Cin) {
            float f[CPC];
            Elem<T>::unpack(raw[u], f);
#pragma unroll
            for (int j = 0; j < CPC; ++j) {
              const float* wr = wL + (size_t)(c0 + j) * K + k0;
#pragma unroll
              for (int k = 0; k < KMAX; ++k)
                if (k0 + k < K) out[k] = fmaf(f[j], wr[k], out[k]);
            }
          }
        }
      }
      int64_t n = r / S, v = r % S;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k0 + k < K) logits[((size_t)n * K + k0 + k) * S + v] = out[k];
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(NT) k_head_bwd_dx(const float* __restrict__ w, const float* __restrict__ dz,
                                                    void* __restrict__ dx, int64_t S, int Cin, int K,
                                                    int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  CBIM_DYN_SMEM(smem);
  float* wL = (float*)smem;  // [K][Cin]
  for (int i = threadIdx.x; i < Cin * K; i += NT) wL[i] = w[i];
  __syncthreads();
  for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < total; r += (int64_t)gridDim.x * NT) {
    int64_t n = r / S, v = r % S;
    for (int c0 = 0; c0 < Cin; c0 += CPC) {
      float f[CPC];
#pragma unroll
      for (int j = 0; j < CPC; ++j) f[j] = 0.f;
      for (int k = 0; k < K; ++k) {
        float g = dz[((size_t)n * K + k) * S + v];
        const float* wr = wL + (size_t)k * Cin + c0;
#pragma unroll
        for (int j = 0; j < CPC; ++j) f[j] = fmaf(g, wr[j], f[j]);
      }
      st_chunk<T>(dx, (size_t)r * Cin + c0, Elem<T>::pack(f));
    }
  }
}

// dw[k][c] = sum_v dz[k][v] x[v][c], db[k] = sum_v dz[k][v] (bias = an extra all-ones column of x).
// Workgroup = strip of voxels staged in 64-voxel LDS tiles ([v][c] and [v][k], rows padded to 16 B);
// a thread owns a 4(k) x 4(c) block of the product and one of VG voxel phases: two ds_read_b128 feed
// 16 FMAs.  Slab per workgroup, fixed-order reduce.
static constexpr int HB_TILE = 64;
template <typename T>
__global__ void __launch_bounds__(NT) k_head_bwd_dw(const void* __restrict__ x, const float* __restrict__ dz,
                                                    float* __restrict__ ws, int64_t S, int Cin, int K,
                                                    int64_t vox_per_block) {
  CBIM_DYN_SMEM(smem);
  const int XS = (Cin + 1 + 3) & ~3, ZS = (K + 3) & ~3;
  const int KB = ZS / 4, CB = XS / 4, PB = KB * CB, VG = NT / PB;   // host guarantees PB <= NT
  float* xL = (float*)smem;               // [HB_TILE][XS]
  float* zL = xL + HB_TILE * XS;          // [HB_TILE][ZS]
  float* red = zL + HB_TILE * ZS;         // [VG][PB][16]
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int pb = tid % PB, vg = tid / PB;
  const int kb = pb / CB, cbk = pb % CB;
  const bool live = vg < VG;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = 0.f;
  int64_t v_begin = (int64_t)blockIdx.x * vox_per_block;
  int64_t v_end = v_begin + vox_per_block;
  if (v_end > S) v_end = S;
  for (int64_t vb = v_begin; vb < v_end; vb += HB_TILE) {
    __syncthreads();
    {   // x tile: whole 16-byte channel chunks per lane; column Cin = 1 (bias), padding columns = 0
      constexpr int CPC = Elem<T>::CPC;
      const int cch = Cin / CPC;
      for (int i = tid; i < HB_TILE * cch; i += NT) {
        int cc = i % cch, m = i / cch;
        int64_t v = vb + m;
        float f[CPC];
#pragma unroll
        for (int j = 0; j < CPC; ++j) f[j] = 0.f;
        if (v < v_end) Elem<T>::unpack(ld_chunk<T>(x, ((size_t)n * S + v) * Cin + (size_t)cc * CPC), f);
#pragma unroll
        for (int j = 0; j < CPC; ++j) xL[m * XS + cc * CPC + j] = f[j];
      }
      for (int i = tid; i < HB_TILE * (XS - Cin); i += NT) {
        int c = Cin + i % (XS - Cin), m = i / (XS - Cin);
        xL[m * XS + c] = (c == Cin && vb + m < v_end) ? 1.f : 0.f;
      }
    }
    for (int i = tid; i < HB_TILE * ZS; i += NT) {
      int m = i % HB_TILE, k = i / HB_TILE;     // coalesced over voxels per class plane
      int64_t v = vb + m;
      zL[m * ZS + k] = (v < v_end && k < K) ? dz[((size_t)n * K + k) * S + v] : 0.f;
    }
    __syncthreads();
    if (live) {
      for (int m = vg; m < HB_TILE; m += VG) {
        f32x4 zv = *(const f32x4*)(zL + m * ZS + kb * 4);
        f32x4 xv = *(const f32x4*)(xL + m * XS + cbk * 4);
        float zz[4] = {zv.x, zv.y, zv.z, zv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = fmaf(zz[a], xx[b2], acc[a][b2]);
      }
    }
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) red[(vg * PB + pb) * 16 + a * 4 + b2] = acc[a][b2];
  }
  __syncthreads();
  const int npairs = K * (Cin + 1);
  float* slab = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * npairs;
  for (int pr = tid; pr < npairs; pr += NT) {
    int c = pr % (Cin + 1), k = pr / (Cin + 1);
    int pbi = (k / 4) * CB + (c / 4), e = (k % 4) * 4 + (c % 4);
    float a = 0.f;
    for (int g = 0; g < VG; ++g) a += red[(g * PB + pbi) * 16 + e];
    slab[pr] = a;
  }
}

__global__ void __launch_bounds__(NT) k_head_bwd_reduce(const float* __restrict__ ws, float* __restrict__ dw,
                                                        float* __restrict__ db, int n_slabs, int Cin, int K) {
  int npairs = K * (Cin + 1);
  for (int i = blockIdx.x * NT + threadIdx.x; i < npairs; i += gridDim.x * NT) {
    int c = i % (Cin + 1), k = i / (Cin + 1);
    float a = 0.f;
    for (int s = 0; s < n_slabs; ++s) a += ws[(size_t)s * npairs + i];
    if (c < Cin) dw[(size_t)k * Cin + c] = a;
    else db[k] = a;
  }
}

// ---- head, K <= 16 classes (every shipped config): compile-time class count ------------------------------------
// forward: one thread per voxel, grid.y = image (no 64-bit divisions in the loop), weights zero-padded to KT classes
// in LDS so the FMA loop carries no predicates.
template <typename T, int KT>
__global__ void __launch_bounds__(NT) k_head_fwd_k(const void* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ logits,
                                                   int64_t S, int Cin, int K) {
  constexpr int CPC = Elem<T>::CPC;
  CBIM_DYN_SMEM(smem);
  float* wL = (float*)smem;  // [Cin][KT]
  for (int i = threadIdx.x; i < Cin * KT; i += NT) {
    int k = i % KT, c = i / KT;
    wL[i] = k < K ? w[(size_t)k * Cin + c] : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.y;
  float bias[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) bias[k] = k < K ? b[k] : 0.f;
  const char* xn = (const char*)x + (size_t)n * S * Cin * Elem<T>::SIZE;
  float* ln = logits + (size_t)n * K * S;
  for (int64_t v = (int64_t)blockIdx.x * NT + threadIdx.x; v < S; v += (int64_t)gridDim.x * NT) {
    float out[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) out[k] = bias[k];
    const char* xr = xn + (size_t)v * Cin * Elem<T>::SIZE;
    for (int cb = 0; cb < Cin; cb += 4 * CPC) {   // 4 chunk loads in flight before any is used
      u32x4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (cb + u * CPC < Cin) raw[u] = *(const u32x4*)(xr + (size_t)(cb + u * CPC) * Elem<T>::SIZE);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c0 = cb + u * CPC;
        if (c0 < Cin) {
          float f[CPC];
          Elem<T>::unpack(raw[u], f);
#pragma unroll
          for (int j = 0; j < CPC; ++j) {
            const f32x4* wr = (const f32x4*)(wL + (size_t)(c0 + j) * KT);
#pragma unroll
            for (int k4 = 0; k4 < KT / 4; ++k4) {
              const f32x4 wv = wr[k4];
              out[4 * k4] = fmaf(f[j], wv.x, out[4 * k4]);
              out[4 * k4 + 1] = fmaf(f[j], wv.y, out[4 * k4 + 1]);
              out[4 * k4 + 2] = fmaf(f[j], wv.z, out[4 * k4 + 2]);
              out[4 * k4 + 3] = fmaf(f[j], wv.w, out[4 * k4 + 3]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) ln[(size_t)k * S + v] = out[k];
  }
}

// backward, one pass over dz and x: dx (optional), dw and db.  Wave w of the workgroup owns the 16-byte channel
// chunks w, w+4, .. (CHW of them) of every voxel row; a lane = one voxel.  dz[k][v] is a coalesced plane read (the four
// waves read the same values: L1), dx_chunk = sum_k dz[k] * W[k][chunk] (weights: LDS broadcasts), and the wave keeps
// dw[k][chunk] as per-lane partial sums over its voxels (16 x CPC x CHW registers), summed over lanes once at the
// end.  Slab per workgroup [K][Cin+1] (last column = db), fixed-order reduce by k_head_bwd_reduce.
template <typename T, int CHW>
__global__ void __launch_bounds__(NT, 1) k_head_bwd_k(const void* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ dz, void* __restrict__ dx,
                                                      float* __restrict__ ws, int64_t S, int Cin, int K,
                                                      int64_t vox_per_block) {
  // One wave per SIMD (16 x CPC x CHW accumulators + the wave's weight columns live in registers: ~300 VGPRs), so
  // memory latency is hidden by the wave itself: the loads of the next two voxel batches are in flight while a batch
  // is computed (ring of three register sets, loop unrolled by three).
  constexpr int CPC = Elem<T>::CPC, KT = 16, RING = 3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.y, cch = Cin / CPC;
  const int q0 = (int)blockIdx.z * 4 * CHW;      // rows of more than 4 * CHW chunks: blockIdx.z = chunk group (dz is re-read, it is small)
  float wr[CHW][KT][CPC];     // W[k][chunk channels] of this wave's chunks (zero rows for k >= K)
  float acc[CHW][KT][CPC];
  float dbv[KT];
#pragma unroll
  for (int h = 0; h < CHW; ++h) {
    const int q = q0 + wave + 4 * h;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        acc[h][k][j] = 0.f;
        wr[h][k][j] = (k < K && q < cch) ? w[(size_t)k * Cin + q * CPC + j] : 0.f;
      }
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) dbv[k] = 0.f;
  const int64_t v_begin = (int64_t)blockIdx.x * vox_per_block;
  int64_t v_end = v_begin + vox_per_block;
  if (v_end > S) v_end = S;
  const float* dzn = dz + (size_t)n * K * S;
  const char* xn = (const char*)x + (size_t)n * S * Cin * Elem<T>::SIZE;
  char* dxn = dx ? (char*)dx + (size_t)n * S * Cin * Elem<T>::SIZE : nullptr;
  float g[RING][KT];
  u32x4 raw[RING][CHW];
  auto load = [&](int slot, int64_t v) {
    if (v < v_end) {
#pragma unroll
      for (int k = 0; k < KT; ++k) g[slot][k] = k < K ? dzn[(size_t)k * S + v] : 0.f;
#pragma unroll
      for (int h = 0; h < CHW; ++h) {
        const int q = q0 + wave + 4 * h;
        if (q < cch) raw[slot][h] = *(const u32x4*)(xn + ((size_t)v * Cin + (size_t)q * CPC) * Elem<T>::SIZE);
      }
    }
  };
  auto compute = [&](int slot, int64_t v) {
    if (v >= v_end) return;
#pragma unroll
    for (int k = 0; k < KT; ++k) dbv[k] += g[slot][k];
#pragma unroll
    for (int h = 0; h < CHW; ++h) {
      const int q = q0 + wave + 4 * h;
      if (q < cch) {
        float xf[CPC];
        Elem<T>::unpack(raw[slot][h], xf);
        if (dxn) {
          float f[CPC];
#pragma unroll
          for (int j = 0; j < CPC; ++j) f[j] = 0.f;
#pragma unroll
          for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int j = 0; j < CPC; ++j) f[j] = fmaf(g[slot][k], wr[h][k][j], f[j]);
          *(u32x4*)(dxn + ((size_t)v * Cin + (size_t)q * CPC) * Elem<T>::SIZE) = Elem<T>::pack(f);
        }
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int j = 0; j < CPC; ++j) acc[h][k][j] = fmaf(g[slot][k], xf[j], acc[h][k][j]);
      }
    }
  };
  int64_t v = v_begin + lane;
  load(0, v);
  load(1, v + 64);
  for (; v < v_end; v += 3 * 64) {
    load(2, v + 128);
    compute(0, v);
    load(0, v + 192);
    compute(1, v + 64);
    load(1, v + 256);
    compute(2, v + 128);
  }
  // lanes -> one value (xor butterfly: every lane ends with the same, order-fixed sum)
  const int npairs = K * (Cin + 1);
  float* slab = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * npairs;
#pragma unroll
  for (int h = 0; h < CHW; ++h) {
    const int q = q0 + wave + 4 * h;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        const float a = wave_sum(acc[h][k][j]);
        if (lane == 0 && q < cch && k < K) slab[(size_t)k * (Cin + 1) + q * CPC + j] = a;
      }
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const float a = wave_sum(dbv[k]);
    if (blockIdx.z == 0 && wave == 0 && lane == 0 && k < K) slab[(size_t)k * (Cin + 1) + Cin] = a;
  }
}

// ---- head backward on the matrix cores: bf16 rows of Cin = 32 * CP <= 128 channels, K <= 16 classes, S % 32 == 0 -----------
// k_head_bwd_k does 2 x 16 x 8 fp32 FMAs per voxel chunk with ONE wave per SIMD (its 256 accumulator / weight registers) and
// runs 4x above its VALU bound (211 us at 128^3 x 32, the tensors move in ~70 us).  Here a wave takes 32 voxels per step:
//   dw[k][c] += dz[k][v] x[v][c]   v_mfma_f32_16x16x32_bf16 contracting over the 32 VOXELS: A = dz rows straight from the
//                                  fp32 planes (two 16-byte loads per lane, split into bf16 hi + lo so dw keeps fp32 accuracy),
//                                  B = the x rows through the LDS transpose read (as k_wgrad_r32 does);
//   dx[v][c]  = sum_k dz[k][v] W[k][c]  contracting over the classes (padded to 32): A = W^T with the rows of a tile pair
//                                  permuted so that a lane ends up with 8 consecutive channels of one voxel (one 16-byte
//                                  store), B = dz[k][v] transposed through a 1 KiB LDS tile of the bf16 planes.
// Registers ~64, 8 waves per workgroup, steps are wave-private (no workgroup barrier inside the loop); the next step's
// global loads are in flight while a step is computed.  Slab per workgroup [K][Cin+1] as before (k_head_bwd_reduce4).
static constexpr int HM_NW = 8;     // waves per workgroup
// HC = 16-channel tiles of a row (Cin = 16 HC: round 5 — the 48-channel head of SwinUNETR ran the vector-ALU kernel because this one
// counted whole 32-channel units); CP = 32-channel units of the dx product (its lane layout is 4 groups x 8 consecutive channels):
// the half unit of an odd HC has zero weight rows and no stores
template <int HC>
__global__ void __launch_bounds__(HM_NW * 64) k_head_bwd_mfma(const void* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ dz, void* __restrict__ dx,
                                                              float* __restrict__ ws, int64_t S, int K, int64_t vox_per_block) {
  constexpr int CP = (HC + 1) / 2, Cin = 16 * HC, ROWB = Cin * 2, XT = 32 * ROWB, DZT = 16 * 64, WT = XT + DZT, NX = HC;
  CBIM_DYN_SMEM(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  unsigned char* xs = smem + wave * WT;
  unsigned char* dzs = xs + XT;
  const int n = blockIdx.y;
  u32x4 wa[CP][2];
#pragma unroll
  for (int p = 0; p < CP; ++p)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c = 32 * p + 8 * (li >> 2) + 4 * t + (li & 3);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (8 * g + j < K && c < Cin) ? w[(size_t)(8 * g + j) * Cin + c] : 0.f;
      wa[p][t] = Elem<bf16_tag>::pack(f);
    }
  f32x4 acc[HC];
#pragma unroll
  for (int ct = 0; ct < HC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbv = 0.f;
  const int64_t v_begin = (int64_t)blockIdx.x * vox_per_block;
  int64_t v_end = v_begin + vox_per_block;
  if (v_end > S) v_end = S;
  const float* dzk = dz + ((size_t)n * K + (li < K ? li : 0)) * S + 8 * g;     // this lane's class plane, at its 8 voxels
  const unsigned char* xn = (const unsigned char*)x + (size_t)n * S * ROWB;
  unsigned char* dxn = dx ? (unsigned char*)dx + (size_t)n * S * ROWB : nullptr;
  u32x4 xr[2][NX];
  f32x4 dr[2][2];
  auto load = [&](int slot, int64_t v) {
    if (v >= v_end) return;
#pragma unroll
    for (int u = 0; u < NX; ++u) xr[slot][u] = *(const u32x4*)(xn + (size_t)v * ROWB + (size_t)(lane + 64 * u) * 16);
    if (li < K) {
      dr[slot][0] = *(const f32x4*)(dzk + v);
      dr[slot][1] = *(const f32x4*)(dzk + v + 4);
    } else {
      dr[slot][0] = dr[slot][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto step = [&](int slot, int64_t v) {
    float d[8], lo[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = dr[slot][0][j]; d[4 + j] = dr[slot][1][j]; }
    dbv += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    const u32x4 ahi = Elem<bf16_tag>::pack(d);
    {
      float h[8];
      Elem<bf16_tag>::unpack(ahi, h);
#pragma unroll
      for (int j = 0; j < 8; ++j) lo[j] = d[j] - h[j];
    }
    const u32x4 alo = Elem<bf16_tag>::pack(lo);
    hd_wave_sync();                                                      // the previous step's LDS reads are done
#pragma unroll
    for (int u = 0; u < NX; ++u) *(u32x4*)(xs + (size_t)(lane + 64 * u) * 16) = xr[slot][u];
    *(u32x4*)(dzs + li * 64 + g * 16) = ahi;
    hd_wave_sync();
    // dw: contraction over the 32 voxels
#pragma unroll
    for (int ct = 0; ct < HC; ++ct) {
      const unsigned char* a = xs + (8 * g + (li >> 2)) * ROWB + ct * 32 + (li & 3) * 8;
      const u32x2 b0 = hd_tr16_b64(a), b1 = hd_tr16_b64(a + 4 * ROWB);
      const u32x4 bf = u32x4{b0.x, b0.y, b1.x, b1.y};
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ahi), __builtin_bit_cast(bf16x8, bf), acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, alo), __builtin_bit_cast(bf16x8, bf), acc[ct], 0, 0, 0);
    }
    // dx: contraction over the classes (k >= 16: zero operand rows)
    if (dxn) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned char* a = dzs + ((8 * g + (li >> 2)) & 15) * 64 + h * 32 + (li & 3) * 8;
        u32x2 e0 = hd_tr16_b64(a), e1 = hd_tr16_b64(a + 4 * 64);
        if (g >= 2) { e0 = u32x2{0u, 0u}; e1 = u32x2{0u, 0u}; }
        const u32x4 ef = u32x4{e0.x, e0.y, e1.x, e1.y};
#pragma unroll
        for (int p = 0; p < CP; ++p) {
          const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[p][0]), __builtin_bit_cast(bf16x8, ef), z, 0, 0, 0);
          const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[p][1]), __builtin_bit_cast(bf16x8, ef), z, 0, 0, 0);
          const float f[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
          if (32 * p + 8 * g < Cin)      // (the half unit of an odd HC)
            *(u32x4*)(dxn + (size_t)(v + 16 * h + li) * ROWB + (size_t)(32 * p + 8 * g) * 2) = Elem<bf16_tag>::pack(f);
        }
      }
    }
  };
  int64_t v = v_begin + wave * 32;
  load(0, v);
  for (; v < v_end; v += 2 * HM_NW * 32) {
    load(1, v + HM_NW * 32);
    step(0, v);
    load(0, v + 2 * HM_NW * 32);
    if (v + HM_NW * 32 < v_end) step(1, v + HM_NW * 32);
  }
  // the 8 waves' partial dw / db -> the workgroup's slab, fixed order
  __syncthreads();
  float* red = (float*)smem;                         // [wave][16][Cin]
  float* rdb = red + HM_NW * 16 * Cin;               // [wave][16]
#pragma unroll
  for (int ct = 0; ct < HC; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * 16 + 4 * g + i) * Cin + 16 * ct + li] = acc[ct][i];
  dbv += __shfl_xor(dbv, 16);
  dbv += __shfl_xor(dbv, 32);
  if (g == 0) rdb[wave * 16 + li] = dbv;
  __syncthreads();
  const int npairs = K * (Cin + 1);
  float* slab = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * npairs;
  for (int o = tid; o < npairs; o += HM_NW * 64) {
    const int k = o / (Cin + 1), c = o % (Cin + 1);
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < HM_NW; ++q) a += c < Cin ? red[(q * 16 + k) * Cin + c] : rdb[q * 16 + k];
    slab[o] = a;
  }
}
static_assert(HM_NW * (32 * 256 + 16 * 64) <= 160 * 1024, "LDS");
template <int HC> static bool hm_raise_lds() {       // more than 64 KiB of dynamic LDS needs the attribute (HC >= 7)
#ifndef CBIM_EMU
  return hipFuncSetAttribute((const void*)k_head_bwd_mfma<HC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
#else
  return true;
#endif
}

// slabs -> dw, db: 64 outputs x 4 slab phases per workgroup, fixed order
__global__ void __launch_bounds__(NT) k_head_bwd_reduce4(const float* __restrict__ ws, float* __restrict__ dw,
                                                         float* __restrict__ db, int n_slabs, int Cin, int K) {
  __shared__ float red[4][64];
  const int npairs = K * (Cin + 1);
  const int o = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float a0 = 0.f, a1 = 0.f;
  if (i < npairs) {
    int s = ph;
    for (; s + 4 < n_slabs; s += 8) { a0 += ws[(size_t)s * npairs + i]; a1 += ws[(size_t)(s + 4) * npairs + i]; }
    if (s < n_slabs) a0 += ws[(size_t)s * npairs + i];
  }
  red[ph][o] = a0 + a1;
  __syncthreads();
  if (ph == 0 && i < npairs) {
    const float a = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    const int c = i % (Cin + 1), k = i / (Cin + 1);
    if (c < Cin) dw[(size_t)k * Cin + c] = a;
    else db[k] = a;
  }
}

static inline int grid_for(int64_t items) {
  int64_t b = (items + NT - 1) / NT;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

struct StemCfg { int tiles_d, tiles_h, tiles_w, strips_per_n, tiles_per_strip; };
static StemCfg stem_cfg(int N, int Do, int Ho, int Wo) {
  StemCfg c;
  c.tiles_d = (Do + 3) / 4; c.tiles_h = (Ho + 7) / 8; c.tiles_w = (Wo + 7) / 8;
  int tiles = c.tiles_d * c.tiles_h * c.tiles_w;
  int want = 1024 / N;
  if (want < 1) want = 1;
  if (want > tiles) want = tiles;
  c.tiles_per_strip = (tiles + want - 1) / want;
  c.strips_per_n = (tiles + c.tiles_per_strip - 1) / c.tiles_per_strip;
  return c;
}

}  // namespace cbim

using namespace cbim;

static int g_stem_mfma = 1;
/* process-wide switch (tests, A/B): 0 = the VALU stem kernels also where the matrix-core ones apply; returns the old value */
extern "C" int cbim_stem_mfma_enable(int on) {
  const int old = g_stem_mfma;
  g_stem_mfma = on ? 1 : 0;
  return old;
}

static int stem_check(int dtype, int Cin, int Cout, int kD, int kH, int kW) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype");
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(Cout % cpc == 0, CBIM_EUNSUPPORTED, "stem Cout %d is not a multiple of %d", Cout, cpc);
  CBIM_CHECK(Cin >= 1 && Cin <= 16, CBIM_EUNSUPPORTED, "stem Cin %d unsupported (1..16)", Cin);
  CBIM_CHECK(kD <= 5 && kH <= 5 && kW <= 5 && kD * kH * kW <= 128, CBIM_EUNSUPPORTED, "stem kernel extent > 5 unsupported");
  return 0;
}

static void stem_fill(StemParams& p, int N, int Cin, int Di, int Hi, int Wi, int Cout, int kD, int kH, int kW,
                      int pD, int pH, int pW, int Do, int Ho, int Wo) {
  p.N = N; p.Cin = Cin; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Cout = Cout; p.kD = kD; p.kH = kH; p.kW = kW;
  p.pD = pD; p.pH = pH; p.pW = pW; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
  StemCfg c = stem_cfg(N, Do, Ho, Wo);
  p.tiles_d = c.tiles_d; p.tiles_h = c.tiles_h; p.tiles_w = c.tiles_w;
  p.hD = 4 + kD - 1; p.hH = 8 + kH - 1; p.hW = 8 + kW - 1; p.taps = kD * kH * kW;
  p.strips_per_n = c.strips_per_n; p.tiles_per_strip = c.tiles_per_strip;
}

extern "C" int cbim_stem_conv_fwd(int dtype_out, const float* x, const float* w, void* y, int N, int Cin,
                                  int Di, int Hi, int Wi, int Cout, int kD, int kH, int kW, int pD, int pH,
                                  int pW, int Do, int Ho, int Wo, void* stream) {
  if (int e = stem_check(dtype_out, Cin, Cout, kD, kH, kW)) return e;
  StemParams p;
  p.x = x; p.w = w; p.y = y; p.dy = nullptr; p.ws = nullptr;
  stem_fill(p, N, Cin, Di, Hi, Wi, Cout, kD, kH, kW, pD, pH, pW, Do, Ho, Wo);
  size_t smem = ((size_t)Cin * p.hD * p.hH * p.hW + (size_t)p.taps * Cin * Cout) * sizeof(float);
  CBIM_CHECK(smem <= 64 * 1024, CBIM_EUNSUPPORTED, "stem needs %zu B of LDS", smem);
  dim3 grid((unsigned)(N * p.tiles_d * p.tiles_h * p.tiles_w));
  hipStream_t st = (hipStream_t)stream;
  if (g_stem_mfma && dtype_out == CBIM_BF16 && Cin == 1 && Cout % 32 == 0 && Cout <= 128 && p.taps <= MAXTAPS &&
      kD <= 3 && kH <= 3 && kW <= 3) {   // matrix-core form
    const size_t sm = (size_t)3 * NT * sizeof(unsigned);
    dim3 pg(grid.x < 2048u ? grid.x : 2048u);             // persistent workgroups: the weight fragments are built once
    switch (Cout / 32) {
      case 1: CBIM_LAUNCH((k_stem_fwd_mfma<1>), pg, dim3(NT), sm, st, p); break;
      case 2: CBIM_LAUNCH((k_stem_fwd_mfma<2>), pg, dim3(NT), sm, st, p); break;
      case 3: CBIM_LAUNCH((k_stem_fwd_mfma<3>), pg, dim3(NT), sm, st, p); break;
      default: CBIM_LAUNCH((k_stem_fwd_mfma<4>), pg, dim3(NT), sm, st, p); break;
    }
    return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
  }
  if (dtype_out == CBIM_BF16) CBIM_LAUNCH((k_stem_fwd<bf16_tag>), grid, dim3(NT), smem, st, p);
  else CBIM_LAUNCH((k_stem_fwd<float>), grid, dim3(NT), smem, st, p);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" size_t cbim_stem_conv_wgrad_workspace(int N, int Cin, int Cout, int kD, int kH, int kW, int Do,
                                                 int Ho, int Wo) {
  StemCfg c = stem_cfg(N, Do, Ho, Wo);
  return (size_t)N * c.strips_per_n * Cin * kD * kH * kW * Cout * sizeof(float);
}

extern "C" int cbim_stem_conv_wgrad(int dtype, const float* x, const void* dy, float* dw, int N, int Cin,
                                    int Di, int Hi, int Wi, int Cout, int kD, int kH, int kW, int pD, int pH,
                                    int pW, int Do, int Ho, int Wo, void* workspace, size_t ws_bytes,
                                    void* stream) {
  if (int e = stem_check(dtype, Cin, Cout, kD, kH, kW)) return e;
  size_t need = cbim_stem_conv_wgrad_workspace(N, Cin, Cout, kD, kH, kW, Do, Ho, Wo);
  CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "stem wgrad workspace %zu < %zu", ws_bytes, need);
  StemParams p;
  p.x = x; p.w = nullptr; p.y = nullptr; p.dy = dy; p.ws = (float*)workspace;
  stem_fill(p, N, Cin, Di, Hi, Wi, Cout, kD, kH, kW, pD, pH, pW, Do, Ho, Wo);
  if (kD > 3 || kH > 3 || kW > 3) {   // beyond the 3x3x3 lattice: one thread per tap
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(N * p.strips_per_n), (unsigned)((Cout + 15) / 16));
    const size_t sm = ((size_t)((p.hD * p.hH * p.hW + 3) & ~3) + 256 * 16) * sizeof(float);
    if (dtype == CBIM_BF16) CBIM_LAUNCH((k_stem_wgrad_taps<bf16_tag>), grid, dim3(NT), sm, st, p);
    else CBIM_LAUNCH((k_stem_wgrad_taps<float>), grid, dim3(NT), sm, st, p);
    if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
    const int total = Cin * p.taps * Cout;
    CBIM_LAUNCH(k_stem_wgrad_reduce, dim3((total + 15) / 16), dim3(NT), 0, st, (const float*)workspace, dw,
                N * p.strips_per_n, Cin, p.taps, Cout);
    return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
  }
  if (g_stem_mfma && dtype == CBIM_BF16 && Cin == 1 && Cout % 32 == 0 && Cout <= 128) {   // matrix-core form
    const size_t tile = (size_t)p.hD * p.hH * 64 + (size_t)256 * Cout * 2, red = (size_t)4 * 32 * Cout * sizeof(float);
    const size_t sm = tile > red ? tile : red;
    dim3 grid((unsigned)(N * p.strips_per_n));
    hipStream_t st = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
#ifndef CBIM_EMU
      hipError_t e1 = hipFuncSetAttribute((const void*)k_stem_wgrad_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      CBIM_CHECK(e1 == hipSuccess, CBIM_ELAUNCH, "stem wgrad: cannot raise the dynamic LDS limit");
#endif
      attr_done = true;
    }
    switch (Cout / 32) {
      case 1: CBIM_LAUNCH((k_stem_wgrad_mfma<1>), grid, dim3(NT), sm, st, p); break;
      case 2: CBIM_LAUNCH((k_stem_wgrad_mfma<2>), grid, dim3(NT), sm, st, p); break;
      case 3: CBIM_LAUNCH((k_stem_wgrad_mfma<3>), grid, dim3(NT), sm, st, p); break;
      default: CBIM_LAUNCH((k_stem_wgrad_mfma<4>), grid, dim3(NT), sm, st, p); break;
    }
    if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
    const int total = p.taps * Cout;
    CBIM_LAUNCH(k_stem_wgrad_reduce, dim3((total + 15) / 16), dim3(NT), 0, st, (const float*)workspace, dw,
                N * p.strips_per_n, 1, p.taps, Cout);
    return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
  }
  const int cg = Cin >= 2 ? 4 : 1;   // input channels sharing one pass over dy
  size_t smem = ((((size_t)p.hD * p.hH * p.hW + 3) & ~(size_t)3) * cg + (size_t)(NT / 32) * MAXTAPS * 32) * sizeof(float);
  CBIM_CHECK(smem <= 64 * 1024, CBIM_EUNSUPPORTED, "stem wgrad needs %zu B of LDS", smem);
  dim3 grid((unsigned)(N * p.strips_per_n), (unsigned)((Cout + 31) / 32));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CBIM_BF16) {
    if (cg == 4) CBIM_LAUNCH((k_stem_wgrad<bf16_tag, 4>), grid, dim3(NT), smem, st, p);
    else CBIM_LAUNCH((k_stem_wgrad<bf16_tag, 1>), grid, dim3(NT), smem, st, p);
  } else {
    if (cg == 4) CBIM_LAUNCH((k_stem_wgrad<float, 4>), grid, dim3(NT), smem, st, p);
    else CBIM_LAUNCH((k_stem_wgrad<float, 1>), grid, dim3(NT), smem, st, p);
  }
  int total = Cin * p.taps * Cout;
  CBIM_LAUNCH(k_stem_wgrad_reduce, dim3((total + 15) / 16), dim3(NT), 0, st, (const float*)workspace, dw,
              N * p.strips_per_n, Cin, p.taps, Cout);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

static int head_check(int dtype, int Cin, int K) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype");
  int cpc = dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(Cin % cpc == 0 && Cin <= 512, CBIM_EUNSUPPORTED, "head Cin %d unsupported", Cin);
  CBIM_CHECK(K >= 1 && ((K + 3) / 4) * ((Cin + 4) / 4) <= NT, CBIM_EUNSUPPORTED, "head K=%d x Cin=%d too large", K, Cin);
  return 0;
}

extern "C" int cbim_head_fwd(int dtype, const void* x, const float* w, const float* b, float* logits, int N,
                             int64_t S, int Cin, int K, void* stream) {
  if (int e = head_check(dtype, Cin, K)) return e;
  int64_t total = (int64_t)N * S;
  size_t smem = (size_t)Cin * K * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (K <= 16 && (size_t)Cin * 16 * sizeof(float) <= 64 * 1024) {   // compile-time class count (k_head_fwd_k)
    const size_t sm = (size_t)Cin * 16 * sizeof(float);
    int64_t bx = (S + NT - 1) / NT;
    const int64_t cap = 2048 / N > 1 ? 2048 / N : 1;
    if (bx > cap) bx = cap;
    dim3 grid((unsigned)bx, (unsigned)N);
    if (dtype == CBIM_BF16) CBIM_LAUNCH((k_head_fwd_k<bf16_tag, 16>), grid, dim3(NT), sm, st, x, w, b, logits, S, Cin, K);
    else CBIM_LAUNCH((k_head_fwd_k<float, 16>), grid, dim3(NT), sm, st, x, w, b, logits, S, Cin, K);
    return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
  }
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_head_fwd<bf16_tag>), dim3(grid_for(total)), dim3(NT), smem, st, x, w, b, logits, S, Cin, K, total);
  else
    CBIM_LAUNCH((k_head_fwd<float>), dim3(grid_for(total)), dim3(NT), smem, st, x, w, b, logits, S, Cin, K, total);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

static int g_head_mfma = 1;
/* process-wide switch (tests, A/B): 0 = the VALU head backward also where the matrix-core kernel applies; returns the old value */
extern "C" int cbim_head_mfma_enable(int on) {
  const int old = g_head_mfma;
  g_head_mfma = on ? 1 : 0;
  return old;
}

static int head_bwd_blocks(int64_t S) {
  int64_t b = (S + 4095) / 4096;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t cbim_head_bwd_workspace(int64_t S, int N, int Cin, int K) {
  return (size_t)N * head_bwd_blocks(S) * K * (Cin + 1) * sizeof(float);
}

extern "C" int cbim_head_bwd(int dtype, const void* x, const float* w, const float* dlogits, void* dx,
                             float* dw, float* db, int N, int64_t S, int Cin, int K, void* workspace,
                             size_t ws_bytes, void* stream) {
  if (int e = head_check(dtype, Cin, K)) return e;
  size_t need = cbim_head_bwd_workspace(S, N, Cin, K);
  CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "head bwd workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  int64_t total = (int64_t)N * S;
  size_t smem = (size_t)Cin * K * sizeof(float);
  {
    // one fused pass (dx, dw, db) when the channel chunks of a row fit the 4 waves x CHW layout of k_head_bwd_k
    const int cpc = dtype == CBIM_BF16 ? 8 : 4, cch = Cin / cpc;
    // (bf16: one chunk per wave — two would need 2 x 16 x 8 accumulators + as many weight registers per lane: spills;
    //  wider rows, e.g. MedFormer's aux head on 128 channels, run as blockIdx.z groups of 4 chunks)
    if (K <= 16 && g_head_mfma && dtype == CBIM_BF16 && Cin % 16 == 0 && Cin <= 128 && S % 32 == 0) {
      // matrix-core form (k_head_bwd_mfma): workgroups of 8 waves x 32-voxel steps, two per CU at full size
      int nb = head_bwd_blocks(S);
      int64_t vpb = (S + nb - 1) / nb;
      vpb = (vpb + HM_NW * 32 - 1) / (HM_NW * 32) * (HM_NW * 32);
      nb = (int)((S + vpb - 1) / vpb);
      const int HC = Cin / 16;
      const size_t sm = (size_t)HM_NW * (32 * Cin * 2 + 16 * 64);
      dim3 grid((unsigned)nb, (unsigned)N);
      float* wsf = (float*)workspace;
#define HM_LAUNCH(H)                                                                                                       \
      case H: {                                                                                                            \
        static bool attr_done = false;                                                                                     \
        if (!attr_done && sm > 64 * 1024) {                                                                                \
          CBIM_CHECK(hm_raise_lds<H>(), CBIM_ELAUNCH, "head bwd: cannot raise the dynamic LDS limit");                     \
          attr_done = true;                                                                                                \
        }                                                                                                                  \
        CBIM_LAUNCH((k_head_bwd_mfma<H>), grid, dim3(HM_NW * 64), sm, st, x, w, dlogits, dx, wsf, S, K, vpb);              \
      } break;
      switch (HC) {
        HM_LAUNCH(1) HM_LAUNCH(2) HM_LAUNCH(3) HM_LAUNCH(4) HM_LAUNCH(5) HM_LAUNCH(6) HM_LAUNCH(7)
        default: HM_LAUNCH(8)
      }
#undef HM_LAUNCH
      if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
      const int npairs = K * (Cin + 1);
      CBIM_LAUNCH(k_head_bwd_reduce4, dim3((npairs + 63) / 64), dim3(NT), 0, st, (const float*)workspace, dw, db,
                  N * nb, Cin, K);
      return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
    }
    if (K <= 16) {
      int nb = head_bwd_blocks(S);      // (the workspace is sized for head_bwd_blocks(S) slabs)
      if (nb > 256) nb = 256;           // one workgroup per CU: fewer slabs for the reduce
      int64_t vpb = (S + nb - 1) / nb;
      vpb = (vpb + 63) / 64 * 64;
      const size_t sm = 0;
      const int per = dtype == CBIM_BF16 ? 4 : 8;     // chunks per workgroup: 4 waves x CHW
      dim3 grid((unsigned)nb, (unsigned)N, (unsigned)((cch + per - 1) / per));
      float* wsf = (float*)workspace;
      if (dtype == CBIM_BF16) CBIM_LAUNCH((k_head_bwd_k<bf16_tag, 1>), grid, dim3(NT), sm, st, x, w, dlogits, dx, wsf, S, Cin, K, vpb);
      else if (cch <= 4) CBIM_LAUNCH((k_head_bwd_k<float, 1>), dim3((unsigned)nb, (unsigned)N, 1), dim3(NT), sm, st, x, w, dlogits, dx, wsf, S, Cin, K, vpb);
      else CBIM_LAUNCH((k_head_bwd_k<float, 2>), grid, dim3(NT), sm, st, x, w, dlogits, dx, wsf, S, Cin, K, vpb);
      if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
      const int npairs = K * (Cin + 1);
      CBIM_LAUNCH(k_head_bwd_reduce4, dim3((npairs + 63) / 64), dim3(NT), 0, st, (const float*)workspace, dw, db,
                  N * nb, Cin, K);
      return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
    }
  }
  if (dx) {
    if (dtype == CBIM_BF16)
      CBIM_LAUNCH((k_head_bwd_dx<bf16_tag>), dim3(grid_for(total)), dim3(NT), smem, st, w, dlogits, dx, S, Cin, K, total);
    else
      CBIM_LAUNCH((k_head_bwd_dx<float>), dim3(grid_for(total)), dim3(NT), smem, st, w, dlogits, dx, S, Cin, K, total);
  }
  int nb = head_bwd_blocks(S);
  int64_t vpb = (S + nb - 1) / nb;
  vpb = (vpb + HB_TILE - 1) / HB_TILE * HB_TILE;
  const int XS = (Cin + 1 + 3) & ~3, ZS = (K + 3) & ~3;
  const int PB = (ZS / 4) * (XS / 4);
  size_t smem2 = ((size_t)HB_TILE * XS + (size_t)HB_TILE * ZS + (size_t)(NT / PB) * PB * 16) * sizeof(float);
  CBIM_CHECK(smem2 <= 64 * 1024, CBIM_EUNSUPPORTED, "head bwd needs %zu B LDS", smem2);
  dim3 grid((unsigned)nb, (unsigned)N);
  if (dtype == CBIM_BF16)
    CBIM_LAUNCH((k_head_bwd_dw<bf16_tag>), grid, dim3(NT), smem2, st, x, dlogits, (float*)workspace, S, Cin, K, vpb);
  else
    CBIM_LAUNCH((k_head_bwd_dw<float>), grid, dim3(NT), smem2, st, x, dlogits, (float*)workspace, S, Cin, K, vpb);
  int npairs = K * (Cin + 1);
  CBIM_LAUNCH(k_head_bwd_reduce, dim3((npairs + NT - 1) / NT), dim3(NT), 0, st, (const float*)workspace, dw, db,
              N * nb, Cin, K);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(stem_head)
