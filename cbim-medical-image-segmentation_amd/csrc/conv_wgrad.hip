// conv_wgrad.hip — weight gradient of the 3-D convolution on the CDNA4 matrix cores.
//
//   dw[co][ci][tap] = sum_{n,v} dy[v][co] * a[v + tap - p][ci],   a = act(InstanceNorm(x)) (zero padded)
//
//   GEMM view per tap: M = co (32), N = ci (32), K = voxels.  The contraction runs over VOXELS while
//   both tensors are channels-last, so each MFMA operand is a transposed read of a [voxel][channel]
//   LDS tile: bf16 uses ds_read_b64_tr_b16 (the gfx950 LDS transpose read: a 16-lane group fetches a
//   [4 voxels][16 channels] block and every lane receives one channel's 4 voxels), f32 needs no
//   transpose because the 32x32x2 MFMA takes a single scalar per lane.
//   Workgroup (256 threads, 4 waves, 54 KiB LDS -> 2 per CU) owns a (32 co x 32 ci) block and a STRIP
//   of spatial tiles; per tile the dy tile and the transformed input halo are staged once into LDS;
//   the taps are dealt round-robin to the 4 waves (TPW accumulators of 16 VGPRs per wave), the dy
//   fragment is re-used by all of a wave's taps; the fragments of voxel step k+1 are fetched into a
//   second register set before the MFMAs of step k issue.  Accumulators live across the whole strip
//   and are written once to a per-strip fp32 slab; a second kernel adds the slabs in fixed order
//   (deterministic, no atomics) into the natural nn.Conv3d [Cout][Cin][kD][kH][kW] fp32 gradient.
//
// Replaces aten::convolution_backward(weight) for nn.Conv3d in ConvNormAct
// (/root/reference/model/dim3/conv_layers.py:29-38).
#include "cbim_common.h"
#include <stdlib.h>
#include "conv_wgrad_r32.h"
#include "conv_r32.h"

namespace cbim {

static constexpr int NT = 256;

struct WgradParams {
  const void* x; int64_t x_stride; const float* in_stats;
  const void* dy; int64_t dy_stride;
  const void* dy2; int64_t dy2_stride; int cout_split;   // couts >= cout_split come from dy2
  float* ws;
  int N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout;
  int kD, kH, kW, pD, pH, pW, act;
  int tD, tH, lgH, tiles_d, tiles_h, tiles_w, hD, hH, hW, taps;
  int strips_per_n, tiles_per_strip, ci_blocks, Cout_pad, Cin_pad;
  unsigned mHW, mW;
#ifdef CBIM_IGEMM_PROF
  unsigned long long* prof;        // tools/ only: per-wave cycle totals of the tile-loop phases
#endif
};

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// 4 consecutive-voxel bf16 of one channel via the LDS transpose read (per-lane address of 4 bf16).
__device__ __forceinline__ u32x2 lds_tr16_b64(const unsigned char* p) {
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

// 24-bit multiply-add (full rate): row and byte offsets inside one tile box
__device__ __forceinline__ unsigned wmad24(unsigned a, unsigned b, unsigned c) {
#ifdef CBIM_EMU
  return a * b + c;
#else
  return __umul24(a, b) + c;
#endif
}

#ifdef CBIM_EMU
#define WG_SCHED_FENCE() ((void)0)
#else
#define WG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

template <int ACT> __device__ __forceinline__ float wg_actf(float x, int rt) {
  if (ACT == CBIM_ACT_RELU) return x > 0.f ? x : 0.f;
  if (ACT == CBIM_ACT_LRELU) return x > 0.f ? x : 0.01f * x;
  if (ACT == CBIM_ACT_NONE) return x;
  return act_fwd(x, rt);
}

template <int TPW> struct WFrag { u32x4 a; u32x4 b[TPW]; };   // bf16: 8 voxels x 1 channel per operand
template <int TPW> struct WFragF { float a; float b[TPW]; };  // f32 : 1 voxel per operand

// K3T: 3x3 taps per plane on the 4x8x8 tile (halo rows 10x10, bf16): every fragment address of the voxel loop is
// a per-tile lane constant + an immediate, the 16 k-steps are unrolled and carry no vector-ALU address arithmetic
template <typename T, int TPW, int ACT, bool K3T>
__global__ void __launch_bounds__(NT, 2) k_conv_wgrad(WgradParams p) {
  constexpr int CPC = Elem<T>::CPC;
  constexpr int ES = Elem<T>::SIZE;
  constexpr int ROWB = 32 * ES;        // LDS row = 32 channels
  constexpr int SLOTS = ROWB / 16;
  constexpr bool IS_BF16 = (ES == 2);
  CBIM_DYN_SMEM(smem);
  const int BMv = p.tD * p.tH * 8;
  const int hV = p.hD * p.hH * p.hW;
  const unsigned aL = (unsigned)BMv * ROWB;   // halo region offset; dy tile at 0
  const int hHW = p.hH * p.hW;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const int n = blockIdx.x / p.strips_per_n, strip = blockIdx.x % p.strips_per_n;
  const int cb = blockIdx.y / p.ci_blocks, ib = blockIdx.y % p.ci_blocks;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int t_begin = strip * p.tiles_per_strip;
  int t_end = t_begin + p.tiles_per_strip;
  if (t_end > tiles_per_n) t_end = tiles_per_n;

  const bool ksplit = TPW == 1 && p.taps == 1;   // wave-uniform
  // taps of this wave: tap = wave + 4*tl (clamped: out-of-range slots redo the last tap and are dropped)
  unsigned tapoff[TPW];
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl) {
    int tap = wave + 4 * tl;
    if (tap > p.taps - 1) tap = p.taps - 1;
    int kw = tap % p.kW, r = tap / p.kW;
    int kh = r % p.kH, kd = r / p.kH;
    tapoff[tl] = (unsigned)((kd * p.hH + kh) * p.hW + kw) * ROWB;
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;

  const int my_slot = tid & (SLOTS - 1);
  const int ci0 = ib * 32 + my_slot * CPC;   // staged input channel chunk of this thread
  const int co0 = cb * 32 + my_slot * CPC;   // staged dy channel chunk of this thread
  const bool dy_from2 = p.dy2 != nullptr && cb * 32 >= p.cout_split;   // block-uniform
  float mean[CPC], rstd[CPC];
  if (p.in_stats && ci0 < p.Cin) {
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      mean[j] = p.in_stats[((size_t)n * p.Cin + ci0 + j) * 2];
      rstd[j] = p.in_stats[((size_t)n * p.Cin + ci0 + j) * 2 + 1];
    }
  }

  // per-lane constants of the fragment reads
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const unsigned colb = (unsigned)(16 * g16 + 4 * (i16 & 3)) * 2;   // bf16: byte offset of the 4-channel group
  const int mq = 8 * half + (i16 >> 2);                             // bf16: voxel of the lane inside a k-step

  auto halo_row = [&](int m) -> unsigned {   // LDS row (bytes) of tile voxel m at tap (0,0,0)
    unsigned tw = m & 7, th = (m >> 3) & (p.tH - 1), td = m >> (3 + p.lgH);
    return wmad24(wmad24(td, (unsigned)p.hH, th), (unsigned)p.hW, tw) * ROWB;
  };
  auto fetch16 = [&](WFrag<TPW>& f, int ks) {
    int m0 = ks * 16 + mq, m1 = m0 + 4;
    u32x2 a0 = lds_tr16_b64(smem + (unsigned)m0 * ROWB + colb);
    u32x2 a1 = lds_tr16_b64(smem + (unsigned)m1 * ROWB + colb);
    f.a = u32x4{a0.x, a0.y, a1.x, a1.y};
    unsigned r0 = aL + halo_row(m0) + colb, r1 = aL + halo_row(m1) + colb;
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
      u32x2 b0 = lds_tr16_b64(smem + r0 + tapoff[tl]);
      u32x2 b1 = lds_tr16_b64(smem + r1 + tapoff[tl]);
      f.b[tl] = u32x4{b0.x, b0.y, b1.x, b1.y};
    }
  };
  auto mma16 = [&](const WFrag<TPW>& f) {
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
      acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a),
                                                        __builtin_bit_cast(bf16x8, f.b[tl]), acc[tl], 0, 0, 0);
  };
  auto fetch32 = [&](WFragF<TPW>& f, int ks) {
    int m = ks * 2 + half;
    f.a = *(const float*)(smem + (unsigned)m * ROWB + li * 4);
    unsigned r0 = aL + halo_row(m) + li * 4;
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) f.b[tl] = *(const float*)(smem + r0 + tapoff[tl]);
  };
  auto mma32 = [&](const WFragF<TPW>& f) {
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a, f.b[tl], acc[tl], 0, 0, 0);
  };

#ifdef CBIM_IGEMM_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long plast = __builtin_readcyclecounter();
#define CBIM_TICK(ph) { unsigned long long t_ = __builtin_readcyclecounter(); pacc[ph] += t_ - plast; plast = t_; }
#else
#define CBIM_TICK(ph) ((void)0)
#endif
  // tile coordinates advanced incrementally (three scalar divisions per tile otherwise)
  int ttd = t_begin / (p.tiles_w * p.tiles_h), tth = (t_begin / p.tiles_w) % p.tiles_h, ttw = t_begin % p.tiles_w;
  for (int t = t_begin; t < t_end; ++t) {
    const int od0 = ttd * p.tD, oh0 = tth * p.tH, ow0 = ttw * 8;
    if (++ttw == p.tiles_w) { ttw = 0; if (++tth == p.tiles_h) { tth = 0; ++ttd; } }
    const int id0 = od0 - p.pD, ih0 = oh0 - p.pH, iw0 = ow0 - p.pW;
    CBIM_TICK(0);
    __syncthreads();
    CBIM_TICK(1);
    // ---- stage the dy tile (dense [BMv][32 co]) and the transformed input halo ([hV][32 ci]).  ALL global
    //      loads of the tile (UD + UA 16-byte chunks per thread) are issued before the first LDS store, addressed
    //      as wave-uniform tile pointer + 32-bit byte offset (24-bit multiply-adds, no 64-bit vector arithmetic) --
    constexpr int UD = SLOTS * 256 / NT, UA = 10;   // dy tile: at most 256 voxels x SLOTS chunks, all prefetched
    const int d_items = BMv * SLOTS, a_items = hV * SLOTS;
    const unsigned dy_sb = (unsigned)(dy_from2 ? p.dy2_stride : p.dy_stride) * ES, x_sb = (unsigned)p.x_stride * ES;
    const long long orow = (((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0;
    const long long irow = (((long long)n * p.Di + id0) * p.Hi + ih0) * p.Wi + iw0;
    const unsigned char* dy_tile = (const unsigned char*)(dy_from2 ? p.dy2 : p.dy) + orow * (long long)dy_sb;
    const unsigned char* x_tile = (const unsigned char*)p.x + irow * (long long)x_sb;
    const unsigned dy_cb = (unsigned)(dy_from2 ? co0 - p.cout_split : co0) * ES, x_cb = (unsigned)ci0 * ES;
    u32x4 vd[UD], va[UA];
    unsigned lda = 0;
    // the item decode below does not depend on the tile; laundering the thread index keeps the compiler from
    // hoisting ~40 registers of it out of the tile loop (they would spill next to the 112 accumulators)
    int tidl = tid;
#ifndef CBIM_EMU
    asm volatile("" : "+v"(tidl));
#endif
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const int item = tidl + u * NT;
      const int m = item / SLOTS;
      const unsigned tw = m & 7, th = (m >> 3) & (p.tH - 1), td = m >> (3 + p.lgH);
      vd[u] = u32x4{0u, 0u, 0u, 0u};
      if (item < d_items && co0 < p.Cout && od0 + (int)td < p.Do && oh0 + (int)th < p.Ho && ow0 + (int)tw < p.Wo)
        vd[u] = *(const u32x4*)(dy_tile + wmad24(wmad24(wmad24(td, (unsigned)p.Ho, th), (unsigned)p.Wo, tw), dy_sb, dy_cb));
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int item = tidl + u * NT;
      const unsigned hv = (unsigned)item / SLOTS;
      const unsigned hd = (hv * p.mHW) >> 20;
      const unsigned r2 = hv - hd * hHW;
      const unsigned hh = (r2 * p.mW) >> 20;
      const unsigned hw = r2 - hh * p.hW;
      const bool ld = item < a_items && ci0 < p.Cin && (unsigned)(id0 + (int)hd) < (unsigned)p.Di &&
                      (unsigned)(ih0 + (int)hh) < (unsigned)p.Hi && (unsigned)(iw0 + (int)hw) < (unsigned)p.Wi;
      va[u] = u32x4{0u, 0u, 0u, 0u};
      if (ld) va[u] = *(const u32x4*)(x_tile + wmad24(wmad24(wmad24(hd, (unsigned)p.Hi, hh), (unsigned)p.Wi, hw), x_sb, x_cb));
      lda |= (ld ? 1u : 0u) << u;
    }
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const int item = tid + u * NT;
      if (item < d_items) *(u32x4*)(smem + (unsigned)item * 16) = vd[u];
    }
    CBIM_TICK(2);
    auto xform = [&](u32x4 w) -> u32x4 {
      float f[CPC];
      Elem<T>::unpack(w, f);
#pragma unroll
      for (int j = 0; j < CPC; ++j) f[j] = wg_actf<ACT>((f[j] - mean[j]) * rstd[j], p.act);
      return Elem<T>::pack(f);
    };
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int item = tid + u * NT;
      if (item < a_items) {
        u32x4 w = va[u];
        if (p.in_stats && ((lda >> u) & 1u)) w = xform(w);
        *(u32x4*)(smem + aL + (unsigned)item * 16) = w;
      }
    }
    // halos larger than NT * UA chunks (kernels beyond 3x3x3 on the 4x8x8 tile): remaining items, same scheme
    for (int base = tid + NT * UA; base < a_items; base += NT) {
      const unsigned hv = (unsigned)base / SLOTS;
      const unsigned hd = (hv * p.mHW) >> 20;
      const unsigned r2 = hv - hd * hHW;
      const unsigned hh = (r2 * p.mW) >> 20;
      const unsigned hw = r2 - hh * p.hW;
      const bool ld = ci0 < p.Cin && (unsigned)(id0 + (int)hd) < (unsigned)p.Di && (unsigned)(ih0 + (int)hh) < (unsigned)p.Hi &&
                      (unsigned)(iw0 + (int)hw) < (unsigned)p.Wi;
      u32x4 w = u32x4{0u, 0u, 0u, 0u};
      if (ld) {
        w = *(const u32x4*)(x_tile + wmad24(wmad24(wmad24(hd, (unsigned)p.Hi, hh), (unsigned)p.Wi, hw), x_sb, x_cb));
        if (p.in_stats) w = xform(w);
      }
      *(u32x4*)(smem + aL + (unsigned)base * 16) = w;
    }
    CBIM_TICK(3);
    __syncthreads();
    CBIM_TICK(4);
    // ---- contraction over the tile's voxels, two k-steps per trip through static register sets --------------
    if (IS_BF16 && K3T) {
      // lane bases: dy row mq, halo row (half, i16 >> 2) of the tile's first plane pair; k-step ks adds
      // 16 dy rows and ((ks >> 2) * 10 + 2 * (ks & 3)) * 10 halo rows — compile-time immediates of the ds_read
      const unsigned dyb = (unsigned)mq * ROWB + colb;
      unsigned tb[TPW];
#pragma unroll
      for (int tl = 0; tl < TPW; ++tl) tb[tl] = aL + (unsigned)(half * 10 + (i16 >> 2)) * ROWB + colb + tapoff[tl];
      auto fetchk = [&](WFrag<TPW>& f, int ks) {
        const unsigned kc = (unsigned)(((ks >> 2) * 10 + 2 * (ks & 3)) * 10) * ROWB;
        u32x2 a0 = lds_tr16_b64(smem + dyb + (unsigned)ks * 16 * ROWB);
        u32x2 a1 = lds_tr16_b64(smem + dyb + (unsigned)ks * 16 * ROWB + 4 * ROWB);
        f.a = u32x4{a0.x, a0.y, a1.x, a1.y};
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
          u32x2 b0 = lds_tr16_b64(smem + tb[tl] + kc);
          u32x2 b1 = lds_tr16_b64(smem + tb[tl] + kc + 4 * ROWB);
          f.b[tl] = u32x4{b0.x, b0.y, b1.x, b1.y};
        }
      };
      // the fences pin "issue the next step's 16 transposed reads, THEN run this step's MFMAs": left alone the
      // scheduler re-uses one B register set and every second MFMA waits for a read issued just before it
      WFrag<TPW> f0, f1;
      fetchk(f0, 0);
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        fetchk(f1, ks + 1);
        WG_SCHED_FENCE();
        mma16(f0);
        WG_SCHED_FENCE();
        if (ks + 2 < 16) fetchk(f0, ks + 2);
        WG_SCHED_FENCE();
        mma16(f1);
        WG_SCHED_FENCE();
      }
    } else if (IS_BF16) {
      // a single tap (1x1x1 convs): the 4 waves split the voxel steps instead of repeating the tap, each writes
      // its own slab (ksplit); otherwise every wave walks all steps for its own taps
      const int nks = BMv / 16;   // multiple of 8 (BMv is 128 or 256)
      const int k_lo = ksplit ? wave * (nks / 4) : 0, k_hi = ksplit ? k_lo + nks / 4 : nks;
      WFrag<TPW> f0, f1;
      fetch16(f0, k_lo);
      for (int ks = k_lo; ks < k_hi; ks += 2) {
        fetch16(f1, ks + 1);
        mma16(f0);
        if (ks + 2 < k_hi) fetch16(f0, ks + 2);
        mma16(f1);
      }
    } else {
      const int nks = BMv / 2;
      const int k_lo = ksplit ? wave * (nks / 4) : 0, k_hi = ksplit ? k_lo + nks / 4 : nks;
      WFragF<TPW> f0, f1;
      fetch32(f0, k_lo);
      for (int ks = k_lo; ks < k_hi; ks += 2) {
        fetch32(f1, ks + 1);
        mma32(f0);
        if (ks + 2 < k_hi) fetch32(f0, ks + 2);
        mma32(f1);
      }
    }
  }

  CBIM_TICK(5);
#ifdef CBIM_IGEMM_PROF
  if (p.prof && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) p.prof[wave * 8 + i] = pacc[i];
#endif
  // ---- write this strip's slab: ws[(n*strips+strip)][tap][Cout_pad][Cin_pad] -------------------------------------
  const size_t slab = (size_t)p.taps * p.Cout_pad * p.Cin_pad;
  float* wsb = p.ws + (ksplit ? (size_t)blockIdx.x * 4 + wave : (size_t)blockIdx.x) * slab;
#pragma unroll
  for (int tl = 0; tl < TPW; ++tl) {
    int tap = ksplit ? 0 : wave + 4 * tl;
    if (tap < p.taps) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        int ci = ib * 32 + li;
        wsb[((size_t)tap * p.Cout_pad + co) * p.Cin_pad + ci] = acc[tl][r];
      }
    }
  }
}

// 64 outputs per workgroup x 4 slab phases: thread (o, ph) adds slabs ph, ph+4, .. (two independent chains), the four
// phase sums are combined through LDS in fixed order.  (One thread per output walking all ~512 slabs left this at 108
// workgroups for a 32x32x27 gradient: under a quarter of the chip, 21 us per call, 34 calls per step.)
// Round 6 — the same fixed-order sum for weights with several taps, as a TRANSPOSE through LDS: a workgroup owns (co, 256 input
// channels): for each tap the slabs' rows ws[s][tap][co][ci0 ..] are summed with coalesced loads (thread = ci) into T[tap][ci], then
// the gradient's own rows dw[co][ci0 ..][0 .. taps) — taps x 256 contiguous floats — leave with coalesced stores.  k_wgrad_reduce
// walks a flat output index (two 64-bit divisions per element) and writes dw with a stride of `taps` floats between lanes: on the
// low-resolution SwinUNETR layers (192^2 ... 768^2 x 27 weights, 4-64 MB) it ran 60 us per launch, 0.6 ms per step.
__global__ void __launch_bounds__(NT) k_wgrad_reduce_t(const float* __restrict__ ws, float* __restrict__ dw, int n_slabs, int taps,
                                                       int Cout, int Cin, int Cout_pad, int Cin_pad) {
  __shared__ float T[28 * 257];
  const int t = threadIdx.x, co = blockIdx.y, ci0 = blockIdx.x * NT;
  const size_t slab = (size_t)taps * Cout_pad * Cin_pad;
  const int ci = ci0 + t;
  for (int tap = 0; tap < taps; ++tap) {
    float a0 = 0.f, a1 = 0.f;
    if (ci < Cin) {
      const float* src = ws + ((size_t)tap * Cout_pad + co) * Cin_pad + ci;
      int sl = 0;
      for (; sl + 1 < n_slabs; sl += 2) {
        a0 += src[(size_t)sl * slab];
        a1 += src[(size_t)(sl + 1) * slab];
      }
      if (sl < n_slabs) a0 += src[(size_t)sl * slab];
    }
    T[tap * 257 + t] = a0 + a1;
  }
  __syncthreads();
  const int n_ci = Cin - ci0 < NT ? Cin - ci0 : NT;
  float* dst = dw + ((size_t)co * Cin + ci0) * taps;
  for (int e = t; e < n_ci * taps; e += NT) {
    const int cl = e / taps, tap = e - cl * taps;
    dst[e] = T[tap * 257 + cl];
  }
}

__global__ void __launch_bounds__(NT) k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw,
                                                     int n_slabs, int taps, int Cout, int Cin, int Cout_pad,
                                                     int Cin_pad, int64_t total) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const size_t slab = (size_t)taps * Cout_pad * Cin_pad;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + o;
    float a0 = 0.f, a1 = 0.f;
    int ci = 0, co = 0, tap = 0;
    if (i < total) {
      ci = (int)(i % Cin);
      const int64_t r = i / Cin;
      co = (int)(r % Cout);
      tap = (int)(r / Cout);
      const float* src = ws + ((size_t)tap * Cout_pad + co) * Cin_pad + ci;
      int s = ph;
      for (; s + 4 < n_slabs; s += 8) {
        a0 += src[(size_t)s * slab];
        a1 += src[(size_t)(s + 4) * slab];
      }
      if (s < n_slabs) a0 += src[(size_t)s * slab];
    }
    red[ph][o] = a0 + a1;
    __syncthreads();
    if (ph == 0 && i < total)
      dw[((size_t)co * Cin + ci) * taps + tap] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    __syncthreads();
  }
}

struct WgCfg { int tD, tH, lgH, tiles_d, tiles_h, tiles_w, strips_per_n, tiles_per_strip, co_blocks, ci_blocks; };

static WgCfg wg_cfg(const cbim_conv_desc* d) {
  WgCfg c;
  int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  if (S >= 32768 && d->Do >= 4 && d->Ho >= 8) { c.tD = 4; c.tH = 8; c.lgH = 3; }
  else if (d->Do <= 2) { c.tD = 2; c.tH = 8; c.lgH = 3; }
  else { c.tD = 4; c.tH = 4; c.lgH = 2; }
  c.tiles_d = (d->Do + c.tD - 1) / c.tD;
  c.tiles_h = (d->Ho + c.tH - 1) / c.tH;
  c.tiles_w = (d->Wo + 7) / 8;
  c.co_blocks = (d->Cout + 31) / 32;
  c.ci_blocks = (d->Cin + 31) / 32;
  int tiles_per_n = c.tiles_d * c.tiles_h * c.tiles_w;
  int64_t pairs = (int64_t)c.co_blocks * c.ci_blocks;
  // ~512 resident workgroups (2 per CU); every extra strip costs one slab of workspace traffic
  static const int64_t wgs = 512;   // (tools: A/B of the strip count)
  int64_t want = (wgs + pairs * d->N - 1) / (pairs * d->N);
  if (want < 1) want = 1;
  if (want > tiles_per_n) want = tiles_per_n;
  int taps = d->kD * d->kH * d->kW;
  size_t slab = (size_t)(taps == 1 ? 4 : taps) * c.co_blocks * 32 * c.ci_blocks * 32 * sizeof(float);
  int64_t cap = (int64_t)((64ull << 20) / (slab * d->N));   // cap the slab workspace at ~64 MiB
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  c.tiles_per_strip = (int)((tiles_per_n + want - 1) / want);
  c.strips_per_n = (tiles_per_n + c.tiles_per_strip - 1) / c.tiles_per_strip;
  return c;
}

}  // namespace cbim

using namespace cbim;

static void launch_reduce(const float* ws, float* dw, int n_slabs, int taps, int Cout, int Cin, int Cout_pad, int Cin_pad, hipStream_t st);

extern "C" size_t cbim_conv3d_wgrad_workspace(const cbim_conv_desc* d) {
  if (!d) return 0;
  WgCfg c = wg_cfg(d);
  int taps = d->kD * d->kH * d->kW;
  // one tap: each of the 4 waves writes its own slab (they split the voxel loop)
  size_t need = (size_t)d->N * c.strips_per_n * (taps == 1 ? 4 : taps) * c.co_blocks * 32 * c.ci_blocks * 32 * sizeof(float);
  // (the kernel is picked at launch, from arguments this query does not see: room for either)
  if (cbim_wgrad_r32_eligible(d, nullptr, nullptr, 0, nullptr, 0)) {
    size_t r = cbim_wgrad_r32_workspace(d);
    if (r > need) need = r;
  }
  if (cbim_conv_pw_wgrad_workspace(d) > need) need = cbim_conv_pw_wgrad_workspace(d);   // pointwise layers: conv_pw.hip
  return need;
}

// which kernel the last cbim_conv3d_wgrad call of this thread launched: 0 = k_conv_wgrad, 1 = k_wgrad_r32 (profiling labels)
static thread_local int g_last_wgrad_kernel = 0;
extern "C" int cbim_conv3d_wgrad_last_kernel(void) { return g_last_wgrad_kernel; }

template <typename T, int TPW, int ACT, bool K3T = false>
static int launch_wgrad(const WgradParams& p, dim3 grid, size_t smem, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv_wgrad<T, TPW, ACT, K3T>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  CBIM_LAUNCH((k_conv_wgrad<T, TPW, ACT, K3T>), grid, dim3(NT), smem, st, p);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "conv wgrad launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

template <typename T, int ACT>
static int dispatch_tpw(int taps, const WgradParams& p, dim3 grid, size_t smem, hipStream_t st) {
  int tpw = (taps + 3) / 4;
  if (tpw <= 1) return launch_wgrad<T, 1, ACT>(p, grid, smem, st);
  if (tpw <= 3) return launch_wgrad<T, 3, ACT>(p, grid, smem, st);
  if (tpw <= 5) return launch_wgrad<T, 5, ACT>(p, grid, smem, st);
  if (sizeof(typename Elem<T>::type) == 2 && p.kH == 3 && p.kW == 3 && p.tD == 4 && p.tH == 8)
    return launch_wgrad<T, 7, ACT, true>(p, grid, smem, st);
  return launch_wgrad<T, 7, ACT>(p, grid, smem, st);
}

extern "C" int cbim_conv3d_wgrad(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2,
                                 int64_t x2_stride, int cin_split,
                                 const float* in_stats, const void* dy, int64_t dy_stride, const void* dy2,
                                 int64_t dy2_stride, int cout_split, float* dw,
                                 void* workspace, size_t ws_bytes, void* stream) {
  CBIM_CHECK(d && x && dy && dw, CBIM_EINVAL, "null argument");
  g_last_wgrad_kernel = 0;
  if (cbim_wgrad_r32_eligible(d, in_stats, x2, cin_split, dy2, cout_split)) {
    size_t need = cbim_wgrad_r32_workspace(d);
    CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "wgrad workspace %zu < %zu", ws_bytes, need);
    g_last_wgrad_kernel = 1;
    if (int rc = cbim_wgrad_r32_launch(d, x, x_stride, x2, x2_stride, cin_split, dy, dy_stride, dy2, dy2_stride, cout_split,
                                       (float*)workspace, dw, stream))
      return rc;
    return cbim_wgrad_r32_reduce(d, (const float*)workspace, dw, stream);
  }
  if (cbim_conv_pw_wgrad_eligible(d, x_stride, x2, dy_stride, dy2)) {      // bf16 1x1x1: conv_pw.hip (round 4)
    size_t need = cbim_conv_pw_wgrad_workspace(d);
    CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "wgrad workspace %zu < %zu", ws_bytes, need);
    g_last_wgrad_kernel = 2;
    int n_slabs = 0, cop = 0, cip = 0;
    if (int rc = cbim_conv_pw_wgrad_launch(d, x, x_stride, in_stats, dy, dy_stride, (float*)workspace, &n_slabs, &cop, &cip, stream)) return rc;
    const int64_t total = (int64_t)d->Cout * d->Cin;
    int64_t blocks = (total + 63) / 64;
    if (blocks > 4096) blocks = 4096;
    CBIM_LAUNCH(k_wgrad_reduce, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, (const float*)workspace, dw, n_slabs, 1, d->Cout,
                d->Cin, cop, cip, total);
    return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
  }
  CBIM_CHECK(!x2, CBIM_EUNSUPPORTED, "wgrad: a second input tensor is only taken by the bf16 3x3x3 kernel on raw (un-normalised) "
             "inputs with channel counts in multiples of 32");
  CBIM_CHECK(d->dtype == CBIM_F32 || d->dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype");
  int cpc = d->dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(d->Cin % cpc == 0 && d->Cout % cpc == 0, CBIM_EUNSUPPORTED,
             "wgrad needs Cin (%d) and Cout (%d) to be multiples of %d", d->Cin, d->Cout, cpc);
  int taps = d->kD * d->kH * d->kW;
  CBIM_CHECK(taps <= 28, CBIM_EUNSUPPORTED, "wgrad supports at most 28 taps");
  size_t need = cbim_conv3d_wgrad_workspace(d);
  CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "wgrad workspace %zu < %zu", ws_bytes, need);
  WgCfg c = wg_cfg(d);
  WgradParams p;
  p.x = x; p.x_stride = x_stride; p.in_stats = in_stats; p.dy = dy; p.dy_stride = dy_stride;
  p.dy2 = dy2; p.dy2_stride = dy2_stride; p.cout_split = cout_split;
  CBIM_CHECK(!dy2 || (cout_split > 0 && cout_split < d->Cout && cout_split % 32 == 0), CBIM_EUNSUPPORTED,
             "second dy: split %d must be a multiple of 32", cout_split);
  p.ws = (float*)workspace;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin;
  p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.kD = d->kD; p.kH = d->kH; p.kW = d->kW; p.pD = d->pD; p.pH = d->pH; p.pW = d->pW; p.act = d->act;
  p.tD = c.tD; p.tH = c.tH; p.lgH = c.lgH; p.tiles_d = c.tiles_d; p.tiles_h = c.tiles_h; p.tiles_w = c.tiles_w;
  p.hD = c.tD + d->kD - 1; p.hH = c.tH + d->kH - 1; p.hW = 8 + d->kW - 1;
  CBIM_CHECK(p.hD * p.hH * p.hW <= 2048, CBIM_EUNSUPPORTED, "halo too large");
  {
    // 32-bit byte offsets inside one tile box, built from 24-bit multiplies (k_conv_wgrad staging)
    const int es_ = d->dtype == CBIM_BF16 ? 2 : 4;
    const int64_t in_rows = (int64_t)p.hD * d->Hi * d->Wi, out_rows = (int64_t)c.tD * d->Ho * d->Wo;
    const int64_t xs = x_stride * es_, ds = (dy2 && dy2_stride > dy_stride ? dy2_stride : dy_stride) * es_;
    CBIM_CHECK(in_rows < (1 << 24) && out_rows < (1 << 24) && xs < (1 << 24) && ds < (1 << 24) &&
               in_rows * xs < ((int64_t)1 << 32) && out_rows * ds < ((int64_t)1 << 32), CBIM_EUNSUPPORTED,
               "wgrad planes %dx%d with row strides %lld/%lld B exceed the 32-bit tile addressing", d->Hi, d->Wi, (long long)xs, (long long)ds);
  }
  p.mHW = ((1u << 20) + (unsigned)(p.hH * p.hW) - 1) / (unsigned)(p.hH * p.hW);
  p.mW = ((1u << 20) + (unsigned)p.hW - 1) / (unsigned)p.hW;
  p.taps = taps; p.strips_per_n = c.strips_per_n; p.tiles_per_strip = c.tiles_per_strip;
  p.ci_blocks = c.ci_blocks; p.Cout_pad = c.co_blocks * 32; p.Cin_pad = c.ci_blocks * 32;
  int es = d->dtype == CBIM_BF16 ? 2 : 4;
  size_t smem = ((size_t)c.tD * c.tH * 8 + (size_t)p.hD * p.hH * p.hW) * 32 * es;
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "wgrad tile needs %zu B of LDS", smem);
  dim3 grid((unsigned)(d->N * c.strips_per_n), (unsigned)(c.co_blocks * c.ci_blocks));
  hipStream_t st = (hipStream_t)stream;
#ifdef CBIM_IGEMM_PROF
  static unsigned long long* prof_dev = nullptr;
  if (!prof_dev) (void)hipMalloc((void**)&prof_dev, 64 * sizeof(unsigned long long));
  (void)hipMemsetAsync(prof_dev, 0, 64 * sizeof(unsigned long long), st);
  p.prof = prof_dev;
  struct ProfDump {
    unsigned long long* dev; hipStream_t st; const cbim_conv_desc* d; int tps;
    ~ProfDump() {
      unsigned long long h[64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
      static const char* nm[6] = {"loop_top", "barrier0", "stage_dy", "stage_halo", "barrier1", "contraction"};
      fprintf(stderr, "[wgrad prof] %d->%d @%d (%d tiles/strip):", d->Cin, d->Cout, d->Do, tps);
      for (int i = 0; i < 6; ++i) {
        unsigned long long s = 0;
        for (int w = 0; w < 4; ++w) s += h[w * 8 + i];
        fprintf(stderr, " %s %llu", nm[i], s / 4);
      }
      fprintf(stderr, "\n");
    }
  } prof_dump{prof_dev, st, d, c.tiles_per_strip};
#endif
  const bool relu = d->act == CBIM_ACT_RELU || !in_stats;
  int rc;
  if (d->dtype == CBIM_BF16)
    // (LeakyReLU as a compile-time case: the monai blocks of SwinUNETR; the run-time switch of the generic instantiation
    //  sits inside the staging loop, which is 64 % of this kernel's cycles)
    rc = relu ? dispatch_tpw<bf16_tag, CBIM_ACT_RELU>(taps, p, grid, smem, st)
              : d->act == CBIM_ACT_LRELU ? dispatch_tpw<bf16_tag, CBIM_ACT_LRELU>(taps, p, grid, smem, st)
                                         : dispatch_tpw<bf16_tag, -1>(taps, p, grid, smem, st);
  else
    rc = relu ? dispatch_tpw<float, CBIM_ACT_RELU>(taps, p, grid, smem, st) : dispatch_tpw<float, -1>(taps, p, grid, smem, st);
  if (rc) return rc;
  launch_reduce((const float*)workspace, dw, d->N * c.strips_per_n * (taps == 1 ? 4 : 1), taps, d->Cout, d->Cin, p.Cout_pad, p.Cin_pad, st);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// taps > 1: the transposing form (k_wgrad_reduce_t); a single tap: the flat form
static void launch_reduce(const float* ws, float* dw, int n_slabs, int taps, int Cout, int Cin, int Cout_pad, int Cin_pad, hipStream_t st) {
  // the transposing form walks the slabs serially per (co, 256 ci) workgroup: for the big low-resolution weights (few slabs,
  // hundreds of workgroups); the many-slab high-resolution layers keep the flat form (measured: Cout 48 x 1024 slabs 1017 us in
  // the transposing form, 768 x 768 x 27 over 2 slabs 35 us against ~100; profiles/r06_s_reduce.txt)
  if (taps > 1 && taps <= 28 && n_slabs <= 16 && Cin >= 128 && (int64_t)Cout * ((Cin + NT - 1) / NT) >= 96) {
    CBIM_LAUNCH(k_wgrad_reduce_t, dim3((unsigned)((Cin + NT - 1) / NT), (unsigned)Cout), dim3(NT), 0, st, ws, dw, n_slabs, taps, Cout, Cin,
                Cout_pad, Cin_pad);
    return;
  }
  const int64_t total = (int64_t)taps * Cout * Cin;
  int64_t blocks = (total + 63) / 64;
  if (blocks > 4096) blocks = 4096;
  CBIM_LAUNCH(k_wgrad_reduce, dim3((unsigned)blocks), dim3(NT), 0, st, ws, dw, n_slabs, taps, Cout, Cin, Cout_pad, Cin_pad, total);
}

// the fixed-order slab reduce for launchers in other files (conv_pw.hip's token Linears): slabs [n_slabs][taps][Cout_pad][Cin_pad]
int cbim_wgrad_reduce_launch(const float* ws, float* dw, int n_slabs, int taps, int Cout, int Cin, int Cout_pad, int Cin_pad,
                             void* stream) {
  launch_reduce(ws, dw, n_slabs, taps, Cout, Cin, Cout_pad, Cin_pad, (hipStream_t)stream);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

CBIM_DEFINE_WARM(wgrad)
