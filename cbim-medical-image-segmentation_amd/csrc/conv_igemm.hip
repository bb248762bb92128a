// conv_igemm.hip — 3-D convolution (forward and dgrad) as an implicit GEMM on the CDNA4 matrix
// cores, written for gfx950 (wave64, v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).
//
//   GEMM view:  M = output voxels, N = Cout, K = taps x Cin.
//   Workgroup:  512 threads = 8 waves (2 per SIMD), PERSISTENT: one workgroup per CU walks a
//               contiguous strip of output tiles (XCD-contiguous, so neighbouring halos hit the same
//               L2).  Tile = tD x 8 x 8 voxels (BM = 256*MT) x BN = 32*NTL couts; wave w owns m-tiles
//               [w*MT, w*MT+MT) (32 voxels = 4 rows x 8 in W) and all NTL n-tiles.
//   K loop:     Cin in chunks of 64 BYTES per voxel (32 bf16 / 16 f32 channels); a "unit" is one
//               (tile, chunk), a "stage" one kd-plane (kH*kW taps) of a unit.
//   Pipeline:   * weights of stage s+1 stream into the other half of a double LDS buffer by LDS-DMA
//                 (global_load_lds_dwordx4: pre-packed in MFMA B-fragment order, so the copy is
//                 contiguous and needs no registers) while stage s computes;
//               * the input halo of unit u+1 is loaded into registers during the last stage of unit
//                 u and written to LDS — InstanceNorm + activation of the producer applied on the way
//                 (pre-activation ConvNormAct, /root/reference/model/dim3/conv_layers.py:48-49; literal
//                 zeros for the padding) — right after it;
//               * inside a stage the fragments of tap t+1 are fetched (ds_read_b128) into a second
//                 register set before the MFMAs of tap t issue.
//   LDS layout: halo row = 64 B = four 16-B slots, slot index XOR (halo_row_h & 3): a 16-lane
//               ds_read_b128 group (4 voxel rows x 4 voxels) then covers 16 distinct 16-B slots.
//   Fragments:  one ds_read_b128 per operand per 16(bf16)/8(f32) channels; element order inside a
//               k-group is (lane-half, j) -> channel 2kg*CPC + half*CPC + j on BOTH operands, which
//               makes bf16 (one 32x32x16 MFMA) and f32 (four 32x32x2 MFMAs) byte-identical in LDS.
//   Epilogue:   + residual, x act'(xh) mask (dgrad through a pre-activation), per-tile partial
//               moments (InstanceNorm statistics of the output) or the two InstanceNorm-backward sums.
//
// Replaces aten::convolution / convolution_backward(input) for nn.Conv3d in ConvNormAct
// (conv_layers.py:29-38), stride 1, groups 1, bias-free; padding k//2 (unet_utils.py:13).
#include "cbim_common.h"
#include "conv_r32.h"
#include <stdlib.h>

// timing ablations of tools/conv_ablate.py (wrong results): compiled in only with `make EXTRA=-DCBIM_IGEMM_DBG_RT`
#ifdef CBIM_IGEMM_DBG_RT
#define I_DBG (p.dbg)
#else
#define I_DBG 0
#endif

namespace cbim {

static constexpr int NT = 512;
static constexpr int NW = NT / 64;
static constexpr int RB = 64;       // bytes per halo row (one Cin chunk)
static constexpr int SLOTS = RB / 16;
static constexpr int KG = RB / 32;  // k-groups (two 16-B slots each) per chunk
static constexpr int UH = 8;        // halo prefetch registers (16-B quads) per thread: covers 1024 rows

struct IgemmParams {
  const void* x; int64_t x_stride;
  const void* x2; int64_t x2_stride; int cin_split;   // channels >= cin_split come from x2 (K-concatenated inputs)
  const float* in_stats;
  const void* w;
  const void* res; int64_t res_stride;
  const void* mx; int64_t mx_stride; const float* m_stats;
  void* y; int64_t y_stride;
  float* partials;
  int N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout;
  int kD, kH, kW, pD, pH, pW, act;
  int tD, tH, lgH;                 // tile = tD x tH x 8, tH = 1 << lgH
  int tiles_d, tiles_h, tiles_w;
  int hD, hH, hW;                  // halo extent = tile + k - 1
  int n_chunks, taps;
  unsigned mHW, mW;                // ceil(2^20 / (hH*hW)), ceil(2^20 / hW): division by multiply
  int dbg;                         // timing ablations of rounds 1-3 (tools/archive); always 0
  int P;                           // records per image of `partials` (cbim_conv3d_num_tiles)
  int ksplit;                      // > 1: blockIdx.z owns a slice of the Cin chunks, raw fp32 partials go to ws
  float* ws;                       // [ksplit][N*Do*Ho*Wo][Cout] fp32
#ifdef CBIM_IGEMM_PROF
  unsigned long long* prof;        // tools/ only (-DCBIM_IGEMM_PROF): per-wave cycle totals of the loop phases
#endif
};

template <typename T> struct Mma;
template <> struct Mma<bf16_tag> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// ACT is a template parameter so the hot ReLU instantiation carries no erf/exp code (code size ->
// instruction cache); ACT < 0 = runtime switch for the rarely used activations.
template <int ACT> __device__ __forceinline__ float actf(float x, int rt) {
  if (ACT == CBIM_ACT_RELU) return x > 0.f ? x : 0.f;
  if (ACT == CBIM_ACT_LRELU) return x > 0.f ? x : 0.01f * x;
  if (ACT == CBIM_ACT_NONE) return x;
  return act_fwd(x, rt);
}
template <int ACT> __device__ __forceinline__ float actg(float x, int rt) {
  if (ACT == CBIM_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (ACT == CBIM_ACT_LRELU) return x > 0.f ? 1.f : 0.01f;
  if (ACT == CBIM_ACT_NONE) return 1.f;
  return act_grad(x, rt);
}

#ifdef CBIM_EMU
#define CBIM_SCHED_FENCE() ((void)0)
#else
#define CBIM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// LDS-DMA: lane l copies 16 bytes from its own global address to (wave-uniform LDS base) + 16*l.
__device__ __forceinline__ void dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
#ifdef CBIM_EMU
  emu_global_load_lds16(gsrc, lds_wave_base);
#else
  // Issued through inline asm ON PURPOSE: with the builtin the compiler knows an LDS-DMA is in flight, treats the
  // LGKM counter as out-of-order and turns every `s_waitcnt lgkmcnt(n)` of the fragment pipeline into
  // lgkmcnt(0) — each MFMA pair then waits a full LDS round trip.  The DMA's completion is covered by the
  // explicit wait_vm0() + workgroup barrier at the end of every stage.
  unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
  a = __builtin_amdgcn_readfirstlane(a);
  // M0 (the LDS base of the instruction) is saved and restored INSIDE the statement: no reserved register in the clobber
  // list (clang: "may lead to undefined behaviour"), nothing about M0 is hidden from the compiler
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(a), "v"(gsrc) : "memory");
#endif
}
// wave-level rendez-vous for LDS data exchanged between lanes of ONE wave (LDS executes a wave's
// instructions in order; the compiler must not reorder across it)
__device__ __forceinline__ void wave_sync() {
#ifdef CBIM_EMU
  int z = 0;
  (void)cbim_emu::wave_exchange(&z, sizeof(z));
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
__device__ __forceinline__ void wait_vm0() {
#ifndef CBIM_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// wait for everything older than the UH halo prefetch loads issued last (vector-memory loads return in order):
// the weight DMA of the next stage must have landed, the next tile's halo may stay in flight
template <int N> __device__ __forceinline__ void wait_vm_halo() {
  static_assert(N == 8 || N == 10, "wait_vm_halo: add the immediate");
#ifndef CBIM_EMU
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
#endif
}

// 24-bit integer multiply-add (full rate; the generic 32-bit multiply is quarter rate) and an optimisation barrier
// that makes a loop-invariant register look freshly computed
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) {
#ifdef CBIM_EMU
  return a * b;
#else
  return __umul24(a, b);
#endif
}
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) { return mul24(a, b) + c; }
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
__device__ __forceinline__ unsigned launder(unsigned v) {
#ifndef CBIM_EMU
  asm volatile("" : "+v"(v));
#endif
  return v;
}

template <int MT, int NTL> struct Frags { u32x4 a[MT]; u32x4 b[NTL]; };   // one k-group of one tap

// NTH = threads per workgroup: 512 (one persistent workgroup per CU) or 256 (two per CU, 4x8x8 tiles: the two
// workgroups drift apart, so one's VALU-heavy halo/epilogue phases overlap the other's MFMA k-loop)
template <typename T, int MT, int NTL, int ACT, bool K3, int NTH = 512>
__global__ void __launch_bounds__(NTH, NTH == 256 ? 2 : 1) k_conv_igemm(IgemmParams p) {
  constexpr int NT = NTH, NW = NTH / 64;
  constexpr int UH = NTH == 256 ? 10 : 8;   // halo prefetch quads per thread (6x10x10 rows x 4 slots / 256 threads)
  constexpr int CPC = Elem<T>::CPC;
  constexpr int KC = SLOTS * CPC;  // channels per chunk
  constexpr int BN = 32 * NTL;
  CBIM_DYN_SMEM(smem);
  const int hV = p.hD * p.hH * p.hW;
  const int hHW = p.hH * p.hW;
  const int ptaps = p.kH * p.kW;                        // taps per stage (one kd-plane)
  const unsigned tap_bytes = KG * 2 * BN * 16;
  const unsigned stage_bytes = (unsigned)ptaps * tap_bytes;
  const unsigned a_bytes = (unsigned)hV * RB;
  const unsigned b_base = a_bytes;                      // two stage buffers follow the halo
  const unsigned red_base = a_bytes + 2 * stage_bytes;  // [NW*MT][BN][3] floats
  const unsigned st_base = red_base + (unsigned)(NW * BN * 3) * 4;   // [KC][2] floats: (mean, rstd) of the staged chunk
  const unsigned scr_base = st_base + 512;                           // NTL == 1 only: NW x 4 KiB epilogue scratch

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int n_tiles = p.N * tiles_per_n;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)(((long long)lb * n_tiles) / gridDim.x);
  const int t_end = (int)(((long long)(lb + 1) * n_tiles) / gridDim.x);
  const int nb = blockIdx.y, co0 = nb * BN;
  if (t_begin >= t_end) return;
  // split-K: this workgroup's slice [q_lo, q_hi) of the Cin chunks
  const int q_lo = (int)(((long long)blockIdx.z * p.n_chunks) / p.ksplit);
  const int q_hi = (int)(((long long)(blockIdx.z + 1) * p.n_chunks) / p.ksplit);
  const int nq = q_hi - q_lo;
  if (nq <= 0) return;
  const int n_units = (t_end - t_begin) * nq;

  unsigned a_off[MT];   // LDS byte offset of the lane's voxel row at tap (0,0,0)
  int thr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = (wave * MT + mt) * 32 + li;
    int tw = m & 7, th = (m >> 3) & (p.tH - 1), td = m >> (3 + p.lgH);
    a_off[mt] = (unsigned)((td * p.hH + th) * p.hW + tw) * RB;
    thr[mt] = th;
  }
  const unsigned b_lane = (unsigned)(half * BN + li) * 16;
  f32x16 acc[MT][NTL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int my_slot = tid & (SLOTS - 1);  // NT % SLOTS == 0: a thread always stages the same slot

  // ---- position of a unit: (image, tile coordinates, Cin chunk), advanced incrementally — the divisions that
  //      decode a linear unit index cost ~150 scalar cycles each and ran ten times per unit ----------------------
  struct UnitPos { int n, td, th, tw, q; };
  auto advance = [&](UnitPos& u) {
    if (++u.q == q_hi) {
      u.q = q_lo;
      if (++u.tw == p.tiles_w) {
        u.tw = 0;
        if (++u.th == p.tiles_h) {
          u.th = 0;
          if (++u.td == p.tiles_d) { u.td = 0; ++u.n; }
        }
      }
    }
  };
  UnitPos cur, nxt;
  {
    const int n0 = t_begin / tiles_per_n, tt = t_begin % tiles_per_n;
    cur.n = n0; cur.td = tt / (p.tiles_w * p.tiles_h); cur.th = (tt / p.tiles_w) % p.tiles_h; cur.tw = tt % p.tiles_w;
    cur.q = q_lo;
    nxt = cur;
    advance(nxt);
  }

  // ---- weight stage s -> LDS buffer (s & 1) by LDS-DMA ------------------------------------------------
  auto dma_stage = [&](int q, int kd, int buf) {
    if (I_DBG & 2) return;
    const unsigned char* src = (const unsigned char*)p.w +
        ((size_t)nb * p.n_chunks + q) * ((size_t)p.kD * stage_bytes) + (size_t)kd * stage_bytes;
    unsigned char* dst = smem + b_base + (unsigned)buf * stage_bytes;
    for (unsigned o = (unsigned)wave * 1024; o < stage_bytes; o += NW * 1024)
      dma16(src + o + lane * 16, dst + o);
  };

  // ---- halo of a unit: issue loads into registers / write them (transformed) to LDS --------------------
  // The (mean, rstd) pairs of the unit's chunk travel through 2 registers of the first KC threads and
  // a 256-byte LDS table (written before the stage-end barrier, read by halo_store after it).
  // Everything about an item that does not change from unit to unit — its (hd, hh, hw) position inside the halo
  // box and whether the box has that many rows — is decoded ONCE into one packed register per item.  Per unit the
  // address is then tile-origin pointer (wave-uniform, scalar registers) + a 32-bit byte offset made of two 24-bit
  // multiply-adds, and the LDS address is a lane constant + an immediate.  (Left to itself the compiler hoists the
  // per-item divisions and 64-bit products out of the unit loop into ~100 spilled registers, reloaded from scratch
  // memory at every stage: launder() keeps the derived values inside the loop.)
  unsigned hpk[UH];     // hd | hh << 8 | hw << 16 | (item exists) << 24
#pragma unroll
  for (int u = 0; u < UH; ++u) {
    const int item = tid + u * NT;
    const unsigned hv = (unsigned)item / SLOTS;
    const unsigned hd = (hv * p.mHW) >> 20;
    const unsigned r2 = hv - hd * hHW;
    const unsigned hh = (r2 * p.mW) >> 20;
    const unsigned hw = r2 - hh * p.hW;
    hpk[u] = hd | (hh << 8) | (hw << 16) | (item < hV * SLOTS ? 1u << 24 : 0u);
  }
  // output voxel of epilogue item (mt, it) inside the tile: td | th << 8 | tw << 16 (same for every tile)
  constexpr int EP_OCH = 32 / CPC, EP_IT = 32 * EP_OCH / 64;
  unsigned opk[MT][EP_IT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
      const int m = (wave * MT + mt) * 32 + (lane + 64 * it) / EP_OCH;
      opk[mt][it] = (unsigned)(m >> (3 + p.lgH)) | ((unsigned)((m >> 3) & (p.tH - 1)) << 8) | ((unsigned)(m & 7) << 16);
    }
  const unsigned l_base = (unsigned)(tid / SLOTS) * RB;   // LDS row of item u: l_base + u * (NT / SLOTS) * RB
  u32x4 hreg[UH];
  unsigned hld = 0;
  float sreg0 = 0.f, sreg1 = 1.f;
  auto halo_load = [&](const UnitPos& up) {
    const int n = up.n, q = up.q;
    const int id0 = up.td * p.tD - p.pD;
    const int ih0 = up.th * p.tH - p.pH;
    const int iw0 = up.tw * 8 - p.pW;
    const int c0 = q * KC + my_slot * CPC;
    const bool c_ok = c0 < p.Cin;
    const bool from2 = p.x2 != nullptr && q * KC >= p.cin_split;   // block-uniform (split is chunk aligned)
    if (p.in_stats && tid < KC && q * KC + tid < p.Cin) {
      sreg0 = p.in_stats[((size_t)n * p.Cin + q * KC + tid) * 2];
      sreg1 = p.in_stats[((size_t)n * p.Cin + q * KC + tid) * 2 + 1];
    }
    // wave-uniform: pointer to the halo box origin (may lie outside the tensor: only in-range rows are read)
    const unsigned stride_b = (unsigned)(from2 ? p.x2_stride : p.x_stride) * Elem<T>::SIZE;
    const long long org = (((long long)n * p.Di + id0) * p.Hi + ih0) * p.Wi + iw0;
    const unsigned char* tbase = (const unsigned char*)(from2 ? p.x2 : p.x) + org * (long long)stride_b;
    const unsigned c_byte = c_ok ? (unsigned)(from2 ? c0 - p.cin_split : c0) * Elem<T>::SIZE : 0u;
    // row read by items that are discarded (padding, rows past the box): the box corner clamped into the tensor
    const int sd = clampi(id0, p.Di - 1) - id0, sh = clampi(ih0, p.Hi - 1) - ih0, sw = clampi(iw0, p.Wi - 1) - iw0;
    const unsigned safe = mul24((unsigned)((sd * p.Hi + sh) * p.Wi + sw), stride_b);
    hld = 0;
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      const unsigned pk = launder(hpk[u]);
      const unsigned hd = pk & 255u, hh = (pk >> 8) & 255u, hw = (pk >> 16) & 255u;
      // every thread issues exactly UH loads (discarded items read the safe row): the stage-end wait can then
      // leave exactly these UH loads in flight (wait_vm_halo)
      const bool ld = !(I_DBG & 1) && (pk >> 24) != 0 && c_ok && (unsigned)(id0 + (int)hd) < (unsigned)p.Di &&
                      (unsigned)(ih0 + (int)hh) < (unsigned)p.Hi && (unsigned)(iw0 + (int)hw) < (unsigned)p.Wi;
      const unsigned rel = mad24(mad24(hd, (unsigned)p.Hi, hh), (unsigned)p.Wi, hw);
      const unsigned voff = ld ? mad24(rel, stride_b, c_byte) : safe;
      hld |= (ld ? 1u : 0u) << u;
      hreg[u] = *(const u32x4*)(tbase + voff);
    }
  };
  auto stats_publish = [&]() {   // before a barrier that precedes halo_store
    if (p.in_stats && tid < KC) {
      float* st = (float*)(smem + st_base);
      st[tid * 2] = sreg0;
      st[tid * 2 + 1] = sreg1;
    }
  };
  auto halo_store = [&]() {
    const float* st = (const float*)(smem + st_base) + my_slot * CPC * 2;
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      const unsigned pk = launder(hpk[u]);
      if (pk >> 24) {
        const bool loaded = (hld >> u) & 1u;
        u32x4 w = loaded ? hreg[u] : u32x4{0u, 0u, 0u, 0u};
        if (p.in_stats && loaded) {
          float f[CPC];
          Elem<T>::unpack(w, f);
#pragma unroll
          for (int j = 0; j < CPC; ++j) f[j] = actf<ACT>((f[j] - st[2 * j]) * st[2 * j + 1], p.act);
          w = Elem<T>::pack(f);
        }
        *(u32x4*)(smem + l_base + (unsigned)(u * (NT / SLOTS) * RB) + (((unsigned)my_slot ^ ((pk >> 8) & (SLOTS - 1))) << 4)) = w;
      }
    }
  };

  // ---- fragment fetch of one k-group of one tap.  The XOR-swizzled slot offset of a lane depends on kh
  //      only through (th+kh)&3, so it is kept in registers (sl[kg][mt]) and refreshed when kh advances:
  //      one v_add3 per ds_read in the steady state -------------------------------------------------------
  unsigned sl[KG][MT];
  auto set_kh = [&](int kh) {
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        sl[kg][mt] = (unsigned)((2 * kg + half) ^ ((thr[mt] + kh) & (SLOTS - 1))) << 4;
  };
  auto fetch = [&](Frags<MT, NTL>& f, int tp, int kg, unsigned toff, unsigned b_buf) {
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
      f.b[nt] = *(const u32x4*)(smem + b_buf + b_lane + (unsigned)tp * tap_bytes + (unsigned)(kg * 2 * BN + nt * 32) * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      f.a[mt] = *(const u32x4*)(smem + a_off[mt] + toff + sl[kg][mt]);
  };
  auto mma = [&](const Frags<MT, NTL>& f) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) Mma<T>::run(f.a[mt], f.b[nt], acc[mt][nt]);
  };
  (void)0;

  // ---- statistics record of this workgroup: partials[n][lb][Cout][3], P = gridDim.x records per image; images the
  //      strip does not touch get an empty record (n = 0 merges as the identity) ----------------------------------
  Moments run = {0.f, 0.f, 0.f};
  int run_n = cur.n;
  auto flush_partial = [&](int n, const Moments& m) {
    const size_t o = (((size_t)n * p.P + lb) * p.Cout + co0 + tid) * 3;
    p.partials[o] = m.n; p.partials[o + 1] = m.mean; p.partials[o + 2] = m.m2;
  };
  if (p.partials && p.ksplit == 1 && tid < BN && co0 + tid < p.Cout) {
    const int n_first = t_begin / tiles_per_n, n_last = (t_end - 1) / tiles_per_n;
    for (int n = 0; n < p.N; ++n) {
      if (n < n_first || n > n_last) flush_partial(n, Moments{0.f, 0.f, 0.f});
      for (unsigned r = lb + gridDim.x; r < (unsigned)p.P; r += gridDim.x) {   // buffer sized for a larger grid
        const size_t o = (((size_t)n * p.P + r) * p.Cout + co0 + tid) * 3;
        p.partials[o] = 0.f; p.partials[o + 1] = 0.f; p.partials[o + 2] = 0.f;
      }
    }
  }

  // ---- prologue: first unit's halo and first stage of weights -----------------------------------------
  dma_stage(cur.q, 0, 0);
  halo_load(cur);
  stats_publish();
  __syncthreads();
  halo_store();
  wait_vm0();
  __syncthreads();

#ifdef CBIM_IGEMM_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long plast = __builtin_readcyclecounter();
#define CBIM_TICK(ph) { unsigned long long t_ = __builtin_readcyclecounter(); pacc[ph] += t_ - plast; plast = t_; }
#else
#define CBIM_TICK(ph) ((void)0)
#endif
  int stage = 0;
  for (int unit = 0; unit < n_units; ++unit) {
    const int q = cur.q;
    for (int kd = 0; kd < p.kD; ++kd, ++stage) {
      const bool last_plane = kd == p.kD - 1;
      // ---- start the next stage's weight DMA before computing; the next unit's halo loads are issued
      //      at the unit's FIRST stage so that they have the whole unit to land ------------------------------
      CBIM_TICK(7);
      if (!last_plane) dma_stage(cur.q, kd + 1, (stage + 1) & 1);
      else if (unit + 1 < n_units) dma_stage(nxt.q, 0, (stage + 1) & 1);
      CBIM_TICK(0);
      if (kd == 0 && unit + 1 < n_units) halo_load(nxt);
      CBIM_TICK(6);
      {
        const unsigned a_plane = (unsigned)(kd * hHW) * RB;
        const unsigned b_buf = b_base + (unsigned)(stage & 1) * stage_bytes;
        Frags<MT, NTL> f0, f1;
        if (K3) {
          // 3x3 plane on an 8x8 tile row pitch (hW = 10): all tap / k-group offsets are compile-time
          // immediates of the ds_read; per kh only the XOR-swizzled lane bases are recomputed.  The 18
          // (tap, k-group) steps are fully unrolled, fragments alternate between two static register sets.
          unsigned ab[KG][MT];
          const unsigned bb = b_buf + b_lane;
          auto set_bases = [&](int kh) {
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                ab[kg][mt] = a_off[mt] + a_plane +
                             ((unsigned)((2 * kg + half) ^ ((thr[mt] + kh) & (SLOTS - 1))) << 4);
          };
          auto fetch3 = [&](Frags<MT, NTL>& f, int kh, int kw, int kg) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
              f.b[nt] = *(const u32x4*)(smem + bb + (unsigned)(((kh * 3 + kw) * KG + kg) * 2 * BN + nt * 32) * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              f.a[mt] = *(const u32x4*)(smem + ab[kg][mt] + (unsigned)(kh * 10 + kw) * RB);
          };
          set_bases(0);
          fetch3(f0, 0, 0, 0);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              fetch3(f1, kh, kw, 1);
              CBIM_SCHED_FENCE();
              mma(f0);
              CBIM_SCHED_FENCE();
              if (kw < 2) fetch3(f0, kh, kw + 1, 0);
              else if (kh < 2) { set_bases(kh + 1); fetch3(f0, kh + 1, 0, 0); }
              CBIM_SCHED_FENCE();
              mma(f1);
              CBIM_SCHED_FENCE();
            }
          }
        } else {
          // generic plane: branch-free tap loop, fragments fetched one k-group step ahead (a second step of
          // lookahead was measured 2x slower: the third register set spills in the hot loop).  The fetch
          // after the last tap reads a few rows past the plane inside the LDS allocation, never used.
          int kh = 0, kw = 0;
          unsigned toff = a_plane;
          set_kh(0);
          fetch(f0, 0, 0, toff, b_buf);
          // CBIM_SCHED_FENCE pins "issue the next fragments' ds_reads, THEN run the MFMAs on the previous set":
          // left to itself the scheduler sinks each ds_read next to its use (register pressure) and every
          // MFMA pair then waits a full LDS round trip (s_waitcnt lgkmcnt(0) right after the reads).
          for (int tp = 0; tp < ptaps; ++tp) {
            fetch(f1, tp, 1, toff, b_buf);
            CBIM_SCHED_FENCE();
            mma(f0);
            CBIM_SCHED_FENCE();
            ++kw;
            toff += RB;
            if (kw == p.kW) {   // wave-uniform
              kw = 0;
              ++kh;
              toff += (unsigned)(p.hW - p.kW) * RB;
              set_kh(kh);
            }
            fetch(f0, tp + 1, 0, toff, b_buf);
            CBIM_SCHED_FENCE();
            mma(f1);
            CBIM_SCHED_FENCE();
          }
        }
      }
      CBIM_TICK(1);
      // ---- stage end -------------------------------------------------------------------------------------------
      const bool tile_done = last_plane && q == q_hi - 1;
      if (last_plane && unit + 1 < n_units) stats_publish();
      // this wave's LDS-DMA (next stage's weights) has landed; the halo prefetch issued in this stage (kd == 0)
      // keeps flying through the following stages
      if (kd == 0 && !last_plane && unit + 1 < n_units) wait_vm_halo<UH>();
      else wait_vm0();
      CBIM_TICK(2);
      __syncthreads();   // every wave is done with this stage's A/B reads
      CBIM_TICK(3);
      if (tile_done) {
        // ---- epilogue: each wave transposes its own 32x32 accumulator tiles through a private 4 KiB LDS
        //      scratch (inside the now dead halo region) so that residual / mask loads and the output
        //      stores are whole 16-byte channel chunks; no workgroup barrier inside -----------------------------
        constexpr int OCH = 32 / CPC;            // 16-byte chunks per 32-cout row
        constexpr int IT = 32 * OCH / 64;        // chunk items per lane per 32x32 tile
        const int n = cur.n;
        const int od0 = cur.td * p.tD, oh0 = cur.th * p.tH, ow0 = cur.tw * 8;
        // wave-uniform pointers to the tile's first output row; a lane adds a 32-bit byte offset (24-bit multiplies)
        const long long orow = (((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0;
        const unsigned y_sb = (unsigned)p.y_stride * Elem<T>::SIZE, res_sb = (unsigned)p.res_stride * Elem<T>::SIZE,
                       mx_sb = (unsigned)p.mx_stride * Elem<T>::SIZE, ws_sb = (unsigned)p.Cout * 4u;
        unsigned char* y_tile = (unsigned char*)p.y + orow * (long long)y_sb;
        const unsigned char* res_tile = (const unsigned char*)p.res + orow * (long long)res_sb;
        const unsigned char* mx_tile = (const unsigned char*)p.mx + orow * (long long)mx_sb;
        unsigned char* ws_tile = (unsigned char*)p.ws + ((long long)blockIdx.z * ((long long)p.N * p.Do * p.Ho * p.Wo) + orow) * (long long)ws_sb;
        // scratch: the stage buffer just consumed (NTL=2: 36 KiB) or a dedicated region (NTL=1), so the
        // next halo can be written while other waves are still in their epilogue
        float* scr = (float*)(smem + ((stage_bytes < NW * 4096u) ? scr_base : b_base + (unsigned)(stage & 1) * stage_bytes) +
                              (unsigned)wave * 4096);
        float* red = (float*)(smem + red_base);
        const int cc = lane % OCH;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          const int cch0 = co0 + nt * 32 + cc * CPC;     // first channel of the lane's chunk
          const bool c_ok = cch0 < p.Cout;
          float mm[CPC], mr[CPC], s0[CPC], s1[CPC], sh[CPC];
          float cnt = 0.f;
#pragma unroll
          for (int j = 0; j < CPC; ++j) { mm[j] = 0.f; mr[j] = 1.f; s0[j] = 0.f; s1[j] = 0.f; sh[j] = 0.f; }
          if (p.mx && p.m_stats && c_ok) {     // (no statistics: the mask tensor is the activated a = relu(IN(x)): xh := a)
#pragma unroll
            for (int j = 0; j < CPC; ++j) {
              mm[j] = p.m_stats[((size_t)n * p.Cout + cch0 + j) * 2];
              mr[j] = p.m_stats[((size_t)n * p.Cout + cch0 + j) * 2 + 1];
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            wave_sync();   // the previous tile's scratch reads are done
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              scr[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[mt][nt][r];
              acc[mt][nt][r] = 0.f;
            }
            wave_sync();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
              const unsigned ok_ = launder(opk[mt][it]);
              const int vr = (lane + 64 * it) / OCH;                 // voxel row inside the 32-voxel m-tile
              const unsigned td = ok_ & 255u, th = (ok_ >> 8) & 255u, tw = (ok_ >> 16) & 255u;
              const bool inb = c_ok && od0 + (int)td < p.Do && oh0 + (int)th < p.Ho && ow0 + (int)tw < p.Wo;
              const unsigned rel = mad24(mad24(td, (unsigned)p.Ho, th), (unsigned)p.Wo, tw);
              float v[CPC];
#pragma unroll
              for (int j4 = 0; j4 < CPC; j4 += 4) {
                f32x4 q4 = *(const f32x4*)(scr + vr * 32 + cc * CPC + j4);
                v[j4] = q4.x; v[j4 + 1] = q4.y; v[j4 + 2] = q4.z; v[j4 + 3] = q4.w;
              }
              if (p.ksplit > 1) {   // split-K: raw fp32 partial, finished by k_splitk_finish
                if (inb) {
                  float* wp = (float*)(ws_tile + mad24(rel, ws_sb, (unsigned)cch0 * 4u));
#pragma unroll
                  for (int j4 = 0; j4 < CPC; j4 += 4) *(f32x4*)(wp + j4) = f32x4{v[j4], v[j4 + 1], v[j4 + 2], v[j4 + 3]};
                }
                continue;
              }
              if (mt == 0 && it == 0 && !p.mx) {
                // common shift per channel for the whole wave: the value lane `cc` holds for its first
                // voxel (any finite value near the data works; shifted sums then simply add across lanes)
#pragma unroll
                for (int j = 0; j < CPC; ++j) sh[j] = __shfl(v[j], cc, 64);
              }
              if (inb) {
                const unsigned cb = (unsigned)cch0 * Elem<T>::SIZE;
                if (p.res) {
                  float f[CPC];
                  Elem<T>::unpack(*(const u32x4*)(res_tile + mad24(rel, res_sb, cb)), f);
#pragma unroll
                  for (int j = 0; j < CPC; ++j) v[j] += f[j];
                }
                if (p.mx) {
                  float f[CPC];
                  Elem<T>::unpack(*(const u32x4*)(mx_tile + mad24(rel, mx_sb, cb)), f);
#pragma unroll
                  for (int j = 0; j < CPC; ++j) {
                    float xh = (f[j] - mm[j]) * mr[j];
                    v[j] *= actg<ACT>(xh, p.act);
                    s0[j] += v[j];
                    s1[j] += v[j] * xh;
                  }
                } else {
#pragma unroll
                  for (int j = 0; j < CPC; ++j) { float d = v[j] - sh[j]; s0[j] += d; s1[j] += d * d; }
                }
                cnt += 1.f;
                if (!(I_DBG & 4)) *(u32x4*)(y_tile + mad24(rel, y_sb, cb)) = Elem<T>::pack(v);
              }
            }
          }
          if (p.partials && p.ksplit == 1) {
            // lanes with the same channel chunk (lane % OCH) hold plain partial sums: xor-shuffle adds
#pragma unroll
            for (int msk = OCH; msk < 64; msk <<= 1) {
              cnt += __shfl_xor(cnt, msk, 64);
#pragma unroll
              for (int j = 0; j < CPC; ++j) {
                s0[j] += __shfl_xor(s0[j], msk, 64);
                s1[j] += __shfl_xor(s1[j], msk, 64);
              }
            }
            if (lane < OCH) {
#pragma unroll
              for (int j = 0; j < CPC; ++j) {
                Moments a;
                if (p.mx) { a.n = 0.f; a.mean = s0[j]; a.m2 = s1[j]; }
                else a = moments_from_shifted(cnt, sh[j], s0[j], s1[j]);
                float* rr = red + ((wave * BN) + nt * 32 + cc * CPC + j) * 3;
                rr[0] = a.n; rr[1] = a.mean; rr[2] = a.m2;
              }
            }
          }
        }
      }
      CBIM_TICK(4);
      const bool more = last_plane && unit + 1 < n_units;
      if (more) halo_store();      // the next unit's halo replaces this one (nobody reads A any more)
      CBIM_TICK(5);
      if (more || tile_done) __syncthreads();   // halo visible; scratch reads done; `red` complete
      if (tile_done && p.partials && p.ksplit == 1 && tid < BN && co0 + tid < p.Cout) {
        // one record per (image, workgroup): the tile's moments join the workgroup's running record (a record per
        // TILE made the finalize kernel walk 4096 scattered 12-byte rows per channel at 128^3)
        const float* red = (const float*)(smem + red_base);
        Moments a = {0.f, 0.f, 0.f};
        for (int g = 0; g < NW; ++g) {
          const float* rr = red + (g * BN + tid) * 3;
          if (p.mx) { a.mean += rr[1]; a.m2 += rr[2]; }
          else { Moments b = {rr[0], rr[1], rr[2]}; a = moments_merge(a, b); }
        }
        if (cur.n != run_n) { flush_partial(run_n, run); run = Moments{0.f, 0.f, 0.f}; run_n = cur.n; }
        if (p.mx) { run.mean += a.mean; run.m2 += a.m2; }
        else run = moments_merge(run, a);
      }
      CBIM_TICK(5);
    }
    cur = nxt;
    advance(nxt);
  }
  if (p.partials && p.ksplit == 1 && tid < BN && co0 + tid < p.Cout) flush_partial(run_n, run);
#ifdef CBIM_IGEMM_PROF
  if (p.prof && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) p.prof[wave * 8 + i] = pacc[i];
#endif
}

// ---- split-K finish: sum the fp32 partials, then the same epilogue as the fused path ------------------
// grid = (row blocks, N); a thread keeps one 16-byte output chunk and strides over voxel rows.
static constexpr int FT = 256;
template <typename T, int ACT>
__global__ void __launch_bounds__(FT) k_splitk_finish(const float* __restrict__ ws, int ksplit, const void* res,
                                                      int64_t res_stride, const void* mx, int64_t mx_stride,
                                                      const float* __restrict__ m_stats, void* y, int64_t y_stride,
                                                      float* partials, int64_t S, int64_t rows_total, int C, int act,
                                                      int P) {
  constexpr int CPC = Elem<T>::CPC;
  const int cch = C / CPC, vlc = FT / cch;
  const int t = threadIdx.x, cc = t % cch, vl = t / cch;
  const int n = blockIdx.y, part = blockIdx.x;
  const bool active = vl < vlc;
  const int64_t per = (S + P - 1) / P;
  const int64_t v0 = (int64_t)part * per;
  int64_t v1 = v0 + per;
  if (v1 > S) v1 = S;
  float mm[CPC], mr[CPC], s0[CPC], s1[CPC], sh[CPC];
  float cnt = 0.f;
#pragma unroll
  for (int j = 0; j < CPC; ++j) { mm[j] = 0.f; mr[j] = 1.f; s0[j] = 0.f; s1[j] = 0.f; sh[j] = 0.f; }
  if (mx && m_stats && active) {
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      mm[j] = m_stats[((size_t)n * C + cc * CPC + j) * 2];
      mr[j] = m_stats[((size_t)n * C + cc * CPC + j) * 2 + 1];
    }
  }
  if (active) {
    for (int64_t v = v0 + vl; v < v1; v += vlc) {
      const size_t row = (size_t)n * S + v;
      float a[CPC];
#pragma unroll
      for (int j = 0; j < CPC; ++j) a[j] = 0.f;
#pragma unroll 4
      for (int z = 0; z < ksplit; ++z) {
        const float* wp = ws + ((size_t)z * rows_total + row) * C + (size_t)cc * CPC;
#pragma unroll
        for (int j4 = 0; j4 < CPC; j4 += 4) {
          f32x4 q4 = *(const f32x4*)(wp + j4);
          a[j4] += q4.x; a[j4 + 1] += q4.y; a[j4 + 2] += q4.z; a[j4 + 3] += q4.w;
        }
      }
      if (res) {
        float f[CPC];
        Elem<T>::unpack(ld_chunk<T>(res, row * res_stride + (size_t)cc * CPC), f);
#pragma unroll
        for (int j = 0; j < CPC; ++j) a[j] += f[j];
      }
      if (mx) {
        float f[CPC];
        Elem<T>::unpack(ld_chunk<T>(mx, row * mx_stride + (size_t)cc * CPC), f);
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          float xh = (f[j] - mm[j]) * mr[j];
          a[j] *= actg<ACT>(xh, act);
          s0[j] += a[j];
          s1[j] += a[j] * xh;
        }
      } else {
        if (cnt == 0.f) {
#pragma unroll
          for (int j = 0; j < CPC; ++j) sh[j] = a[j];
        }
#pragma unroll
        for (int j = 0; j < CPC; ++j) { float d = a[j] - sh[j]; s0[j] += d; s1[j] += d * d; }
      }
      cnt += 1.f;
      st_chunk<T>(y, row * y_stride + (size_t)cc * CPC, Elem<T>::pack(a));
    }
  }
  if (!partials) return;
  __shared__ float red[FT * 3 * 8];
#pragma unroll
  for (int j = 0; j < CPC; ++j) {
    Moments m;
    if (mx) { m.n = 0.f; m.mean = s0[j]; m.m2 = s1[j]; }
    else m = moments_from_shifted(cnt, sh[j], s0[j], s1[j]);
    red[(t * CPC + j) * 3 + 0] = m.n;
    red[(t * CPC + j) * 3 + 1] = m.mean;
    red[(t * CPC + j) * 3 + 2] = m.m2;
  }
  __syncthreads();
  for (int ch = t; ch < cch * CPC; ch += FT) {     // one thread per channel merges the voxel lanes in fixed order
    const int c2 = ch / CPC, j = ch % CPC;
    Moments acc = {0.f, 0.f, 0.f};
    for (int q = 0; q < vlc; ++q) {
      const float* r = red + ((q * cch + c2) * CPC + j) * 3;
      if (mx) { acc.mean += r[1]; acc.m2 += r[2]; }
      else { Moments b = {r[0], r[1], r[2]}; acc = moments_merge(acc, b); }
    }
    const size_t o = (((size_t)n * P + part) * C + c2 * CPC + j) * 3;
    partials[o] = acc.n; partials[o + 1] = acc.mean; partials[o + 2] = acc.m2;
  }
}

// ---- weight re-layout into B-fragment order ------------------------------------------------------------
// packed[nb][chunk][tap][kg][half][BN][CPC];  channel = chunk*KC + (2kg+half)*CPC + j, cout = nb*BN + nn.
// mode 0: K = Cin, N = Cout, value w[cout][cin][tap]; mode 1 (dgrad): K = Cout_fwd, N = Cin_fwd,
// value w[k_ch][n_ch][taps-1-tap]  (w is always the forward [Cout][Cin][taps] tensor).
// `w` rows (forward output channels) >= rows0 come from `w1` (row r - rows0): a Cout-concatenated convolution
// (conv1 | shortcut) is packed straight from its two parameter tensors; w1 == nullptr: everything is in `w`.
//
// One workgroup = one block of 32 forward output channels x KC input channels x all taps, transposed through LDS:
// the block is read as 32 contiguous runs of KC*taps floats (a thread-per-chunk version gathered every float at a
// stride of `taps` floats and re-fetched each cache line ~27 times: 778 us for the ResUNet's 19 M weights, 0.1 TB/s,
// 4.5 % of the training step) and leaves as runs of 32 (KC) sixteen-byte chunks = 512 contiguous bytes per (tap, k-slot),
// in BOTH layouts from the one staged copy.
// lo (bf16 only, round 5): pack the ROUNDING RESIDUE w - bf16(w) instead of w — the second image of a weight whose GEMM keeps
// fp32 accuracy (x . w = x . w_hi + x . w_lo: the token Linears outside SwinUNETR's autocast region, cbim_token_linear)
struct PackGeom { const float* w0; const float* w1; void* p0; void* p1; int rows0, Cout, Cin, taps, BN0, nch0, BN1, nch1; int lo; };

template <typename T>
__device__ __forceinline__ void pack_block(const PackGeom& g, int blk, unsigned char* smem) {
  constexpr int CPC = Elem<T>::CPC, ES = Elem<T>::SIZE;
  constexpr int KC = SLOTS * CPC;                 // input channels per block (32 bf16 / 16 f32): 64-byte LDS rows
  constexpr int G = 32 / CPC;                     // groups of CPC output channels in the block
  const int tid = threadIdx.x, taps = g.taps;
  const int n_nblk0 = (g.Cout + g.BN0 - 1) / g.BN0, n_nblk1 = (g.Cin + g.BN1 - 1) / g.BN1;
  const int ci_blks = n_nblk1 * g.BN1 / KC;
  const int co_b = blk / ci_blks, ci_b = blk - co_b * ci_blks;
  const int co0 = co_b * 32, ci0 = ci_b * KC;
  // ---- load: row co = one contiguous, 16-byte aligned run of KC*taps floats (Cin % 4 == 0); eight float4 per thread
  //      in flight per trip (one load per trip left the kernel latency bound); LDS element (tap, co_l, ci_l) ----------
  const int seg = KC * taps, seg4 = seg / 4;
  const unsigned mdiv = ((1u << 20) + (unsigned)taps - 1) / (unsigned)taps;    // idx / taps for idx < 2048, taps <= 64
  const unsigned mseg = ((1u << 22) + (unsigned)seg4 - 1) / (unsigned)seg4;    // e4 / seg4 for e4 < 32 * seg4 <= 16384
  const int n_ci = g.Cin - ci0 < KC ? g.Cin - ci0 : KC;                          // valid input channels of the block (<= 0: none)
  const int total4 = 32 * seg4;
  constexpr int UL = 8;
  for (int base = 0; base < total4; base += 256 * UL) {
    f32x4 v4[UL];
    int rowv[UL], idxv[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int e4 = base + u * 256 + tid;
      const int co_l = (int)(((unsigned)e4 * mseg) >> 22);
      const int idx = (e4 - co_l * seg4) * 4;
      rowv[u] = co_l; idxv[u] = idx;
      const int co = co0 + co_l;
      v4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e4 < total4 && co < g.Cout && n_ci > 0 && idx < n_ci * taps) {   // (n_ci * taps is a multiple of 4)
        const float* src = ((g.w1 && co >= g.rows0) ? g.w1 + (size_t)(co - g.rows0) * g.Cin * taps : g.w0 + (size_t)co * g.Cin * taps) +
                           (size_t)ci0 * taps;
        v4[u] = *(const f32x4*)(src + idx);
      }
    }
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      if (base + u * 256 + tid < total4) {
        const float vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = idxv[u] + j;
          const int ci_l = (int)(((unsigned)idx * mdiv) >> 20), tap = idx - ci_l * taps;
          unsigned char* cell = smem + (size_t)((tap * 32 + rowv[u]) * KC + ci_l) * ES;
          if (ES == 2) {
            const bf16_t hi = (bf16_t)pk_bf16(vv[j], 0.f);
            *(bf16_t*)cell = g.lo ? (bf16_t)pk_bf16(vv[j] - bf2f(hi), 0.f) : hi;
          } else *(float*)cell = vv[j];
        }
      }
    }
  }
  __syncthreads();
  // ---- forward layout: chunk (tap, k-slot, cout) = 16 contiguous LDS bytes ------------------------------------------
  if (g.p0 && ci_b < g.nch0) {
    const int nb = co0 / g.BN0, nn0 = co0 - nb * g.BN0;
    for (int c = tid; c < taps * 128; c += 256) {
      const int co_l = c & 31, slot = (c >> 5) & 3, tap = c >> 7;
      const u32x4 v = *(const u32x4*)(smem + (size_t)(tap * 32 + co_l) * 64 + slot * 16);
      const size_t ic = ((((size_t)(nb * g.nch0 + ci_b) * taps + tap) * KG + (slot >> 1)) * 2 + (slot & 1)) * g.BN0 + nn0 + co_l;
      *(u32x4*)((unsigned char*)g.p0 + ic * 16) = v;
    }
  }
  // ---- dgrad layout: K = forward cout (groups of CPC), N = forward cin, taps flipped --------------------------------
  if (g.p1) {
    for (int c = tid; c < taps * G * KC; c += 256) {
      const int ci_l = c % KC, gg = (c / KC) % G, tap = c / (KC * G);
      const int k0 = co0 + gg * CPC, q1 = k0 / KC, slot = (k0 - q1 * KC) / CPC;
      const int ci = ci0 + ci_l, nb1 = ci / g.BN1, nn1 = ci - nb1 * g.BN1;
      if (q1 >= g.nch1 || nb1 >= n_nblk1) continue;
      u32x4 v;
      if (ES == 2) {
        unsigned h[CPC];
#pragma unroll
        for (int j = 0; j < CPC; ++j) h[j] = *(const bf16_t*)(smem + (size_t)((tap * 32 + gg * CPC + j) * KC + ci_l) * 2);
        v = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4 % CPC] | (h[5 % CPC] << 16), h[6 % CPC] | (h[7 % CPC] << 16)};
      } else {
        unsigned h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = *(const unsigned*)(smem + (size_t)((tap * 32 + gg * CPC + (j % CPC)) * KC + ci_l) * 4);
        v = u32x4{h[0], h[1], h[2], h[3]};
      }
      const size_t ic = ((((size_t)(nb1 * g.nch1 + q1) * taps + (taps - 1 - tap)) * KG + (slot >> 1)) * 2 + (slot & 1)) * g.BN1 + nn1;
      *(u32x4*)((unsigned char*)g.p1 + ic * 16) = v;
    }
  }
}
// blocks of one weight: (32-cout blocks incl. the n-block padding of the forward layout) x (KC-cin blocks incl. that of
// the dgrad layout)
// n-block width of a layer: 64 channels, 32 for narrow layers and for 5x5 planes (25 taps per weight stage: LDS)
static int bn_of(int Ndim, int kplane) { return (Ndim <= 32 || kplane > 9) ? 32 : 64; }
static int pack_blocks(int dtype, int Cout, int Cin, int kplane) {
  const int KC = RB / (dtype == CBIM_BF16 ? 2 : 4);
  const int BN0 = bn_of(Cout, kplane), BN1 = bn_of(Cin, kplane);
  return (((Cout + BN0 - 1) / BN0) * BN0 / 32) * (((Cin + BN1 - 1) / BN1) * BN1 / KC);
}

// One launch packs the forward layout, the dgrad layout, or both (p0/p1 may be null).
template <typename T>
__global__ void __launch_bounds__(256) k_pack_weights(PackGeom g) {
  CBIM_DYN_SMEM(smem);
  pack_block<T>(g, (int)blockIdx.x, smem);
}

// Every convolution weight of a model in ONE launch (the weights change once per optimizer step: 34 pack launches +
// the torch.cat of each conv1|shortcut pair were 0.5 ms of a 19 ms ResUNet step).  The table lives in device memory;
// blockIdx.x -> item by binary search over the items' first block.
__global__ void __launch_bounds__(256) k_pack_weights_table(const cbim_pack_item* __restrict__ items, int n_items) {
  CBIM_DYN_SMEM(smem);
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {   // last item whose block_begin <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const cbim_pack_item it = items[lo];
  PackGeom g;
  g.w0 = it.w0; g.w1 = it.w1; g.p0 = it.p0; g.p1 = it.p1; g.rows0 = it.rows0; g.Cout = it.Cout; g.Cin = it.Cin;
  g.taps = it.taps; g.BN0 = it.BN0; g.nch0 = it.nch0; g.BN1 = it.BN1; g.nch1 = it.nch1; g.lo = 0;
  const int blk = (int)blockIdx.x - it.block_begin;
  if (it.dtype == CBIM_BF16) pack_block<bf16_tag>(g, blk, smem);
  else pack_block<float>(g, blk, smem);
}

struct TileCfg { int MT, NTL, tD, tH, lgH, nth; };

static TileCfg pick_cfg(const cbim_conv_desc* d) {
  TileCfg c;
  c.NTL = bn_of(d->Cout, d->kH * d->kW) / 32;
  int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  // 8x8x8 tiles (BM = 512) when that still leaves >= 2 tiles per CU; otherwise 4x8x8 (BM = 256)
  c.MT = (S >= 262144 && d->Do >= 8 && d->Ho >= 8) ? 2 : 1;
  c.tH = 8; c.lgH = 3;
  c.tD = c.MT == 2 ? 8 : 4;
  c.nth = 512;
  if (d->kH * d->kW > 9) {   // 5x5 planes (VNet): 25 taps per weight stage -> 32-cout blocks (bn_of) on 4x8x8 tiles to fit 160 KB of LDS
    c.MT = 1; c.tD = 4;
  }
  // two 256-thread workgroups per CU on 4x8x8 tiles (phase overlap): 3x3x3, Cout <= 32 layers at full resolution
  static const int half_on = 0;
  if (half_on && c.MT == 2 && c.NTL == 1 && d->dtype == CBIM_BF16 && d->kD == 3 && d->kH == 3 && d->kW == 3 &&
      d->act == CBIM_ACT_RELU) {
    c.nth = 256;
    c.tD = 4;
  }
  return c;
}

static int elem_size(int dtype) { return dtype == CBIM_BF16 ? 2 : 4; }

// split-K factor: low-resolution layers have too few (tile, n-block) pairs to fill 256 CUs; the Cin
// chunks are then shared out over blockIdx.z and summed by k_splitk_finish.
static int pick_ksplit(const cbim_conv_desc* d, const TileCfg& c) {
  int KC = RB / elem_size(d->dtype);
  int n_chunks = (d->Cin + KC - 1) / KC;
  int64_t tiles = (int64_t)d->N * ((d->Do + c.tD - 1) / c.tD) * ((d->Ho + c.tH - 1) / c.tH) * ((d->Wo + 7) / 8);
  int BN = 32 * c.NTL;
  int64_t wgs = tiles * ((d->Cout + BN - 1) / BN);
  // the finish kernel keeps one 16-byte output chunk per thread (FT = 256 threads)
  static const int target = 192;
  if (wgs >= target / 2 || n_chunks < 2 || d->Cout / (d->dtype == CBIM_BF16 ? 8 : 4) > 256) return 1;
  int64_t s = (target + wgs - 1) / wgs;
  if (s > n_chunks) s = n_chunks;
  if (s > 16) s = 16;
  return (int)s;
}
static int finish_parts(int64_t S) {
  int64_t p = (S + 15) / 16;
  if (p > 512) p = 512;
  if (p < 1) p = 1;
  return (int)p;
}
static int kc_of(int dtype) { return RB / elem_size(dtype); }
// persistent grid: about one workgroup per CU (256 CUs), never more workgroups than tiles
static int64_t igemm_grid_x(const cbim_conv_desc* d, const TileCfg& c) {
  const int64_t n_tiles = (int64_t)d->N * ((d->Do + c.tD - 1) / c.tD) * ((d->Ho + c.tH - 1) / c.tH) * ((d->Wo + 7) / 8);
  const int BN = 32 * c.NTL, n_nblk = (d->Cout + BN - 1) / BN;
  int64_t G = (c.nth == 256 ? 512 : 256) / n_nblk;
  if (G < 1) G = 1;
  if (G > n_tiles) G = n_tiles;
  return G;
}

}  // namespace cbim

using namespace cbim;

static int validate(const cbim_conv_desc* d) {
  CBIM_CHECK(d != nullptr, CBIM_EINVAL, "null conv descriptor");
  CBIM_CHECK(d->dtype == CBIM_F32 || d->dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", d->dtype);
  int cpc = d->dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(d->Cin > 0 && d->Cin % cpc == 0, CBIM_EUNSUPPORTED, "conv Cin %d is not a multiple of %d", d->Cin, cpc);
  CBIM_CHECK(d->Cout > 0 && d->Cout % cpc == 0, CBIM_EUNSUPPORTED, "conv Cout %d is not a multiple of %d", d->Cout, cpc);
  CBIM_CHECK(d->kD >= 1 && d->kH >= 1 && d->kW >= 1 && d->kD * d->kH * d->kW <= 64, CBIM_EUNSUPPORTED, "kernel extent unsupported");
  CBIM_CHECK(d->N >= 1 && d->Do >= 1 && d->Ho >= 1 && d->Wo >= 1, CBIM_EINVAL, "empty conv output");
  return 0;
}

extern "C" size_t cbim_conv3d_packed_bytes(const cbim_conv_desc* d, int mode) {
  if (!d) return 0;
  int Kdim = mode == 0 ? d->Cin : d->Cout, Ndim = mode == 0 ? d->Cout : d->Cin;
  int BN = bn_of(Ndim, d->kH * d->kW);
  int n_nblk = (Ndim + BN - 1) / BN;
  int KC = kc_of(d->dtype);
  int n_chunks = (Kdim + KC - 1) / KC;
  int taps = d->kD * d->kH * d->kW;
  return (size_t)n_nblk * n_chunks * taps * KG * 2 * BN * 16;
}

static int pack_smem_attr() {
#ifndef CBIM_EMU
  static bool done = false;
  if (!done) {
    hipError_t e1 = hipFuncSetAttribute((const void*)k_pack_weights<bf16_tag>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute((const void*)k_pack_weights<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e3 = hipFuncSetAttribute((const void*)k_pack_weights_table, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute (pack)");
    done = true;
  }
#endif
  return CBIM_OK;
}

static int pack_launch(const cbim_conv_desc* d, const float* w, void* p0, void* p1, void* stream, int lo = 0) {
  if (int e = validate(d)) return e;
  if (int e = pack_smem_attr()) return e;
  const int KC = kc_of(d->dtype), taps = d->kD * d->kH * d->kW;
  PackGeom g;
  g.w0 = w; g.w1 = nullptr; g.p0 = p0; g.p1 = p1; g.rows0 = d->Cout; g.Cout = d->Cout; g.Cin = d->Cin; g.taps = taps;
  g.BN0 = bn_of(d->Cout, d->kH * d->kW); g.BN1 = bn_of(d->Cin, d->kH * d->kW);
  g.nch0 = (d->Cin + KC - 1) / KC; g.nch1 = (d->Cout + KC - 1) / KC; g.lo = lo;
  const int blocks = pack_blocks(d->dtype, d->Cout, d->Cin, d->kH * d->kW);
  const size_t smem = (size_t)taps * 2048;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CBIM_BF16) CBIM_LAUNCH((k_pack_weights<bf16_tag>), dim3((unsigned)blocks), dim3(256), smem, st, g);
  else CBIM_LAUNCH((k_pack_weights<float>), dim3((unsigned)blocks), dim3(256), smem, st, g);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_conv3d_pack_weights(const cbim_conv_desc* d, int mode, const float* w, void* packed,
                                        void* stream) {
  CBIM_CHECK(mode == 0 || mode == 1, CBIM_EINVAL, "bad pack mode");
  CBIM_CHECK(w && packed, CBIM_EINVAL, "null argument");
  return mode == 0 ? pack_launch(d, w, packed, nullptr, stream) : pack_launch(d, w, nullptr, packed, stream);
}

extern "C" int cbim_conv3d_pack_weights_both(const cbim_conv_desc* d, const float* w, void* packed_fwd,
                                             void* packed_dgrad, void* stream) {
  CBIM_CHECK(w && packed_fwd && packed_dgrad, CBIM_EINVAL, "null argument");
  return pack_launch(d, w, packed_fwd, packed_dgrad, stream);
}

extern "C" int cbim_conv3d_pack_weights_lo(const cbim_conv_desc* d, const float* w, void* packed_fwd, void* packed_dgrad,
                                           void* stream) {
  CBIM_CHECK(d && d->dtype == CBIM_BF16, CBIM_EINVAL, "residue images exist for bf16 only");
  CBIM_CHECK(w && (packed_fwd || packed_dgrad), CBIM_EINVAL, "null argument");
  return pack_launch(d, w, packed_fwd, packed_dgrad, stream, 1);
}

extern "C" int cbim_conv3d_pack_item_fill(const cbim_conv_desc* d, const float* w0, const float* w1, int rows0,
                                          void* packed_fwd, void* packed_dgrad, int block_begin, cbim_pack_item* out) {
  if (int e = validate(d)) return e;
  CBIM_CHECK(w0 && out && (packed_fwd || packed_dgrad), CBIM_EINVAL, "null argument");
  CBIM_CHECK(!w1 || (rows0 > 0 && rows0 < d->Cout), CBIM_EINVAL, "second weight tensor: bad split %d", rows0);
  const int KC = kc_of(d->dtype), es = elem_size(d->dtype);
  out->w0 = w0; out->w1 = w1; out->p0 = packed_fwd; out->p1 = packed_dgrad;
  out->rows0 = w1 ? rows0 : d->Cout; out->Cout = d->Cout; out->Cin = d->Cin; out->taps = d->kD * d->kH * d->kW;
  out->BN0 = bn_of(d->Cout, d->kH * d->kW); out->BN1 = bn_of(d->Cin, d->kH * d->kW);
  out->nch0 = (d->Cin + KC - 1) / KC; out->nch1 = (d->Cout + KC - 1) / KC;
  out->total0 = packed_fwd ? (int64_t)(cbim_conv3d_packed_bytes(d, 0) / es) : 0;
  out->total1 = packed_dgrad ? (int64_t)(cbim_conv3d_packed_bytes(d, 1) / es) : 0;
  const int nb = pack_blocks(d->dtype, d->Cout, d->Cin, d->kH * d->kW);   // one workgroup per 32 couts x KC cins (k_pack_weights_table)
  out->block_begin = block_begin; out->n_blocks = (int)nb; out->dtype = d->dtype;
  return CBIM_OK;
}

extern "C" int cbim_conv3d_pack_weights_table(const cbim_pack_item* items_dev, int n_items, int total_blocks,
                                              int max_taps, void* stream) {
  CBIM_CHECK(items_dev && n_items > 0 && total_blocks > 0, CBIM_EINVAL, "empty pack table");
  CBIM_CHECK(max_taps >= 1 && max_taps <= 64, CBIM_EUNSUPPORTED, "pack table: %d taps", max_taps);
  if (int e = pack_smem_attr()) return e;
  CBIM_LAUNCH(k_pack_weights_table, dim3((unsigned)total_blocks), dim3(256), (size_t)max_taps * 2048, (hipStream_t)stream, items_dev, n_items);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_conv3d_tile_config(const cbim_conv_desc* d, int out[4]) {
  CBIM_CHECK(d && out, CBIM_EINVAL, "null argument");
  if (cbim_conv_r32_eligible(d, nullptr, 0, nullptr, nullptr, nullptr)) {   // (when the call has one input tensor) conv_r32.hip: 4 m-tiles per wave
    out[0] = 4; out[1] = 1; out[2] = 8; out[3] = 8;
    return CBIM_OK;
  }
  TileCfg c = pick_cfg(d);
  out[0] = c.MT; out[1] = c.NTL; out[2] = c.tD; out[3] = c.tH;
  return CBIM_OK;
}

// bf16 3x3x3 layers in multiples of 32 channels with too few tiles for cbim_conv_r32_eligible: k_conv3_rw's split-K form
// (conv_rw.hip) instead of k_conv_igemm's
static int rw_split_of(const cbim_conv_desc* d) {
  if (d->dtype != CBIM_BF16 || d->kD != 3 || d->kH != 3 || d->kW != 3 || d->Cin % 32 != 0 || d->Cout % 32 != 0) return 0;
  if (d->Do < 8 || d->Ho < 8 || d->Wo < 8 || d->Do != d->Di || d->Ho != d->Hi || d->Wo != d->Wi) return 0;
  return cbim_conv_rw_ksplit(d);
}

extern "C" size_t cbim_conv3d_igemm_workspace(const cbim_conv_desc* d) {
  if (!d) return 0;
  TileCfg c = pick_cfg(d);
  int ks = pick_ksplit(d, c);
  const int ks_rw = rw_split_of(d);
  if (ks_rw > ks) ks = ks_rw;
  if (ks <= 1) return 0;
  return (size_t)ks * d->N * d->Do * d->Ho * d->Wo * d->Cout * sizeof(float);
}

extern "C" int cbim_conv3d_num_tiles(const cbim_conv_desc* d) {
  if (!d) return 0;
  TileCfg c = pick_cfg(d);
  if (pick_ksplit(d, c) > 1) return finish_parts((int64_t)d->Do * d->Ho * d->Wo);
  int64_t g = igemm_grid_x(d, c);   // one record per (image, persistent workgroup)
  if (rw_split_of(d) > 1 && finish_parts((int64_t)d->Do * d->Ho * d->Wo) > g) g = finish_parts((int64_t)d->Do * d->Ho * d->Wo);
  // the same layer may run on conv_r32.hip (8x8x8 tiles whatever pick_cfg says): room for either grid
  if (cbim_conv_r32_eligible(d, nullptr, 0, nullptr, nullptr, nullptr)) {
    if (cbim_conv_r32_grid(d) > g) g = cbim_conv_r32_grid(d);
    if (cbim_conv_rw_grid(d) > g) g = cbim_conv_rw_grid(d);
  }
  if (cbim_conv_pw_records(d) > g) g = cbim_conv_pw_records(d);   // pointwise layers: conv_pw.hip's strips
  if (cbim_conv_rw48_takes(d) && cbim_conv_rw_grid(d) > g) g = cbim_conv_rw_grid(d);   // k_conv3_rw48
  return (int)g;
}

template <typename T, int MT, int NTL, int ACT, bool K3, int NTH = 512>
static int launch_igemm(const IgemmParams& p, dim3 grid, size_t smem, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv_igemm<T, MT, NTL, ACT, K3, NTH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  CBIM_LAUNCH((k_conv_igemm<T, MT, NTL, ACT, K3, NTH>), grid, dim3(NTH), smem, st, p);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "conv igemm launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

template <typename T, int ACT, bool K3>
static int dispatch_tiles(const TileCfg& c, const IgemmParams& p, dim3 grid, size_t smem, hipStream_t st) {
  if (c.nth == 256) {
    if constexpr (ACT == CBIM_ACT_RELU && K3 && sizeof(typename Elem<T>::type) == 2)
      return launch_igemm<T, 2, 1, ACT, K3, 256>(p, grid, smem, st);
    CBIM_CHECK(false, CBIM_EUNSUPPORTED, "256-thread igemm variant not instantiated for this configuration");
  }
  if (c.MT == 2 && c.NTL == 1) return launch_igemm<T, 2, 1, ACT, K3>(p, grid, smem, st);
  if (c.MT == 2 && c.NTL == 2) return launch_igemm<T, 2, 2, ACT, K3>(p, grid, smem, st);
  if (c.MT == 1 && c.NTL == 1) return launch_igemm<T, 1, 1, ACT, K3>(p, grid, smem, st);
  return launch_igemm<T, 1, 2, ACT, K3>(p, grid, smem, st);
}

// activation -> compile-time instantiation (ReLU: UNet/MedFormer; none: raw and 1x1 projections; LeakyReLU: the
// monai blocks of SwinUNETR); the rarely used ones share a runtime-switch instantiation
template <typename T>
static int dispatch_act(int act, bool k3, const TileCfg& c, const IgemmParams& p, dim3 grid, size_t smem, hipStream_t st) {
  switch (act) {
    case CBIM_ACT_RELU:
      return k3 ? dispatch_tiles<T, CBIM_ACT_RELU, true>(c, p, grid, smem, st) : dispatch_tiles<T, CBIM_ACT_RELU, false>(c, p, grid, smem, st);
    case CBIM_ACT_NONE:
      return k3 ? dispatch_tiles<T, CBIM_ACT_NONE, true>(c, p, grid, smem, st) : dispatch_tiles<T, CBIM_ACT_NONE, false>(c, p, grid, smem, st);
    case CBIM_ACT_LRELU:
      return k3 ? dispatch_tiles<T, CBIM_ACT_LRELU, true>(c, p, grid, smem, st) : dispatch_tiles<T, CBIM_ACT_LRELU, false>(c, p, grid, smem, st);
    default:
      return dispatch_tiles<T, -1, false>(c, p, grid, smem, st);
  }
}

// which kernel the last cbim_conv3d_igemm call of this thread launched: 0 = k_conv_igemm, 1 = k_conv3_r32, 2 = k_conv3_rw, 3 = k_conv3_rw split-K + finish, 4 = k_conv_pw, 5 = k_conv3_rw48 (profiling labels)
static thread_local int g_last_conv_kernel = 0;
extern "C" int cbim_conv3d_last_kernel(void) { return g_last_conv_kernel; }

// the finish pass of a split-K convolution (k_conv_igemm's or k_conv3_rw's): fixed-order sum of the `ksplit` fp32 slabs, then
// the fused epilogue (residual, act' mask, statistics / InstanceNorm-backward sums into `P` records per image, store)
static int launch_finish(const cbim_conv_desc* d, const void* workspace, int ksplit, const void* res, int64_t res_stride,
                         const void* mask_x, int64_t mask_stride, const float* mask_stats, void* y, int64_t y_stride,
                         float* partials, int P_records, hipStream_t st) {
  const bool relu = d->act == CBIM_ACT_RELU;
  int cpc = d->dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(d->Cout / cpc <= FT, CBIM_EUNSUPPORTED, "split-K finish: Cout %d too large", d->Cout);
  const int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  const int P = finish_parts(S);
  CBIM_CHECK(!partials || P_records == P, CBIM_EINVAL, "split-K finish: %d partial records per image, %d parts", P_records, P);
  dim3 fg((unsigned)P, (unsigned)d->N);
  if (d->dtype == CBIM_BF16) {
    if (relu) CBIM_LAUNCH((k_splitk_finish<bf16_tag, CBIM_ACT_RELU>), fg, dim3(FT), 0, st, (const float*)workspace, ksplit, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials, S, S * d->N, d->Cout, d->act, P);
    else CBIM_LAUNCH((k_splitk_finish<bf16_tag, -1>), fg, dim3(FT), 0, st, (const float*)workspace, ksplit, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials, S, S * d->N, d->Cout, d->act, P);
  } else {
    if (relu) CBIM_LAUNCH((k_splitk_finish<float, CBIM_ACT_RELU>), fg, dim3(FT), 0, st, (const float*)workspace, ksplit, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials, S, S * d->N, d->Cout, d->act, P);
    else CBIM_LAUNCH((k_splitk_finish<float, -1>), fg, dim3(FT), 0, st, (const float*)workspace, ksplit, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials, S, S * d->N, d->Cout, d->act, P);
  }
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_conv3d_igemm(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2,
                                 int64_t x2_stride, int cin_split,
                                 const float* in_stats, const void* w_packed, const void* res,
                                 int64_t res_stride, const void* mask_x, int64_t mask_stride,
                                 const float* mask_stats, void* y, int64_t y_stride, float* partials,
                                 void* workspace, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  CBIM_CHECK(x && w_packed && y, CBIM_EINVAL, "null tensor");
  // mask_x without statistics: the mask tensor is the caller's materialised a = relu(IN(x)) (act'(xh) = [a > 0], xh = a
  // wherever the mask is open) — defined for ReLU only
  const bool rw48 = cbim_conv_rw48_eligible(d, x, x_stride, x2, x2_stride, cin_split, in_stats, mask_x, mask_stats);
  CBIM_CHECK(!mask_x || mask_stats || d->act == CBIM_ACT_RELU || rw48, CBIM_EINVAL,
             "mask_x without mask_stats (an activated mask tensor) needs act = ReLU (LeakyReLU: k_conv3_rw48 layers only)");
  g_last_conv_kernel = 0;
  if (rw48) {   // round 6: 48-channel-granular layers (conv_rw.hip k_conv3_rw48)
    g_last_conv_kernel = 5;
    return cbim_conv_rw_launch(d, x, x_stride, x2, x2_stride, cin_split, w_packed, res, res_stride, mask_x, mask_stride, y, y_stride, partials,
                               stream);
  }
  if (cbim_conv_pw_eligible(d, x_stride, x2, res_stride, mask_stride, y_stride, res, mask_x, mask_stats)) {   // 1x1x1: conv_pw.hip (round 4)
    g_last_conv_kernel = 4;
    return cbim_conv_pw_launch(d, x, x_stride, in_stats, w_packed, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials,
                               cbim_conv3d_num_tiles(d), stream);
  }
  if (cbim_conv_r32_eligible(d, x2, cin_split, in_stats, res, mask_x)) {  // channels in multiples of 32 at high resolution: weights in registers
    if (cbim_conv_rw_eligible(d, x, x_stride, x2, x2_stride, in_stats, mask_x, mask_stats)) {   // round 4: conv_rw.hip
      g_last_conv_kernel = 2;
      return cbim_conv_rw_launch(d, x, x_stride, x2, x2_stride, cin_split, w_packed, res, res_stride, mask_x, mask_stride, y, y_stride, partials,
                                 stream);
    }
    g_last_conv_kernel = 1;
    return cbim_conv_r32_launch(d, x, x_stride, x2, x2_stride, cin_split, in_stats, w_packed, res, res_stride, mask_x, mask_stride, mask_stats, y,
                                y_stride, partials, stream);
  }
  if (rw_split_of(d) > 1 && (!x2 || (cin_split > 0 && cin_split < d->Cin && cin_split % 32 == 0)) &&
      cbim_conv_rw_eligible(d, x, x_stride, x2, x2_stride, in_stats, mask_x, mask_stats) &&
      (!partials || cbim_conv3d_num_tiles(d) == finish_parts((int64_t)d->Do * d->Ho * d->Wo))) {
    // low-resolution layer: k_conv3_rw over slices of the Cin chunks + the finish pass (round 4)
    const int ks = rw_split_of(d);
    const size_t need = (size_t)ks * d->N * d->Do * d->Ho * d->Wo * d->Cout * sizeof(float);
    CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "conv split-K workspace %zu < %zu", ws_bytes, need);
    const int P_records = cbim_conv3d_num_tiles(d);
    g_last_conv_kernel = 3;
    if (int rc = cbim_conv_rw_split_launch(d, x, x_stride, x2, x2_stride, cin_split, w_packed, (float*)workspace, stream)) return rc;
    return launch_finish(d, workspace, ks, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials, P_records,
                         (hipStream_t)stream);
  }
  TileCfg c = pick_cfg(d);
  IgemmParams p;
  p.x = x; p.x_stride = x_stride; p.in_stats = in_stats; p.w = w_packed;
  p.x2 = x2; p.x2_stride = x2_stride; p.cin_split = cin_split;
  CBIM_CHECK(!x2 || (cin_split > 0 && cin_split < d->Cin && cin_split % kc_of(d->dtype) == 0), CBIM_EUNSUPPORTED,
             "second input: split %d must be a multiple of the %d-channel chunk", cin_split, kc_of(d->dtype));
  p.res = res; p.res_stride = res_stride; p.mx = mask_x; p.mx_stride = mask_stride; p.m_stats = mask_stats;
  p.y = y; p.y_stride = y_stride; p.partials = partials;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin;
  p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.kD = d->kD; p.kH = d->kH; p.kW = d->kW; p.pD = d->pD; p.pH = d->pH; p.pW = d->pW; p.act = d->act;
  p.tD = c.tD; p.tH = c.tH; p.lgH = c.lgH;
  p.tiles_d = (d->Do + c.tD - 1) / c.tD; p.tiles_h = (d->Ho + c.tH - 1) / c.tH; p.tiles_w = (d->Wo + 7) / 8;
  p.hD = c.tD + d->kD - 1; p.hH = c.tH + d->kH - 1; p.hW = 8 + d->kW - 1;
  const int nw = c.nth / 64, uh = c.nth == 256 ? 10 : UH;
  CBIM_CHECK(p.hD * p.hH * p.hW * SLOTS <= uh * c.nth, CBIM_EUNSUPPORTED, "halo of %d rows too large", p.hD * p.hH * p.hW);
  {
    // 32-bit halo byte offsets from two 24-bit multiplies (k_conv_igemm::halo_load)
    const int64_t box_rows = (int64_t)p.hD * d->Hi * d->Wi;
    const int64_t max_stride_b = (int64_t)(x2 && x2_stride > x_stride ? x2_stride : x_stride) * elem_size(d->dtype);
    CBIM_CHECK(box_rows < (1 << 24) && max_stride_b < (1 << 24) && box_rows * max_stride_b < ((int64_t)1 << 32) &&
               p.hD < 256 && p.hH < 256, CBIM_EUNSUPPORTED,
               "conv input plane %dx%d with row stride %lld B exceeds the 32-bit halo addressing", d->Hi, d->Wi, (long long)max_stride_b);
    // same for the epilogue: byte offsets relative to the tile's first output row
    const int64_t tile_rows = (int64_t)c.tD * d->Ho * d->Wo;
    int64_t so = y_stride * elem_size(d->dtype);
    if (res && res_stride * elem_size(d->dtype) > so) so = res_stride * elem_size(d->dtype);
    if (mask_x && mask_stride * elem_size(d->dtype) > so) so = mask_stride * elem_size(d->dtype);
    if ((int64_t)d->Cout * 4 > so) so = (int64_t)d->Cout * 4;
    CBIM_CHECK(tile_rows < (1 << 24) && so < (1 << 24) && tile_rows * so < ((int64_t)1 << 32), CBIM_EUNSUPPORTED,
               "conv output plane %dx%d with row stride %lld B exceeds the 32-bit epilogue addressing", d->Ho, d->Wo, (long long)so);
  }
  p.mHW = ((1u << 20) + (unsigned)(p.hH * p.hW) - 1) / (unsigned)(p.hH * p.hW);
  p.mW = ((1u << 20) + (unsigned)p.hW - 1) / (unsigned)p.hW;
  int KC = kc_of(d->dtype);
  p.n_chunks = (d->Cin + KC - 1) / KC;
  p.taps = d->kD * d->kH * d->kW;
  p.dbg = 0;
  int BN = 32 * c.NTL;
  size_t smem = (size_t)p.hD * p.hH * p.hW * RB + 2 * (size_t)d->kH * d->kW * KG * 2 * BN * 16 +
                (size_t)nw * BN * 3 * sizeof(float) + 64 * 2 * sizeof(float) +
                ((size_t)d->kH * d->kW * KG * 2 * BN * 16 < (size_t)nw * 4096 ? (size_t)nw * 4096 : 0);
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "conv tile needs %zu B of LDS", smem);
  int64_t n_tiles = (int64_t)d->N * p.tiles_d * p.tiles_h * p.tiles_w;
  int n_nblk = (d->Cout + BN - 1) / BN;
  const int64_t G = igemm_grid_x(d, c);
  (void)n_tiles;
  p.P = cbim_conv3d_num_tiles(d);
  const int P_records = p.P;
  p.ksplit = pick_ksplit(d, c);
  p.ws = (float*)workspace;
  if (p.ksplit > 1) {
    size_t need = cbim_conv3d_igemm_workspace(d);
    CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "conv split-K workspace %zu < %zu", ws_bytes, need);
  }
  dim3 grid((unsigned)G, (unsigned)n_nblk, (unsigned)p.ksplit);
  hipStream_t st = (hipStream_t)stream;
  const bool relu = d->act == CBIM_ACT_RELU;
#ifdef CBIM_IGEMM_PROF
  static unsigned long long* prof_dev = nullptr;
  if (!prof_dev) (void)hipMalloc((void**)&prof_dev, 64 * sizeof(unsigned long long));
  (void)hipMemsetAsync(prof_dev, 0, 64 * sizeof(unsigned long long), st);
  p.prof = prof_dev;
  struct ProfDump {
    unsigned long long* dev; hipStream_t st; const cbim_conv_desc* d;
    ~ProfDump() {
      unsigned long long h[64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
      static const char* nm[8] = {"dma_issue", "mainloop", "wait_vm", "barrier1", "epilogue", "halo_store+barrier2+partials", "halo_load", "loop_top"};
      fprintf(stderr, "[igemm prof] %d->%d @%d:", d->Cin, d->Cout, d->Do);
      for (int i = 0; i < 8; ++i) {
        unsigned long long s = 0;
        for (int w = 0; w < 8; ++w) s += h[w * 8 + i];
        fprintf(stderr, " %s %llu", nm[i], s / 8);
      }
      fprintf(stderr, "\n");
    }
  } prof_dump{prof_dev, st, d};
#endif
  if (p.ksplit > 1) {
    // main kernel writes raw partials; the finish kernel owns residual / mask / statistics / store
    IgemmParams q = p;
    q.res = nullptr; q.mx = nullptr; q.partials = nullptr;
    static const bool k3s_on = true;
    const bool k3s = k3s_on && d->kH == 3 && d->kW == 3 && c.tH == 8;
    int rc = d->dtype == CBIM_BF16 ? dispatch_act<bf16_tag>(d->act, k3s, c, q, grid, smem, st)
                                   : dispatch_act<float>(d->act, k3s, c, q, grid, smem, st);
    if (rc) return rc;
    return launch_finish(d, workspace, p.ksplit, res, res_stride, mask_x, mask_stride, mask_stats, y, y_stride, partials, P_records, st);
  }
  static const bool k3_on = true;
  const bool k3 = k3_on && d->kH == 3 && d->kW == 3 && c.tH == 8;   // hW == hH == 10: compile-time tap offsets
  return d->dtype == CBIM_BF16 ? dispatch_act<bf16_tag>(d->act, k3, c, p, grid, smem, st)
                               : dispatch_act<float>(d->act, k3, c, p, grid, smem, st);
}

CBIM_DEFINE_WARM(igemm)
