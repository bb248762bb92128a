// conv_rw.hip — 3x3x3 convolution (forward and dgrad), bf16, channel counts in multiples of 32: the round-4 form of the
// "weights in registers" kernel (conv_r32.hip), rebuilt around what tools/ubench/lds_dma.hip measured on the MI355X
// (profiles/r04_a_lds_dma.txt):
//   * the loop skeleton of k_conv3_r32 — 27 weight fragments in registers, 8 n-tiles stacked along D, a ring of fragment
//     reads, LDS-DMA of the next halo, one barrier per tile — runs at 58-64 % of the bf16 matrix peak (the bare MFMA stream at
//     80 %: the chip clocks to ~1.9 GHz under it); k_conv3_r32 reached 36-38 %.  Its ISA shows why: every LDS-DMA piece was
//     its own basic block of ~100 scalar / vector instructions (position decode from an LDS table, three range tests as
//     exec-mask branches, a 64-bit tile origin multiplied out per piece), eight to ten of them per tile and wave, none of
//     which the scheduler could move under the MFMAs;
//   * one LDS-DMA piece costs its issuing wave ~130 cycles whatever is done about it, and a wave cannot have more than one in
//     issue (one wave alone: 5 B/clk/CU, eight waves: 30 B/clk/CU from L2), so the pieces are dealt evenly to seven waves and
//     their address arithmetic has to be (almost) nothing.
// What changed:
//   * halo pieces are PLANE-major: piece j of halo plane p = rows 16 j .. 16 j + 15 of the plane's 100 (the seventh piece
//     re-covers rows 84..99), wave w fetches piece w of every plane.  A lane's (h, w) position inside the box, its swizzled
//     16-byte slot and so its source offset are then the SAME for all ten pieces: one VGPR, computed once per kernel;
//   * the pieces are `buffer_load_dwordx4 ... offen lds`: the tile / plane origin travels in the scalar offset, the lane
//     offset in the vector offset, and zero padding is the buffer range check — a lane outside the tensor gets the offset
//     0x80000000 (out of range: zeros land in LDS, checked on the hardware), a plane outside the tensor a descriptor with
//     num_records = 0.  Per tile: two compares and a select; per piece: three scalar instructions and the load.  No zero page,
//     no branch, no table in LDS;
//   * WIDE workgroups (HP = 2): a workgroup owns 64 output channels, a wave 16 of them over HALF the tile (16 n-tiles, 64
//     accumulator registers): one halo, one weight stream and one barrier per 2 x the MFMA work (Cout multiples of 64);
//   * per-lane statistics sums live in LDS (32 KiB: [4][512 lanes] 16-byte cells), added to once per tile — neither 24
//     registers through the MFMA phase (spills in the multi-chunk kernels) nor a 5-round butterfly per tile.
// Same C ABI entry (cbim_conv3d_igemm picks this kernel when the call qualifies), same packed-weight layout, same
// partial-record format as conv_r32.hip / conv_igemm.hip.  Replaces aten::convolution / convolution_backward(input) of
// nn.Conv3d in ConvNormAct (/root/reference/model/dim3/conv_layers.py:29-38, 48-49: zero padding after the activation).
#include "cbim_common.h"
#include "conv_r32.h"
#include <stdlib.h>
#include <stdio.h>

namespace cbim {

static constexpr unsigned W_HB = 64000u;                 // one halo buffer: 1000 rows x 64 B
static constexpr unsigned W_ACC = 2u * W_HB;             // per-lane statistics cells: [4][512] x 16 B
static constexpr unsigned W_SHF = W_ACC + 32768u;        // shift of the sums: [8 waves][2 lane halves][8] floats
static constexpr unsigned W_RED = W_SHF + 512u;          // wave records [8][16 channels][3] floats
static constexpr unsigned W_SMEM = W_RED + 1536u;
static_assert(W_SMEM <= 160 * 1024, "LDS");
static constexpr unsigned W_OOB = 0x80000000u;           // vector offset of a lane outside the tensor (num_records = 2^31)

typedef __attribute__((ext_vector_type(4))) float w_f32x4;
typedef __attribute__((ext_vector_type(4))) int w_i32x4;

#ifdef CBIM_EMU
#define W_SCHED_FENCE() ((void)0)
#define W_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define W_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// one 1 KiB LDS-DMA piece: lane l copies the 16 bytes at base + soff + voff (zeros when soff + voff + 16 > nrec) to LDS
// byte lds + 16 l.  M0 (the LDS base of the instruction) is saved and restored inside the statement.
__device__ __forceinline__ void w_dma16(unsigned voff, unsigned long long base, unsigned nrec, unsigned soff, unsigned char* smem,
                                        unsigned lds_base, unsigned lds_off) {
#ifdef CBIM_EMU
  (void)lds_base;
  emu_buffer_load_lds16((const unsigned char*)base, nrec, voff, soff, smem + lds_off);
#else
  (void)smem;
  w_i32x4 rs = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), (int)nrec, 0x00020000};
  rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
  rs.z = __builtin_amdgcn_readfirstlane(rs.z);
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_base + lds_off), so = __builtin_amdgcn_readfirstlane(soff);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(a), "v"(voff), "s"(rs), "s"(so) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void w_wait_vm() {
#ifndef CBIM_EMU
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ unsigned w_mul24(unsigned a, unsigned b) {
#ifdef CBIM_EMU
  return a * b;
#else
  return __umul24(a, b);
#endif
}
__device__ __forceinline__ unsigned w_swz(unsigned hh) { return (hh & 1u) << 1; }   // (conv_r32.hip r_swz)
template <int MSK>
__device__ __forceinline__ float w_bfly(float v) {                                  // (conv_r32.hip r_bfly)
#ifdef CBIM_EMU
  return __shfl_xor(v, MSK, 64);
#else
  if (MSK == 16) return __shfl_xor(v, 16, 64);
  constexpr int ctrl = MSK == 1 ? 0xB1 : MSK == 2 ? 0x4E : MSK == 4 ? 0x141 : 0x140;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false));
#endif
}
__device__ __forceinline__ void w_swap16(float& a, float& b) {                      // (conv_r32.hip r_swap16)
#ifdef CBIM_EMU
  struct P { float a, b; } mine = {a, b};
  const P* buf = (const P*)cbim_emu::wave_exchange(&mine, sizeof(P));
  const int l = CBIM_EMU_LANE_ID();
  if (l & 16) a = buf[l - 16].b;
  else b = buf[l + 16].a;
#else
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
#endif
}
__device__ __forceinline__ unsigned w_launder(unsigned v) {     // keeps a loop-invariant value (and what is derived from it) inside the loop
#ifndef CBIM_EMU
  asm volatile("" : "+v"(v));
#endif
  return v;
}
__device__ __forceinline__ int w_uniform(int v) {
#ifdef CBIM_EMU
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

// cycle profile of the phases of a unit (make EXTRA=-DCBIM_RW_PROF; tools/r04/run_prof.sh): s_memtime stamps of every wave of
// workgroup 0, summed over its units, printed per launch by the launcher
#ifdef CBIM_RW_PROF
__device__ unsigned long long g_rw_prof[8][8];
#define W_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_t[i] += now_ - prof_last; prof_last = now_; } while (0)
#else
#define W_STAMP(i) ((void)0)
#endif

// MX: dgrad epilogue — the mask tensor is the ACTIVATED tensor a = relu(IN(x)) (act'(xh) = [a != 0], xh = a wherever the mask
//     is open) + the two InstanceNorm-backward sums; !MX: forward epilogue (optional residual, moments of the output)
// HP: h-pairs per wave = 32-channel chunks of Cout per workgroup (1: wave = (cout half, h-pair); 2: wave = (cout quarter of
//     64, half of the tile's rows))
// MC: several 32-channel chunks of Cin (units = (tile, chunk), streamed weights, optional second input tensor)
// SK: split-K — blockIdx.z owns the Cin chunks [NC z / ksplit, NC (z + 1) / ksplit) and writes its raw fp32 partial sums
//     to p.ws (no residual, mask or statistics: k_splitk_finish owns the epilogue)
// CBW: output channels per workgroup — 32 HP for the symmetric kernels (every wave the same share), 48 for the round-6 form
//     (k_conv3_rw48 below: the body is instantiated twice in one kernel, HP = 2 for waves 0..3 and HP = 1 for waves 4..7)
// LR: the dgrad mask tensor is a = lrelu(IN(x)), slope 0.01 (monai's UnetResBlock): act'(xh) = a > 0 ? 1 : 0.01 and
//     xh = a > 0 ? a : a / 0.01
// cb16 / hp0 / pj: the wave's role — its 16-cout block inside the workgroup's CBW channels, its first h-pair, and the index of
//     the LDS-DMA piece it fetches of every halo plane (7: none)
template <bool MX, int HP, bool MC, bool SK, int CBW, bool LR>
__device__ __forceinline__ void rw_body(const R32Params& p, unsigned char* const smem, const unsigned lds_base, const int cb16, const int hp0,
                                        const int pj) {
  static_assert(!SK || (MC && !MX), "split-K: forward-style epilogue over several Cin chunks");
  static_assert(!LR || MX, "the leaky mask belongs to the dgrad epilogue");
  constexpr int NT = 512, NW = 8, NTL = 8 * HP, NPAIR = 4 * HP, CB = CBW;
  const int tid = threadIdx.x, wave = w_uniform(tid >> 6), lane = tid & 63, lv = lane & 15, lq = lane >> 4;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int n_tiles = p.N * tiles_per_n;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)(((long long)lb * n_tiles) / gridDim.x);
  const int t_end = (int)(((long long)(lb + 1) * n_tiles) / gridDim.x);
  if (t_begin >= t_end) return;
  const int oc = blockIdx.y;                          // this workgroup's block of CB output channels
  const int NC = MC ? p.NC : 1;
  const int cc_lo = SK ? (int)(((long long)blockIdx.z * NC) / p.ksplit) : 0;         // this workgroup's Cin chunks
  const int cc_hi = SK ? (int)(((long long)(blockIdx.z + 1) * NC) / p.ksplit) : NC;

  // ---- wave = (16-cout block cb16, h-pairs hp0 .. hp0 + HP - 1); lane = (voxel lv of a 2x8 patch, k-group lq) ---------
  const int tw = lv & 7;
  // weights: packed image [cout block of BN][Cin chunk][tap][kg = lq >> 1][half = lq & 1][BN couts][8] (conv_igemm.hip)
  u32x4 wf[27];
  const unsigned w_tap = 4u * (unsigned)p.BN * 16u;    // bytes per tap
  const int co0 = oc * CB + 16 * cb16;                 // the wave's first output channel
  const unsigned char* const w_lane = (const unsigned char*)p.w + (size_t)(co0 / p.BN) * (size_t)NC * 27u * w_tap +
                                      (unsigned)((lq * p.BN) + co0 % p.BN + lv) * 16;
  // B operand (voxels): fragment of plane i at tap (kh, kw) = 16 bytes at row (i, th + kh, tw + kw), slot lq ^ swz(th + kh)
  int thp[HP];
  unsigned fb[HP][3];
#pragma unroll
  for (int hp = 0; hp < HP; ++hp) {
    thp[hp] = 2 * (hp0 + hp) + (lv >> 3);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
      fb[hp][kh] = (unsigned)((thp[hp] + kh) * 10 + tw) * 64u + (((unsigned)lq ^ w_swz((unsigned)(thp[hp] + kh))) << 4);
  }

  // ---- this lane's halo item: row r = r0 + (lane >> 2) of a plane, physical slot lane & 3 ------------------------------
  const unsigned x_sb = (unsigned)p.x_stride * 2u, x2_sb = (unsigned)p.x2_stride * 2u;
  const int r0 = pj < 6 ? 16 * pj : 84;
  const unsigned hrow = (unsigned)r0 + ((unsigned)lane >> 2);
  const unsigned hh = (hrow * 205u) >> 11, hw = hrow - hh * 10u;           // hrow / 10, hrow % 10 (hrow < 100)
  const unsigned slot_src = (((unsigned)lane & 3u) ^ w_swz(hh)) << 4;
  const unsigned rows_hw = w_mul24(hh, (unsigned)p.Wi) + hw;
  const unsigned lane_off = w_mul24(rows_hw, x_sb) + slot_src;
  const unsigned lane_off2 = MC ? w_mul24(rows_hw, x2_sb) + slot_src : 0u;
  const unsigned piece_lds = (unsigned)r0 * 64u;                           // + plane * 6400 + buffer

  struct TilePos { int n, td, th, tw, cc; };            // a unit: tile + Cin chunk
  auto advance = [&](TilePos& u) {
    if (++u.cc < cc_hi) return;
    u.cc = cc_lo;
    if (++u.tw == p.tiles_w) { u.tw = 0; if (++u.th == p.tiles_h) { u.th = 0; if (++u.td == p.tiles_d) { u.td = 0; ++u.n; } } }
  };
  TilePos cur, nxt;
  {
    const int tt = t_begin % tiles_per_n;
    cur.n = t_begin / tiles_per_n; cur.td = tt / (p.tiles_w * p.tiles_h); cur.th = (tt / p.tiles_w) % p.tiles_h; cur.tw = tt % p.tiles_w;
    cur.cc = cc_lo;
    nxt = cur;
    advance(nxt);
  }
  // halo source of a unit: descriptor base = image + chunk - (Wi + 1) rows (so that the scalar offset of a plane inside the
  // tensor is never negative: its first box row is at least row -1, column -1), scalar offset of plane 0, plane step, the
  // lane offset (out of range outside the tensor in h / w)
  struct Halo { unsigned long long base; unsigned soff0, plane_b, voff; int id0; };
  auto halo_of = [&](const TilePos& tp) -> Halo {
    Halo h;
    const bool second = MC && tp.cc >= p.c_split;
    const unsigned sb = second ? x2_sb : x_sb;
    const unsigned char* t0 = second ? (const unsigned char*)p.x2 + (tp.cc - p.c_split) * 64 : (const unsigned char*)p.x + tp.cc * 64;
    const int ih0 = tp.th * 8 - p.pH, iw0 = tp.tw * 8 - p.pW;
    h.id0 = tp.td * 8 - p.pD;
    h.plane_b = (unsigned)(p.Hi * p.Wi) * sb;
    h.base = (unsigned long long)t0 + (unsigned long long)tp.n * p.Di * h.plane_b - (unsigned long long)(p.Wi + 1) * sb;
    h.soff0 = (unsigned)((ih0 + 1) * p.Wi + iw0 + 1) * sb;              // + (id0 + plane) * plane_b
    // (a last chunk of fewer than 32 channels — Cin = 48: the slots past the tensor's channels are the next voxel's, their
    //  weights are zero in the packed image, and zeros land in LDS instead: no 0 x NaN)
    const bool ok = (unsigned)(ih0 + (int)hh) < (unsigned)p.Hi && (unsigned)(iw0 + (int)hw) < (unsigned)p.Wi &&
                    (unsigned)(tp.cc * 64) + slot_src < (unsigned)p.cin_bytes;
    h.voff = ok ? (second ? lane_off2 : lane_off) : W_OOB;
    return h;
  };
  auto dma_plane = [&](const Halo& h, int pl, unsigned buf) {             // this wave's piece of halo plane `pl` (waves 0..6)
    const int d = h.id0 + pl;
    const bool in = (unsigned)d < (unsigned)p.Di;
    const unsigned soff = in ? h.soff0 + (unsigned)d * h.plane_b : 0u;
    w_dma16(h.voff, h.base, in ? 0x80000000u : 0u, soff, smem, lds_base, buf + (unsigned)pl * 6400u + piece_lds);
  };

  // ---- statistics -------------------------------------------------------------------------------------------------------
  // after the epilogue exchange a lane owns the 8-channel chunk cidx = 2 cb16 + (lq >> 1) of the workgroup's CB channels
  const int cidx = 2 * cb16 + (lq >> 1);
  const bool c_ok = oc * CB + cidx * 8 < p.Cout;
  const bool want_part = !SK && p.partials != nullptr;
  float* const red = (float*)(smem + W_RED);
  u32x4* const cell0 = (u32x4*)(smem + W_ACC) + tid;                       // + 512 j, j = 0..3: (s0[0..3], s0[4..7], s1[0..3], s1[4..7])
  const float* const shf0 = (const float*)(smem + W_SHF) + (wave * 2 + (lane >> 5)) * 8;
  float cnt = 0.f;
  bool shift_set = false;                                                  // workgroup-uniform
  int run_n = cur.n;
#pragma unroll
  for (int j = 0; j < 4; ++j) cell0[512 * j] = u32x4{0u, 0u, 0u, 0u};
  // combine the lanes' sums of image n into this workgroup's records and reset them (all threads call it)
  auto flush_stats = [&](int n) {
    float t0[8], t1[8];
    {
      const u32x4 a = cell0[0], b = cell0[512], c = cell0[1024], d = cell0[1536];
      t0[0] = __uint_as_float(a.x); t0[1] = __uint_as_float(a.y); t0[2] = __uint_as_float(a.z); t0[3] = __uint_as_float(a.w);
      t0[4] = __uint_as_float(b.x); t0[5] = __uint_as_float(b.y); t0[6] = __uint_as_float(b.z); t0[7] = __uint_as_float(b.w);
      t1[0] = __uint_as_float(c.x); t1[1] = __uint_as_float(c.y); t1[2] = __uint_as_float(c.z); t1[3] = __uint_as_float(c.w);
      t1[4] = __uint_as_float(d.x); t1[5] = __uint_as_float(d.y); t1[6] = __uint_as_float(d.z); t1[7] = __uint_as_float(d.w);
#pragma unroll
      for (int j = 0; j < 4; ++j) cell0[512 * j] = u32x4{0u, 0u, 0u, 0u};
    }
    float tc = cnt;
#define W_ROUND(msk)                                                                                              \
    {                                                                                                             \
      float u0[8], u1[8];                                                                                         \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { u0[j] = w_bfly<msk>(t0[j]); u1[j] = w_bfly<msk>(t1[j]); }    \
      const float uc = w_bfly<msk>(tc);                                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { t0[j] += u0[j]; t1[j] += u1[j]; }                           \
      tc += uc;                                                                                                   \
    }
    W_ROUND(1) W_ROUND(2) W_ROUND(4) W_ROUND(8) W_ROUND(16)
#undef W_ROUND
    if ((lane & 31) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Moments a;
        if (MX) { a.n = 0.f; a.mean = t0[j]; a.m2 = t1[j]; }
        else a = moments_from_shifted(tc, shf0[j], t0[j], t1[j]);
        float* rr = red + ((wave * 16) + (lq >> 1) * 8 + j) * 3;
        rr[0] = a.n; rr[1] = a.mean; rr[2] = a.m2;
      }
    }
    cnt = 0.f;
    shift_set = false;
    __syncthreads();
    if (tid < CB && oc * CB + tid < p.Cout) {
      Moments a = {0.f, 0.f, 0.f};
      const int blk = tid >> 4;                         // the 16-cout block of this channel
      // the waves that hold it (CBW = 48: blocks 0 / 1 on waves {0, 1} / {2, 3}, block 2 on waves 4..7)
      const int wv0 = CBW == 48 ? (blk < 2 ? 2 * blk : 4) : blk, wvn = CBW == 48 ? (blk < 2 ? 2 : 4) : NW / (2 * HP);
      const int wvs = CBW == 48 ? 1 : 2 * HP;
      for (int g = 0; g < wvn; ++g) {
        const int wv = wv0 + g * wvs;
        const float* rr = red + ((wv * 16) + (tid & 15)) * 3;
        if (MX) { a.mean += rr[1]; a.m2 += rr[2]; }
        else { Moments b = {rr[0], rr[1], rr[2]}; a = moments_merge(a, b); }
      }
      const size_t o = (((size_t)n * p.P + lb) * p.Cout + oc * CB + tid) * 3;
      p.partials[o] = a.n; p.partials[o + 1] = a.mean; p.partials[o + 2] = a.m2;
    }
    __syncthreads();
  };
  if (want_part && tid < CB && oc * CB + tid < p.Cout) {
    // empty records (n = 0 merges as the identity): images this strip does not touch, and the records lb + k * grid of a
    // buffer sized for more workgroups than this launch has (p.P = cbim_conv3d_num_tiles records per image)
    const int n_first = t_begin / tiles_per_n, n_last = (t_end - 1) / tiles_per_n;
    for (int n = 0; n < p.N; ++n)
      for (unsigned r = lb; r < (unsigned)p.P; r += gridDim.x)
        if (r != lb || n < n_first || n > n_last) {
          const size_t o = (((size_t)n * p.P + r) * p.Cout + oc * CB + tid) * 3;
          p.partials[o] = 0.f; p.partials[o + 1] = 0.f; p.partials[o + 2] = 0.f;
        }
  }

  // ---- prologue: first unit's halo ------------------------------------------------------------------------------------
  {
    const Halo h = halo_of(cur);
    if (pj < 7) {
#pragma unroll
      for (int pl = 0; pl < 10; ++pl) dma_plane(h, pl, 0);
    }
    // the first unit's weight fragments, requested BEHIND the halo pieces: the two latencies overlap (requested first and
    // waited for before the pieces were issued, they cost every launch a second memory round trip)
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) wf[tp] = *(const u32x4*)(w_lane + ((size_t)cc_lo * 27u + tp) * w_tap);
#ifndef CBIM_EMU
    // (the compiler waits for these once, here, and forgets them: conv_r32.hip)
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) asm volatile("" : "+v"(wf[tp]));
#endif
    w_wait_vm<0>();
    __syncthreads();
  }

  w_f32x4 acc[NTL];
  const int n_my = (t_end - t_begin) * (cc_hi - cc_lo);  // units
#ifdef CBIM_RW_PROF
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = __builtin_readcyclecounter();
#endif
  for (int t = 0; t < n_my; ++t) {
    const unsigned buf = (unsigned)(t & 1) * W_HB, obuf = W_HB - buf;
    const bool more = t + 1 < n_my;
    const bool first_cc = cur.cc == cc_lo, last_cc = cur.cc == cc_hi - 1;
    TilePos nx;                                          // (the strip's last unit re-fetches itself into the idle buffer)
    nx.n = more ? nxt.n : cur.n; nx.td = more ? nxt.td : cur.td; nx.th = more ? nxt.th : cur.th; nx.tw = more ? nxt.tw : cur.tw;
    nx.cc = more ? nxt.cc : cur.cc;
    const Halo hn = halo_of(nx);
    if (first_cc) {
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) acc[nt] = w_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- epilogue addressing of this tile (used by the operand requests inside the MFMA loop and by the epilogue) ----------
    const int n = cur.n;
    const int od0 = cur.td * 8, oh0 = cur.th * 8, ow0 = cur.tw * 8;
    const long long orow = (((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0;
    const unsigned y_sb = (unsigned)p.y_stride * 2u, q_sb = MX ? (unsigned)p.mx_stride * 2u : (unsigned)p.res_stride * 2u;
    unsigned char* const y_tile = (unsigned char*)p.y + orow * (long long)y_sb;
    const unsigned char* const q_tile = (MX ? (const unsigned char*)p.mx : (const unsigned char*)p.res) + orow * (long long)q_sb;
    const bool has_q = !SK && (MX || p.res != nullptr);  // workgroup-uniform
    const unsigned cbyte = (unsigned)(oc * (CB / 8) + cidx) * 16u;
    // pair pr = (hp, pp): after the exchange this lane owns chunk cidx of voxel (plane 2 pp + (lq & 1), thp[hp], tw)
    // (the offsets below depend on the lane only: laundering the lane index keeps the compiler from hoisting them — and the
    //  64-bit addresses built on them — out of the unit loop, where they would be spilled and reloaded in every epilogue)
    const unsigned lane_l = w_launder((unsigned)lane);
    const unsigned e_lq1 = (lane_l >> 4) & 1u, e_tw = lane_l & 7u, e_h8 = (lane_l >> 3) & 1u;
    u32x4* const cell = (u32x4*)(smem + W_ACC) + (wave * 64 + lane_l);                        // (as cell0 / shf0 above)
    const float* const shf = (const float*)(smem + W_SHF) + (wave * 2 + (lane_l >> 5)) * 8;
    auto pair_rows = [&](int pr) -> unsigned {
      const int hp = pr / 4, pp = pr % 4;
      return w_mul24(w_mul24((unsigned)(2 * pp) + e_lq1, (unsigned)p.Ho) + (unsigned)(2 * (hp0 + hp)) + e_h8, (unsigned)p.Wo) + e_tw;
    };
    auto pair_in = [&](int pr) -> bool {
      const int hp = pr / 4, pp = pr % 4;
      return c_ok && oh0 + 2 * (hp0 + hp) + (int)e_h8 < p.Ho && ow0 + (int)e_tw < p.Wo && od0 + 2 * pp + (int)e_lq1 < p.Do;
    };
    // operand (residual / mask) rows of the lane's output voxels: requested INSIDE the MFMA loop (step RQ_STEP), so that their
    // memory latency is over when the epilogue starts (requested after the loop they cost ~750 cycles of issue behind the LDS-DMA
    // pieces and ~1700 of exposed latency per tile: profiles/r04_d_rw_phase_cycles.txt).  Every request is issued whatever the
    // lane's position (a lane outside the tensor re-reads the tile's first row): the number of vector-memory operations behind
    // the last LDS-DMA piece is then a constant the wait before the barrier can count on.
    constexpr int QD = HP == 2 ? 2 : 4;                  // operand requests in flight
    constexpr int RQ_STEP = 6;
    u32x4 rq[QD];
    auto request = [&](int pr) -> u32x4 {
      const unsigned off = pair_in(pr) ? w_mul24(pair_rows(pr), q_sb) + cbyte : 0u;
      return *(const u32x4*)(q_tile + off);
    };
    const unsigned char* const w_next = w_lane + (size_t)nx.cc * 27u * w_tap;
    W_STAMP(0);                                          // unit set-up
    // 9 (kh, kw) steps x HP patches: the 10 plane fragments of a patch stream through a ring of 5 registers, plane i feeds
    // the MFMAs (n-tile i, kd 0), (i-1, kd 1), (i-2, kd 2).  The next unit's halo: two pieces per step in steps 0..4
    // (measured: early issue beats an even spread once the source is HBM).
    {
      constexpr int RING = 5, PLN = 10, SEQ = 9 * HP * PLN;
      u32x4 xr[RING];
      auto frag_addr = [&](int e) -> unsigned {                   // e = ((kh*3 + kw) * HP + hp) * PLN + plane
        const int i = e % PLN, hp = (e / PLN) % HP, s = e / (PLN * HP);
        return buf + fb[hp][s / 3] + (unsigned)((s % 3) * 64) + (unsigned)(i * 6400);
      };
#pragma unroll
      for (int e = 0; e < RING - 1; ++e) xr[e] = *(const u32x4*)(smem + frag_addr(e));
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int kh = s / 3, kw = s % 3;
        if (s < 5 && pj < 7) {
          dma_plane(hn, 2 * s, obuf);
          dma_plane(hn, 2 * s + 1, obuf);
        }
        if (s == RQ_STEP && last_cc && has_q) {   // (a branch, not a predicate: the units in between must not touch any of this)
#pragma unroll
          for (int pr = 0; pr < QD; ++pr) rq[pr] = request(pr);
        }
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) {
#pragma unroll
          for (int i = 0; i < PLN; ++i) {
            const int e = (s * HP + hp) * PLN + i;
            if (e + RING - 1 < SEQ) xr[(e + RING - 1) % RING] = *(const u32x4*)(smem + frag_addr(e + RING - 1));
            W_SCHED_FENCE();
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
              const int pl = i - kd;
              if (pl >= 0 && pl < 8)
                acc[hp * 8 + pl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[(kd * 3 + kh) * 3 + kw]),
                                                                           __builtin_bit_cast(bf16x8, xr[e % RING]), acc[hp * 8 + pl], 0, 0, 0);
            }
            W_SCHED_FENCE();
          }
        }
        // the three fragments of this (kh, kw) are dead until step s of the next unit: reload them for it now
        if (MC) {
#pragma unroll
          for (int kd = 0; kd < 3; ++kd)
            wf[(kd * 3 + kh) * 3 + kw] = *(const u32x4*)(w_next + (size_t)((kd * 3 + kh) * 3 + kw) * w_tap);
        }
      }
    }
    // ---- ONE barrier per unit: every wave is done with `buf`, the other buffer is complete (own pieces landed; vector
    //      memory operations complete in order, so the 15 weight-fragment loads / the QD operand requests issued after the last
    //      piece may stay in flight)
    W_STAMP(1);                                          // MFMA loop (+ piece issue, weight reloads, operand requests)
    if (MC) w_wait_vm<15>();
    else if (last_cc && has_q) w_wait_vm<QD>();          // (the QD operand requests are the youngest operations)
    else w_wait_vm<0>();
    W_STAMP(3);                                          // own pieces landed
    __syncthreads();
    W_STAMP(4);                                          // barrier
    // ---- epilogue of this tile: its stores drain under the next unit's MFMAs ---------------------------------------------
    if (SK && last_cc) {
      // raw fp32 partial sums of this Cin slice: ws[z][row][Cout], 32 bytes (the lane's 8 channels of one voxel) per pair
      float* const wz = p.ws + ((size_t)blockIdx.z * ((size_t)p.N * p.Do * p.Ho * p.Wo) + (size_t)orow) * p.Cout + oc * CB + cidx * 8;
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = acc[2 * pr][r], b = acc[2 * pr + 1][r];
          w_swap16(a, b);
          v[r] = a;
          v[4 + r] = b;
        }
        if (pair_in(pr)) {
          float* dst = wz + (size_t)pair_rows(pr) * p.Cout;
          *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
          *(f32x4*)(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
      }
    }
    if (!SK && last_cc) {
      if (want_part && n != run_n) { flush_stats(run_n); run_n = n; }
      typedef float f2_t __attribute__((ext_vector_type(2)));
      f2_t l0[4], l1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { l0[j] = f2_t{0.f, 0.f}; l1[j] = f2_t{0.f, 0.f}; }
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) {
        const bool in = pair_in(pr);
        const unsigned rows = pair_rows(pr);
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = acc[2 * pr][r], b = acc[2 * pr + 1][r];
          w_swap16(a, b);
          v[r] = a;
          v[4 + r] = b;
        }
        u32x4 q = u32x4{0u, 0u, 0u, 0u};
        if (has_q) {
          if (in) q = rq[pr % QD];
          if (pr + QD < NPAIR) rq[pr % QD] = request(pr + QD);      // next request into the slot just read
        }
        const float live = in ? 1.f : 0.f;
        const f2_t live2 = {live, live};
        const unsigned rw[4] = {q.x, q.y, q.z, q.w};
        if (MX) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f2_t a = {__uint_as_float(rw[j] << 16), __uint_as_float(rw[j] & 0xffff0000u)};
            f2_t g;
            if (LR) {
              // a = lrelu(xh): a > 0 passes the gradient, anything else (negative, -0, 0) takes the slope — torch's
              // leaky_relu_backward tests x > 0; xh is recovered from a for the InstanceNorm-backward sum
              const bool p0 = (int)(rw[j] << 16) > 0, p1 = (int)(rw[j] & 0xffff0000u) > 0;
              g.x = p0 ? v[2 * j] : 0.01f * v[2 * j];
              g.y = p1 ? v[2 * j + 1] : 0.01f * v[2 * j + 1];
              a.x = p0 ? a.x : 100.f * a.x;
              a.y = p1 ? a.y : 100.f * a.y;
            } else {
              g.x = (rw[j] & 0xffffu) != 0u ? v[2 * j] : 0.f;
              g.y = (rw[j] >> 16) != 0u ? v[2 * j + 1] : 0.f;
            }
            v[2 * j] = g.x;
            v[2 * j + 1] = g.y;
            const f2_t gl = g * live2;
            l0[j] = l0[j] + gl;
            l1[j] = __builtin_elementwise_fma(gl, a, l1[j]);
#ifndef CBIM_EMU
            // (the sums are consumed under `want_part` only: without this the compiler sinks ALL of the arithmetic above to
            //  the end of the epilogue and keeps every pair's gradient and mask registers alive until then — 83 spilled VGPRs)
            asm volatile("" : "+v"(l0[j]), "+v"(l1[j]));
#endif
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // residual (zeros when there is none)
            v[2 * j] += __uint_as_float(rw[j] << 16);
            v[2 * j + 1] += __uint_as_float(rw[j] & 0xffff0000u);
          }
          if (want_part) {
            if (pr == 0 && !shift_set) {
              // common shift of the 32 lanes that hold a channel chunk: this tile's first voxel in the group's first lane (any
              // finite value near the data works).  Written and read by the same wave (LDS operations of a wave are in order).
              shift_set = true;
              if ((lane_l & 31u) == 0u) {
                float* sw = (float*)(smem + W_SHF) + (wave * 2 + (lane_l >> 5)) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) sw[j] = v[j];
              }
#ifdef CBIM_EMU
              (void)__shfl(0, 0, 64);                    // (the executor's lanes are not in lockstep: the write lands before the reads)
#endif
            }
            // (the shift is read per pair — two broadcast LDS reads — instead of living in 8 registers through the epilogue)
            const float* const shp = shf + w_launder(0u);   // (a fresh address per pair: the two reads are not merged into 8 live registers)
            const f32x4 sa = *(const f32x4*)shp, sb4 = *(const f32x4*)(shp + 4);
            const f2_t sh2[4] = {f2_t{sa.x, sa.y}, f2_t{sa.z, sa.w}, f2_t{sb4.x, sb4.y}, f2_t{sb4.z, sb4.w}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f2_t x = {v[2 * j], v[2 * j + 1]};
              const f2_t d = (x - sh2[j]) * live2;
              l0[j] = l0[j] + d;
              l1[j] = __builtin_elementwise_fma(d, d, l1[j]);
#ifndef CBIM_EMU
              asm volatile("" : "+v"(l0[j]), "+v"(l1[j]));   // (computed here, not sunk to the end of the epilogue)
#endif
            }
          }
        }
        if (in) *(u32x4*)(y_tile + (w_mul24(rows, y_sb) + cbyte)) = Elem<bf16_tag>::pack(v);
        cnt += live;
      }
      if (want_part) {                                   // this tile's sums into the lane's cells
        u32x4 a = cell[0], b = cell[512], c = cell[1024], d = cell[1536];
        a.x = __float_as_uint(__uint_as_float(a.x) + l0[0].x); a.y = __float_as_uint(__uint_as_float(a.y) + l0[0].y);
        a.z = __float_as_uint(__uint_as_float(a.z) + l0[1].x); a.w = __float_as_uint(__uint_as_float(a.w) + l0[1].y);
        b.x = __float_as_uint(__uint_as_float(b.x) + l0[2].x); b.y = __float_as_uint(__uint_as_float(b.y) + l0[2].y);
        b.z = __float_as_uint(__uint_as_float(b.z) + l0[3].x); b.w = __float_as_uint(__uint_as_float(b.w) + l0[3].y);
        c.x = __float_as_uint(__uint_as_float(c.x) + l1[0].x); c.y = __float_as_uint(__uint_as_float(c.y) + l1[0].y);
        c.z = __float_as_uint(__uint_as_float(c.z) + l1[1].x); c.w = __float_as_uint(__uint_as_float(c.w) + l1[1].y);
        d.x = __float_as_uint(__uint_as_float(d.x) + l1[2].x); d.y = __float_as_uint(__uint_as_float(d.y) + l1[2].y);
        d.z = __float_as_uint(__uint_as_float(d.z) + l1[3].x); d.w = __float_as_uint(__uint_as_float(d.w) + l1[3].y);
        cell[0] = a; cell[512] = b; cell[1024] = c; cell[1536] = d;
      }
    }
    W_STAMP(5);                                          // epilogue
    cur = nxt;
    advance(nxt);
  }
  if (want_part) flush_stats(run_n);
#ifdef CBIM_RW_PROF
  W_STAMP(6);
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) g_rw_prof[wave][i] = prof_t[i];
    g_rw_prof[wave][7] = (unsigned long long)n_my;
  }
#endif
}

template <bool MX, int HP, bool MC, bool SK = false>
__global__ void __launch_bounds__(512, 1) k_conv3_rw(R32Params p) {
  W_DYN_SMEM(smem);
#ifdef CBIM_EMU
  const unsigned lds_base = 0;
#else
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#endif
  const int wave = w_uniform((int)threadIdx.x >> 6);
  rw_body<MX, HP, MC, SK, 32 * HP, false>(p, smem, lds_base, HP == 1 ? (wave & 1) : (wave & 3), HP == 1 ? (wave >> 1) : 2 * (wave >> 2), wave);
}

// Round 6 — 48 output channels per workgroup (SwinUNETR's feature_size = 48 layers: monai UnetResBlock convolutions 48 -> 48 and
// 96 -> 48 at 128^3 / 64^3, /root/reference/model/dim3/swin_unetr.py:129-228).  Three 16-cout blocks do not divide over eight
// equal waves, and a wave holds the 27 weight fragments of ONE block; what has to balance is the matrix pipe of each SIMD, and
// waves w and w + 4 of a 512-thread workgroup share a SIMD (tools/ubench/simd_map.hip, profiles/r06_a_simd_map.txt).  So the tile's
// twelve (cout block, h-pair) jobs go 2 + 1 per SIMD: waves 0..3 take two h-pairs of block 0 / 1 (the HP = 2 register image: 64
// accumulators), waves 4..7 one h-pair of block 2 (HP = 1) — every SIMD issues 3 x 216 MFMAs per unit, none of them on padding.
// The halo pieces go to the light waves first (pieces 0..3 on waves 4..7, 4..6 on waves 0..2).
template <bool MX, bool LR>
__global__ void __launch_bounds__(512, 1) k_conv3_rw48(R32Params p) {
  W_DYN_SMEM(smem);
#ifdef CBIM_EMU
  const unsigned lds_base = 0;
#else
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#endif
  const int wave = w_uniform((int)threadIdx.x >> 6);
  if (wave < 4) rw_body<MX, 2, true, false, 48, LR>(p, smem, lds_base, wave >> 1, 2 * (wave & 1), wave + 4);
  else rw_body<MX, 1, true, false, 48, LR>(p, smem, lds_base, 2, wave - 4, wave - 4);
}

}  // namespace cbim

using namespace cbim;

// g_rw_on (cbim_conv_rw_enable): 0 off (every call stays on k_conv3_r32 / k_conv_igemm), 1 (default) on
static int g_rw_on = 1;
// g_rw_wide (cbim_conv_rw_enable): 0 never, 1 (default) the wide form wherever Cout is a multiple of 64 and the grid still fills the chip,
// 2 wherever Cout is a multiple of 64
static int g_rw_wide = 1;
extern "C" int cbim_conv_rw_enable(int on, int wide) {
  const int old = g_rw_on | (g_rw_wide << 1);
  if (on >= 0) g_rw_on = on & 1;
  if (wide >= 0) g_rw_wide = wide;
  return old;
}

static int64_t rw_tiles(const cbim_conv_desc* d) {
  return (int64_t)d->N * ((d->Do + 7) / 8) * ((d->Ho + 7) / 8) * ((d->Wo + 7) / 8);
}
// 64 output channels per workgroup: Cout in multiples of 64, still >= 192 (strip, block) workgroups, and at least three
// Cin chunks per tile — the wide epilogue (8 voxel pairs per lane, 64 accumulator registers next to the 108 weight registers)
// is slower per output than the narrow one and has to be amortised: 96->64 @128^3 687 -> 609 us, 192->128 @64^3 269 -> 251,
// 384->256 @32^3 136 -> 127, but 64->64 @64^3 59 -> 63 (profiles/r04_g_conv_rw_ab.txt)
static bool rw_wide(const cbim_conv_desc* d) {
  if (!g_rw_wide || d->Cout % 64 != 0) return false;
  if (g_rw_wide == 2) return true;                                  // (forced: tests)
  return d->Cin >= 96 && rw_tiles(d) * (d->Cout / 64) >= 192;
}

// the calls k_conv3_rw takes from k_conv3_r32 (cbim_conv_r32_eligible has already said yes): the input used as it is, a
// forward epilogue (optional residual) or the activated-mask dgrad epilogue, tile depth 8, "same" padding of at most 1
bool cbim_conv_rw_eligible(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                           const float* in_stats, const void* mask_x, const float* mask_stats) {
  (void)x; (void)x2;
  if (!g_rw_on || in_stats) return false;
  if (mask_x && (mask_stats || d->act != CBIM_ACT_RELU)) return false;
  if (d->pD > 1 || d->pH > 1 || d->pW > 1 || d->pD < 0 || d->pH < 0 || d->pW < 0) return false;
  // scalar + vector offset of the buffer loads stay below 2^31 inside one image (num_records = 2^31)
  const int64_t sb = (x2 && x2_stride > x_stride ? x2_stride : x_stride) * 2;
  const int64_t img = ((int64_t)d->Di + 2) * d->Hi * d->Wi * sb;
  if (img >= ((int64_t)1 << 31) - 65536 || sb >= (1 << 24) || (int64_t)10 * d->Wi + 10 >= (1 << 24)) return false;
  return true;
}

// Round 6: 48 output channels per workgroup (k_conv3_rw48) — Cout in multiples of 48 where the 32-channel kernels do not apply
// (Cin or Cout not a multiple of 32: SwinUNETR's 48 -> 48, 96 -> 48 and the 48 -> 96 input gradient), Cin any multiple of 8
// (the last 32-channel chunk is zero-filled past Cin).  CBIM_CONV_RW48=0 keeps those layers on k_conv_igemm (A/B runs).
static int g_rw48 = getenv("CBIM_CONV_RW48") ? atoi(getenv("CBIM_CONV_RW48")) : 1;
extern "C" int cbim_conv_rw48_enable(int on) {
  const int old = g_rw48;
  if (on >= 0) g_rw48 = on;
  return old;
}
static bool rw48_shape(const cbim_conv_desc* d) {
  if (!g_rw_on || !g_rw48 || d->dtype != CBIM_BF16 || d->kD != 3 || d->kH != 3 || d->kW != 3) return false;
  if (d->Cout % 48 != 0 || d->Cin % 8 != 0 || (d->Cin % 32 == 0 && d->Cout % 32 == 0)) return false;
  if (d->Do < 8 || d->Ho < 8 || d->Wo < 8) return false;
  if (g_rw48 == 2) return true;                                     // (forced: tests)
  return rw_tiles(d) * (d->Cout / 48) >= 128;
}
bool cbim_conv_rw48_eligible(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride, int cin_split,
                             const float* in_stats, const void* mask_x, const float* mask_stats) {
  if (!rw48_shape(d) || in_stats) return false;
  if (mask_x && (mask_stats || !(d->act == CBIM_ACT_RELU || d->act == CBIM_ACT_LRELU))) return false;
  if (x2 && (cin_split <= 0 || cin_split >= d->Cin || cin_split % 32 != 0)) return false;
  const int keep = g_rw_on;                                          // (the addressing limits of k_conv3_rw)
  const bool ok = cbim_conv_rw_eligible(d, x, x_stride, x2, x2_stride, nullptr, nullptr, nullptr);
  (void)keep;
  return ok;
}
// the same question without the call's tensors (Python picks the materialised-activation path of a block by it)
extern "C" int cbim_conv_rw48_takes(const cbim_conv_desc* d) {
  if (!d || !rw48_shape(d)) return 0;
  if (d->pD > 1 || d->pH > 1 || d->pW > 1 || d->pD < 0 || d->pH < 0 || d->pW < 0) return 0;
  const int64_t sb = (int64_t)d->Cin * 2, img = ((int64_t)d->Di + 2) * d->Hi * d->Wi * sb;
  return img < ((int64_t)1 << 31) - 65536 && (int64_t)10 * d->Wi + 10 < (1 << 24);
}

int64_t cbim_conv_rw_grid(const cbim_conv_desc* d) {
  const int64_t n_tiles = rw_tiles(d);
  const int n_cb = rw48_shape(d) ? d->Cout / 48 : rw_wide(d) ? d->Cout / 64 : (d->Cout + 31) / 32;
  int64_t cap = 256 / n_cb;      // about one workgroup per CU over all Cout blocks
  if (cap < 1) cap = 1;
  int64_t g = n_tiles < cap ? n_tiles : cap;
  // several Cout blocks: strips in multiples of 8 put the blocks of one strip (the same input halo, Cout-block after
  // Cout-block) on the same XCD (workgroup b runs on XCD b % 8, b = x + grid.x * y), where the second reader hits the L2.
  // Measured on 64 -> 96 @128^3 dgrad, grid (85, 3): 3.66 GB fetched per launch for 0.67 GB of tensors — the launch ran at the
  // HBM bandwidth (profiles/r04_f_pmc_traffic_rw.txt)
  if (n_cb > 1 && g >= 8) g &= ~(int64_t)7;
  return g;
}

template <bool MX, bool LR>
static int rw48_launch_k(const R32Params& p, dim3 grid, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv3_rw48<MX, LR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  CBIM_LAUNCH((k_conv3_rw48<MX, LR>), grid, dim3(512), (size_t)W_SMEM, st, p);
#ifdef CBIM_RW_PROF
  {
    (void)hipStreamSynchronize(st);
    unsigned long long h[8][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rw_prof), sizeof(h));
    static const char* nm[7] = {"setup", "mfma", "requests", "wait_vm", "barrier", "epilogue", "flush"};
    fprintf(stderr, "[rw48 prof] MX %d Cin %d Cout %d @%d units %llu | cycles per unit, waves 0..3 (two h-pairs) / 4..7 (one):", (int)MX,
            p.cin_bytes / 2, p.Cout, p.Do, h[0][7]);
    for (int i = 0; i < 7; ++i) {
      unsigned long long a = 0, b = 0;
      for (int w = 0; w < 4; ++w) { a += h[w][i]; b += h[w + 4][i]; }
      const unsigned long long n = h[0][7] ? h[0][7] : 1;
      fprintf(stderr, " %s %llu / %llu", nm[i], a / 4 / n, b / 4 / n);
    }
    fprintf(stderr, "\n");
  }
#endif
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "conv rw48 launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

template <bool MX, int HP, bool MC, bool SK = false>
static int rw_launch_k(const R32Params& p, dim3 grid, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv3_rw<MX, HP, MC, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  CBIM_LAUNCH((k_conv3_rw<MX, HP, MC, SK>), grid, dim3(512), (size_t)W_SMEM, st, p);
#ifdef CBIM_RW_PROF
  {
    (void)hipStreamSynchronize(st);
    unsigned long long h[8][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rw_prof), sizeof(h));
    static const char* nm[7] = {"setup", "mfma", "requests", "wait_vm", "barrier", "epilogue", "flush"};
    fprintf(stderr, "[rw prof] MX %d HP %d MC %d Cin %d Cout %d @%d units %llu | cycles per unit:", (int)MX, HP, (int)MC, p.NC * 32, p.Cout, p.Do, h[0][7]);
    for (int i = 0; i < 7; ++i) {
      unsigned long long sum = 0, mx = 0;
      for (int w = 0; w < 8; ++w) { sum += h[w][i]; if (h[w][i] > mx) mx = h[w][i]; }
      fprintf(stderr, " %s %llu (max %llu)", nm[i], sum / 8 / (h[0][7] ? h[0][7] : 1), mx / (h[0][7] ? h[0][7] : 1));
    }
    fprintf(stderr, "\n");
  }
#endif
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "conv rw launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

int cbim_conv_rw_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                        int cin_split, const void* w_packed, const void* res, int64_t res_stride, const void* mask_x,
                        int64_t mask_stride, void* y, int64_t y_stride, float* partials, void* stream) {
  R32Params p;
  p.x = x; p.x_stride = x_stride; p.in_stats = nullptr; p.w = w_packed;
  p.NC = (d->Cin + 31) / 32; p.cin_bytes = d->Cin * 2;
  p.x2 = x2; p.x2_stride = x2 ? x2_stride : x_stride; p.c_split = x2 ? cin_split / 32 : p.NC;
  p.BN = d->Cout <= 32 ? 32 : 64;
  p.res = res; p.res_stride = res_stride; p.mx = mask_x; p.mx_stride = mask_stride; p.m_stats = nullptr;
  p.y = y; p.y_stride = y_stride; p.partials = partials;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.pD = d->pD; p.pH = d->pH; p.pW = d->pW; p.act = d->act;
  p.tiles_d = (d->Do + 7) / 8; p.tiles_h = (d->Ho + 7) / 8; p.tiles_w = (d->Wo + 7) / 8;
  p.dbg = 0;
  p.ksplit = 1; p.ws = nullptr;
  p.P = cbim_conv3d_num_tiles(d);
  const bool c48 = rw48_shape(d);
  const bool wide = !c48 && rw_wide(d);
  dim3 grid((unsigned)cbim_conv_rw_grid(d), (unsigned)(c48 ? d->Cout / 48 : wide ? d->Cout / 64 : (d->Cout + 31) / 32));
  CBIM_CHECK(!partials || p.P >= (int)grid.x, CBIM_EINVAL, "conv rw: %d partial records < grid", p.P);
  {
    // 32-bit byte offsets inside one output tile, built from 24-bit multiplies
    const int64_t tile_rows = (int64_t)8 * d->Ho * d->Wo;
    int64_t so = y_stride * 2;
    if (res && res_stride * 2 > so) so = res_stride * 2;
    if (mask_x && mask_stride * 2 > so) so = mask_stride * 2;
    CBIM_CHECK(tile_rows < (1 << 24) && so < (1 << 24) && tile_rows * so < ((int64_t)1 << 32), CBIM_EUNSUPPORTED,
               "conv rw: output plane %dx%d with row stride %lld B exceeds the 32-bit epilogue addressing", d->Ho, d->Wo, (long long)so);
  }
  hipStream_t st = (hipStream_t)stream;
  if (c48) {
    if (mask_x) return d->act == CBIM_ACT_LRELU ? rw48_launch_k<true, true>(p, grid, st) : rw48_launch_k<true, false>(p, grid, st);
    return rw48_launch_k<false, false>(p, grid, st);
  }
  const bool mc = p.NC > 1;
  if (mask_x) {
    if (wide) return mc ? rw_launch_k<true, 2, true>(p, grid, st) : rw_launch_k<true, 2, false>(p, grid, st);
    return mc ? rw_launch_k<true, 1, true>(p, grid, st) : rw_launch_k<true, 1, false>(p, grid, st);
  }
  if (wide) return mc ? rw_launch_k<false, 2, true>(p, grid, st) : rw_launch_k<false, 2, false>(p, grid, st);
  return mc ? rw_launch_k<false, 1, true>(p, grid, st) : rw_launch_k<false, 1, false>(p, grid, st);
}

// split-K factor of a low-resolution layer: 0 when the layer already has >= 128 (tile, 32-cout block) workgroups or a single
// Cin chunk; otherwise the divisor-free share-out of the Cin chunks that brings the launch closest to one workgroup per CU
// (k_splitk_finish keeps one 16-byte output chunk per thread of 256: Cout <= 2048)
static int g_rw_split = 1;
int cbim_conv_rw_ksplit(const cbim_conv_desc* d) {
  if (!g_rw_on || !g_rw_split || d->dtype != CBIM_BF16) return 0;
  const int NC = d->Cin / 32;
  const int64_t wgs = rw_tiles(d) * ((d->Cout + 31) / 32);
  if (NC < 2 || wgs >= 128 || d->Cout / 8 > 256) return 0;
  int64_t ks = (256 + wgs - 1) / wgs;
  if (ks > NC) ks = NC;
  if (ks > 32) ks = 32;
  while (ks > 1 && wgs * ks > 288) --ks;       // (one round of workgroups)
  return ks > 1 ? (int)ks : 0;
}

int cbim_conv_rw_split_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                              int cin_split, const void* w_packed, float* ws, void* stream) {
  R32Params p;
  p.x = x; p.x_stride = x_stride; p.in_stats = nullptr; p.w = w_packed;
  p.NC = d->Cin / 32; p.cin_bytes = d->Cin * 2;
  p.x2 = x2; p.x2_stride = x2 ? x2_stride : x_stride; p.c_split = x2 ? cin_split / 32 : p.NC;
  p.BN = d->Cout <= 32 ? 32 : 64;
  p.res = nullptr; p.res_stride = 0; p.mx = nullptr; p.mx_stride = 0; p.m_stats = nullptr;
  p.y = nullptr; p.y_stride = d->Cout; p.partials = nullptr;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.pD = d->pD; p.pH = d->pH; p.pW = d->pW; p.act = d->act;
  p.tiles_d = (d->Do + 7) / 8; p.tiles_h = (d->Ho + 7) / 8; p.tiles_w = (d->Wo + 7) / 8;
  p.dbg = 0; p.P = 0;
  p.ksplit = cbim_conv_rw_ksplit(d); p.ws = ws;
  CBIM_CHECK(p.ksplit > 1 && ws, CBIM_EINVAL, "conv rw split-K: not a split-K layer");
  CBIM_CHECK((int64_t)8 * d->Ho * d->Wo < (1 << 24), CBIM_EUNSUPPORTED, "conv rw split-K: output plane too large");
  dim3 grid((unsigned)rw_tiles(d), (unsigned)((d->Cout + 31) / 32), (unsigned)p.ksplit);   // one tile per workgroup
  return rw_launch_k<false, 1, true, true>(p, grid, (hipStream_t)stream);
}

CBIM_DEFINE_WARM(rw)
