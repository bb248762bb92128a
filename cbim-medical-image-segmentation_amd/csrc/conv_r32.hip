// conv_r32.hip — 3x3x3 convolution (forward and dgrad), bf16, channel counts in multiples of 32: "weights in registers".
// Written for the single-chunk layers 32 -> 32 @ 128^3 that own half of the ResUNet's FLOPs at full resolution
// (conv_layers.py:71-94 with base_ch 32, unet.py:22,44), then generalised: blockIdx.y = 32-channel chunk of Cout, and a
// workgroup walks the 32-channel chunks of Cin (one "unit" = one (tile, Cin chunk)) with the accumulators kept across
// the chunks.  The 27 weight fragments of the NEXT unit stream from L2 into the registers of the CURRENT unit as they
// fall dead: a (kh, kw) step is the only user of its three fragments, so right after step s they are reloaded for the
// next unit, which needs them one whole unit later — no second register set, no exposed latency.
//
// Why a second kernel: k_conv_igemm<2,1> ran these layers at 19 % of the MFMA peak.  With N = 32 every A fragment
// feeds ONE MFMA, so the LDS fragment traffic (3 ds_read_b128 per 2 MFMAs) plus halo stores and the transposing
// epilogue saturate the LDS pipe, and every phase (halo load, transform, weight stage, epilogue) is serialised by
// workgroup barriers.  Here
//   * the layer's weights stay in REGISTERS for the life of the persistent workgroup — no weight staging, no stage
//     barriers, ONE barrier per tile.  To fit two waves per SIMD (256 VGPRs each; a first version with one 512-register
//     wave per SIMD was instruction-issue bound: 2 700 vector-ALU instructions per 216 MFMAs) the matrix product runs
//     on v_mfma_f32_16x16x32_bf16: a wave owns 16 of the 32 output channels, its 27 weight fragments (K = all 32 input
//     channels in one instruction) are 108 VGPRs;
//   * a wave owns 8 n-tiles STACKED ALONG D (a 2x8 (h,w) patch of 16 voxels on each of the tile's 8 planes): the
//     fragment of input plane i serves (n-tile i, kd 0), (n-tile i-1, kd 1) and (n-tile i-2, kd 2), so per (kh,kw)
//     10 fragment reads feed 24 MFMAs; reads stream through a small register ring, plane-major;
//   * the halo (10x10x10 rows x 64 B) is DOUBLE buffered in LDS (2 x 64 KiB): the next tile's halo is fetched while
//     this tile computes — by LDS-DMA (global_load_lds_dwordx4, no registers, no VALU) when the input is used as
//     it is (dgrad, or a raw convolution), through registers with InstanceNorm + activation applied on the way
//     (pre-activation ConvNormAct, conv_layers.py:48-49; zero padding after the transform) otherwise; the per-item
//     work is spread over the (kh,kw) steps of the MFMA loop;
//   * the MFMA operands are swapped (A = weights, B = voxels), so an accumulator holds 4 output channels of ONE voxel
//     per lane: one v_permlane16_swap per register between two n-tiles leaves every lane with a whole 16-byte
//     channel chunk of one voxel — residual add, act' mask, statistics and the store need no LDS transpose; the
//     per-channel statistics are per-lane running sums, combined across lanes once per workgroup.
// Same C ABI entry (cbim_conv3d_igemm picks this kernel when the layer qualifies), same packed-weight layout, same
// partial-record format as conv_igemm.hip.
#include "cbim_common.h"
#include "conv_r32.h"
#include <stdlib.h>

namespace cbim {

static constexpr int R_RB = 64;                 // bytes per halo row (32 bf16 channels)
// timing ablations of tools/r32_ablate.py (wrong results): compiled in only with `make EXTRA=-DCBIM_R32_DBG_RT` — as a
// run-time parameter every test is a scalar branch, and the 90 of them inside the MFMA loop (one per fragment read) cut
// the loop into basic blocks the scheduler cannot interleave across
#ifdef CBIM_R32_DBG_RT
#define R_DBG (p.dbg)
#else
#define R_DBG 0
#endif
#ifndef R32_LS_SINGLE
#define R32_LS_SINGLE 0   // measured: 218 vs 183 us on 32->32 @128^3 (the per-tile butterfly costs more than the 10 spilled registers)
#endif
// Geometry by tile depth TD.  TD = 8: 8x8x8 tile, 512 threads, ONE workgroup per CU (2 x 64 KiB of halo).
// TD = 4: 4x8x8 tile, 256 threads, TWO workgroups per CU (2 x 40 KiB of halo each, 80 KiB per workgroup): each SIMD then
// hosts one wave of each workgroup, the two workgroups drift apart, and one's epilogue / halo transform (vector ALU
// and memory) overlaps the other's MFMA loop — with one workgroup per CU all eight waves leave the MFMA loop together
// at the tile barrier and the matrix pipes idle through every epilogue.
template <int TD> struct RGeom {
  static constexpr int NT = TD == 8 ? 512 : 256;
  static constexpr int NW = NT / 64;
  static constexpr int HP = TD == 8 ? 1 : 2;                 // (h-pair) patches of 2x8 voxels per wave and plane
  static constexpr int HROWS = (TD + 2) * 100;               // halo rows: (TD+2) x 10 x 10
  static constexpr int ITEMS = HROWS * 4;                    // 16-byte items
  static constexpr int UH = (ITEMS + NT - 1) / NT;           // items per thread: 8 / 10
  static constexpr int PIECES = (ITEMS + 63) / 64;           // 1 KiB LDS-DMA pieces: 63 / 38
  static constexpr unsigned HBUF = (unsigned)UH * NT * 16;   // bytes reserved per halo buffer: 65536 / 40960
  static constexpr unsigned TAB = 1536 + 256 + 256;          // statistics scratch + two (mean, rstd) tables
  // TD = 8: tables after the two buffers; TD = 4: in the tail of buffer 1 that no LDS-DMA piece touches
  static constexpr unsigned TAB_BASE = TD == 8 ? 2 * HBUF : 2 * HBUF - TAB;
  // TD = 8 only (there is no LDS left with two workgroups per CU): per halo item its byte offset from the halo box
  // origin (u32) and its position (hd | hh << 4 | hw << 8 | exists << 12, u16), filled once per workgroup
  static constexpr bool ITAB = TD == 8;
  static constexpr unsigned ITAB_BASE = 2 * HBUF + TAB;
  // TD = 8: (mean, rstd) of every input channel, one 256-byte row per 32-channel chunk (Cin <= 32 * MAXC)
  static constexpr int MAXC = TD == 8 ? 18 : 1;
  static constexpr unsigned IST_BASE = TD == 8 ? ITAB_BASE + (unsigned)UH * NT * 6 : TAB_BASE + 1536 + 256;
  static constexpr unsigned SMEM = TD == 8 ? IST_BASE + MAXC * 256 : 2 * HBUF;
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert((unsigned)PIECES * 1024 + 1024 <= HBUF, "no room for the 1 KiB dump behind the pieces of buffer 0");
  static_assert((unsigned)PIECES * 1024 <= HBUF - (TD == 8 ? 0 : TAB), "LDS-DMA pieces overlap the tables");
};

__device__ __attribute__((aligned(64))) unsigned int g_r32_zero[16];   // source of padding rows for the LDS-DMA

typedef __attribute__((ext_vector_type(4))) float r_f32x4;

#ifdef CBIM_EMU
#define R_SCHED_GROUP(mask, n) ((void)0)
#define R_SCHED_FENCE() ((void)0)
#define R_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
// one group of the instruction-interleave pattern of a scheduling region: `n` instructions of class `mask`
// (0x2 VALU, 0x4 SALU, 0x8 MFMA, 0x100 DS read, 0x200 DS write) come next
#define R_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define R_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define R_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

__device__ __forceinline__ void r_dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
#ifdef CBIM_EMU
  emu_global_load_lds16(gsrc, lds_wave_base);
#else
  // inline asm on purpose (conv_igemm.hip dma16): the compiler must not treat LGKM as out of order
  unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
  a = __builtin_amdgcn_readfirstlane(a);
  // M0 (the LDS base of the instruction) is saved and restored INSIDE the statement: no reserved register in the clobber
  // list (clang: "may lead to undefined behaviour"), nothing about M0 is hidden from the compiler
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(a), "v"(gsrc) : "memory");
#endif
}
__device__ __forceinline__ void r_wait_vm0() {
#ifndef CBIM_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ unsigned r_launder(unsigned v) {
#ifndef CBIM_EMU
  asm volatile("" : "+v"(v));
#endif
  return v;
}
__device__ __forceinline__ unsigned r_mul24(unsigned a, unsigned b) {
#ifdef CBIM_EMU
  return a * b;
#else
  return __umul24(a, b);
#endif
}
// XOR key of the 16-byte slot inside halo row (hd, hh, hw): a ds_read_b128 lane group of the B-fragment read covers the
// two h rows of the wave's patch and two k-groups; keys 0 / 2 on alternating rows give it 16 distinct cells of the
// 256-byte bank window (key hh & 3, right for the 4-row patches of conv_igemm.hip, measured 37 % conflict cycles here)
__device__ __forceinline__ unsigned r_swz(unsigned hh) { return (hh & 1u) << 1; }
// partner value of the butterfly step `msk` (1, 2, 4, 8, 16) of an all-reduce SUM over 32 lanes.  Steps 1 and 2 are quad
// permutes, steps 4 and 8 the half-row / row mirrors of the data-parallel-primitive path (vector-ALU rate): after steps 1, 2
// the four lanes of a quad hold the same value, so the mirror partner (other quad, any lane) is as good as lane ^ 4 —
// bit-identical to the xor butterfly; only step 16 crosses 16-lane rows and goes through the LDS crossbar.  (All five
// steps as ds_bpermute were 85 LDS round trips per tile in the multi-chunk epilogues.)
template <int MSK>
__device__ __forceinline__ float r_bfly(float v) {
#ifdef CBIM_EMU
  return __shfl_xor(v, MSK, 64);
#else
  if (MSK == 16) return __shfl_xor(v, 16, 64);
  constexpr int ctrl = MSK == 1 ? 0xB1 : MSK == 2 ? 0x4E : MSK == 4 ? 0x141 : 0x140;   // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false));
#endif
}
#define R_BFLY_ROUNDS(...) { constexpr int msk = 1; __VA_ARGS__ } { constexpr int msk = 2; __VA_ARGS__ } { constexpr int msk = 4; __VA_ARGS__ } { constexpr int msk = 8; __VA_ARGS__ } { constexpr int msk = 16; __VA_ARGS__ }

// exchange between 16-lane rows: a's odd rows (lanes 16..31, 48..63) <-> b's even rows (lanes 0..15, 32..47)
// (v_permlane16_swap_b32)
__device__ __forceinline__ void r_swap16(float& a, float& b) {
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

// TR: the input is transformed (InstanceNorm + ACT) in place in LDS after the LDS-DMA; !TR: used as it is
// MX: dgrad epilogue (x act'(xh) mask + the two InstanceNorm-backward sums); !MX: forward epilogue (moments)
// p.m_stats == nullptr with MX: the mask tensor is the ACTIVATED tensor a = relu(IN(x)) the caller materialised:
// act'(xh) = [a != 0] straight from the bf16 bits and, wherever the mask is open, a is the normalised value itself —
// no statistics table, no normalisation arithmetic in the epilogue
// TD: tile depth 8 (one 512-thread workgroup per CU) or 4 (two 256-thread workgroups per CU), see RGeom
// MC: several 32-channel chunks of Cin (units = (tile, chunk), streamed weights, optional second input tensor)
template <int ACT, bool TR, bool MX, int TD, bool MC>
__global__ void __launch_bounds__(RGeom<TD>::NT, TD == 8 ? 1 : 2) k_conv3_r32(R32Params p) {
  typedef RGeom<TD> G;
  constexpr int NT = G::NT, NW = G::NW, HP = G::HP, UH = G::UH;
  constexpr unsigned HBUF = G::HBUF;
  R_DYN_SMEM(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lv = lane & 15, lq = lane >> 4;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int n_tiles = p.N * tiles_per_n;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)(((long long)lb * n_tiles) / gridDim.x);
  const int t_end = (int)(((long long)(lb + 1) * n_tiles) / gridDim.x);
  if (t_begin >= t_end) return;
  const unsigned red_base = G::TAB_BASE;              // [NW][16][3] floats
  const unsigned mst_base = red_base + 1536;          // MX: (mean, rstd) of the 32 mask channels, 256 B
  const unsigned ist_base = G::IST_BASE;              // TR: (mean, rstd) of the input channels, 256 B per 32-channel chunk
  const int oc = blockIdx.y;                          // this workgroup's 32-channel chunk of Cout
  const int NC = MC ? p.NC : 1;                       // 32-channel chunks of Cin; a unit = (tile, chunk)
  // tile statistics reduced in the epilogue into LDS records (no per-lane running sums held through the MFMA phase):
  // multi-chunk kernels, and (R32_LS) the single-chunk forward whose transform already fills the register file
  constexpr bool LS = MC || (R32_LS_SINGLE && TR && !MX);
  constexpr bool stream_w = MC;                       // weights of the next unit replace this unit's as they fall dead

  // ---- wave = (cout half ch, voxel group vg); lane = (voxel lv of a 2x8 patch, k-group / row-group lq) -----------
  // TD = 8: 4 voxel groups of one h-pair; TD = 4: 2 voxel groups of two h-pairs.  n-tile nt = hp * TD + plane.
  const int ch = wave & 1, vg = wave >> 1;
  const int tw = lv & 7;
  // weights: fragment of tap tp = A operand [16 couts][32 channels]: lane (cout lv, channels 8*lq..+7) = 16 bytes of
  // the packed image [tap][kg = lq>>1][half = lq&1][32 couts][8]
  // packed image (conv_igemm.hip): [cout block of BN][Cin chunk][tap][kg = lq>>1][half = lq&1][BN couts][8], BN = 32 / 64
  u32x4 wf[27];
  const unsigned w_tap = 4u * (unsigned)p.BN * 16u;    // bytes per tap
  const unsigned char* const w_lane = (const unsigned char*)p.w +
      (size_t)(p.BN == 64 ? oc >> 1 : oc) * (size_t)NC * 27u * w_tap +
      (unsigned)((lq * p.BN) + (p.BN == 64 ? (oc & 1) * 32 : 0) + 16 * ch + lv) * 16;
  {
    const unsigned char* wp = w_lane;
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) wf[tp] = *(const u32x4*)(wp + (size_t)tp * w_tap);
#ifndef CBIM_EMU
    // The compiler tracks these loads as possibly pending at the loop back-edge and guards the first MFMA of every tile
    // with s_waitcnt vmcnt(1) — which, with the LDS-DMA pieces it cannot see in flight, waits for the DMA to land.
    // Passing the registers through an empty asm makes it wait once, here, and forget the loads.
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) asm volatile("" : "+v"(wf[tp]));
#endif
  }
  // B operand (voxels): fragment of plane i at tap (kh, kw) = 16 bytes at row (i, th + kh, tw + kw), slot
  // lq ^ r_swz(th + kh); base per (h-pair, kh); plane and kw are immediates (i * 6400 + kw * 64)
  int thp[HP];
  unsigned fb[HP][3];
#pragma unroll
  for (int hp = 0; hp < HP; ++hp) {
    thp[hp] = 2 * (vg * HP + hp) + (lv >> 3);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
      fb[hp][kh] = (unsigned)((thp[hp] + kh) * 10 + tw) * R_RB + (((unsigned)lq ^ r_swz((unsigned)(thp[hp] + kh))) << 4);
  }

  // ---- halo items of this thread (the same for every tile): position inside the (TD+2)x10x10 box ---------------
  // LDS-DMA item q = tid + NT u is PHYSICAL (LDS byte q*16 = row q>>2, slot q&3), source slot = (q&3) ^ r_swz(hh).
  // The position is decoded per tile from the laundered thread index (~10 VALU per item) instead of living in
  // registers next to the weights.
  auto item_pos = [&](unsigned tl, int u) -> unsigned {   // hd | hh << 8 | hw << 16 | exists << 24
    const unsigned row = (tl + (unsigned)NT * (unsigned)u) >> 2;
    const unsigned hd = (row * 5243u) >> 19;              // row / 100 for row < 1024
    const unsigned r2 = row - hd * 100u;
    const unsigned hh = (r2 * 205u) >> 11;                // r2 / 10 for r2 < 100
    const unsigned hw = r2 - hh * 10u;
    return hd | (hh << 8) | (hw << 16) | (row < (unsigned)G::HROWS ? 1u << 24 : 0u);
  };
  const unsigned my_slot = (unsigned)tid & 3u;

  struct TilePos { int n, td, th, tw, cc; };            // a unit: tile + Cin chunk
  auto advance = [&](TilePos& u) {
    if (++u.cc < NC) return;
    u.cc = 0;
    if (++u.tw == p.tiles_w) { u.tw = 0; if (++u.th == p.tiles_h) { u.th = 0; if (++u.td == p.tiles_d) { u.td = 0; ++u.n; } } }
  };
  TilePos cur, nxt;
  {
    const int tt = t_begin % tiles_per_n;
    cur.n = t_begin / tiles_per_n; cur.td = tt / (p.tiles_w * p.tiles_h); cur.th = (tt / p.tiles_w) % p.tiles_h; cur.tw = tt % p.tiles_w;
    cur.cc = 0;
    nxt = cur;
    advance(nxt);
  }

  const unsigned x_sb = (unsigned)p.x_stride * 2u, x2_sb = (unsigned)p.x2_stride * 2u;
  // per-item constants in LDS (TD = 8): decoding the position and multiplying out the offset for every item of every
  // tile was ~50 vector/scalar instructions per item in which the wave issues no MFMA; with the table an item costs two
  // LDS reads, the range test and a 64-bit add
  unsigned* const tab_off = (unsigned*)(smem + G::ITAB_BASE);
  unsigned short* const tab_pos = (unsigned short*)(smem + G::ITAB_BASE + (unsigned)UH * NT * 4);
  if (G::ITAB) {
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      const unsigned pk = item_pos((unsigned)tid, u);
      const unsigned hd = pk & 255u, hh = (pk >> 8) & 255u, hw = (pk >> 16) & 255u;
      const unsigned rel = r_mul24(r_mul24(hd, (unsigned)p.Hi) + hh, (unsigned)p.Wi) + hw;
      // MC: row of the box (bytes = rel * row stride of the chunk's tensor); else the byte offset itself
      tab_off[tid + NT * u] = MC ? rel : r_mul24(rel, x_sb) + ((my_slot ^ r_swz(hh)) << 4);
      tab_pos[tid + NT * u] = (unsigned short)(hd | (hh << 4) | (hw << 8) | ((pk >> 24) << 12));
    }
    // (read back by the same thread only: no barrier needed, the prologue has one anyway)
  }
  // position (hd, hh, hw, exists) and source offset of item u
  auto item_get = [&](int u, unsigned sb, unsigned& hd, unsigned& hh, unsigned& hw, bool& exists, unsigned& off) {
    unsigned rel;
    if (G::ITAB) {
      const unsigned tl = r_launder((unsigned)tid);      // (keeps the loop-invariant reads inside the tile loop)
      const unsigned ps = tab_pos[tl + NT * u];
      rel = tab_off[tl + NT * u];
      hd = ps & 15u; hh = (ps >> 4) & 15u; hw = (ps >> 8) & 15u; exists = (ps >> 12) != 0;
    } else {
      const unsigned pk = item_pos(r_launder((unsigned)tid), u);
      hd = pk & 255u; hh = (pk >> 8) & 255u; hw = (pk >> 16) & 255u; exists = (pk >> 24) != 0;
      rel = r_mul24(r_mul24(hd, (unsigned)p.Hi) + hh, (unsigned)p.Wi) + hw;
    }
    off = (G::ITAB && !MC) ? rel : r_mul24(rel, sb) + ((my_slot ^ r_swz(hh)) << 4);
  };
  // source address of halo item u of tile `tp` (in range: the tensor; padding: 64 zero bytes) ---------------------
  auto item_src = [&](const TilePos& tp, int u) -> const unsigned char* {
    const int id0 = tp.td * TD - p.pD, ih0 = tp.th * 8 - p.pH, iw0 = tp.tw * 8 - p.pW;
    const long long org = (((long long)tp.n * p.Di + id0) * p.Hi + ih0) * p.Wi + iw0;
    // chunk cc of the (virtually concatenated) input: chunks below c_split come from x, the others from x2
    const bool second = MC && tp.cc >= p.c_split;
    const unsigned sb = second ? x2_sb : x_sb;
    const unsigned char* tbase = (second ? (const unsigned char*)p.x2 + (tp.cc - p.c_split) * R_RB
                                         : (const unsigned char*)p.x + tp.cc * R_RB) + org * (long long)sb;   // wave-uniform
    unsigned hd, hh, hw, off;
    bool exists;
    item_get(u, sb, hd, hh, hw, exists, off);
    const bool ld = !(R_DBG & 1) && exists && (unsigned)(id0 + (int)hd) < (unsigned)p.Di &&
                    (unsigned)(ih0 + (int)hh) < (unsigned)p.Hi && (unsigned)(iw0 + (int)hw) < (unsigned)p.Wi;
    return (ld ? tbase : (const unsigned char*)g_r32_zero) + (ld ? off : 0u);
  };
  // item u of tile `tp` into LDS buffer `buf` (an asynchronous 1 KiB piece per wave).  A piece that lies wholly past
  // the box (last u of the upper waves) copies zeros into a 1 KiB dump at the end of buffer 0 that nothing reads:
  // no branch, and every wave has the same number of pieces in flight (the TR path counts them).
  auto dma_item = [&](const TilePos& tp, int u, unsigned buf) {
    const bool past = (wave * 64 + NT * u) >= G::PIECES * 64;     // wave-uniform
    const unsigned char* src = item_src(tp, u);                   // (items past the box: the zero page)
    r_dma16(src, smem + (past ? (unsigned)G::PIECES * 1024u : buf + (unsigned)(wave * 64 + NT * u) * 16));
  };
  // TR path: the raw halo arrives by the same LDS-DMA; a wave transforms IN PLACE exactly the 1 KiB pieces it has
  // fetched itself (lane = the piece's 16-byte item), so the only ordering needed is the wave's own counted vmcnt —
  // no workgroup barrier between the fetch and the transform.  ds_read_b128 -> (x - mean) * rstd as packed f32 pairs
  // -> bf16 pairs -> ReLU as v_pk_max_i16 against 0 -> ds_write_b128.  The statistics of the item's channel chunk
  // are one 64-byte row of a 256-byte LDS table.  Padding rows stay zero.
  auto tr_xform = [&](const TilePos& tp, int u, unsigned buf) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef short s2_t __attribute__((ext_vector_type(2)));
    const int id0 = tp.td * TD - p.pD, ih0 = tp.th * 8 - p.pH, iw0 = tp.tw * 8 - p.pW;
    unsigned hd, hh, hw, off;
    bool exists;
    item_get(u, x_sb, hd, hh, hw, exists, off);
    const bool ld = exists && (unsigned)(id0 + (int)hd) < (unsigned)p.Di && (unsigned)(ih0 + (int)hh) < (unsigned)p.Hi &&
                    (unsigned)(iw0 + (int)hw) < (unsigned)p.Wi;
    if (exists) {                                     // (items past the box do not exist in LDS)
      unsigned char* cell = smem + buf + ((unsigned)tid + (unsigned)NT * (unsigned)u) * 16u;
      const float* is = (const float*)(smem + ist_base) + tp.cc * 64 + ((my_slot ^ r_swz(hh)) << 4);   // logical chunk of this cell
      const u32x4 raw = *(const u32x4*)cell;
      const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
      unsigned ow[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 q4 = *(const f32x4*)(is + 4 * j);   // (mean, rstd) of channels 2j, 2j+1
        const f2_t x = {__uint_as_float(rw[j] << 16), __uint_as_float(rw[j] & 0xffff0000u)};
        const f2_t m = {q4.x, q4.z}, r = {q4.y, q4.w};
        const f2_t y = (x - m) * r;
        unsigned o = pk_bf16(y.x, y.y);
        if (ACT == CBIM_ACT_RELU) {
          s2_t h = __builtin_bit_cast(s2_t, o);
          const s2_t z = {0, 0};
          h = __builtin_elementwise_max(h, z);      // a negative bf16 is a negative int16
          o = __builtin_bit_cast(unsigned, h);
        }
        ow[j] = ld ? o : 0u;                        // padding cells (zeros from the DMA) stay zero
      }
      *(u32x4*)cell = u32x4{ow[0], ow[1], ow[2], ow[3]};
    }
  };
  auto wait_vm = [&](int n) {   // at most n vector-memory operations of this wave still in flight
#ifndef CBIM_EMU
    if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    (void)n;
#endif
  };

  // ---- per-lane statistics: after the epilogue exchange a lane owns channel chunk cidx = 2*ch + (lq >> 1) -------
  const int cidx = 2 * ch + (lq >> 1);
  const bool c_ok = oc * 32 + cidx * 8 < p.Cout;
  float s0[8], s1[8], sh[MX ? 1 : 8];
  float cnt = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; sh[MX ? 0 : j] = 0.f; }
  int run_n = cur.n;
  bool shift_set = false;     // workgroup-uniform
  const bool want_part = p.partials != nullptr;
  // combine the lanes' sums of image n into this workgroup's record and reset them (all threads call it).  The 32 lanes
  // that hold a channel chunk share ONE shift (taken from the group's first lane when the sums start), so their sums
  // simply add: 5 butterfly rounds over 17 independent values, one conversion to (n, mean, M2) per channel and wave.
  // MC: the per-lane sums would be ~25 more registers alive through the MFMA phase (measured: 21-29 spilled VGPRs whose
  // reloads next to the unit barrier drain the weight stream).  The statistics of a tile are reduced over the lanes in its
  // epilogue — once per NC units — and merged into the wave's running record in LDS (red, wave-private).
  auto tile_stats = [&](float (&t0)[8], float (&t1)[8], const float (&tsh)[MX ? 1 : 8], float tcnt) {
    float* red = (float*)(smem + red_base);
    R_BFLY_ROUNDS(
      float u0[8], u1[8];
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { u0[j] = r_bfly<msk>(t0[j]); u1[j] = r_bfly<msk>(t1[j]); }
      const float tc = r_bfly<msk>(tcnt);
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { t0[j] += u0[j]; t1[j] += u1[j]; }
      tcnt += tc;
    )
    if ((lane & 31) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float* rr = red + ((wave * 16) + (lq >> 1) * 8 + j) * 3;
        if (MX) { rr[1] += t0[j]; rr[2] += t1[j]; }
        else {
          const Moments a = {rr[0], rr[1], rr[2]};
          const Moments b = moments_from_shifted(tcnt, tsh[MX ? 0 : j], t0[j], t1[j]);
          const Moments m = moments_merge(a, b);
          rr[0] = m.n; rr[1] = m.mean; rr[2] = m.m2;
        }
      }
    }
  };
  auto flush_stats = [&](int n) {
    __syncthreads();
    float* red = (float*)(smem + red_base);
    if (!LS) {
    R_BFLY_ROUNDS(                             // lanes differing in bits 0..4 (voxel, lq & 1) hold the same channels
      float t0[8], t1[8];
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { t0[j] = r_bfly<msk>(s0[j]); t1[j] = r_bfly<msk>(s1[j]); }
      const float tc = r_bfly<msk>(cnt);
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { s0[j] += t0[j]; s1[j] += t1[j]; }
      cnt += tc;
    )
    if ((lane & 31) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Moments a;
        if (MX) { a.n = 0.f; a.mean = s0[j]; a.m2 = s1[j]; }
        else a = moments_from_shifted(cnt, sh[MX ? 0 : j], s0[j], s1[j]);
        float* rr = red + ((wave * 16) + (lq >> 1) * 8 + j) * 3;
        rr[0] = a.n; rr[1] = a.mean; rr[2] = a.m2;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    cnt = 0.f;
    shift_set = false;
    }
    __syncthreads();
    if (tid < 32 && oc * 32 + tid < p.Cout) {
      Moments a = {0.f, 0.f, 0.f};
      for (int g = 0; g < NW / 2; ++g) {   // the voxel groups of this channel's cout half
        const float* rr = red + (((2 * g + (tid >> 4)) * 16) + (tid & 15)) * 3;
        if (MX) { a.mean += rr[1]; a.m2 += rr[2]; }
        else { Moments b = {rr[0], rr[1], rr[2]}; a = moments_merge(a, b); }
      }
      const size_t o = (((size_t)n * p.P + lb) * p.Cout + oc * 32 + tid) * 3;
      p.partials[o] = a.n; p.partials[o + 1] = a.mean; p.partials[o + 2] = a.m2;
    }
    if (LS) {                                 // the next image starts from empty records
      __syncthreads();
      for (int i = tid; i < NW * 16 * 3; i += NT) red[i] = 0.f;
    }
  };
  if (LS) {
    for (int i = tid; i < NW * 16 * 3; i += NT) ((float*)(smem + red_base))[i] = 0.f;   // (the prologue barriers follow)
  }
  if (want_part && tid < 32 && oc * 32 + tid < p.Cout) {
    // empty records (n = 0 merges as the identity): images this strip does not touch, and the records lb + k * grid
    // of a buffer sized for more workgroups than this launch has (p.P = cbim_conv3d_num_tiles records per image)
    const int n_first = t_begin / tiles_per_n, n_last = (t_end - 1) / tiles_per_n;
    for (int n = 0; n < p.N; ++n)
      for (unsigned r = lb; r < (unsigned)p.P; r += gridDim.x)
        if (r != lb || n < n_first || n > n_last) {
          const size_t o = (((size_t)n * p.P + r) * p.Cout + oc * 32 + tid) * 3;
          p.partials[o] = 0.f; p.partials[o + 1] = 0.f; p.partials[o + 2] = 0.f;
        }
  }
  // (mean, rstd) tables in LDS: the mask tensor's channels (MX) / the input's channels (TR); refreshed at an image
  // change, always followed by a workgroup barrier before they are read
  int mst_n = -1, ist_n = -1;
  const bool mask_is_act = MX && p.m_stats == nullptr;      // workgroup-uniform
  auto load_mstats = [&](int n) {
    if (MX && !mask_is_act && n != mst_n) {
      if (tid < 64) {
        const int c = tid >> 1;
        ((float*)(smem + mst_base))[tid] = oc * 32 + c < p.Cout ? p.m_stats[((size_t)n * p.Cout + oc * 32 + c) * 2 + (tid & 1)] : (float)(tid & 1);
      }
      mst_n = n;
    }
  };
  auto load_istats = [&](int n) {
    if (TR && n != ist_n) {
      for (int i = tid; i < NC * 64; i += NT) ((float*)(smem + ist_base))[i] = p.in_stats[(size_t)n * (NC * 64) + i];
      ist_n = n;
    }
  };

  // ---- prologue: first tile's halo ---------------------------------------------------------------------------------
  load_mstats(cur.n);
  load_istats(cur.n);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < UH; ++u) dma_item(cur, u, 0);
  r_wait_vm0();
  __syncthreads();
  if (TR) {
#pragma unroll
    for (int u = 0; u < UH; ++u) tr_xform(cur, u, 0);
    __syncthreads();
  }

  r_f32x4 acc[8];
  const int n_my = (t_end - t_begin) * NC;              // units
  for (int t = 0; t < n_my; ++t) {
    const unsigned buf = (unsigned)(t & 1) * HBUF, obuf = HBUF - buf;
    const bool more = t + 1 < n_my;
    const bool first_cc = cur.cc == 0, last_cc = cur.cc == NC - 1;
    // the next tile's transforms run during THIS tile: at an image change the statistics table is rewritten first (no
    // reader is active here: the previous tile's transforms ended before its barrier)
    if (TR && more && nxt.n != ist_n) { load_istats(nxt.n); __syncthreads(); }
    TilePos nx;
    nx.n = more ? nxt.n : cur.n; nx.td = more ? nxt.td : cur.td; nx.th = more ? nxt.th : cur.th; nx.tw = more ? nxt.tw : cur.tw;
    nx.cc = more ? nxt.cc : cur.cc;
    if (first_cc) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[nt] = r_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* const w_next = w_lane + (size_t)nx.cc * 27u * w_tap;
    // (B) 9 (kh, kw) steps x HP patches: the TD+2 plane fragments of a patch stream through a ring of 5 registers,
    //     plane i feeds the MFMAs (n-tile i, kd 0), (i-1, kd 1), (i-2, kd 2).  The halo of the next tile is fetched by
    //     LDS-DMA during the first steps; TR: a piece is transformed in place four steps after its fetch.
    //     Branch-free (the strip's last tile re-fetches itself into the idle buffer).
    if (!(R_DBG & 2)) {
      constexpr int RING = 5, PLN = TD + 2, SEQ = 9 * HP * PLN;   // fragment reads of a tile, in order
      u32x4 xr[RING];
      auto frag_addr = [&](int e) -> unsigned {                   // e = ((kh*3 + kw) * HP + hp) * PLN + plane
        const int i = e % PLN, hp = (e / PLN) % HP, s = e / (PLN * HP);
        return buf + fb[hp][s / 3] + (unsigned)((s % 3) * R_RB) + (unsigned)(i * 100 * R_RB);
      };
#pragma unroll
      for (int e = 0; e < RING - 1; ++e) xr[e] = *(const u32x4*)(smem + frag_addr(e));
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int kh = s / 3, kw = s % 3;
        {
          // per-step share of the next tile's halo work: UH items over 9 steps.  DMA: items 2s, 2s+1 in the first
          // steps; TR: transform of items 2(s-4).. four steps later (own pieces: counted vmcnt is the whole sync)
          constexpr int FS = (UH + 1) / 2;                   // fetch steps: 4 / 5
          if (s < FS) {
            dma_item(nx, 2 * s, obuf);
            if (2 * s + 1 < UH) dma_item(nx, 2 * s + 1, obuf);
          }
          if (TR && s >= 9 - FS) {
            const int k = s - (9 - FS);                      // 0 .. FS-1: transforms items 2k, 2k+1
            // own pieces younger than item 2k+1, plus (streamed weights) the 3 fragment loads of each step since
            wait_vm((UH - 2 * k - 2 > 0 ? UH - 2 * k - 2 : 0) + (stream_w ? 3 * (9 - FS) : 0));
            tr_xform(nx, 2 * k, obuf);
            if (2 * k + 1 < UH) tr_xform(nx, 2 * k + 1, obuf);
          }
        }
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) {
#pragma unroll
          for (int i = 0; i < PLN; ++i) {
            const int e = (s * HP + hp) * PLN + i;
            // read RING-1 entries ahead; the fences keep the compiler from sinking the read next to its use (it then
            // waits a full LDS round trip every third MFMA: measured 57 % of the MFMA rate)
            if (e + RING - 1 < SEQ && !(R_DBG & 128)) xr[(e + RING - 1) % RING] = *(const u32x4*)(smem + frag_addr(e + RING - 1));
            R_SCHED_FENCE();
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
              const int pl = i - kd;
              if (pl >= 0 && pl < TD)
                acc[hp * TD + pl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[(kd * 3 + kh) * 3 + kw]),
                                                                            __builtin_bit_cast(bf16x8, xr[e % RING]), acc[hp * TD + pl], 0, 0, 0);
            }
            R_SCHED_FENCE();
          }
        }
        // the three fragments of this (kh, kw) are dead until step s of the next unit: reload them for it now
        if (stream_w) {
#pragma unroll
          for (int kd = 0; kd < 3; ++kd)
            wf[(kd * 3 + kh) * 3 + kw] = *(const u32x4*)(w_next + (size_t)((kd * 3 + kh) * 3 + kw) * w_tap);
        }
      }
    }
    // epilogue operands (mask / residual rows of the lane's four output voxels): requested BEFORE the tile barrier, so
    // their memory latency overlaps the barrier and the wait for the LDS-DMA (requested inside the epilogue, one pair
    // ahead, every pair paid most of a global-memory round trip: the epilogue was 38 % of the masked dgrad)
    const int n = cur.n;
    const int od0 = cur.td * TD, oh0 = cur.th * 8, ow0 = cur.tw * 8;
    const long long orow = (((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0;
    const unsigned y_sb = (unsigned)p.y_stride * 2u, res_sb = (unsigned)p.res_stride * 2u, mx_sb = (unsigned)p.mx_stride * 2u;
    unsigned char* y_tile = (unsigned char*)p.y + orow * (long long)y_sb;
    const unsigned char* res_tile = (const unsigned char*)p.res + orow * (long long)res_sb;
    const unsigned char* mx_tile = (const unsigned char*)p.mx + orow * (long long)mx_sb;
    const unsigned plane2 = 2u * r_mul24((unsigned)p.Ho, (unsigned)p.Wo);
    const unsigned cb = (unsigned)(oc * 4 + cidx) * 16u;
    // after the exchange this lane owns, for pair pr = (hp, pp): chunk cidx of voxel (2pp + (lq&1), th(hp), tw)
    constexpr int NPAIR = 4;                               // 8 n-tiles
    // (the offsets below depend on the lane only: laundering the lane's plane bit keeps the compiler from hoisting the
    //  four of them — and the 64-bit addresses built on them — out of the tile loop, where they were spilled to scratch
    //  and reloaded in every epilogue: 12-17 spilled VGPRs in the masked multi-chunk instantiations)
    const unsigned lq1 = r_launder((unsigned)(lq & 1));
    auto pair_rel = [&](int pr) -> unsigned {
      const int hp = pr / (TD / 2), pp = pr % (TD / 2);
      return r_mul24(r_mul24(lq1, (unsigned)p.Ho) + (unsigned)thp[hp], (unsigned)p.Wo) + (unsigned)tw + (unsigned)pp * plane2;
    };
    auto pair_in = [&](int pr) -> bool {
      const int hp = pr / (TD / 2), pp = pr % (TD / 2);
      return oh0 + thp[hp] < p.Ho && ow0 + tw < p.Wo && od0 + 2 * pp + (lq & 1) < p.Do;
    };
    u32x4 rq[NPAIR];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) rq[pr] = u32x4{0u, 0u, 0u, 0u};
    if (last_cc) {      // (a branch, not a predicate: the units in between must not touch — reload — any of this)
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) {
        const bool in = pair_in(pr) && c_ok && !(R_DBG & (4 | 8));
        const unsigned rel = pair_rel(pr);
        if (in) {
          if (MX) rq[pr] = *(const u32x4*)(mx_tile + (r_mul24(rel, mx_sb) + cb));
          else if (p.res) rq[pr] = *(const u32x4*)(res_tile + (r_mul24(rel, res_sb) + cb));
        }
      }
      load_mstats(cur.n);
    }
    // (D) ONE barrier per tile: every wave is done with `buf`, the other buffer is complete (own LDS-DMA waited for,
    //     LDS stores drained)
    // own LDS-DMA pieces landed.  Streamed weights: the fragment loads issued after the last piece (steps FS-1 .. 8) may
    // stay in flight (vector-memory operations complete in order)
    if (stream_w) wait_vm(3 * (10 - (UH + 1) / 2));
    else r_wait_vm0();
    __syncthreads();
    // (C) epilogue of this tile — no barrier inside (except at an image change); its stores drain under the next
    //     tile's MFMAs
    if (last_cc && !(R_DBG & 4)) {
      if (want_part && n != run_n) { flush_stats(run_n); run_n = n; }
      // statistics of this tile: running per-lane sums (single chunk) or tile-local sums reduced below (MC)
      float l0[8], l1[8], lsh[MX ? 1 : 8];
      float lcnt = 0.f;
      bool lset = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) { l0[j] = 0.f; l1[j] = 0.f; lsh[MX ? 0 : j] = 0.f; }
      float (&S0)[8] = LS ? l0 : s0;
      float (&S1)[8] = LS ? l1 : s1;
      float (&SH)[MX ? 1 : 8] = LS ? lsh : sh;
      float& CNT = LS ? lcnt : cnt;
      bool& SHSET = LS ? lset : shift_set;
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) {
        const bool in = pair_in(pr);
        const unsigned rel = pair_rel(pr);
        // accumulator register r of n-tile nt = channel 16*ch + 4*lq + r of voxel (plane, th, tw).  Swapping the
        // registers of the pair's first n-tile in the odd 16-lane rows with those of its second n-tile in the even
        // rows leaves the lane with 8 consecutive channels (chunk cidx) of ONE voxel: plane 2pp (lq even) / 2pp+1 (odd)
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = acc[2 * pr][r], b = acc[2 * pr + 1][r];
          r_swap16(a, b);
          v[r] = a;
          v[4 + r] = b;
        }
        // packed f32 pairs from here on (v_pk_add / v_pk_mul / v_pk_fma); lanes outside the tensor or beyond Cout
        // contribute with weight 0 instead of branching
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const float live = (in && c_ok) ? 1.f : 0.f;
        const f2_t live2 = {live, live};
        const unsigned rw[4] = {rq[pr].x, rq[pr].y, rq[pr].z, rq[pr].w};
        if (MX && mask_is_act) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f2_t a = {__uint_as_float(rw[j] << 16), __uint_as_float(rw[j] & 0xffff0000u)};
            f2_t g;
            g.x = (rw[j] & 0xffffu) != 0u ? v[2 * j] : 0.f;
            g.y = (rw[j] >> 16) != 0u ? v[2 * j + 1] : 0.f;
            v[2 * j] = g.x;
            v[2 * j + 1] = g.y;
            const f2_t gl = g * live2;
            f2_t a0 = {S0[2 * j], S0[2 * j + 1]}, a1 = {S1[2 * j], S1[2 * j + 1]};
            a0 = a0 + gl;
            a1 = __builtin_elementwise_fma(gl, a, a1);
            S0[2 * j] = a0.x; S0[2 * j + 1] = a0.y; S1[2 * j] = a1.x; S1[2 * j + 1] = a1.y;
          }
        } else if (MX) {
          const float* ms = (const float*)(smem + mst_base) + cidx * 16;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 q4 = *(const f32x4*)(ms + 4 * j);   // (mean, rstd) of channels 2j, 2j+1
            const f2_t x = {__uint_as_float(rw[j] << 16), __uint_as_float(rw[j] & 0xffff0000u)};
            const f2_t m = {q4.x, q4.z}, r = {q4.y, q4.w};
            const f2_t xh = (x - m) * r;
            f2_t g;
            if (ACT == CBIM_ACT_RELU) { g.x = xh.x > 0.f ? v[2 * j] : 0.f; g.y = xh.y > 0.f ? v[2 * j + 1] : 0.f; }
            else { g.x = v[2 * j]; g.y = v[2 * j + 1]; }
            v[2 * j] = g.x;
            v[2 * j + 1] = g.y;
            const f2_t gl = g * live2;
            f2_t a0 = {S0[2 * j], S0[2 * j + 1]}, a1 = {S1[2 * j], S1[2 * j + 1]};
            a0 = a0 + gl;
            a1 = __builtin_elementwise_fma(gl, xh, a1);
            S0[2 * j] = a0.x; S0[2 * j + 1] = a0.y; S1[2 * j] = a1.x; S1[2 * j + 1] = a1.y;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // residual (zeros when there is none)
            v[2 * j] += __uint_as_float(rw[j] << 16);
            v[2 * j + 1] += __uint_as_float(rw[j] & 0xffff0000u);
          }
          if (want_part && !(R_DBG & 32)) {
            if (!SHSET) {            // common shift of the 32 lanes that hold this chunk: any finite value near
              SHSET = true;             // the data works (shifted moments); taken from the group's first lane
#pragma unroll
              for (int j = 0; j < 8; ++j) SH[MX ? 0 : j] = __shfl(v[j], lane & 32, 64);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f2_t x = {v[2 * j], v[2 * j + 1]}, hs = {SH[MX ? 0 : 2 * j], SH[MX ? 0 : 2 * j + 1]};
              const f2_t d = (x - hs) * live2;
              f2_t a0 = {S0[2 * j], S0[2 * j + 1]}, a1 = {S1[2 * j], S1[2 * j + 1]};
              a0 = a0 + d;
              a1 = __builtin_elementwise_fma(d, d, a1);
              S0[2 * j] = a0.x; S0[2 * j + 1] = a0.y; S1[2 * j] = a1.x; S1[2 * j + 1] = a1.y;
            }
          }
        }
        if (in && c_ok && !(R_DBG & 16)) *(u32x4*)(y_tile + (r_mul24(rel, y_sb) + cb)) = Elem<bf16_tag>::pack(v);
        CNT += live;
      }
      if (LS && want_part && !(R_DBG & 32)) tile_stats(l0, l1, lsh, lcnt);
    }
    cur = nxt;
    advance(nxt);
  }
  if (want_part && !(R_DBG & 64)) flush_stats(run_n);
}

}  // namespace cbim

using namespace cbim;

static int64_t g_r32_min_voxels = 262144;   // same threshold as the 8x8x8 tile configuration of k_conv_igemm
static int r32_tile_depth();

// mode (frozen at 2 since round 3): 0 off, 1 the single-chunk layers only (Cin = 32, Cout <= 32), 2 (default) every multiple of 32
static int r32_mode() {
  static const int m = 2;
  return m;
}
bool cbim_conv_r32_eligible(const cbim_conv_desc* d, const void* x2, int cin_split, const float* in_stats, const void* res,
                            const void* mask_x) {
  const int mode = r32_mode();
  if (!mode || d->dtype != CBIM_BF16) return false;
  if (d->kD != 3 || d->kH != 3 || d->kW != 3 || d->Cin % 32 != 0 || !(d->Cout <= 32 || d->Cout % 32 == 0)) return false;
  if (mode == 1 && (d->Cin != 32 || d->Cout > 32 || x2)) return false;
  if (d->Cin > 32 * RGeom<8>::MAXC || (d->Cin > 32 && r32_tile_depth() != 8)) return false;
  if (x2 && (cin_split <= 0 || cin_split >= d->Cin || cin_split % 32 != 0)) return false;
  if (d->Do < 8 || d->Ho < 8 || d->Wo < 8) return false;
  if (!(d->act == CBIM_ACT_RELU || d->act == CBIM_ACT_NONE)) return false;
  if (mask_x && (in_stats || res)) return false;   // masked epilogue: raw input, no accumulate tensor
  const int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  if (g_r32_min_voxels == 0) return true;            // forced (tests, tools)
  if (d->Cin == 32 && d->Cout <= 32) return S >= g_r32_min_voxels;   // at least ~2 tiles per CU
  // several chunks, measured against k_conv_igemm (profiles/r02_p_conv_bench_ab.txt).  Raw input (dgrad, LDS-DMA only):
  // ahead on every shape, 2-24 %.  Transformed input (forward: InstanceNorm + activation applied in LDS once per unit
  // and per Cout chunk): ahead for single-chunk inputs, where the 64-wide n-blocks of k_conv_igemm are half empty
  // (Cout = 32, 96, ...) and at 32^3 where its tiles are too few; 10-15 % behind on 64-multiples at >= 64^3.
  const int64_t tiles = (int64_t)d->N * ((d->Do + 7) / 8) * ((d->Ho + 7) / 8) * ((d->Wo + 7) / 8);
  const int n_oc = (d->Cout + 31) / 32;
  if (S >= g_r32_min_voxels) return !in_stats || d->Cin == 32 || d->Cout % 64 != 0;
  // low-resolution layers: enough (tile strip, Cout chunk) workgroups to fill half the chip.  Raw inputs (round 3,
  // profiles/r03_t_small_layers_r32.txt): 576->512 @16^3 138 -> 97 us forward, 124 -> 86 us dgrad; 128->512 @16^3 forward
  // 46 -> 35 us; 256->64 @32^3 dgrad 68 -> 50 us; below 128 workgroups (256->256 @16^3, everything at 8^3) the split-K
  // k_conv_igemm stays ahead
  if (!in_stats) return tiles * n_oc >= 128;
  return S >= g_r32_min_voxels / 8 && tiles * n_oc >= 192;
}

extern "C" int64_t cbim_conv_r32_min_voxels(int64_t v) {
  const int64_t old = g_r32_min_voxels;
  if (v >= 0) g_r32_min_voxels = v;
  return old;
}

// 8: one 512-thread workgroup per CU on 8x8x8 tiles (default); 4: two 256-thread workgroups per CU on 4x8x8 tiles —
// measured 5-15 % slower on 32->32 @128^3 (more halo per voxel, and the MFMA phase, not the overlap, is what limits)
static int g_r32_td = 8;
static int r32_tile_depth() { return g_r32_td; }
extern "C" int cbim_conv_r32_tile_depth(int td) {
  const int old = g_r32_td;
  if (td == 4 || td == 8) g_r32_td = td;
  return old;
}

int64_t cbim_conv_r32_grid(const cbim_conv_desc* d) {
  const int td = r32_tile_depth();
  const int64_t n_tiles = (int64_t)d->N * ((d->Do + td - 1) / td) * ((d->Ho + 7) / 8) * ((d->Wo + 7) / 8);
  const int n_oc = (d->Cout + 31) / 32;
  int64_t cap = (td == 8 ? 256 : 512) / n_oc;      // about one workgroup per CU over all Cout chunks
  if (cap < 1) cap = 1;
  int64_t g = n_tiles < cap ? n_tiles : cap;
  // several Cout chunks: strips in multiples of 8 keep the chunks of one strip on one XCD (workgroup b runs on XCD b % 8,
  // b = x + grid.x * y): the second reader of a halo hits that XCD's L2 (conv_rw.hip: cbim_conv_rw_grid)
  if (n_oc > 1 && g >= 8) g &= ~(int64_t)7;
  return g;
}

template <int ACT, bool TR, bool MX, int TD, bool MC>
static int r32_launch_mc(const R32Params& p, dim3 grid, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv3_r32<ACT, TR, MX, TD, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  CBIM_LAUNCH((k_conv3_r32<ACT, TR, MX, TD, MC>), grid, dim3(RGeom<TD>::NT), (size_t)RGeom<TD>::SMEM, st, p);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "conv r32 launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}
template <int ACT, bool TR, bool MX>
static int r32_launch(const R32Params& p, dim3 grid, hipStream_t st) {
  if (p.NC > 1) return r32_launch_mc<ACT, TR, MX, 8, true>(p, grid, st);     // (eligibility: tile depth 8 only)
  return r32_tile_depth() == 8 ? r32_launch_mc<ACT, TR, MX, 8, false>(p, grid, st) : r32_launch_mc<ACT, TR, MX, 4, false>(p, grid, st);
}

int cbim_conv_r32_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                         int cin_split, const float* in_stats, const void* w_packed, const void* res, int64_t res_stride, const void* mask_x,
                         int64_t mask_stride, const float* mask_stats, void* y, int64_t y_stride, float* partials,
                         void* stream) {
  R32Params p;
  p.x = x; p.x_stride = x_stride; p.in_stats = in_stats; p.w = w_packed;
  p.NC = d->Cin / 32; p.cin_bytes = d->Cin * 2;
  p.x2 = x2; p.x2_stride = x2 ? x2_stride : x_stride; p.c_split = x2 ? cin_split / 32 : p.NC;
  p.BN = d->Cout <= 32 ? 32 : 64;
  p.res = res; p.res_stride = res_stride; p.mx = mask_x; p.mx_stride = mask_stride; p.m_stats = mask_stats;
  p.y = y; p.y_stride = y_stride; p.partials = partials;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.pD = d->pD; p.pH = d->pH; p.pW = d->pW; p.act = d->act;
  const int td = r32_tile_depth();
  p.tiles_d = (d->Do + td - 1) / td; p.tiles_h = (d->Ho + 7) / 8; p.tiles_w = (d->Wo + 7) / 8;
  p.dbg = 0;   // tools/r32_ablate.py timing ablations; 0 in production
  p.P = cbim_conv3d_num_tiles(d);
  CBIM_CHECK(!partials || p.P >= (int)cbim_conv_r32_grid(d), CBIM_EINVAL, "conv r32: %d partial records < grid", p.P);
  {
    // 32-bit byte offsets inside one halo box / one output tile, built from 24-bit multiplies
    const int64_t box_rows = (int64_t)(td + 2) * d->Hi * d->Wi, xs = (x2 && x2_stride > x_stride ? x2_stride : x_stride) * 2;
    CBIM_CHECK(box_rows < (1 << 24) && xs < (1 << 24) && box_rows * xs < ((int64_t)1 << 32), CBIM_EUNSUPPORTED,
               "conv r32: input plane %dx%d with row stride %lld B exceeds the 32-bit halo addressing", d->Hi, d->Wi, (long long)xs);
    const int64_t tile_rows = (int64_t)td * d->Ho * d->Wo;
    int64_t so = y_stride * 2;
    if (res && res_stride * 2 > so) so = res_stride * 2;
    if (mask_x && mask_stride * 2 > so) so = mask_stride * 2;
    CBIM_CHECK(tile_rows < (1 << 24) && so < (1 << 24) && tile_rows * so < ((int64_t)1 << 32), CBIM_EUNSUPPORTED,
               "conv r32: output plane %dx%d with row stride %lld B exceeds the 32-bit epilogue addressing", d->Ho, d->Wo, (long long)so);
  }
  dim3 grid((unsigned)cbim_conv_r32_grid(d), (unsigned)((d->Cout + 31) / 32));
  hipStream_t st = (hipStream_t)stream;
  const bool relu = d->act == CBIM_ACT_RELU;
  // (eligibility: act is ReLU or none; a transformed input comes with the forward epilogue, a mask with an input
  //  that is used as it is — the four combinations the pre-activation blocks produce)
  if (in_stats) return relu ? r32_launch<CBIM_ACT_RELU, true, false>(p, grid, st) : r32_launch<CBIM_ACT_NONE, true, false>(p, grid, st);
  if (mask_x) return relu ? r32_launch<CBIM_ACT_RELU, false, true>(p, grid, st) : r32_launch<CBIM_ACT_NONE, false, true>(p, grid, st);
  return r32_launch<CBIM_ACT_NONE, false, false>(p, grid, st);
}

CBIM_DEFINE_WARM(r32)
