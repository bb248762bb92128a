// conv_wgrad_r32.hip — weight gradient of the 3x3x3 convolution, bf16, channel counts in multiples of 32:
// "accumulators in registers", the mirror image of conv_r32.hip (round 3).
//
//   dw[tap][co][ci] = sum_v dy[v][co] * a[v + tap - 1][ci]        a = the conv's input AS IT IS in memory: the caller
//                                                                  materialised act(IN(x)) once (functional.BasicBlockFn)
//
// Why a second kernel: k_conv_wgrad (conv_wgrad.hip) ran at 21 % of the MFMA peak and took 30 % of the ResUNet step.
// Its 4x8x8 tile is staged global -> registers -> (normalise) -> LDS between two workgroup barriers, a tile's
// contraction (112 MFMAs) is shorter than one HBM round trip, and every k-step re-reads both operands through the LDS
// transpose (16 ds_read_b64_tr_b16 per 7 MFMAs): three rewrites of the staging left the tile time where it was
// (DESIGN.md "measured and rejected").  The shape of the work is what changes here:
//   * persistent 512-thread workgroup per CU on 8x8x8 tiles; the 10x10x10 input halo is DOUBLE buffered in LDS and
//     arrives by LDS-DMA (global_load_lds_dwordx4: no registers, no vector ALU, one barrier per tile) — possible
//     because the input needs no transform any more;
//   * the 27 tap accumulators of a (16 co x 16 ci) quadrant live in registers for the whole strip of tiles
//     (27 x 4 VGPRs = the register image of conv_r32's weight fragments); a wave = (quadrant, half of the tile's rows);
//   * v_mfma_f32_16x16x32_bf16 with K = 32 VOXELS = 4 rows x 8 w of one plane.  The wave reads its 8 dy fragments
//     (one per plane) ONCE per tile; an input fragment of halo plane p at (kh, kw) then feeds the taps (kd = 0, 1, 2)
//     against dy planes p, p-1, p-2: 10 fragment reads per 24 MFMAs, the ratio conv_r32 runs at;
//   * both operands are channels-last rows in LDS and contract over voxels: every fragment is two
//     ds_read_b64_tr_b16 (a 16-lane group fetches [4 voxels][16 channels] and each lane keeps one channel's 4 voxels);
//   * the dy tile (32 KiB) is single buffered: it is dead once the 8 fragments are in registers, so the next tile's
//     dy streams in behind a barrier at the top of the tile.
// LDS: 2 x 63 KiB halo + 32 KiB dy = 158 KiB.  Per-strip fp32 slabs + the fixed-order k_wgrad_reduce as before.
//
// Replaces aten::convolution_backward(weight) for nn.Conv3d in ConvNormAct
// (/root/reference/model/dim3/conv_layers.py:29-38).
#include "cbim_common.h"
#include "conv_wgrad_r32.h"
#include <stdlib.h>
#include <stdio.h>

namespace cbim {

// WV = 8: 512 threads, two waves per SIMD (256 registers each); a wave = (co half, ci half, row group): 27 x 4 accumulators.
// WV = 4: 256 threads, ONE wave per SIMD with the whole 512-register file; a wave = (ci half, row group) and keeps the
//         accumulators of BOTH co halves (2 x 27 x 4): every input fragment read feeds 6 MFMAs instead of 3, which halves
//         the LDS traffic per MFMA (at WV = 8 the LDS pipe is ~90 % busy at the full MFMA rate).
static constexpr unsigned WR_HB = 63u * 1024u;          // one halo buffer: 1000 rows x 64 B in 63 LDS-DMA pieces of 1 KiB
static constexpr unsigned WR_DY = 2u * WR_HB;            // dy tile: 512 rows x 64 B
static constexpr unsigned WR_SMEM = WR_DY + 32u * 1024u;
static_assert(WR_SMEM <= 160 * 1024, "LDS");

__device__ __attribute__((aligned(64))) unsigned int g_wr32_zero[16];   // source of padding rows for the LDS-DMA

struct WR32Params {
  const void* x; int64_t x_stride;
  const void* x2; int64_t x2_stride; int ci_split;     // ci chunks >= ci_split come from x2
  const void* dy; int64_t dy_stride;
  const void* dy2; int64_t dy2_stride; int co_split;   // co chunks >= co_split come from dy2
  float* ws;
  int N, Di, Hi, Wi, Do, Ho, Wo;
  int tiles_d, tiles_h, tiles_w;
  int ci_blocks, Cout_pad, Cin_pad;   // 32-channel blocks of Cin; the gradient's own Cout, Cin (slab layout [Cout][Cin][27])
  int cin_bytes, cout_bytes;          // 2 Cin, 2 Cout: the 16-byte slots of a last block past them are zero-filled (round 6: 48 channels)
  int dbg;   // timing ablations of round 3 (tools/archive); always 0
  int diag;  // depthwise form: blockIdx.y = 32-channel group, only the diagonal (co == ci) of the 32 x 32 block is kept
};

// timing ablations of tools/wr32_ablate.py (wrong results): a COMPILE-TIME parameter of the kernel (as run-time tests they
// were ~300 scalar branches per tile that cut the MFMA loop into basic blocks: +12 % kernel time, measured); the ablated
// instantiations exist only in builds with `make EXTRA=-DCBIM_WR32_ABLATE`
#define WR_DBG DBG
// tuning knobs (tools/run_wr32_variants.sh builds and times the alternatives)
#ifndef WR32_RING
#define WR32_RING 5          // input-fragment reads in flight ahead of their MFMAs (+1)
#endif
#ifndef WR32_DMA_STEPS
#define WR32_DMA_STEPS 0     // 0: the next halo is requested two pieces per (kh, kw) step from step 0; n > 0: one piece per step from step n
#endif
#ifndef WR32_SETPRIO
#define WR32_SETPRIO 0       // 1: raise the wave's priority around each MFMA group
#endif
#ifdef CBIM_EMU
#define WR_SCHED_FENCE() ((void)0)
#define WR_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define WR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WR_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// one 1 KiB LDS-DMA piece: lane l copies 16 bytes from its own global address to lds_wave_base + 16 l.  M0 (the LDS
// base of the instruction) is saved and restored inside the statement, so nothing is hidden from the compiler.
__device__ __forceinline__ void wr_dma16(const unsigned char* gsrc, unsigned char* smem, unsigned lds_base, unsigned off) {
#ifdef CBIM_EMU
  (void)lds_base;
  emu_global_load_lds16(gsrc, smem + off);
#else
  (void)smem;
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_base + off);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(a), "v"(gsrc) : "memory");
#endif
}
// the same piece through a buffer descriptor (conv_rw.hip w_dma16): lane l copies the 16 bytes at base + soff + voff, ZEROS when
// soff + voff + 16 > nrec — zero padding is the range check (a lane outside the tensor carries voff = 0x80000000, a plane
// outside the tensor nrec = 0), the tile / plane origin travels in the scalar offset
typedef __attribute__((ext_vector_type(4))) int wr_i32x4;
__device__ __forceinline__ void wr_bdma16(unsigned voff, unsigned long long base, unsigned nrec, unsigned soff, unsigned char* smem,
                                          unsigned lds_base, unsigned off) {
#ifdef CBIM_EMU
  (void)lds_base;
  emu_buffer_load_lds16((const unsigned char*)base, nrec, voff, soff, smem + off);
#else
  (void)smem;
  wr_i32x4 rs = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), (int)nrec, 0x00020000};
  rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
  rs.z = __builtin_amdgcn_readfirstlane(rs.z);
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_base + off), so = __builtin_amdgcn_readfirstlane(soff);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(a), "v"(voff), "s"(rs), "s"(so) : "memory");
#endif
}
__device__ __forceinline__ void wr_wait_vm0() {
#ifndef CBIM_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void wr_wait_lgkm0() {
#ifndef CBIM_EMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ unsigned wr_mul24(unsigned a, unsigned b) {
#ifdef CBIM_EMU
  return a * b;
#else
  return __umul24(a, b);
#endif
}
// 4 consecutive-voxel bf16 of one channel via the LDS transpose read (per-lane address of 4 bf16)
__device__ __forceinline__ u32x2 wr_tr16_b64(const unsigned char* p) {
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
// XOR key of the 16-byte slot inside an LDS row, by the row's h coordinate: the four 16-lane groups of a transposed
// read sit on four consecutive h rows (640 / 512 bytes apart = 128 / 0 modulo the 256-byte bank window); alternating
// the 32-byte halves spreads them over all banks (same key as the conv_r32 halo)
__device__ __forceinline__ unsigned wr_swz(unsigned h) { return (h & 1u) << 1; }

typedef __attribute__((ext_vector_type(4))) float wr_f32x4;

// cycle profile of the phases of a tile (make EXTRA=-DCBIM_WR32_PROF): s_memtime stamps of every wave of workgroup (0, 0),
// summed over its tiles, printed per launch by the launcher
#ifdef CBIM_WR32_PROF
__device__ unsigned long long g_wr32_prof[8][8];
#define WR_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_t[i] += now_ - prof_last; prof_last = now_; } while (0)
#else
#define WR_STAMP(i) ((void)0)
#endif

template <int WV, int DBG = 0>
__global__ void __launch_bounds__(WV * 64, 1) k_wgrad_r32(WR32Params p) {
  constexpr int WR_NT = WV * 64;
  constexpr int NCH = WV == 8 ? 1 : 2;                                // co halves per wave
  WR_DYN_SMEM(smem);
#ifdef CBIM_EMU
  const unsigned lds_base = 0;
#else
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#endif
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lv = lane & 15, lq = lane >> 4;
  // co half (WV = 8 only), ci half, row group of the 8x8 plane
  const int ch0 = WV == 8 ? (wave & 1) : 0, cih = WV == 8 ? ((wave >> 1) & 1) : (wave & 1), vg = WV == 8 ? (wave >> 2) : (wave >> 1);
  const int cb = p.diag ? (int)blockIdx.y : (int)blockIdx.y / p.ci_blocks, ib = p.diag ? (int)blockIdx.y : (int)blockIdx.y % p.ci_blocks;
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int n_tiles = p.N * tiles_per_n;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)(((long long)lb * n_tiles) / gridDim.x);
  const int t_end = (int)(((long long)(lb + 1) * n_tiles) / gridDim.x);

  // ---- this workgroup's 32-channel chunks of the input and of dy (wave-uniform) ----------------------------------
  const bool x_second = p.x2 != nullptr && ib >= p.ci_split;
  const bool dy_second = p.dy2 != nullptr && cb >= p.co_split;
  const unsigned x_sb = (unsigned)(x_second ? p.x2_stride : p.x_stride) * 2u;
  const unsigned dy_sb = (unsigned)(dy_second ? p.dy2_stride : p.dy_stride) * 2u;
  const unsigned char* const x_chunk = x_second ? (const unsigned char*)p.x2 + (ib - p.ci_split) * 64
                                                : (const unsigned char*)p.x + ib * 64;
  const unsigned char* const dy_chunk = dy_second ? (const unsigned char*)p.dy2 + (cb - p.co_split) * 64
                                                  : (const unsigned char*)p.dy + cb * 64;

  // ---- LDS-DMA pieces, PLANE-major (round 4, as conv_rw.hip): piece j of a halo plane = its rows 16 j .. 16 j + 15 (of 100; the
  //      seventh re-covers rows 84..99), piece j of a dy plane = its rows 16 j .. 16 j + 15 (of 64).  A wave fetches the SAME
  //      piece index of every plane it serves, so a lane's (h, w) position, swizzled slot and source offset are constants:
  //      one register per piece set instead of a (position, offset) pair per piece, no per-piece range tests — a lane outside
  //      the tensor in h / w carries an out-of-range vector offset, a plane outside it a descriptor with num_records = 0, and
  //      the tile origin travels in the scalar offset of the buffer load.  (k_wgrad_r32 spent ~280 cycles per dy piece and
  //      ~100 per halo piece on address arithmetic: profiles/r04_p_wr32_phase_cycles.txt)
  constexpr int HS = WV == 8 ? 1 : 2;                 // halo piece sets per wave: j = wave (+ 4)
  constexpr int DP = 32 / WV;                         // dy pieces per wave and tile
  constexpr unsigned WR_OOB = 0x80000000u;
#ifdef CBIM_EMU
  const int wv_u = wave;
#else
  const int wv_u = __builtin_amdgcn_readfirstlane(wave);               // (wave-uniform by construction: keeps the piece state scalar)
#endif
  unsigned h_off[HS], h_hh[HS], h_hw[HS], h_lds[HS];
  bool h_on[HS];
#pragma unroll
  for (int q = 0; q < HS; ++q) {
    const int j = wv_u + 4 * q * (WV == 4 ? 1 : 0);
    h_on[q] = j < 7;
    const int r0 = j < 6 ? 16 * j : 84;
    const unsigned row = (unsigned)r0 + ((unsigned)lane >> 2);
    h_hh[q] = (row * 205u) >> 11;                     // row / 10 (row < 100)
    h_hw[q] = row - h_hh[q] * 10u;
    const unsigned slot_src = (((unsigned)lane & 3u) ^ wr_swz(h_hh[q])) << 4;
    h_off[q] = wr_mul24(wr_mul24(h_hh[q], (unsigned)p.Wi) + h_hw[q], x_sb) + slot_src;
    // (a last block of fewer than 32 channels — Cin = 48: the slots past the tensor's channels hold the next voxel's; zeros land in
    //  LDS instead and the gradient rows / columns they feed are not written)
    if ((unsigned)(ib * 64) + slot_src >= (unsigned)p.cin_bytes) h_off[q] = WR_OOB;
    h_lds[q] = (unsigned)r0 * 64u;
  }
  // dy: WV = 8: piece j = wave & 3 of the planes (wave >> 2) + 2 k; WV = 4: piece j = wave of every plane
  const int dj = WV == 8 ? (wv_u & 3) : wv_u, dp0 = WV == 8 ? (wv_u >> 2) : 0, dstep = WV == 8 ? 2 : 1;
  const unsigned d_row = 16u * (unsigned)dj + ((unsigned)lane >> 2);        // row (h, w) of the 8x8 plane
  const unsigned d_h = d_row >> 3, d_w = d_row & 7u;
  const unsigned d_slot_src = (((unsigned)lane & 3u) ^ wr_swz(d_h)) << 4;
  const unsigned d_off = (unsigned)(cb * 64) + d_slot_src < (unsigned)p.cout_bytes
                             ? wr_mul24(wr_mul24(d_h, (unsigned)p.Wo) + d_w, dy_sb) + d_slot_src : WR_OOB;

  struct TilePos { int n, td, th, tw; };
  auto advance = [&](TilePos& u) {
    if (++u.tw == p.tiles_w) { u.tw = 0; if (++u.th == p.tiles_h) { u.th = 0; if (++u.td == p.tiles_d) { u.td = 0; ++u.n; } } }
  };
  TilePos cur, nxt;
  {
    const int tt = t_begin % tiles_per_n;
    cur.n = t_begin / tiles_per_n; cur.td = tt / (p.tiles_w * p.tiles_h); cur.th = (tt / p.tiles_w) % p.tiles_h; cur.tw = tt % p.tiles_w;
    nxt = cur;
    advance(nxt);
  }

  // per-tile source state: descriptor bases (the halo's is shifted down by one row + one voxel so that the scalar offset of a
  // plane inside the tensor is never negative), scalar offsets of plane 0, the lanes' vector offsets
  struct Src { unsigned long long hbase, dbase; unsigned hsoff0, dsoff0, hplane, dplane, hv[HS], dv; int id0, od0; };
  auto src_of = [&](const TilePos& tp) -> Src {
    Src r;
    const int ih0 = tp.th * 8 - 1, iw0 = tp.tw * 8 - 1, oh0 = tp.th * 8, ow0 = tp.tw * 8;
    r.id0 = tp.td * 8 - 1; r.od0 = tp.td * 8;
    r.hplane = (unsigned)(p.Hi * p.Wi) * x_sb;
    r.dplane = (unsigned)(p.Ho * p.Wo) * dy_sb;
    r.hbase = (unsigned long long)x_chunk + (unsigned long long)tp.n * p.Di * r.hplane - (unsigned long long)(p.Wi + 1) * x_sb;
    r.dbase = (unsigned long long)dy_chunk + (unsigned long long)tp.n * p.Do * r.dplane;
    r.hsoff0 = (unsigned)((ih0 + 1) * p.Wi + iw0 + 1) * x_sb;               // + (id0 + plane) * hplane
    r.dsoff0 = (unsigned)(oh0 * p.Wo + ow0) * dy_sb;                        // + (od0 + plane) * dplane
#pragma unroll
    for (int q = 0; q < HS; ++q) {
      const bool ok = (unsigned)(ih0 + (int)h_hh[q]) < (unsigned)p.Hi && (unsigned)(iw0 + (int)h_hw[q]) < (unsigned)p.Wi;
      r.hv[q] = ok ? h_off[q] : WR_OOB;                                     // (h_off itself may be WR_OOB: channel tail)
    }
    r.dv = (oh0 + (int)d_h < p.Ho && ow0 + (int)d_w < p.Wo) ? d_off : WR_OOB;
    return r;
  };
  // this wave's piece (set q) of halo plane `pl` into the halo buffer at LDS offset `buf`
  auto dma_halo = [&](const Src& r, int q, int pl, unsigned buf) {
    const int d = r.id0 + pl;
    const bool in = (unsigned)d < (unsigned)p.Di;
    wr_bdma16(r.hv[q], r.hbase, in ? 0x80000000u : 0u, in ? r.hsoff0 + (unsigned)d * r.hplane : 0u, smem, lds_base,
              buf + (unsigned)pl * 6400u + h_lds[q]);
  };
  // this wave's k-th dy piece of the tile
  auto dma_dy = [&](const Src& r, int k) {
    const int pl = dp0 + dstep * k, d = r.od0 + pl;
    const bool in = d < p.Do;
    wr_bdma16(r.dv, r.dbase, in ? 0x80000000u : 0u, in ? r.dsoff0 + (unsigned)d * r.dplane : 0u, smem, lds_base,
              WR_DY + (unsigned)(pl * 64 + 16 * dj) * 64u);
  };

  // ---- fragment addresses.  A transposed read: lane s of a 16-lane group supplies the address of voxel (s >> 2) [+4],
  //      channels 4 (s & 3) .. +3 of the group's 16; it receives channel s, 4 voxels.  Group lq = row 4 vg + lq of the
  //      plane, the two reads of a fragment are w 0..3 and w 4..7: lane (lv, lq) ends up with its channel's 8 voxels
  //      (plane, 4 vg + lq, w = 0..7) = k-group lq of the 32-voxel k-step.
  const unsigned sub = ((unsigned)lv & 1u) * 8u;                       // 8-byte half of the 16-byte slot
  const unsigned a_h = (unsigned)(4 * vg + lq);
  unsigned a_base[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    a_base[c] = WR_DY + (a_h * 8u + ((unsigned)lv >> 2)) * 64u + ((((unsigned)(2 * (ch0 + c)) + (((unsigned)lv & 3u) >> 1)) ^ wr_swz(a_h)) << 4) + sub;
  unsigned b_base[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const unsigned hh = a_h + (unsigned)kh;
    b_base[kh] = (hh * 10u + ((unsigned)lv >> 2)) * 64u + ((((unsigned)(2 * cih) + (((unsigned)lv & 3u) >> 1)) ^ wr_swz(hh)) << 4) + sub;
  }

  wr_f32x4 acc[NCH][27];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) acc[c][tp] = wr_f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_my = t_end - t_begin;
  if (n_my > 0) {
    // ---- prologue: first tile's dy and halo -------------------------------------------------------------------------
    const Src r0s = src_of(cur);
#pragma unroll
    for (int k = 0; k < DP; ++k) dma_dy(r0s, k);
#pragma unroll
    for (int q = 0; q < HS; ++q)
      if (h_on[q]) {
#pragma unroll
        for (int pl = 0; pl < 10; ++pl) dma_halo(r0s, q, pl, 0);
      }
    wr_wait_vm0();
    __syncthreads();
  }
#ifdef CBIM_WR32_PROF
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = __builtin_readcyclecounter();
#endif
  for (int t = 0; t < n_my; ++t) {
    const unsigned buf = (unsigned)(t & 1) * WR_HB, obuf = WR_HB - buf;
    const bool more = t + 1 < n_my;                                    // workgroup-uniform
    WR_STAMP(0);
    // (A) the wave's 8 dy fragments (one per plane); the dy region is dead afterwards
    u32x4 af[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u32x2 a0 = wr_tr16_b64(smem + a_base[c] + (unsigned)i * 4096u);
        const u32x2 a1 = wr_tr16_b64(smem + a_base[c] + (unsigned)i * 4096u + 256u);
        af[c][i] = u32x4{a0.x, a0.y, a1.x, a1.y};
      }
    if (more) {
      wr_wait_lgkm0();                                                 // the reads have returned their data
      WR_STAMP(1);                                                     // dy fragment reads
      __syncthreads();
      WR_STAMP(2);                                                     // barrier (dy region free)
      WR_STAMP(3);
    }
    const Src rn = src_of(more ? nxt : cur);                           // (wave-uniform scalars + two selects: cheap)
    // (B) 9 (kh, kw) steps: the 10 halo-plane fragments stream through a ring of 5 registers; plane p feeds the taps
    //     (kd 0, dy plane p), (kd 1, p-1), (kd 2, p-2).  The next tile's halo is fetched during the first four steps.
    if (!(WR_DBG & 2)) {
      constexpr int RING = WR32_RING, PLN = 10, SEQ = 9 * PLN;
      u32x4 xr[RING];
      auto frag = [&](int e) -> u32x4 {                                // e = (kh*3 + kw) * PLN + plane
        const int pl = e % PLN, s = e / PLN;
        const unsigned a = buf + b_base[s / 3] + (unsigned)((s % 3) * 64) + (unsigned)(pl * 6400);
        const u32x2 b0 = wr_tr16_b64(smem + a);
        const u32x2 b1 = wr_tr16_b64(smem + a + 256u);
        return u32x4{b0.x, b0.y, b1.x, b1.y};
      };
#pragma unroll
      for (int e = 0; e < RING - 1; ++e) xr[e] = frag(e);
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int kh = s / 3, kw = s % 3;
        // the next tile's pieces: halo planes 2 s, 2 s + 1 in steps 0..4, the dy pieces (the dy region is free: every wave
        // passed the barrier above with its fragments in registers) one per step in steps 1..4
        if (more && !(WR_DBG & 1)) {
          if (s < 5) {
#pragma unroll
            for (int q = 0; q < HS; ++q)
              if (h_on[q]) { dma_halo(rn, q, 2 * s, obuf); dma_halo(rn, q, 2 * s + 1, obuf); }
          }
          if (s >= 1 && s < 5) {                                       // (issued late — steps 5..8 — the last pieces were still in
            constexpr int per = (DP + 3) / 4;                          //  flight at the tile barrier: 570 cycles of vmcnt wait per tile)
#pragma unroll
            for (int k = (s - 1) * per; k < s * per && k < DP; ++k) dma_dy(rn, k);
          }
        }
#pragma unroll
        for (int pl = 0; pl < PLN; ++pl) {
          const int e = s * PLN + pl;
          if (e + RING - 1 < SEQ && !(WR_DBG & 4)) xr[(e + RING - 1) % RING] = frag(e + RING - 1);
          WR_SCHED_FENCE();
#if WR32_SETPRIO && !defined(CBIM_EMU)
          __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
            const int i = pl - kd;
#ifndef CBIM_EMU
            if (WR_DBG & 8) asm volatile("" ::"v"(xr[e % RING]));   // (ablation: the fragment reads stay alive without the MFMAs)
#endif
            if (i >= 0 && i < 8 && !(WR_DBG & 8)) {
              const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
              for (int c = 0; c < NCH; ++c)
                acc[c][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c][i]), __builtin_bit_cast(bf16x8, xr[e % RING]),
                                                                      acc[c][tap], 0, 0, 0);
            }
          }
#if WR32_SETPRIO && !defined(CBIM_EMU)
          __builtin_amdgcn_s_setprio(0);
#endif
          WR_SCHED_FENCE();
        }
      }
    }
    // (C) ONE barrier per tile (two with the dy hand-over): every wave is done with `buf`, the other buffer and the dy
    //     tile are complete (own LDS-DMA pieces waited for)
    WR_STAMP(4);                                                       // MFMA loop (+ halo pieces)
    wr_wait_vm0();
    WR_STAMP(5);                                                       // own pieces landed
    __syncthreads();
    WR_STAMP(6);                                                       // barrier
    cur = nxt;
    advance(nxt);
  }
#ifdef CBIM_WR32_PROF
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) g_wr32_prof[wave][i] = prof_t[i];
    g_wr32_prof[wave][7] = (unsigned long long)n_my;
  }
#endif

  // ---- the two row groups of a quadrant hold partial sums of the same (tap, co, ci): they meet in an LDS tile laid out
  //      like the gradient itself, T[co][ci][tap] (vg 1 stores, vg 0 adds: fixed order), and the tile leaves as 32 rows
  //      of 864 contiguous floats — into this strip's slab ws[strip][Cout][Cin][27], or straight into dw when the launch
  //      has a single strip (no reduce kernel then) ----------------------------------------------------------------------
  float* T = (float*)smem;                                              // 32 x 32 x 27 floats = 108 KiB (the halos are dead)
  const int t_ci = 16 * cih + lv;
  if (vg == 1) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* row = T + ((16 * (ch0 + c) + 4 * lq + r) * 32 + t_ci) * 27;
#pragma unroll
        for (int tp = 0; tp < 27; ++tp) row[tp] = acc[c][tp][r];
      }
  }
  __syncthreads();
  if (vg == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* row = T + ((16 * (ch0 + c) + 4 * lq + r) * 32 + t_ci) * 27;
#pragma unroll
        for (int tp = 0; tp < 27; ++tp) row[tp] += acc[c][tp][r];
      }
  }
  __syncthreads();
  if (p.diag) {
    // depthwise: T[c][c][tap] of this 32-channel group -> slab [strip][C][27]
    float* dst = p.ws + ((size_t)lb * p.Cout_pad + (size_t)cb * 32) * 27;
    for (int i = tid; i < 32 * 27; i += WR_NT) {
      const int c = i / 27, tp = i % 27;
      dst[i] = T[(c * 32 + c) * 27 + tp];
    }
  } else {
    const size_t total = (size_t)27 * p.Cout_pad * p.Cin_pad;
    float* dst = p.ws + (size_t)lb * total;              // (a single strip: p.ws is dw itself)
    constexpr int ROW4 = 32 * 27 / 4;                    // float4 per co row of the tile
    const int ci_n = p.Cin_pad - ib * 32 < 32 ? p.Cin_pad - ib * 32 : 32;   // channels of this block inside the gradient (a multiple of 4)
    for (int i = tid; i < 32 * ROW4; i += WR_NT) {
      const int co = i / ROW4, k = i % ROW4;
      if (cb * 32 + co < p.Cout_pad && 4 * k < ci_n * 27)
        *(f32x4*)(dst + ((size_t)(cb * 32 + co) * p.Cin_pad + ib * 32) * 27 + 4 * k) = *(const f32x4*)(T + co * 864 + 4 * k);
    }
  }
}

// Fixed-order sum of the strips' slabs (each already in the gradient's own layout [Cout][Cin][27]) into dw.  512 threads =
// LW lanes x SP slab phases: lane l owns 4 consecutive values (one 16-byte load per slab), phase ph adds slabs ph, ph + SP,
// ... in two alternating chains; the SP phase sums meet in LDS and are added in phase order.
template <int SP>
__global__ void __launch_bounds__(512) k_wgrad_r32_reduce(const float* __restrict__ ws, float* __restrict__ dw, int n_slabs,
                                                          int64_t total) {
  constexpr int LW = 512 / SP;
  __shared__ f32x4 red[512];
  const int l = threadIdx.x % LW, ph = threadIdx.x / LW;
  const int64_t i = ((int64_t)blockIdx.x * LW + l) * 4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    const float* src = ws + i;
    int sl = ph;
    for (; sl + SP < n_slabs; sl += 2 * SP) {
      const f32x4 v0 = *(const f32x4*)(src + (size_t)sl * total);
      const f32x4 v1 = *(const f32x4*)(src + (size_t)(sl + SP) * total);
      a0 += v0;
      a1 += v1;
    }
    if (sl < n_slabs) a0 += *(const f32x4*)(src + (size_t)sl * total);
  }
  red[threadIdx.x] = a0 + a1;
  __syncthreads();
  if (ph == 0 && i < total) {
    f32x4 r = red[l];
#pragma unroll
    for (int q = 1; q < SP; ++q) r += red[q * LW + l];
    *(f32x4*)(dw + i) = r;
  }
}

}  // namespace cbim

using namespace cbim;

// cbim_wgrad_r32_enable(0) keeps every weight gradient on k_conv_wgrad (tests: two-sided checks)
static int g_wr32_on = 1;
static int wr32_on() { return g_wr32_on; }
// CBIM_WGRAD_R32_C16=0 keeps the layers whose channel counts are not multiples of 32 on k_conv_wgrad (A/B runs)
static int g_wr32_c16 = getenv("CBIM_WGRAD_R32_C16") ? atoi(getenv("CBIM_WGRAD_R32_C16")) : 1;
extern "C" int cbim_wgrad_r32_enable(int on) {
  const int old = g_wr32_on;
  if (on >= 0) g_wr32_on = on;
  return old;
}

// cbim_wgrad_r32_waves(4 | 8): waves per workgroup (see k_wgrad_r32); process-wide knob for tests
static int g_wr32_waves = 8;
static int wr32_waves() { return g_wr32_waves; }
extern "C" int cbim_wgrad_r32_waves(int w) {
  const int old = g_wr32_waves;
  if (w == 4 || w == 8) g_wr32_waves = w;
  return old;
}

bool cbim_wgrad_r32_eligible(const cbim_conv_desc* d, const float* in_stats, const void* x2, int cin_split, const void* dy2,
                             int cout_split) {
  if (!wr32_on() || d->dtype != CBIM_BF16 || in_stats) return false;
  if (d->kD != 3 || d->kH != 3 || d->kW != 3 || d->pD != 1 || d->pH != 1 || d->pW != 1) return false;
  // round 6: channel counts in multiples of 16 run with a zero-filled last 32-channel block (SwinUNETR's 48 / 96-channel layers:
  // 48 x 48 = four (32 x 32) pairs, nine sixteenths of them inside the gradient) — still ahead of k_conv_wgrad, which pads the same way
  if (d->Cin % 32 != 0 || d->Cout % 32 != 0) {
    // (Cin in multiples of 8 down to 8: the 8-channel padded network input of SwinUNETR's encoder1 — one quarter-filled block)
    if (!g_wr32_c16 || d->Cin % 8 != 0 || d->Cout % 16 != 0 || d->Cout < 32) return false;
  }
  if (x2 && (cin_split <= 0 || cin_split >= d->Cin || cin_split % 32 != 0)) return false;
  if (dy2 && (cout_split <= 0 || cout_split >= d->Cout || cout_split % 32 != 0)) return false;
  if (d->Do != d->Di || d->Ho != d->Hi || d->Wo != d->Wi) return false;
  // below 8 in a dimension most of an 8x8x8 tile is padding: k_conv_wgrad's 4x4x8 / 2x8x8 tiles fit better
  return d->Do >= 8 && d->Ho >= 8 && d->Wo >= 8;
}

static void wr32_tiles(const cbim_conv_desc* d, int& td, int& th, int& tw) {
  td = (d->Do + 7) / 8; th = (d->Ho + 7) / 8; tw = (d->Wo + 7) / 8;
}

// strips per (co chunk, ci chunk) pair: whole rounds of 256 persistent workgroups, a tile and a half of fixed cost per
// workgroup (exposed first load, slab write), slabs capped at 96 MiB
static int wr32_strips_for(int64_t n_tiles, int64_t pairs, int64_t slab);
int cbim_wgrad_r32_strips(const cbim_conv_desc* d) {
  int td, th, tw;
  wr32_tiles(d, td, th, tw);
  return wr32_strips_for((int64_t)d->N * td * th * tw, (int64_t)((d->Cout + 31) / 32) * ((d->Cin + 31) / 32), (int64_t)27 * d->Cout * d->Cin * 4);
}
// depthwise form: one (group, group) pair per 32 channels, slabs [C][27]
int cbim_wgrad_r32_dw_strips(const cbim_conv_desc* d) {
  int td, th, tw;
  wr32_tiles(d, td, th, tw);
  return wr32_strips_for((int64_t)d->N * td * th * tw, (int64_t)(d->Cout / 32), (int64_t)27 * d->Cout * 4);
}
static int g_wr32_small_strips = 1;
static int wr32_strips_for(int64_t n_tiles, int64_t pairs, int64_t slab) {
  int64_t gmax = (96ll << 20) / slab;
  if (gmax < 1) gmax = 1;
  if (gmax > n_tiles) gmax = n_tiles;
  if (gmax > 256) gmax = 256;
  int best = 1;
  double best_cost = 1e30;
  for (int64_t g = 1; g <= gmax; ++g) {
    // several pairs: strips in multiples of 8 put the pairs of one strip (same dy tile / same input halo) on the same
    // XCD (workgroup k runs on XCD k % 8), where the second reader hits the L2
    // (small layers — at most 64 tiles, everything L2 / MALL resident — may break that rule when it saves a round of
    //  workgroups: 256 -> 256 @16^3 = 8 tiles x 64 pairs ran as 512 one-tile workgroups in two rounds)
    const bool off8 = pairs > 1 && gmax >= 8 && g % 8 != 0;
    if (off8 && (n_tiles > 64 || !g_wr32_small_strips)) continue;
    const double rounds = (double)((pairs * g + 255) / 256);
    const double cost = rounds * ((double)((n_tiles + g - 1) / g) + 1.5) + (off8 ? 0.25 : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)g; }
  }
  return best;
}

size_t cbim_wgrad_r32_workspace(const cbim_conv_desc* d) {
  return (size_t)cbim_wgrad_r32_strips(d) * 27 * d->Cout * d->Cin * sizeof(float);
}

static int wr32_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                       int cin_split, const void* dy, int64_t dy_stride, const void* dy2, int64_t dy2_stride,
                       int cout_split, float* workspace, float* dw, void* stream, int diag);
int cbim_wgrad_r32_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                          int cin_split, const void* dy, int64_t dy_stride, const void* dy2, int64_t dy2_stride,
                          int cout_split, float* workspace, float* dw, void* stream) {
  return wr32_launch(d, x, x_stride, x2, x2_stride, cin_split, dy, dy_stride, dy2, dy2_stride, cout_split, workspace, dw, stream, 0);
}
// depthwise weight gradient (d->Cin == d->Cout == C, groups = C): slabs [strips][C][27] into `workspace`
int cbim_wgrad_r32_dw_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* dy, int64_t dy_stride,
                             float* workspace, void* stream) {
  return wr32_launch(d, x, x_stride, nullptr, 0, 0, dy, dy_stride, nullptr, 0, 0, workspace, workspace, stream, 1);
}
bool cbim_wgrad_r32_dw_eligible(const cbim_conv_desc* d) {
  return cbim_wgrad_r32_eligible(d, nullptr, nullptr, 0, nullptr, 0) && d->Cin == d->Cout;
}
static int wr32_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                       int cin_split, const void* dy, int64_t dy_stride, const void* dy2, int64_t dy2_stride,
                       int cout_split, float* workspace, float* dw, void* stream, int diag) {
  WR32Params p;
  p.diag = diag;
  const int strips = diag ? cbim_wgrad_r32_dw_strips(d) : cbim_wgrad_r32_strips(d);
  p.x = x; p.x_stride = x_stride; p.x2 = x2; p.x2_stride = x2 ? x2_stride : x_stride; p.ci_split = x2 ? cin_split / 32 : d->Cin / 32;
  p.dy = dy; p.dy_stride = dy_stride; p.dy2 = dy2; p.dy2_stride = dy2 ? dy2_stride : dy_stride;
  p.co_split = dy2 ? cout_split / 32 : d->Cout / 32;
  p.ws = strips == 1 ? dw : workspace;     // a single strip writes the gradient itself
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo;
  wr32_tiles(d, p.tiles_d, p.tiles_h, p.tiles_w);
  p.ci_blocks = (d->Cin + 31) / 32; p.Cout_pad = d->Cout; p.Cin_pad = d->Cin;
  p.cin_bytes = d->Cin * 2; p.cout_bytes = d->Cout * 2;
  p.dbg = 0;
  {
    // 32-bit byte offsets inside one halo box / one dy tile, built from 24-bit multiplies
    const int64_t box_rows = (int64_t)10 * d->Hi * d->Wi, xs = (x2 && x2_stride > x_stride ? x2_stride : x_stride) * 2;
    const int64_t ds = (dy2 && dy2_stride > dy_stride ? dy2_stride : dy_stride) * 2;
    CBIM_CHECK(box_rows < (1 << 24) && xs < (1 << 24) && ds < (1 << 24) && box_rows * xs < ((int64_t)1 << 32) &&
               box_rows * ds < ((int64_t)1 << 32), CBIM_EUNSUPPORTED,
               "wgrad r32: plane %dx%d with row strides %lld/%lld B exceeds the 32-bit tile addressing", d->Hi, d->Wi, (long long)xs, (long long)ds);
    // buffer-addressed pieces: scalar + vector offset inside one image stay below 2^31 (num_records of the descriptors)
    const int64_t img = ((int64_t)d->Di + 2) * d->Hi * d->Wi * (xs > ds ? xs : ds);
    CBIM_CHECK(img < ((int64_t)1 << 31) - 65536, CBIM_EUNSUPPORTED, "wgrad r32: one image of %lld B exceeds the 2 GiB buffer addressing", (long long)img);
  }
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wgrad_r32<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_wgrad_r32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  dim3 grid((unsigned)strips, (unsigned)(diag ? d->Cout / 32 : ((d->Cout + 31) / 32) * ((d->Cin + 31) / 32)));
#ifdef CBIM_WR32_ABLATE
#define WR_ABL(W, D)                                                                                                   \
  if (wr32_waves() == W && p.dbg == D) {                                                                               \
    (void)hipFuncSetAttribute((const void*)k_wgrad_r32<W, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    CBIM_LAUNCH((k_wgrad_r32<W, D>), grid, dim3(W * 64), (size_t)WR_SMEM, (hipStream_t)stream, p);                     \
  } else
  WR_ABL(8, 1) WR_ABL(8, 2) WR_ABL(8, 4) WR_ABL(8, 8) WR_ABL(8, 5) WR_ABL(8, 9) WR_ABL(8, 12) WR_ABL(8, 13) WR_ABL(8, 3)
  WR_ABL(4, 1) WR_ABL(4, 2) WR_ABL(4, 4) WR_ABL(4, 8) WR_ABL(4, 5) WR_ABL(4, 9) WR_ABL(4, 12) WR_ABL(4, 13) WR_ABL(4, 3)
#undef WR_ABL
#endif
  if (wr32_waves() == 4) CBIM_LAUNCH(k_wgrad_r32<4>, grid, dim3(256), (size_t)WR_SMEM, (hipStream_t)stream, p);
  else CBIM_LAUNCH(k_wgrad_r32<8>, grid, dim3(512), (size_t)WR_SMEM, (hipStream_t)stream, p);
#ifdef CBIM_WR32_PROF
  {
    (void)hipStreamSynchronize((hipStream_t)stream);
    unsigned long long h[8][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wr32_prof), sizeof(h));
    static const char* nm[7] = {"top", "dy_frag_reads", "barrier1", "dy_dma_issue", "mfma+halo_dma", "wait_vm", "barrier2"};
    fprintf(stderr, "[wr32 prof] Cin %d Cout %d @%d strips %d tiles %llu | cycles per tile:", d->Cin, d->Cout, d->Do, strips, h[0][7]);
    for (int i = 0; i < 7; ++i) {
      unsigned long long sum = 0, mx = 0;
      for (int w = 0; w < 8; ++w) { sum += h[w][i]; if (h[w][i] > mx) mx = h[w][i]; }
      fprintf(stderr, " %s %llu (max %llu)", nm[i], sum / 8 / (h[0][7] ? h[0][7] : 1), mx / (h[0][7] ? h[0][7] : 1));
    }
    fprintf(stderr, "\n");
  }
#endif
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "wgrad r32 launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

int cbim_wgrad_r32_reduce(const cbim_conv_desc* d, const float* workspace, float* dw, void* stream) {
  const int G = cbim_wgrad_r32_strips(d);
  if (G == 1) return CBIM_OK;                                // the single strip wrote dw itself (cbim_wgrad_r32_launch)
  const int64_t total = (int64_t)27 * d->Cout * d->Cin;      // a multiple of 4 (Cin is a multiple of 32)
  hipStream_t st = (hipStream_t)stream;
#define WR_RED(SPV)                                                                                              \
  do {                                                                                                           \
    const int64_t per = (int64_t)(512 / SPV) * 4;                                                                \
    CBIM_LAUNCH((k_wgrad_r32_reduce<SPV>), dim3((unsigned)((total + per - 1) / per)), dim3(512), 0, st, workspace, dw, G, \
                total);                                                                                          \
  } while (0)
  if (G >= 32) WR_RED(32);
  else if (G >= 16) WR_RED(16);
  else if (G >= 8) WR_RED(8);
  else if (G >= 4) WR_RED(4);
  else WR_RED(2);
#undef WR_RED
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "wgrad r32 reduce launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(wgrad_r32)
