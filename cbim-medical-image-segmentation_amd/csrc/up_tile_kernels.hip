// up_tile_kernels.hip — the decoder level's up-sampling path on LDS tiles (round 3).
//
// up_block (/root/reference/model/dim3/unet_utils.py:69-71): F.interpolate(low, size=skip size, 'trilinear',
// align_corners=True) -> cat([skip, up]) -> first pre-activation block.  The engine never stores the concatenation: one
// pass writes a = act(IN([skip | up(low)])), the backward re-forms the concatenation to apply the InstanceNorm backward
// and splits it into dskip and the fine-resolution gradient of the up-sampled part, which the transposed interpolation
// gathers into dlow (functional.UpBlockFirstFn).
//
// The first versions of these kernels (pool_up_kernels.hip: k_upcat_fwd_stats, k_upcat_act_fwd, k_upcat_norm_bwd,
// k_upcat_bwd_low) form every output chunk from 8 gathered 16-byte global loads plus ~100 vector instructions of index
// and weight arithmetic: 1.5-1.7 TB/s at the 128^3 level (profiles/r03_d_resunet_kernels.txt: 1.8 ms of a 13.5 ms
// step).  Here a workgroup owns a 4x8x8 tile of the FINE grid:
//   * the <= 3x5x5 box of coarse voxels the tile interpolates from is staged ONCE into LDS (whole channel rows,
//     coalesced), together with three small per-axis tables (relative source indices and weights of the tile's 4+8+8
//     fine coordinates): an output chunk is 8 ds_read_b128 + 3 table reads, no index arithmetic, no global gathers;
//   * thread = (fixed 16-byte channel chunk, voxel lane), so per-channel statistics / sums stay in registers and a wave
//     writes whole contiguous rows;
//   * the transposed gather (dup -> dlow) stays on k_upcat_bwd_low: staging the fine box of a coarse tile re-reads every
//     fine voxel ((t+2)/t)^3 = 3-4.5 times for LDS-sized tiles, which is no cheaper than its L2-served gathers.
// Shapes whose boxes do not fit the LDS budget stay on the gather kernels (the launchers return CBIM_EUNSUPPORTED and the
// host side falls back — both paths are tested against each other bit for bit).
#include "cbim_common.h"
#include "up_lerp.h"

namespace cbim {

static constexpr int UT = 256;                        // threads
static constexpr int FD = 4, FH = 8, FW = 8;          // fine tile
static constexpr int FV = FD * FH * FW;

#ifdef CBIM_EMU
#define UT_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define UT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

__device__ __forceinline__ unsigned ut_mad24(unsigned a, unsigned b, unsigned c) {
#ifdef CBIM_EMU
  return a * b + c;
#else
  return __umul24(a, b) + c;
#endif
}

// trilinear align_corners source index (the arithmetic of pool_up_kernels.hip / ATen: float scale, truncation)
struct ULin { int i0, i1; float l0, l1; };
__device__ __forceinline__ float ulin_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
__device__ __forceinline__ ULin ulin_src(int dst, float scale, int in) {
  float src = scale * (float)dst;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  float l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  ULin r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in - 1 ? 1 : 0);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}

struct UpTileParams {
  const void* low; const void* skip; const void* g;
  const float* stats; const float* sums;
  void* out; void* out2;                 // act_fwd: a | norm_bwd: dskip, dup
  float* partials;                       // stats: [N][P][Cl][3]
  int Dl, Hl, Wl, Cl, D, H, W, Cs, skip_first, act;
  int tiles_d, tiles_h, tiles_w;
  int bd, bh, bw;                        // LDS box extents (upper bounds over all tiles)
  int P;                                 // stats: records per image (workgroups per image)
};

struct AxTab { int r0, r1; float l0, l1; };          // BYTE offsets of the two source rows / planes / columns inside the LDS box + weights

// ---- stage the coarse box of fine tile (td, th, tw) of image n and the axis tables --------------------------------
// LDS layout: [box rows][Cl * ES bytes] | AxTab d[FD] | AxTab h[FH] | AxTab w[FW]
template <typename T>
__device__ __forceinline__ void stage_box(const UpTileParams& p, unsigned char* smem, int n, int td, int th, int tw,
                                          float sd, float sh, float sw) {
  constexpr int CPC = Elem<T>::CPC, ES = Elem<T>::SIZE;
  const int tid = threadIdx.x;
  const int d0 = td * FD, h0 = th * FH, w0 = tw * FW;
  const int dl = d0 + FD - 1 < p.D ? d0 + FD - 1 : p.D - 1, hl = h0 + FH - 1 < p.H ? h0 + FH - 1 : p.H - 1,
            wl = w0 + FW - 1 < p.W ? w0 + FW - 1 : p.W - 1;
  const int od = ulin_src(d0, sd, p.Dl).i0, oh = ulin_src(h0, sh, p.Hl).i0, ow = ulin_src(w0, sw, p.Wl).i0;   // box origin
  const int ed = ulin_src(dl, sd, p.Dl).i1 - od + 1, eh = ulin_src(hl, sh, p.Hl).i1 - oh + 1, ew = ulin_src(wl, sw, p.Wl).i1 - ow + 1;
  const int cch = p.Cl / CPC;
  const unsigned rowb = (unsigned)p.Cl * ES;
  const int items = ed * eh * ew * cch;
  for (int i = tid; i < items; i += UT) {
    const int cc = i % cch, r = i / cch;
    const int bwi = r % ew, q = r / ew, bhi = q % eh, bdi = q / eh;
    const size_t src = ((((size_t)n * p.Dl + od + bdi) * p.Hl + oh + bhi) * p.Wl + ow + bwi) * p.Cl + (size_t)cc * CPC;
    *(u32x4*)(smem + (unsigned)((bdi * p.bh + bhi) * p.bw + bwi) * rowb + (unsigned)cc * 16u) = ld_chunk<T>(p.low, src);
  }
  AxTab* tab = (AxTab*)(smem + (unsigned)(p.bd * p.bh * p.bw) * rowb);
  if (tid < FD + FH + FW) {
    int dst, in, org;
    float sc;
    if (tid < FD) { dst = d0 + tid; if (dst > p.D - 1) dst = p.D - 1; in = p.Dl; org = od; sc = sd; }
    else if (tid < FD + FH) { dst = h0 + tid - FD; if (dst > p.H - 1) dst = p.H - 1; in = p.Hl; org = oh; sc = sh; }
    else { dst = w0 + tid - FD - FH; if (dst > p.W - 1) dst = p.W - 1; in = p.Wl; org = ow; sc = sw; }
    const ULin l = ulin_src(dst, sc, in);
    // byte offset of a box coordinate along this axis: the per-voxel address is then three adds (the products of run-time
    // extents were 16 quarter-rate 32-bit multiplies per output chunk)
    const int unit = (int)rowb * (tid < FD ? p.bh * p.bw : tid < FD + FH ? p.bw : 1);
    tab[tid] = AxTab{(l.i0 - org) * unit, (l.i1 - org) * unit, l.l0, l.l1};
  }
}

// value of up(low) at tile voxel (fd, fh, fw), channel chunk `cl` (channels), in float32 (same arithmetic as up_chunk of
// pool_up_kernels.hip; never rounded to the storage type: the up-sampled tensor is not stored)
template <typename T>
__device__ __forceinline__ void up_from_box(const UpTileParams& p, const unsigned char* smem, int fd, int fh, int fw, int cl, float* f) {
  constexpr int CPC = Elem<T>::CPC, ES = Elem<T>::SIZE;
  const unsigned rowb = (unsigned)p.Cl * ES;
  const AxTab* tab = (const AxTab*)(smem + (unsigned)(p.bd * p.bh * p.bw) * rowb);
  const AxTab ad = tab[fd], ah = tab[FD + fh], aw = tab[FD + FH + fw];
  u32x4 cn[8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned rbase = (unsigned)((a ? ad.r1 : ad.r0) + (b ? ah.r1 : ah.r0)) + (unsigned)cl * ES;
      cn[(a * 2 + b) * 2] = *(const u32x4*)(smem + rbase + (unsigned)aw.r0);
      cn[(a * 2 + b) * 2 + 1] = *(const u32x4*)(smem + rbase + (unsigned)aw.r1);
    }
  trilerp<T>(cn, ad.l0, ad.l1, ah.l0, ah.l1, aw.l0, aw.l1, f);
}

// MODE 0: statistics of up(low) (partials)   MODE 1: a = act(IN([skip | up]))   MODE 2: IN backward -> dskip, dup
// grid = (workgroups per image, N); a workgroup walks tiles wg, wg + gridDim.x, ... of its image
template <typename T, int MODE>
__global__ void __launch_bounds__(UT) k_up_tile(UpTileParams p) {
  constexpr int CPC = Elem<T>::CPC;
  UT_DYN_SMEM(smem);
  const int tid = threadIdx.x, n = blockIdx.y;
  const int Ct = MODE == 0 ? p.Cl : p.Cs + p.Cl;
  const int cch = Ct / CPC, vlc = UT / cch;
  const int cc = tid % cch, vl = tid / cch;
  const bool active = vl < vlc;
  const int skip_lo = p.skip_first ? 0 : p.Cl, low_lo = MODE == 0 ? 0 : (p.skip_first ? p.Cs : 0);
  const int c0 = cc * CPC;
  const bool is_skip = MODE != 0 && c0 >= skip_lo && c0 < skip_lo + p.Cs;
  const float sd = ulin_scale(p.Dl, p.D), sh = ulin_scale(p.Hl, p.H), sw = ulin_scale(p.Wl, p.W);
  float mean[CPC], rstd[CPC], m1[CPC], m2[CPC];
  float s0[CPC], s1[CPC], shift[CPC];
  float cnt = 0.f;
#pragma unroll
  for (int j = 0; j < CPC; ++j) { mean[j] = 0.f; rstd[j] = 1.f; m1[j] = 0.f; m2[j] = 0.f; s0[j] = 0.f; s1[j] = 0.f; shift[j] = 0.f; }
  if (MODE != 0 && active) {
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      mean[j] = p.stats[((size_t)n * Ct + c0 + j) * 2];
      rstd[j] = p.stats[((size_t)n * Ct + c0 + j) * 2 + 1];
      if (MODE == 2) { m1[j] = p.sums[((size_t)n * Ct + c0 + j) * 2]; m2[j] = p.sums[((size_t)n * Ct + c0 + j) * 2 + 1]; }
    }
  }
  const int tiles = p.tiles_d * p.tiles_h * p.tiles_w;
  const size_t S = (size_t)p.D * p.H * p.W;
  // per-image base pointers (64-bit, once) + 32-bit byte offsets from 24-bit multiplies per voxel: the 64-bit element
  // arithmetic per item was ~10 quarter-rate instructions
  constexpr unsigned ES = Elem<T>::SIZE;
  const unsigned cat_rb = (unsigned)Ct * ES, skip_rb = (unsigned)p.Cs * ES, low_rb = (unsigned)p.Cl * ES;
  const unsigned cat_cb = (unsigned)c0 * ES, skip_cb = (unsigned)(c0 - skip_lo) * ES, low_cb = (unsigned)(c0 - low_lo) * ES;
  const unsigned char* const skip_n = (const unsigned char*)p.skip + (size_t)n * S * skip_rb;
  const unsigned char* const g_n = (const unsigned char*)p.g + (size_t)n * S * cat_rb;
  unsigned char* const out_n = (unsigned char*)p.out + (size_t)n * S * (MODE == 2 ? skip_rb : cat_rb);
  unsigned char* const out2_n = (unsigned char*)p.out2 + (size_t)n * S * low_rb;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int tw = t % p.tiles_w, q = t / p.tiles_w, th = q % p.tiles_h, td = q / p.tiles_h;
    __syncthreads();                                   // the previous tile's box is no longer read
    stage_box<T>(p, smem, n, td, th, tw, sd, sh, sw);
    __syncthreads();
    if (!active) continue;
    // B voxels per trip: every global load of the batch (skip rows, gradient rows) is in flight before the first use —
    // one load -> use -> store chain per voxel left a skip-chunk thread waiting a full memory round trip 12 times per
    // tile (the kernel ran at 2.4 TB/s)
    constexpr int B = 4;
    for (int vb = vl; vb < FV; vb += vlc * B) {
      u32x4 sk[B], gq[B];
      unsigned vox[B];                                  // voxel index inside the image (< 2^24, checked by the launcher)
      int fdv[B], fhv[B], fwv[B];
      bool ok[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int v = vb + b * vlc;
        fwv[b] = v % FW; fhv[b] = (v / FW) % FH; fdv[b] = v / (FW * FH);
        const int d = td * FD + fdv[b], h = th * FH + fhv[b], w = tw * FW + fwv[b];
        ok[b] = v < FV && d < p.D && h < p.H && w < p.W;
        vox[b] = ut_mad24(ut_mad24((unsigned)d, (unsigned)p.H, (unsigned)h), (unsigned)p.W, (unsigned)w);
        sk[b] = u32x4{0u, 0u, 0u, 0u};
        gq[b] = u32x4{0u, 0u, 0u, 0u};
        if (ok[b]) {
          if (is_skip) sk[b] = *(const u32x4*)(skip_n + ut_mad24(vox[b], skip_rb, skip_cb));
          if (MODE == 2) gq[b] = *(const u32x4*)(g_n + ut_mad24(vox[b], cat_rb, cat_cb));
        }
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (!ok[b]) continue;
        float f[CPC];
        if (is_skip) Elem<T>::unpack(sk[b], f);
        else up_from_box<T>(p, smem, fdv[b], fhv[b], fwv[b], c0 - low_lo, f);
        if (MODE == 0) {
          if (cnt == 0.f) {
#pragma unroll
            for (int j = 0; j < CPC; ++j) shift[j] = f[j];
          }
          cnt += 1.f;
#pragma unroll
          for (int j = 0; j < CPC; ++j) { const float dlt = f[j] - shift[j]; s0[j] += dlt; s1[j] += dlt * dlt; }
        } else if (MODE == 1) {
          if (p.act == CBIM_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < CPC; ++j) { const float y = (f[j] - mean[j]) * rstd[j]; f[j] = y > 0.f ? y : 0.f; }
          } else {
#pragma unroll
            for (int j = 0; j < CPC; ++j) f[j] = act_fwd((f[j] - mean[j]) * rstd[j], p.act);
          }
          *(u32x4*)(out_n + ut_mad24(vox[b], cat_rb, cat_cb)) = Elem<T>::pack(f);
        } else {
          float gg[CPC];
          Elem<T>::unpack(gq[b], gg);
#pragma unroll
          for (int j = 0; j < CPC; ++j) {
            const float xh = (f[j] - mean[j]) * rstd[j];
            gg[j] = rstd[j] * (gg[j] - m1[j] - xh * m2[j]);
          }
          if (is_skip) *(u32x4*)(out_n + ut_mad24(vox[b], skip_rb, skip_cb)) = Elem<T>::pack(gg);
          else *(u32x4*)(out2_n + ut_mad24(vox[b], low_rb, low_cb)) = Elem<T>::pack(gg);
        }
      }
    }
  }
  if (MODE == 0) {
    // one (n, mean, M2) record per channel and workgroup: voxel lanes merged in fixed order (as k_upcat_fwd_stats)
    __syncthreads();
    float* red = (float*)smem;                         // UT * CPC * 3 floats <= 24 KiB (the box is dead)
    if (active) {
#pragma unroll
      for (int j = 0; j < CPC; ++j) {
        const Moments m = moments_from_shifted(cnt, shift[j], s0[j], s1[j]);
        red[(tid * CPC + j) * 3 + 0] = m.n;
        red[(tid * CPC + j) * 3 + 1] = m.mean;
        red[(tid * CPC + j) * 3 + 2] = m.m2;
      }
    }
    __syncthreads();
    for (int chn = tid; chn < cch * CPC; chn += UT) {
      const int c2 = chn / CPC, j = chn % CPC;
      Moments acc = {0.f, 0.f, 0.f};
      for (int q = 0; q < vlc; ++q) {
        const float* r = red + ((q * cch + c2) * CPC + j) * 3;
        const Moments b = {r[0], r[1], r[2]};
        acc = moments_merge(acc, b);
      }
      const size_t o = (((size_t)n * p.P + blockIdx.x) * p.Cl + c2 * CPC + j) * 3;
      p.partials[o] = acc.n; p.partials[o + 1] = acc.mean; p.partials[o + 2] = acc.m2;
    }
  }
}

}  // namespace cbim

using namespace cbim;

extern "C" int cbim_stats_finalize(const float* partials, int N, int P, int C, double count, float eps, int mode,
                                   float* out, void* stream);

static constexpr size_t UP_LDS_CAP = 96 * 1024;      // leaves room for a second workgroup on the CU
static int64_t g_up_tile_min = 512;                  // fewest fine tiles per image the tiled kernels take (0: any; tests)
extern "C" int64_t cbim_up_tile_min_tiles(int64_t v) {
  const int64_t old = g_up_tile_min;
  if (v >= 0) g_up_tile_min = v;
  return old;
}

// largest coarse-box extent along one axis over all fine tiles of length t: the kernel's own index arithmetic (float
// scale, truncation) replayed on the host
static int box_extent(int in, int out, int t) {
  const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  auto i0 = [&](int dst) { int i = (int)(sc * (float)dst); return i > in - 1 ? in - 1 : i; };
  int best = 1;
  for (int f0 = 0; f0 < out; f0 += t) {
    const int fl = f0 + t - 1 < out ? f0 + t - 1 : out - 1;
    const int a = i0(f0), b = i0(fl), e = b + (b < in - 1 ? 1 : 0) - a + 1;
    if (e > best) best = e;
  }
  return best;
}

static int up_fill(UpTileParams& p, int dtype, int Dl, int Hl, int Wl, int Cl, int D, int H, int W, int Cs, size_t* smem) {
  const int cpc = dtype == CBIM_BF16 ? 8 : 4, es = dtype == CBIM_BF16 ? 2 : 4;
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(Cl > 0 && Cl % cpc == 0 && Cs >= 0 && Cs % cpc == 0, CBIM_EUNSUPPORTED, "up tile: channel counts %d / %d must be multiples of %d", Cl, Cs, cpc);
  CBIM_CHECK((Cs + Cl) / cpc <= UT, CBIM_EUNSUPPORTED, "up tile: %d channels unsupported", Cs + Cl);
  // 32-bit byte offsets inside one image, built from 24-bit multiplies
  CBIM_CHECK((int64_t)D * H * W < (1 << 24) && (int64_t)(Cs + Cl) * es < (1 << 24) && (int64_t)D * H * W * (Cs + Cl) * es < ((int64_t)1 << 32),
             CBIM_EUNSUPPORTED, "up tile: a %dx%dx%d image with %d channels exceeds the 32-bit addressing", D, H, W, Cs + Cl);
  p.Dl = Dl; p.Hl = Hl; p.Wl = Wl; p.Cl = Cl; p.D = D; p.H = H; p.W = W; p.Cs = Cs;
  p.tiles_d = (D + FD - 1) / FD; p.tiles_h = (H + FH - 1) / FH; p.tiles_w = (W + FW - 1) / FW;
  p.bd = box_extent(Dl, D, FD); p.bh = box_extent(Hl, H, FH); p.bw = box_extent(Wl, W, FW);
  size_t need = (size_t)p.bd * p.bh * p.bw * Cl * es + (FD + FH + FW) * sizeof(AxTab);
  const size_t red = (size_t)UT * cpc * 3 * sizeof(float);
  if (need < red) need = red;
  CBIM_CHECK(need <= UP_LDS_CAP, CBIM_EUNSUPPORTED, "up tile: coarse box of %dx%dx%d rows x %d channels exceeds the LDS budget", p.bd, p.bh, p.bw, Cl);
  // a workgroup per 4x8x8 fine tile: below ~512 tiles the chip is under-filled and the wide rows of the low levels leave a
  // thread 50-100 serial items (measured 75-115 us against 11-29 us of the gather kernels at 32^3 / 16^3)
  CBIM_CHECK(g_up_tile_min <= 0 || (int64_t)p.tiles_d * p.tiles_h * p.tiles_w >= g_up_tile_min, CBIM_EUNSUPPORTED,
             "up tile: %d tiles are too few for the tiled kernels", p.tiles_d * p.tiles_h * p.tiles_w);
  *smem = need;
  return CBIM_OK;
}

template <int MODE>
static int up_launch(int dtype, const UpTileParams& p, dim3 grid, size_t smem, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_up_tile<bf16_tag, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)UP_LDS_CAP);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_up_tile<float, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)UP_LDS_CAP);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  if (dtype == CBIM_BF16) CBIM_LAUNCH((k_up_tile<bf16_tag, MODE>), grid, dim3(UT), smem, st, p);
  else CBIM_LAUNCH((k_up_tile<float, MODE>), grid, dim3(UT), smem, st, p);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "up tile launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

// records per image of cbim_up_stats_tile's partials (= workgroups per image)
extern "C" int cbim_up_tile_parts(int D, int H, int W) {
  const int64_t tiles = (int64_t)((D + FD - 1) / FD) * ((H + FH - 1) / FH) * ((W + FW - 1) / FW);
  return (int)(tiles < 1024 ? tiles : 1024);
}

extern "C" int cbim_up_stats_tile(int dtype, const void* low, int N, int Dl, int Hl, int Wl, int Cl, int D, int H, int W,
                                  float eps, float* partials, int P, float* stats, void* stream) {
  UpTileParams p = {};
  size_t smem;
  if (int e = up_fill(p, dtype, Dl, Hl, Wl, Cl, D, H, W, 0, &smem)) return e;
  CBIM_CHECK(low && partials && stats && P == cbim_up_tile_parts(D, H, W), CBIM_EINVAL, "up_stats_tile: partials must have cbim_up_tile_parts records");
  p.low = low; p.partials = partials; p.P = P; p.skip_first = 1;
  if (int e = up_launch<0>(dtype, p, dim3((unsigned)P, (unsigned)N), smem, (hipStream_t)stream)) return e;
  return cbim_stats_finalize(partials, N, P, Cl, (double)D * H * W, eps, 0, stats, stream);
}

extern "C" int cbim_upcat_act_fwd_tile(int dtype, const void* low, const void* skip, const float* stats, void* out, int N,
                                       int Dl, int Hl, int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, int act,
                                       void* stream) {
  UpTileParams p = {};
  size_t smem;
  if (int e = up_fill(p, dtype, Dl, Hl, Wl, Cl, D, H, W, Cs, &smem)) return e;
  CBIM_CHECK(low && skip && stats && out && Cs > 0, CBIM_EINVAL, "null argument");
  p.low = low; p.skip = skip; p.stats = stats; p.out = out; p.skip_first = skip_first; p.act = act;
  const int64_t tiles = (int64_t)p.tiles_d * p.tiles_h * p.tiles_w;
  return up_launch<1>(dtype, p, dim3((unsigned)(tiles < 65535 ? tiles : 65535), (unsigned)N), smem, (hipStream_t)stream);
}

extern "C" int cbim_upcat_norm_bwd_tile(int dtype, const void* g, const void* low, const void* skip, const float* stats,
                                        const float* sums, void* dskip, void* dlow, void* dup_scratch, int N, int Dl, int Hl,
                                        int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream) {
  UpTileParams p = {};
  size_t smem;
  if (int e = up_fill(p, dtype, Dl, Hl, Wl, Cl, D, H, W, Cs, &smem)) return e;
  CBIM_CHECK(g && low && skip && stats && sums && dskip && dup_scratch && Cs > 0, CBIM_EINVAL, "null argument");
  p.g = g; p.low = low; p.skip = skip; p.stats = stats; p.sums = sums; p.out = dskip; p.out2 = dup_scratch; p.skip_first = skip_first;
  const int64_t tiles = (int64_t)p.tiles_d * p.tiles_h * p.tiles_w;
  hipStream_t st = (hipStream_t)stream;
  if (int e = up_launch<2>(dtype, p, dim3((unsigned)(tiles < 65535 ? tiles : 65535), (unsigned)N), smem, st)) return e;
  if (!dlow) return CBIM_OK;   // the caller reduces dup_scratch with cbim_lin_adjoint_axis
  // transposed interpolation of the fine-resolution gradient: the gather kernel on dense Cl-channel rows
  return cbim_upcat_bwd(dtype, dup_scratch, dlow, nullptr, N, Dl, Hl, Wl, Cl, D, H, W, 0, 1, stream);
}

CBIM_DEFINE_WARM(up_tile)
