// conv_pw.hip — pointwise (1x1x1) convolutions as a row GEMM on the matrix cores (round 4).
//
//   y[r][co] = sum_ci  act(IN(x))[r][ci] * w[co][ci]   (+ residual)         forward  (conv_layers.py:48-49, 1x1x1 ConvNormAct:
//                                                                           MBConv expand / project, shortcuts, the pointwise
//                                                                           half of DepthwiseSeparableConv, MedFormer / SwinUNETR /
//                                                                           VNet down- and up-sampling GEMMs)
//   g[r][ci] = (sum_co dy[r][co] * w[co][ci]) * act'(IN(x))[r][ci]          its input gradient, with the two InstanceNorm-backward sums
//
// Why not k_conv_igemm: its pipeline hides the global round trip of the NEXT (tile, Cin chunk) unit behind the 54 MFMA steps of a
// 3x3x3 unit; a 1x1x1 unit has two, so every unit waits the round trip out (3.5 us per unit, profiles/r04_x_igemm_floor.txt: 56 us
// for 32^3 128->512 = 42 MB).  A pointwise layer is a streaming GEMM: what it needs is loads in flight, i.e. occupancy.
//
//   * no LDS staging of the operands at all: the A fragment of v_mfma_f32_32x32x16_bf16 for (row, k-group, lane half) IS 16
//     contiguous bytes of the channels-last row (slot 2 kg + half of the 64-byte chunk: the fragment order conv_igemm.hip packs
//     the weights for), the B fragment is 16 contiguous bytes of the packed weight image; both are plain global loads, the
//     weights stay in L1 / L2 (HBM-bound layers ask < 10 % of the matrix rate);
//   * InstanceNorm + activation of the producer are applied to the fragment in registers (statistics of the image in LDS);
//   * 256-thread workgroups (4 waves x 32 rows x NTW*32 output channels), <= 128 registers: four workgroups per CU;
//   * measured and left out (profiles/r04_y_pw_*.txt): the whole A row block of a tile requested at once with the weight block in
//     LDS (Cin <= 128: 64^3 64->256 53.7 -> 49.4 us, 32^3 128->512 23.2 -> 25.3, MedFormer step unchanged), the next tile's first
//     fragments requested before the epilogue (no change), 32 / 64 / 128 output channels per wave forced (step within 0.2 ms):
//     what remains is the store path — 134 MB of 16-byte stores in 50 us is 2.7 TB/s of a ~4.3 TB/s drain rate;
//   * epilogue as in conv_igemm.hip: accumulators transposed through a private 4 KiB LDS tile per wave into 16-byte channel
//     chunks; residual add, activated mask, shifted moments / backward sums; one statistics record per (image, strip of row
//     tiles), fixed merge order.
#include "cbim_common.h"
#include "conv_r32.h"
#include <stdlib.h>

namespace cbim {

static constexpr int PW_NT = 256, PW_NW = 4, PW_ROWS = 32 * PW_NW;

struct PwParams {
  const void* x; int64_t x_stride; const float* in_stats; const void* w;
  const void* res; int64_t res_stride; const void* mx; int64_t mx_stride; const float* m_stats;
  void* y; int64_t y_stride; float* partials;
  int N; int64_t S; int Cin, Cout, act;
  int nch;                 // 32-channel chunks of Cin
  int BNp;                 // n-block width of the packed weight image (32 | 64)
  int P, Pn;               // records per image of `partials` (allocated / written by this launch)
  int tiles_per_n, tiles_per_strip, n_wblk;
  // token mode (TK: nn.Linear over channels-last token rows, cbim_token_linear): bias float [Cout]; fp32 residual rows; the
  // activation whose derivative at the mask tensor (the pre-activation h of the MLP) multiplies the result; fp32 output rows
  const float* bias; const float* res32; int64_t res32_stride; int mask_act; int y32;
  const void* w_lo;        // X32 only: the residue image of the weight (cbim_conv3d_pack_weights_lo) or NULL
};

#ifdef CBIM_EMU
#define PW_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define PW_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

__device__ __forceinline__ void pw_wave_sync() {
#ifdef CBIM_EMU
  int z = 0;
  (void)cbim_emu::wave_exchange(&z, sizeof(z));
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

template <int ACT> __device__ __forceinline__ float pw_actf(float x, int rt) {
  if (ACT == CBIM_ACT_RELU) return x > 0.f ? x : 0.f;
  if (ACT == CBIM_ACT_NONE) return x;
  return act_fwd(x, rt);
}
template <int ACT> __device__ __forceinline__ float pw_actg(float x, int rt) {
  if (ACT == CBIM_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (ACT == CBIM_ACT_NONE) return 1.f;
  return act_grad(x, rt);
}

// grid.x = N * Pn * n_wblk (output-channel block fastest: the workgroups that share rows are neighbours in time)
// KS = 1: a workgroup tile is 128 rows, wave w owns rows 32 w .. 32 w + 31 over the whole K.
// KS = 4 (low-resolution layers with a long K — 8^3 1280->320 is 4 row tiles x 5 channel blocks = 20 workgroups of 80 dependent
//         k-groups each): a tile is 32 rows, the four waves split the k-groups, their partial accumulators meet in the LDS
//         transposition tiles and wave (nt mod 4) finishes n-tile nt — four times the workgroups, a quarter of the chain.
// TK (round 5): token mode — the same row GEMM as nn.Linear of the SwinUNETR trunk (qkv / proj / MLP / patch merging,
//   /root/reference/model/dim3/swin_unetr.py:467-490,552,640-643,707-731): x rows in bf16 or fp32 (X32: converted in
//   registers, the fp32 residual-stream gradient feeds the input-gradient GEMMs as it is), ACT applied to x on load WITHOUT
//   statistics (GELU of the MLP: the activated tensor is never stored), epilogue = + bias, * act'(mask) (the MLP's GELU' at the
//   stored pre-activation), + fp32 residual, bf16 or fp32 store — replaces aten::linear (hipBLASLt), aten::gelu(_backward),
//   the bias / residual adds and the fp32 <-> bf16 casts around them
template <int NTW, int ACT, int KS, bool TK = false, bool X32 = false>
__global__ void __launch_bounds__(PW_NT, 2) k_conv_pw(PwParams p) {
  static_assert(TK || !X32, "fp32 rows: token mode only");
  typedef bf16_tag T;
  constexpr int CPC = 8, OCH = 4, IT = 2;
  PW_DYN_SMEM(smem);
  float* scr_all = (float*)smem;                                  // [4 waves][32 x 32] fp32
  float* red = (float*)(smem + PW_NW * 4096);                     // [4 waves][NTW * 32][3]
  float* stL = (float*)(smem + PW_NW * 4096 + PW_NW * NTW * 32 * 3 * 4);   // [Cin][2]: mean, rstd of image n
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  float* scr = scr_all + wave * 1024;
  const unsigned b = blockIdx.x;
  const int nbw = (int)(b % (unsigned)p.n_wblk);
  const unsigned t_ = b / (unsigned)p.n_wblk;
  const int strip = (int)(t_ % (unsigned)p.Pn), n = (int)(t_ / (unsigned)p.Pn);
  const int co_base = nbw * NTW * 32;
  if (p.in_stats) {
    for (int i = tid; i < p.Cin * 2; i += PW_NT) stL[i] = p.in_stats[(size_t)n * p.Cin * 2 + i];
  }
  if (KS == 4)      // only the wave that finishes an n-tile writes its record: the other waves' slots stay empty
    for (int i = tid; i < PW_NW * NTW * 32 * 3; i += PW_NT) red[i] = 0.f;
  __syncthreads();
  constexpr int XES = X32 ? 4 : 2;                                 // bytes per element of an x row
  const unsigned char* const xn = (const unsigned char*)p.x + (size_t)n * p.S * p.x_stride * XES;
  const unsigned char* const wb = (const unsigned char*)p.w;
  // byte offsets of the lane's B fragments inside a (chunk, k-group) slab of the packed image
  unsigned boff[NTW];
  bool bok[NTW];
  const int cout_pad = (p.Cout + p.BNp - 1) / p.BNp * p.BNp;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int co0 = co_base + nt * 32;
    bok[nt] = co0 < cout_pad;
    const int nb = co0 / p.BNp, nn = co0 - nb * p.BNp;
    // 16-byte index = (((nb * nch + q) * KG + kg) * 2 + half) * BNp + nn + li, KG = 2
    boff[nt] = bok[nt] ? (unsigned)((((nb * p.nch) * 2) * 2 + half) * p.BNp + nn + li) * 16u : 0u;
  }
  const unsigned b_step = (unsigned)(2 * p.BNp) * 16u;            // one k-group further
  const int nsteps = p.nch * 2;
  constexpr int ROWS = KS == 4 ? 32 : PW_ROWS;
  const int spw = (nsteps + 3) / 4;
  const int s_lo = KS == 4 ? wave * spw : 0;
  const int s_hi = KS == 4 ? (s_lo + spw < nsteps ? s_lo + spw : nsteps) : nsteps;
  const int t_begin = strip * p.tiles_per_strip;
  int t_end = t_begin + p.tiles_per_strip;
  if (t_end > p.tiles_per_n) t_end = p.tiles_per_n;
  Moments run = {0.f, 0.f, 0.f};
  const int cc = lane % OCH;

  for (int t = t_begin; t < t_end; ++t) {
    const int64_t row0 = (int64_t)t * ROWS + (KS == 4 ? 0 : wave * 32);
    const int64_t arow = row0 + li;
    const bool a_in = arow < p.S;
    const unsigned char* const ap = xn + (size_t)(a_in ? arow : 0) * p.x_stride * XES + half * (8 * XES);
    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    auto load_a = [&](int s) -> u32x4 {          // s = 2 * chunk + kg: slot (2 kg + half) of the chunk = bytes s * 32 + half * 16
      const int c0 = s * 16 + half * 8;
      if (!a_in || c0 >= p.Cin) return u32x4{0u, 0u, 0u, 0u};
      return *(const u32x4*)(ap + (size_t)s * (16 * XES));
    };
    auto load_a2 = [&](int s) -> u32x4 {         // X32: channels 4..7 of the lane's 8 (the second 16 bytes of its fp32 slot)
      const int c0 = s * 16 + half * 8;
      if (!X32 || !a_in || c0 >= p.Cin) return u32x4{0u, 0u, 0u, 0u};
      return *(const u32x4*)(ap + (size_t)s * (16 * XES) + 16);
    };
    auto load_b = [&](int s, int nt) -> u32x4 {
      if (!bok[nt]) return u32x4{0u, 0u, 0u, 0u};
      return *(const u32x4*)(wb + boff[nt] + (size_t)s * b_step);
    };
    const unsigned char* const wlo = X32 ? (const unsigned char*)p.w_lo : nullptr;    // (workgroup-uniform)
    auto load_b_lo = [&](int s, int nt) -> u32x4 {
      if (!X32 || !wlo || !bok[nt]) return u32x4{0u, 0u, 0u, 0u};
      return *(const u32x4*)(wlo + boff[nt] + (size_t)s * b_step);
    };
    // fragments of PD k-groups in flight (a ring of PD register sets, the step loop unrolled by PD): with one k-group of
    // lookahead the deep-stage layers (Cin 1280: 80 steps of a few hundred cycles of MFMA work) waited out an L2 round trip per
    // step — 30 us for 8^3 1280->320
    constexpr int PD = NTW == 4 ? 2 : 4;           // (four sets of the 128-channel form do not fit 256 registers)
    u32x4 a_q[PD], a_q2[X32 ? PD : 1], b_q[PD][NTW];
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      a_q[u] = s_lo + u < s_hi ? load_a(s_lo + u) : u32x4{0u, 0u, 0u, 0u};
      if (X32) a_q2[u] = s_lo + u < s_hi ? load_a2(s_lo + u) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) b_q[u][nt] = s_lo + u < s_hi ? load_b(s_lo + u, nt) : u32x4{0u, 0u, 0u, 0u};
    }
    for (int s0 = s_lo; s0 < s_hi; s0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s = s0 + u;
        if (s < s_hi) {                                // wave-uniform
          u32x4 a = a_q[u], a_lo = u32x4{0u, 0u, 0u, 0u}, bb[NTW];
          if (X32) {                                   // fp32 row slot -> bf16 hi + lo fragments: the rows keep fp32 accuracy (two
            const u32x4 a2 = a_q2[u];                  //  MFMAs per n-tile: these GEMMs are bound by their row traffic).  Raw values
            const float f8[CPC] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),   // ride the ring:
                                   __uint_as_float(a2.x), __uint_as_float(a2.y), __uint_as_float(a2.z), __uint_as_float(a2.w)};  // the conversion
            a = Elem<T>::pack(f8);                     //  waits for the load only here, one ring depth later
            float fh[CPC], fl[CPC];
            Elem<T>::unpack(a, fh);
#pragma unroll
            for (int j = 0; j < CPC; ++j) fl[j] = f8[j] - fh[j];
            a_lo = Elem<T>::pack(fl);
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) bb[nt] = b_q[u][nt];
          if (s + PD < s_hi) {
            a_q[u] = load_a(s + PD);
            if (X32) a_q2[u] = load_a2(s + PD);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) b_q[u][nt] = load_b(s + PD, nt);
          }
          if (TK && ACT != CBIM_ACT_NONE) {            // activation on load, no statistics (the MLP's GELU)
            const int c0 = s * 16 + half * 8;
            if (a_in && c0 < p.Cin) {
              float f[CPC];
              Elem<T>::unpack(a, f);
#pragma unroll
              for (int j = 0; j < CPC; ++j) f[j] = pw_actf<ACT>(f[j], p.act);
              a = Elem<T>::pack(f);
            }
          }
          if (!TK && p.in_stats) {
            const int c0 = s * 16 + half * 8;
            if (a_in && c0 < p.Cin) {
              float f[CPC];
              Elem<T>::unpack(a, f);
              const float* st = stL + c0 * 2;
#pragma unroll
              for (int j = 0; j < CPC; ++j) f[j] = pw_actf<ACT>((f[j] - st[2 * j]) * st[2 * j + 1], p.act);
              a = Elem<T>::pack(f);
            }
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            if (X32 && wlo)     // (the residue image is read where it is used: L1 / L2 hits next to the hi image, few layers)
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, load_b_lo(s, nt)), acc[nt], 0, 0, 0);
            if (X32) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_lo), __builtin_bit_cast(bf16x8, bb[nt]), acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb[nt]), acc[nt], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue -------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int cch0 = co_base + nt * 32 + cc * CPC;
      const bool c_ok = cch0 < p.Cout;
      float mm[CPC], mr[CPC], s0[CPC], s1[CPC], sh[CPC];
      float cnt = 0.f;
#pragma unroll
      for (int j = 0; j < CPC; ++j) { mm[j] = 0.f; mr[j] = 1.f; s0[j] = 0.f; s1[j] = 0.f; sh[j] = 0.f; }
      if (p.mx && p.m_stats && c_ok) {
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          mm[j] = p.m_stats[((size_t)n * p.Cout + cch0 + j) * 2];
          mr[j] = p.m_stats[((size_t)n * p.Cout + cch0 + j) * 2 + 1];
        }
      }
      float bv[CPC];
      if (TK) {
#pragma unroll
        for (int j = 0; j < CPC; ++j) bv[j] = (p.bias && c_ok) ? p.bias[cch0 + j] : 0.f;
      }
      if (KS == 4) __syncthreads(); else pw_wave_sync();   // the previous n-tile's scratch reads are done
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[nt][r];
      if (KS == 4) __syncthreads(); else pw_wave_sync();
      const bool mine = KS != 4 || wave == (nt & 3);        // KS: wave (nt mod 4) adds the four partial tiles and finishes nt
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        if (!mine) continue;
        const int vr = (lane + 64 * it) / OCH;
        const int64_t row = row0 + vr;
        const bool inb = c_ok && row < p.S;
        float v[CPC];
#pragma unroll
        for (int j4 = 0; j4 < CPC; j4 += 4) {
          f32x4 q4 = *(const f32x4*)((KS == 4 ? scr_all : scr) + vr * 32 + cc * CPC + j4);
          if (KS == 4) {
#pragma unroll
            for (int g = 1; g < PW_NW; ++g) {             // wave order
              const f32x4 o4 = *(const f32x4*)(scr_all + g * 1024 + vr * 32 + cc * CPC + j4);
              q4.x += o4.x; q4.y += o4.y; q4.z += o4.z; q4.w += o4.w;
            }
          }
          v[j4] = q4.x; v[j4 + 1] = q4.y; v[j4 + 2] = q4.z; v[j4 + 3] = q4.w;
        }
        if (it == 0 && !p.mx) {
          // common shift per channel for the wave: the value lane `cc` holds for its first row
#pragma unroll
          for (int j = 0; j < CPC; ++j) sh[j] = __shfl(v[j], cc, 64);
        }
        if (TK) {
          if (inb) {
            const size_t grow = (size_t)n * p.S + (size_t)row;
#pragma unroll
            for (int j = 0; j < CPC; ++j) v[j] += bv[j];
            if (p.mx) {                                 // * act'(h): the input gradient of the MLP's second Linear at the stored
              float f[CPC];                             //   pre-activation (d/dh of act(h) . W2)
              Elem<T>::unpack(*(const u32x4*)((const unsigned char*)p.mx + (grow * p.mx_stride + cch0) * 2), f);
#pragma unroll
              for (int j = 0; j < CPC; ++j) v[j] *= act_grad(f[j], p.mask_act);
            }
            if (p.res32) {
              const float* rp = p.res32 + grow * p.res32_stride + cch0;
              const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
              v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
              v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
            if (p.y32) {
              float* yp = (float*)p.y + grow * p.y_stride + cch0;
              *(f32x4*)yp = f32x4{v[0], v[1], v[2], v[3]};
              *(f32x4*)(yp + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
              *(u32x4*)((unsigned char*)p.y + (grow * p.y_stride + cch0) * 2) = Elem<T>::pack(v);
            }
          }
          continue;
        }
        if (inb) {
          const size_t grow = (size_t)n * p.S + (size_t)row;
          if (p.res) {
            float f[CPC];
            Elem<T>::unpack(*(const u32x4*)((const unsigned char*)p.res + (grow * p.res_stride + cch0) * 2), f);
#pragma unroll
            for (int j = 0; j < CPC; ++j) v[j] += f[j];
          }
          if (p.mx) {
            float f[CPC];
            Elem<T>::unpack(*(const u32x4*)((const unsigned char*)p.mx + (grow * p.mx_stride + cch0) * 2), f);
#pragma unroll
            for (int j = 0; j < CPC; ++j) {
              const float xh = (f[j] - mm[j]) * mr[j];
              v[j] *= pw_actg<ACT>(xh, p.act);
              s0[j] += v[j];
              s1[j] += v[j] * xh;
            }
          } else {
#pragma unroll
            for (int j = 0; j < CPC; ++j) { const float d = v[j] - sh[j]; s0[j] += d; s1[j] += d * d; }
          }
          cnt += 1.f;
          *(u32x4*)((unsigned char*)p.y + (grow * p.y_stride + cch0) * 2) = Elem<T>::pack(v);
        }
      }
      if (p.partials && mine) {
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
            float* rr = red + ((wave * NTW + nt) * 32 + cc * CPC + j) * 3;
            rr[0] = a.n; rr[1] = a.mean; rr[2] = a.m2;
          }
        }
      }
    }
    if (p.partials) {
      __syncthreads();                                  // `red` complete
      if (tid < NTW * 32) {
        Moments a = {0.f, 0.f, 0.f};
        for (int g = 0; g < PW_NW; ++g) {
          const float* rr = red + ((g * NTW) * 32 + tid) * 3;
          if (p.mx) { a.mean += rr[1]; a.m2 += rr[2]; }
          else { const Moments bq = {rr[0], rr[1], rr[2]}; a = moments_merge(a, bq); }
        }
        if (p.mx) { run.mean += a.mean; run.m2 += a.m2; }
        else run = moments_merge(run, a);
      }
      __syncthreads();                                  // before the next tile's records overwrite `red`
    }
  }
  if (p.partials && tid < NTW * 32 && co_base + tid < p.Cout) {
    const size_t o = (((size_t)n * p.P + strip) * p.Cout + co_base + tid) * 3;
    p.partials[o] = run.n; p.partials[o + 1] = run.mean; p.partials[o + 2] = run.m2;
    for (int r = strip + p.Pn; r < p.P; r += p.Pn) {      // buffer sized for another kernel's grid: empty records
      const size_t o2 = (((size_t)n * p.P + r) * p.Cout + co_base + tid) * 3;
      p.partials[o2] = 0.f; p.partials[o2 + 1] = 0.f; p.partials[o2 + 2] = 0.f;
    }
  }
}


// ---- weight gradient of the pointwise layers ------------------------------------------------------------------------------
//   dw[co][ci] = sum_r dy[r][co] * act(IN(x))[r][ci]
// k_conv_wgrad (conv_wgrad.hip) gives a workgroup ONE (32 co x 32 ci) block, stages a 4x8x8 tile of both tensors per barrier pair
// and lets its four waves split the tile's 16 k-steps: every dy / x element is re-read by Cin/32 resp. Cout/32 workgroups and a
// stage is 4 MFMAs per wave.  Here a workgroup owns a (64 co x 64 ci) block (wave = one 32 x 32 quadrant), stages 128 rows of
// both tensors per barrier (the next stage's sixteen 16-byte loads per thread in flight while the current one computes, the
// InstanceNorm + activation applied on the way into LDS) and runs 8 k-steps per stage; both MFMA operands are transposed LDS
// reads (ds_read_b64_tr_b16, the fragment scheme of conv_wgrad.hip).  Strips of rows -> slabs [strip][Cout_pad][Cin_pad] -> the
// fixed-order k_wgrad_reduce of conv_wgrad.hip.
static constexpr int PWG_VT = 128;                    // rows per stage
__device__ __forceinline__ u32x2 pw_lds_tr16_b64(const unsigned char* p) {
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
  u32x2 r;
  r.x = (unsigned)(unsigned short)v.x | ((unsigned)(unsigned short)v.y << 16);
  r.y = (unsigned)(unsigned short)v.z | ((unsigned)(unsigned short)v.w << 16);
  return r;
#endif
}

struct PwgParams {
  const void* x; int64_t x_stride; const float* in_stats; const void* dy; int64_t dy_stride; float* ws;
  int N; int64_t S; int Cin, Cout, act;
  int strips, rows_per_strip, ci_blocks, Cout_pad, Cin_pad;
  int act_raw;             // token mode: ACT applied to x without statistics (the MLP's GELU on the stored pre-activation)
};

// X32 / DY32 (round 5, token Linears): the operand rows are fp32 (the residual-stream gradient; the patch-embedding input) and
// are rounded to bf16 on the way into LDS — no separate cast pass
template <int ACT, bool X32 = false, bool DY32 = false>
__global__ void __launch_bounds__(PW_NT, 2) k_pw_wgrad(PwgParams p) {
  typedef bf16_tag T;
  constexpr int CPC = 8;
  // LDS: two buffers x (dy sub-tiles co 0-31 | co 32-63, x sub-tiles ci 0-31 | ci 32-63), each [PWG_VT rows][64 B]
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][4][PWG_VT * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const int n = blockIdx.x / p.strips, strip = blockIdx.x % p.strips;
  const int cb = blockIdx.y / p.ci_blocks, ib = blockIdx.y % p.ci_blocks;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t r_begin = (int64_t)strip * p.rows_per_strip;
  int64_t r_end = r_begin + p.rows_per_strip;
  if (r_end > p.S) r_end = p.S;
  // staging: chunk q = tid + 256 u -> row (tid >> 3) + 32 u, 16-byte slot s = tid & 7 of the 64-channel row segment
  const int slot = tid & 7, vrow = tid >> 3;
  const int co_c = cb * 64 + slot * CPC, ci_c = ib * 64 + slot * CPC;
  const bool co_ok = co_c < p.Cout, ci_ok = ci_c < p.Cin;
  float mean[CPC], rstd[CPC];
#pragma unroll
  for (int j = 0; j < CPC; ++j) { mean[j] = 0.f; rstd[j] = 1.f; }
  if (p.in_stats && ci_ok) {
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      mean[j] = p.in_stats[((size_t)n * p.Cin + ci_c + j) * 2];
      rstd[j] = p.in_stats[((size_t)n * p.Cin + ci_c + j) * 2 + 1];
    }
  }
  constexpr int DES = DY32 ? 4 : 2, XES = X32 ? 4 : 2;
  const unsigned char* const dyn = (const unsigned char*)p.dy + ((size_t)n * p.S * p.dy_stride + co_c) * DES;
  const unsigned char* const xn = (const unsigned char*)p.x + ((size_t)n * p.S * p.x_stride + ci_c) * XES;
  const unsigned sub_off = (unsigned)(slot >> 2), in_off = (unsigned)(slot & 3) * 16u;
  u32x4 gd[4], gx[4], gd2[DY32 ? 4 : 1], gx2[X32 ? 4 : 1];
  auto issue = [&](int64_t r0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + vrow + 32 * u;
      const bool in = r < r_end;
      gd[u] = (in && co_ok) ? *(const u32x4*)(dyn + (size_t)r * p.dy_stride * DES) : u32x4{0u, 0u, 0u, 0u};
      gx[u] = (in && ci_ok) ? *(const u32x4*)(xn + (size_t)r * p.x_stride * XES) : u32x4{0u, 0u, 0u, 0u};
      if (DY32) gd2[u] = (in && co_ok) ? *(const u32x4*)(dyn + (size_t)r * p.dy_stride * DES + 16) : u32x4{0u, 0u, 0u, 0u};
      if (X32) gx2[u] = (in && ci_ok) ? *(const u32x4*)(xn + (size_t)r * p.x_stride * XES + 16) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto to_bf16 = [](const u32x4& a, const u32x4& b) -> u32x4 {      // 8 fp32 -> 8 bf16
    const float f8[CPC] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                           __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
    return Elem<T>::pack(f8);
  };
  auto commit = [&](int buf, int64_t r0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = vrow + 32 * u;
      const int64_t r = r0 + v;
      if (DY32) gd[u] = to_bf16(gd[u], gd2[u]);
      u32x4 xa = X32 ? to_bf16(gx[u], gx2[u]) : gx[u];
      if ((p.in_stats || p.act_raw) && r < r_end && ci_ok) {
        float f[CPC];
        Elem<T>::unpack(xa, f);
#pragma unroll
        for (int j = 0; j < CPC; ++j) f[j] = pw_actf<ACT>((f[j] - mean[j]) * rstd[j], p.act);
        xa = Elem<T>::pack(f);
      }
      *(u32x4*)(&lds[buf][sub_off][v * 64 + in_off]) = gd[u];
      *(u32x4*)(&lds[buf][2 + sub_off][v * 64 + in_off]) = xa;
    }
  };
  // fragment addressing of ds_read_b64_tr_b16 (conv_wgrad.hip): a 16-lane group reads a [4 rows][16 channels] block, lane i
  // receives channel i's 4 rows; two reads (rows m, m + 4) give the lane 8 consecutive k of its channel
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const unsigned colb = (unsigned)(16 * g16 + 4 * (i16 & 3)) * 2;
  const int mq = 8 * half + (i16 >> 2);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  issue(r_begin);
  commit(0, r_begin);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += PWG_VT) {
    const bool more = r0 + PWG_VT < r_end;
    if (more) issue(r0 + PWG_VT);
    const unsigned char* A = &lds[buf][wm][0];
    const unsigned char* B = &lds[buf][2 + wn][0];
#pragma unroll
    for (int ks = 0; ks < PWG_VT / 16; ++ks) {
      const unsigned m0 = (unsigned)(ks * 16 + mq) * 64u + colb, m1 = m0 + 4u * 64u;
      const u32x2 a0 = pw_lds_tr16_b64(A + m0), a1 = pw_lds_tr16_b64(A + m1);
      const u32x2 b0 = pw_lds_tr16_b64(B + m0), b1 = pw_lds_tr16_b64(B + m1);
      const u32x4 af = u32x4{a0.x, a0.y, a1.x, a1.y}, bf = u32x4{b0.x, b0.y, b1.x, b1.y};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
    }
    if (more) commit(buf ^ 1, r0 + PWG_VT);
    __syncthreads();
    buf ^= 1;
  }
  // slab of this strip: ws[(n * strips + strip)][Cout_pad][Cin_pad]
  float* wsb = p.ws + (size_t)blockIdx.x * p.Cout_pad * p.Cin_pad;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cb * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int ci = ib * 64 + wn * 32 + li;
    wsb[(size_t)co * p.Cin_pad + ci] = acc[r];
  }
}

}  // namespace cbim

using namespace cbim;

static int g_pw_on = 1;
/* process-wide switch (tests, A/B): 0 = pointwise convolutions stay on k_conv_igemm; returns the old value */
extern "C" int cbim_conv_pw_enable(int on) {
  const int old = g_pw_on;
  if (on >= 0) g_pw_on = on ? 1 : 0;
  return old;
}

static void pw_strips(int64_t S, int rows, int* tiles, int* tps, int* Pn) {
  const int64_t t = (S + rows - 1) / rows;
  int64_t per = (t + 511) / 512;
  if (per < 1) per = 1;
  *tiles = (int)t; *tps = (int)per; *Pn = (int)((t + per - 1) / per);
}

static bool pw_desc_ok(const cbim_conv_desc* d) {
  return g_pw_on && d->dtype == CBIM_BF16 && d->kD == 1 && d->kH == 1 && d->kW == 1 && d->pD == 0 && d->pH == 0 && d->pW == 0 &&
         d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->Cin <= 4096 && d->Di == d->Do && d->Hi == d->Ho && d->Wi == d->Wo &&
         (int64_t)d->Do * d->Ho * d->Wo * PW_ROWS < ((int64_t)1 << 40);
}

// output channels per wave and the K-split form, from the descriptor alone (cbim_conv_pw_records must agree with the launch)
static void pw_plan(const cbim_conv_desc* d, int* ntw, int* ks) {
  static const int ks_env = 1;   // tools/ only: 0 = never split K
  const int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  const int64_t row_wgs = (int64_t)d->N * ((S + PW_ROWS - 1) / PW_ROWS);
  int n = d->Cout <= 32 ? 1 : (d->Cout <= 64 ? 2 : 4);
  // 128 channels per wave when that still gives every CU a few workgroups, else 64 / 32
  if (n == 4 && row_wgs * ((d->Cout + 127) / 128) < 512) n = 2;
  const int nsteps = (d->Cin + 31) / 32 * 2;
  *ks = 1;
  if (ks_env && n <= 2 && nsteps >= 8 && row_wgs * ((d->Cout + n * 32 - 1) / (n * 32)) < 384) *ks = 4;
  *ntw = n;
}

int cbim_conv_pw_records(const cbim_conv_desc* d) {
  if (!pw_desc_ok(d)) return 0;
  int tiles, tps, Pn, ntw, ks;
  pw_plan(d, &ntw, &ks);
  pw_strips((int64_t)d->Do * d->Ho * d->Wo, ks == 4 ? 32 : PW_ROWS, &tiles, &tps, &Pn);
  return Pn;
}

bool cbim_conv_pw_eligible(const cbim_conv_desc* d, int64_t x_stride, const void* x2, int64_t res_stride, int64_t mask_stride,
                           int64_t y_stride, const void* res, const void* mask_x, const float* mask_stats) {
  if (!pw_desc_ok(d) || x2) return false;
  if (x_stride % 8 || y_stride % 8 || (res && res_stride % 8) || (mask_x && mask_stride % 8)) return false;
  if (mask_x && !mask_stats && d->act != CBIM_ACT_RELU) return false;
  return true;
}

template <int NTW, int KS>
static int pw_launch_n(const PwParams& p, dim3 grid, size_t smem, int act, hipStream_t st) {
  if (act == CBIM_ACT_RELU) CBIM_LAUNCH((k_conv_pw<NTW, CBIM_ACT_RELU, KS>), grid, dim3(PW_NT), smem, st, p);
  else if (act == CBIM_ACT_NONE) CBIM_LAUNCH((k_conv_pw<NTW, CBIM_ACT_NONE, KS>), grid, dim3(PW_NT), smem, st, p);
  else CBIM_LAUNCH((k_conv_pw<NTW, -1, KS>), grid, dim3(PW_NT), smem, st, p);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

int cbim_conv_pw_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const float* in_stats, const void* w_packed,
                        const void* res, int64_t res_stride, const void* mask_x, int64_t mask_stride, const float* mask_stats,
                        void* y, int64_t y_stride, float* partials, int P, void* stream) {
  PwParams p;
  p.x = x; p.x_stride = x_stride; p.in_stats = in_stats; p.w = w_packed;
  p.res = res; p.res_stride = res_stride; p.mx = mask_x; p.mx_stride = mask_stride; p.m_stats = mask_stats;
  p.y = y; p.y_stride = y_stride; p.partials = partials;
  p.N = d->N; p.S = (int64_t)d->Do * d->Ho * d->Wo; p.Cin = d->Cin; p.Cout = d->Cout; p.act = d->act;
  p.nch = (d->Cin + 31) / 32;
  p.BNp = d->Cout <= 32 ? 32 : 64;
  p.bias = nullptr; p.res32 = nullptr; p.res32_stride = 0; p.mask_act = 0; p.y32 = 0; p.w_lo = nullptr;
  int ntw, ks;
  pw_plan(d, &ntw, &ks);
  pw_strips(p.S, ks == 4 ? 32 : PW_ROWS, &p.tiles_per_n, &p.tiles_per_strip, &p.Pn);
  p.P = P > p.Pn ? P : p.Pn;
  CBIM_CHECK(!partials || P >= p.Pn, CBIM_EINVAL, "pointwise conv: %d statistics records < %d", P, p.Pn);
  const int64_t row_wgs = (int64_t)d->N * p.Pn;
  p.n_wblk = (d->Cout + ntw * 32 - 1) / (ntw * 32);
  const size_t smem = (size_t)PW_NW * 4096 + (size_t)PW_NW * ntw * 32 * 3 * 4 + (size_t)d->Cin * 2 * 4;
  CBIM_CHECK(row_wgs * p.n_wblk < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "pointwise conv: grid too large");
  dim3 grid((unsigned)(row_wgs * p.n_wblk));
  hipStream_t st = (hipStream_t)stream;
  // a mask tensor without statistics is the activated tensor itself (ReLU): xh := a, as in conv_igemm.hip
  if (ks == 4) return ntw == 1 ? pw_launch_n<1, 4>(p, grid, smem, d->act, st) : pw_launch_n<2, 4>(p, grid, smem, d->act, st);
  switch (ntw) {
    case 1: return pw_launch_n<1, 1>(p, grid, smem, d->act, st);
    case 2: return pw_launch_n<2, 1>(p, grid, smem, d->act, st);
    default: return pw_launch_n<4, 1>(p, grid, smem, d->act, st);
  }
}


// ---- pointwise weight gradient: host side (the reduce pass is conv_wgrad.hip's k_wgrad_reduce) -----------------------------
static void pwg_cfg(const cbim_conv_desc* d, int* strips, int* rps, int* co_blocks, int* ci_blocks) {
  const int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  *co_blocks = (d->Cout + 63) / 64; *ci_blocks = (d->Cin + 63) / 64;
  const int64_t pairs = (int64_t)*co_blocks * *ci_blocks * d->N;
  const int64_t stages = (S + PWG_VT - 1) / PWG_VT;
  static const int64_t wgs = 768;   // (tools: A/B of the strip count)
  int64_t want = (wgs + pairs - 1) / pairs;                 // ~3 workgroups per CU
  if (want > stages) want = stages;
  const size_t slab = (size_t)*co_blocks * 64 * *ci_blocks * 64 * sizeof(float);
  int64_t cap = (int64_t)((64ull << 20) / (slab * d->N));   // slab workspace <= 64 MiB
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  const int64_t per = (stages + want - 1) / want;           // stages per strip
  *rps = (int)(per * PWG_VT);
  *strips = (int)((stages + per - 1) / per);
}

bool cbim_conv_pw_wgrad_eligible(const cbim_conv_desc* d, int64_t x_stride, const void* x2, int64_t dy_stride, const void* dy2) {
  return pw_desc_ok(d) && !x2 && !dy2 && x_stride % 8 == 0 && dy_stride % 8 == 0 && (int64_t)d->Do * d->Ho * d->Wo < ((int64_t)1 << 31);
}

size_t cbim_conv_pw_wgrad_workspace(const cbim_conv_desc* d) {
  if (!pw_desc_ok(d)) return 0;
  int strips, rps, cob, cib;
  pwg_cfg(d, &strips, &rps, &cob, &cib);
  return (size_t)d->N * strips * cob * 64 * cib * 64 * sizeof(float);
}

int cbim_conv_pw_wgrad_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const float* in_stats, const void* dy,
                              int64_t dy_stride, float* ws, int* n_slabs, int* Cout_pad, int* Cin_pad, void* stream) {
  PwgParams p;
  p.x = x; p.x_stride = x_stride; p.in_stats = in_stats; p.dy = dy; p.dy_stride = dy_stride; p.ws = ws;
  p.N = d->N; p.S = (int64_t)d->Do * d->Ho * d->Wo; p.Cin = d->Cin; p.Cout = d->Cout; p.act = d->act;
  p.act_raw = 0;
  int cob;
  pwg_cfg(d, &p.strips, &p.rows_per_strip, &cob, &p.ci_blocks);
  p.Cout_pad = cob * 64; p.Cin_pad = p.ci_blocks * 64;
  *n_slabs = d->N * p.strips; *Cout_pad = p.Cout_pad; *Cin_pad = p.Cin_pad;
  dim3 grid((unsigned)(d->N * p.strips), (unsigned)(cob * p.ci_blocks));
  hipStream_t st = (hipStream_t)stream;
  const int act = in_stats ? d->act : CBIM_ACT_NONE;
  if (act == CBIM_ACT_RELU) CBIM_LAUNCH((k_pw_wgrad<CBIM_ACT_RELU>), grid, dim3(PW_NT), 0, st, p);
  else if (act == CBIM_ACT_NONE) CBIM_LAUNCH((k_pw_wgrad<CBIM_ACT_NONE>), grid, dim3(PW_NT), 0, st, p);
  else CBIM_LAUNCH((k_pw_wgrad<-1>), grid, dim3(PW_NT), 0, st, p);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

// ---- token mode: nn.Linear over channels-last token rows (SwinUNETR trunk) ---------------------------------------------------
int cbim_wgrad_reduce_launch(const float* ws, float* dw, int n_slabs, int taps, int Cout, int Cin, int Cout_pad, int Cin_pad,
                             void* stream);      // conv_wgrad.hip

static void tok_desc(cbim_conv_desc* d, int64_t rows, int Cin, int Cout) {
  // rows as a [1][D][H][W] volume (the descriptor carries 32-bit extents)
  int64_t W = rows, H = 1, D = 1;
  while (W >= ((int64_t)1 << 30)) { W = (W + 1) / 2; H *= 2; }
  d->dtype = CBIM_BF16; d->N = 1; d->Di = d->Do = (int)D; d->Hi = d->Ho = (int)H; d->Wi = d->Wo = (int)W;
  d->Cin = Cin; d->Cout = Cout; d->kD = d->kH = d->kW = 1; d->pD = d->pH = d->pW = 0; d->act = CBIM_ACT_NONE;
}

template <int NTW, int KS, int ACT, bool X32>
static int tok_launch_a(const PwParams& p, dim3 grid, size_t smem, hipStream_t st) {
  CBIM_LAUNCH((k_conv_pw<NTW, ACT, KS, true, X32>), grid, dim3(PW_NT), smem, st, p);
  return CBIM_LAST_LAUNCH() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}
template <int NTW, int KS>
static int tok_launch_n(const PwParams& p, dim3 grid, size_t smem, int act_in, int x32, hipStream_t st) {
  if (x32) return tok_launch_a<NTW, KS, CBIM_ACT_NONE, true>(p, grid, smem, st);
  if (act_in == CBIM_ACT_NONE) return tok_launch_a<NTW, KS, CBIM_ACT_NONE, false>(p, grid, smem, st);
  return tok_launch_a<NTW, KS, -1, false>(p, grid, smem, st);
}

extern "C" int cbim_token_linear(const void* x, int x_dtype, int64_t x_stride, int act_in, const void* w_packed, const void* w_lo_packed,
                                 const float* bias,
                                 const float* res, int64_t res_stride, const void* mask, int64_t mask_stride, int mask_act,
                                 void* y, int y_dtype, int64_t y_stride, int64_t rows, int Cin, int Cout, void* stream) {
  CBIM_CHECK(x && w_packed && y && rows >= 1, CBIM_EINVAL, "token linear: null operand / no rows");
  CBIM_CHECK((x_dtype == CBIM_F32 || x_dtype == CBIM_BF16) && (y_dtype == CBIM_F32 || y_dtype == CBIM_BF16), CBIM_EINVAL, "token linear: bad dtype");
  CBIM_CHECK(Cin % 8 == 0 && Cout % 8 == 0 && Cin >= 8 && Cout >= 8 && Cin <= 4096, CBIM_EUNSUPPORTED,
             "token linear: %d -> %d features (multiples of 8, at most 4096 inputs)", Cin, Cout);
  CBIM_CHECK(x_stride % 8 == 0 && y_stride % 8 == 0 && (!res || res_stride % 4 == 0) && (!mask || mask_stride % 8 == 0), CBIM_EUNSUPPORTED,
             "token linear: row strides must keep 16-byte chunks aligned");
  CBIM_CHECK(!(x_dtype == CBIM_F32 && act_in != CBIM_ACT_NONE), CBIM_EUNSUPPORTED, "token linear: activation on load takes bf16 rows");
  CBIM_CHECK(!w_lo_packed || x_dtype == CBIM_F32, CBIM_EUNSUPPORTED, "token linear: a residue weight image goes with fp32 rows");
  cbim_conv_desc d;
  tok_desc(&d, rows, Cin, Cout);
  PwParams p;
  p.w_lo = w_lo_packed;
  p.x = x; p.x_stride = x_stride; p.in_stats = nullptr; p.w = w_packed;
  p.res = nullptr; p.res_stride = 0; p.mx = mask; p.mx_stride = mask_stride; p.m_stats = nullptr;
  p.y = y; p.y_stride = y_stride; p.partials = nullptr;
  p.N = 1; p.S = rows; p.Cin = Cin; p.Cout = Cout; p.act = act_in;
  p.nch = (Cin + 31) / 32;
  p.BNp = Cout <= 32 ? 32 : 64;
  p.bias = bias; p.res32 = res; p.res32_stride = res_stride; p.mask_act = mask_act; p.y32 = y_dtype == CBIM_F32;
  int ntw, ks;
  pw_plan(&d, &ntw, &ks);
  pw_strips(p.S, ks == 4 ? 32 : PW_ROWS, &p.tiles_per_n, &p.tiles_per_strip, &p.Pn);
  p.P = p.Pn;
  p.n_wblk = (Cout + ntw * 32 - 1) / (ntw * 32);
  const size_t smem = (size_t)PW_NW * 4096 + (size_t)PW_NW * ntw * 32 * 3 * 4 + (size_t)Cin * 2 * 4;
  CBIM_CHECK((int64_t)p.Pn * p.n_wblk < ((int64_t)1 << 31), CBIM_EUNSUPPORTED, "token linear: grid too large");
  dim3 grid((unsigned)((int64_t)p.Pn * p.n_wblk));
  hipStream_t st = (hipStream_t)stream;
  const int x32 = x_dtype == CBIM_F32;
  if (ks == 4) return ntw == 1 ? tok_launch_n<1, 4>(p, grid, smem, act_in, x32, st) : tok_launch_n<2, 4>(p, grid, smem, act_in, x32, st);
  switch (ntw) {
    case 1: return tok_launch_n<1, 1>(p, grid, smem, act_in, x32, st);
    case 2: return tok_launch_n<2, 1>(p, grid, smem, act_in, x32, st);
    default: return tok_launch_n<4, 1>(p, grid, smem, act_in, x32, st);
  }
}

extern "C" size_t cbim_token_linear_wgrad_workspace(int64_t rows, int Cin, int Cout) {
  cbim_conv_desc d;
  tok_desc(&d, rows, Cin, Cout);
  int strips, rps, cob, cib;
  pwg_cfg(&d, &strips, &rps, &cob, &cib);
  return (size_t)strips * cob * 64 * cib * 64 * sizeof(float);
}

extern "C" int cbim_token_linear_wgrad(const void* x, int x_dtype, int64_t x_stride, int act_in, const void* dy, int dy_dtype,
                                       int64_t dy_stride, float* dw, void* workspace, size_t ws_bytes, int64_t rows, int Cin,
                                       int Cout, void* stream) {
  CBIM_CHECK(x && dy && dw && rows >= 1 && rows < ((int64_t)1 << 31), CBIM_EINVAL, "token linear wgrad: null operand / bad row count");
  CBIM_CHECK(Cin % 8 == 0 && Cout % 8 == 0 && x_stride % 8 == 0 && dy_stride % 8 == 0, CBIM_EUNSUPPORTED,
             "token linear wgrad: %d -> %d features, strides %lld / %lld", Cin, Cout, (long long)x_stride, (long long)dy_stride);
  CBIM_CHECK(!(x_dtype == CBIM_F32 && act_in != CBIM_ACT_NONE), CBIM_EUNSUPPORTED, "token linear wgrad: activation on load takes bf16 rows");
  const size_t need = cbim_token_linear_wgrad_workspace(rows, Cin, Cout);
  CBIM_CHECK(workspace && ws_bytes >= need, CBIM_EWORKSPACE, "token linear wgrad workspace %zu < %zu", ws_bytes, need);
  cbim_conv_desc d;
  tok_desc(&d, rows, Cin, Cout);
  PwgParams p;
  p.x = x; p.x_stride = x_stride; p.in_stats = nullptr; p.dy = dy; p.dy_stride = dy_stride; p.ws = (float*)workspace;
  p.N = 1; p.S = rows; p.Cin = Cin; p.Cout = Cout; p.act = act_in; p.act_raw = act_in != CBIM_ACT_NONE;
  int cob;
  pwg_cfg(&d, &p.strips, &p.rows_per_strip, &cob, &p.ci_blocks);
  p.Cout_pad = cob * 64; p.Cin_pad = p.ci_blocks * 64;
  dim3 grid((unsigned)p.strips, (unsigned)(cob * p.ci_blocks));
  hipStream_t st = (hipStream_t)stream;
  const bool x32 = x_dtype == CBIM_F32, d32 = dy_dtype == CBIM_F32;
  if (act_in != CBIM_ACT_NONE) {
    if (d32) CBIM_LAUNCH((k_pw_wgrad<-1, false, true>), grid, dim3(PW_NT), 0, st, p);
    else CBIM_LAUNCH((k_pw_wgrad<-1, false, false>), grid, dim3(PW_NT), 0, st, p);
  } else if (x32) {
    if (d32) CBIM_LAUNCH((k_pw_wgrad<CBIM_ACT_NONE, true, true>), grid, dim3(PW_NT), 0, st, p);
    else CBIM_LAUNCH((k_pw_wgrad<CBIM_ACT_NONE, true, false>), grid, dim3(PW_NT), 0, st, p);
  } else {
    if (d32) CBIM_LAUNCH((k_pw_wgrad<CBIM_ACT_NONE, false, true>), grid, dim3(PW_NT), 0, st, p);
    else CBIM_LAUNCH((k_pw_wgrad<CBIM_ACT_NONE, false, false>), grid, dim3(PW_NT), 0, st, p);
  }
  if (CBIM_LAST_LAUNCH() != hipSuccess) return CBIM_ELAUNCH;
  return cbim_wgrad_reduce_launch((const float*)workspace, dw, p.strips, 1, Cout, Cin, p.Cout_pad, p.Cin_pad, stream);
}

CBIM_DEFINE_WARM(pw)
