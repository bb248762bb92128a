// conv_igemm.hip — 3-D convolution (forward and dgrad) as an implicit GEMM on the CDNA4 matrix
// cores, written for gfx950 (wave64, v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).
//
//   GEMM view:  M = output voxels, N = Cout, K = taps x Cin.
//   Workgroup:  256 threads = 4 waves; output tile = tD x tH x 8 voxels (BM = 128*MT) x BN = 32*NTL
//               couts; wave w owns m-tiles [w*MT, w*MT+MT) (32 voxels = 4 rows x 8 in W) and all NTL
//               n-tiles, i.e. MT*NTL accumulators of 16 VGPRs.  LDS <= 75 KiB -> 2 workgroups per CU
//               (2 waves per SIMD): one workgroup's staging overlaps the other's MFMA phase.
//   K loop:     Cin in chunks of 64 BYTES per voxel (32 bf16 / 16 f32 channels).  Per chunk the
//               (tD+kD-1)(tH+kH-1)(8+kW-1) input halo is staged ONCE into LDS — InstanceNorm +
//               activation of the producer are applied on that load (pre-activation ConvNormAct,
//               /root/reference/model/dim3/conv_layers.py:48-49; literal zeros for the padding) —
//               and re-used by all taps; the weights are staged one kd-plane (kH*kW taps) at a time
//               in MFMA B-fragment order (a contiguous block copy from the pre-packed buffer).
//   Inner loop: one tap per step, fragments of the NEXT tap are fetched (ds_read_b128) into a second
//               register set before the MFMAs of the current tap issue (software pipelining by hand:
//               with 1-2 waves per SIMD nothing else hides the LDS latency).
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

namespace cbim {

static constexpr int NT = 256;
static constexpr int RB = 64;       // bytes per halo row (one Cin chunk)
static constexpr int SLOTS = RB / 16;
static constexpr int KG = RB / 32;  // k-groups (two 16-B slots each) per chunk

struct IgemmParams {
  const void* x; int64_t x_stride;
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
  if (ACT == CBIM_ACT_NONE) return x;
  return act_fwd(x, rt);
}
template <int ACT> __device__ __forceinline__ float actg(float x, int rt) {
  if (ACT == CBIM_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (ACT == CBIM_ACT_NONE) return 1.f;
  return act_grad(x, rt);
}

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

template <int MT, int NTL> struct Frags { u32x4 a[KG][MT]; u32x4 b[KG][NTL]; };

template <typename T, int MT, int NTL, int ACT>
__global__ void __launch_bounds__(NT, 2) k_conv_igemm(IgemmParams p) {
  constexpr int CPC = Elem<T>::CPC;
  constexpr int KC = SLOTS * CPC;  // channels per chunk
  constexpr int BN = 32 * NTL;
  CBIM_DYN_SMEM(smem);
  const int hV = p.hD * p.hH * p.hW;
  const unsigned a_bytes = (unsigned)hV * RB;           // B region starts here
  const int ptaps = p.kH * p.kW;                        // taps per staged kd-plane
  const unsigned tap_bytes = KG * 2 * BN * 16;
  const unsigned plane_bytes = (unsigned)ptaps * tap_bytes;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_per_n = p.tiles_d * p.tiles_h * p.tiles_w;
  const int n = bid / tiles_per_n, t = bid % tiles_per_n;
  const int od0 = (t / (p.tiles_w * p.tiles_h)) * p.tD;
  const int oh0 = ((t / p.tiles_w) % p.tiles_h) * p.tH;
  const int ow0 = (t % p.tiles_w) * 8;
  const int id0 = od0 - p.pD, ih0 = oh0 - p.pH, iw0 = ow0 - p.pW;
  const int nb = blockIdx.y, co0 = nb * BN;

  unsigned a_off[MT];   // LDS byte offset of the lane's voxel row at tap (0,0,0)
  int thr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = (wave * MT + mt) * 32 + li;
    int tw = m & 7, th = (m >> 3) & (p.tH - 1), td = m >> (3 + p.lgH);
    a_off[mt] = (unsigned)((td * p.hH + th) * p.hW + tw) * RB;
    thr[mt] = th;
  }
  const unsigned b_lane = a_bytes + (unsigned)(half * BN + li) * 16;
  f32x16 acc[MT][NTL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int my_slot = tid & (SLOTS - 1);  // NT % SLOTS == 0: a thread always stages the same slot
  const size_t nbase_in = (size_t)n * p.Di * p.Hi * p.Wi;
  const int hHW = p.hH * p.hW;

  // fragment fetch of one tap (kh, kw inside the staged plane; kd folded into a_plane)
  auto fetch = [&](Frags<MT, NTL>& f, int tp, int kh, int kw, unsigned a_plane) {
    const unsigned toff = a_plane + (unsigned)(kh * p.hW + kw) * RB;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
        f.b[kg][nt] = *(const u32x4*)(smem + b_lane + (unsigned)tp * tap_bytes + (unsigned)(kg * 2 * BN + nt * 32) * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        unsigned slot = (unsigned)((2 * kg + half) ^ ((thr[mt] + kh) & (SLOTS - 1)));
        f.a[kg][mt] = *(const u32x4*)(smem + a_off[mt] + toff + (slot << 4));
      }
    }
  };
  auto mma = [&](const Frags<MT, NTL>& f) {
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) Mma<T>::run(f.a[kg][mt], f.b[kg][nt], acc[mt][nt]);
  };

  for (int q = 0; q < p.n_chunks; ++q) {
    __syncthreads();  // previous chunk's fragments are consumed
    // ---- stage the input halo (with fused InstanceNorm + activation of the producer) --------------
    // loads are issued in batches (UA independent 16-byte loads in flight per thread) before any is
    // consumed; (hd,hh,hw) decode by multiply-shift, no integer division.
    {
      const int c0 = q * KC + my_slot * CPC;
      const bool c_ok = c0 < p.Cin;
      float mean[CPC], rstd[CPC];
      if (p.in_stats && c_ok) {
#pragma unroll
        for (int j = 0; j < CPC; ++j) {
          mean[j] = p.in_stats[((size_t)n * p.Cin + c0 + j) * 2];
          rstd[j] = p.in_stats[((size_t)n * p.Cin + c0 + j) * 2 + 1];
        }
      }
      constexpr int UA = 5;
      const int a_items = hV * SLOTS;
      for (int base = tid; base < a_items; base += NT * UA) {
        u32x4 v[UA];
        int dst[UA];
        bool ld[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
          int item = base + u * NT;
          unsigned hv = (unsigned)item / SLOTS;
          unsigned hd = (hv * p.mHW) >> 20;
          unsigned r2 = hv - hd * hHW;
          unsigned hh = (r2 * p.mW) >> 20;
          unsigned hw = r2 - hh * p.hW;
          int id = id0 + (int)hd, ih = ih0 + (int)hh, iw = iw0 + (int)hw;
          dst[u] = item < a_items ? (int)(hv * RB + ((my_slot ^ (hh & (SLOTS - 1))) << 4)) : -1;
          ld[u] = item < a_items && c_ok && id >= 0 && id < p.Di && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          if (ld[u]) {
            size_t row = nbase_in + ((size_t)id * p.Hi + ih) * p.Wi + iw;
            v[u] = ld_chunk<T>(p.x, row * p.x_stride + c0);
          }
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) {
          if (dst[u] >= 0) {
            u32x4 w = v[u];
            if (p.in_stats && ld[u]) {
              float f[CPC];
              Elem<T>::unpack(w, f);
#pragma unroll
              for (int j = 0; j < CPC; ++j) f[j] = actf<ACT>((f[j] - mean[j]) * rstd[j], p.act);
              w = Elem<T>::pack(f);
            }
            *(u32x4*)(smem + dst[u]) = w;
          }
        }
      }
    }
    const unsigned char* wq = (const unsigned char*)p.w + ((size_t)nb * p.n_chunks + q) * ((size_t)p.kD * plane_bytes);
    for (int kd = 0; kd < p.kD; ++kd) {
      if (kd > 0) __syncthreads();  // the previous plane's weights are consumed
      // ---- stage this kd-plane's weights (already in fragment order): contiguous block copy ------------
      {
        constexpr int UB = 5;
        const unsigned char* wsrc = wq + (size_t)kd * plane_bytes;
        for (unsigned ob = (unsigned)tid * 16; ob < plane_bytes; ob += NT * 16 * UB) {
          u32x4 v[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            unsigned o = ob + (unsigned)u * NT * 16;
            if (o < plane_bytes) v[u] = *(const u32x4*)(wsrc + o);
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            unsigned o = ob + (unsigned)u * NT * 16;
            if (o < plane_bytes) *(u32x4*)(smem + a_bytes + o) = v[u];
          }
        }
      }
      __syncthreads();
      // ---- the plane's taps, two per trip through statically named register sets ----------------------
      const unsigned a_plane = (unsigned)(kd * hHW) * RB;
      Frags<MT, NTL> f0, f1;
      int kh = 0, kw = 0;   // (kh, kw) of the NEXT tap to fetch
      fetch(f0, 0, 0, 0, a_plane);
      if (++kw == p.kW) { kw = 0; ++kh; }
      for (int tp = 0; tp < ptaps; tp += 2) {
        if (tp + 1 < ptaps) {
          fetch(f1, tp + 1, kh, kw, a_plane);
          if (++kw == p.kW) { kw = 0; ++kh; }
        }
        mma(f0);
        if (tp + 2 < ptaps) {
          fetch(f0, tp + 2, kh, kw, a_plane);
          if (++kw == p.kW) { kw = 0; ++kh; }
        }
        if (tp + 1 < ptaps) mma(f1);
      }
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  __syncthreads();
  float* red = (float*)smem;  // [4 waves * MT][BN][3]
  const size_t nbase_out = (size_t)n * p.Do * p.Ho * p.Wo;
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int co = co0 + nt * 32 + li;
    const bool co_ok = co < p.Cout;
    float mmean = 0.f, mrstd = 1.f;
    if (p.mx && co_ok) {
      mmean = p.m_stats[((size_t)n * p.Cout + co) * 2];
      mrstd = p.m_stats[((size_t)n * p.Cout + co) * 2 + 1];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      // forward: shifted sums -> (count, mean, M2); dgrad: plain (sum g, sum g*xh)
      float s0 = 0.f, s1 = 0.f, cnt = 0.f, shift = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = (wave * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        int tw = m & 7, th = (m >> 3) & (p.tH - 1), td = m >> (3 + p.lgH);
        int od = od0 + td, oh = oh0 + th, ow = ow0 + tw;
        if (co_ok && od < p.Do && oh < p.Ho && ow < p.Wo) {
          size_t row = nbase_out + ((size_t)od * p.Ho + oh) * p.Wo + ow;
          float v = acc[mt][nt][r];
          if (p.res) v += Elem<T>::load1(p.res, row * p.res_stride + co);
          if (p.mx) {
            float xh = (Elem<T>::load1(p.mx, row * p.mx_stride + co) - mmean) * mrstd;
            v *= actg<ACT>(xh, p.act);
            s0 += v;
            s1 += v * xh;
          } else {
            if (cnt == 0.f) shift = v;
            float d = v - shift;
            s0 += d;
            s1 += d * d;
          }
          cnt += 1.f;
          Elem<T>::store1(p.y, row * p.y_stride + co, v);
        }
      }
      if (p.partials) {
        Moments a;
        if (p.mx) { a.n = 0.f; a.mean = s0; a.m2 = s1; }
        else a = moments_from_shifted(cnt, shift, s0, s1);
        Moments b;
        b.n = __shfl_xor(a.n, 32, 64);
        b.mean = __shfl_xor(a.mean, 32, 64);
        b.m2 = __shfl_xor(a.m2, 32, 64);
        if (half == 0) {
          if (p.mx) { a.mean += b.mean; a.m2 += b.m2; }
          else a = moments_merge(a, b);
          float* rr = red + (((wave * MT + mt) * BN) + nt * 32 + li) * 3;
          rr[0] = a.n; rr[1] = a.mean; rr[2] = a.m2;
        }
      }
    }
  }
  if (p.partials) {
    __syncthreads();
    if (tid < BN && co0 + tid < p.Cout) {
      Moments a = {0.f, 0.f, 0.f};
      for (int g = 0; g < 4 * MT; ++g) {
        const float* rr = red + (g * BN + tid) * 3;
        if (p.mx) { a.mean += rr[1]; a.m2 += rr[2]; }
        else { Moments b = {rr[0], rr[1], rr[2]}; a = moments_merge(a, b); }
      }
      size_t o = (((size_t)n * tiles_per_n + t) * p.Cout + co0 + tid) * 3;
      p.partials[o] = a.n;
      p.partials[o + 1] = a.mean;
      p.partials[o + 2] = a.m2;
    }
  }
}

// ---- weight re-layout into B-fragment order ------------------------------------------------------------
// packed[nb][chunk][tap][kg][half][BN][CPC];  channel = chunk*KC + (2kg+half)*CPC + j, cout = nb*BN + nn.
// mode 0: K = Cin, N = Cout, value w[cout][cin][tap]; mode 1 (dgrad): K = Cout_fwd, N = Cin_fwd,
// value w[k_ch][n_ch][taps-1-tap]  (w is always the forward [Cout][Cin][taps] tensor).
template <typename T>
__global__ void __launch_bounds__(NT) k_pack_weights(const float* __restrict__ w, void* __restrict__ packed,
                                                     int Cout_f, int Cin_f, int taps, int mode, int BN,
                                                     int n_chunks, int64_t total) {
  constexpr int CPC = Elem<T>::CPC;
  constexpr int KC = SLOTS * CPC;
  const int Kdim = mode == 0 ? Cin_f : Cout_f;
  const int Ndim = mode == 0 ? Cout_f : Cin_f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int64_t r = i;
    int j = (int)(r % CPC); r /= CPC;
    int nn = (int)(r % BN); r /= BN;
    int half = (int)(r % 2); r /= 2;
    int kg = (int)(r % KG); r /= KG;
    int tap = (int)(r % taps); r /= taps;
    int q = (int)(r % n_chunks);
    int nb = (int)(r / n_chunks);
    int kc = q * KC + (2 * kg + half) * CPC + j;
    int nc = nb * BN + nn;
    float v = 0.f;
    if (kc < Kdim && nc < Ndim) {
      if (mode == 0) v = w[((size_t)nc * Cin_f + kc) * taps + tap];
      else v = w[((size_t)kc * Cin_f + nc) * taps + (taps - 1 - tap)];
    }
    Elem<T>::store1(packed, (size_t)i, v);
  }
}

struct TileCfg { int MT, NTL, tD, tH, lgH; };

static TileCfg pick_cfg(const cbim_conv_desc* d) {
  TileCfg c;
  c.NTL = d->Cout <= 32 ? 1 : 2;
  int64_t S = (int64_t)d->Do * d->Ho * d->Wo;
  c.MT = (S >= 32768 && d->Do >= 4 && d->Ho >= 8) ? 2 : 1;
  if (c.MT == 2) { c.tD = 4; c.tH = 8; c.lgH = 3; }
  else if (d->Do <= 2) { c.tD = 2; c.tH = 8; c.lgH = 3; }
  else { c.tD = 4; c.tH = 4; c.lgH = 2; }
  return c;
}

static int elem_size(int dtype) { return dtype == CBIM_BF16 ? 2 : 4; }
static int kc_of(int dtype) { return RB / elem_size(dtype); }

}  // namespace cbim

using namespace cbim;

static int validate(const cbim_conv_desc* d) {
  CBIM_CHECK(d != nullptr, CBIM_EINVAL, "null conv descriptor");
  CBIM_CHECK(d->dtype == CBIM_F32 || d->dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", d->dtype);
  int cpc = d->dtype == CBIM_BF16 ? 8 : 4;
  CBIM_CHECK(d->Cin > 0 && d->Cin % cpc == 0, CBIM_EUNSUPPORTED, "conv Cin %d is not a multiple of %d", d->Cin, cpc);
  CBIM_CHECK(d->Cout > 0, CBIM_EINVAL, "conv Cout %d", d->Cout);
  CBIM_CHECK(d->kD >= 1 && d->kH >= 1 && d->kW >= 1 && d->kD * d->kH * d->kW <= 64, CBIM_EUNSUPPORTED, "kernel extent unsupported");
  CBIM_CHECK(d->N >= 1 && d->Do >= 1 && d->Ho >= 1 && d->Wo >= 1, CBIM_EINVAL, "empty conv output");
  return 0;
}

extern "C" size_t cbim_conv3d_packed_bytes(const cbim_conv_desc* d, int mode) {
  if (!d) return 0;
  int Kdim = mode == 0 ? d->Cin : d->Cout, Ndim = mode == 0 ? d->Cout : d->Cin;
  int NTL = Ndim <= 32 ? 1 : 2, BN = 32 * NTL;
  int n_nblk = (Ndim + BN - 1) / BN;
  int KC = kc_of(d->dtype);
  int n_chunks = (Kdim + KC - 1) / KC;
  int taps = d->kD * d->kH * d->kW;
  return (size_t)n_nblk * n_chunks * taps * KG * 2 * BN * 16;
}

extern "C" int cbim_conv3d_pack_weights(const cbim_conv_desc* d, int mode, const float* w, void* packed,
                                        void* stream) {
  if (int e = validate(d)) return e;
  CBIM_CHECK(mode == 0 || mode == 1, CBIM_EINVAL, "bad pack mode");
  int Kdim = mode == 0 ? d->Cin : d->Cout, Ndim = mode == 0 ? d->Cout : d->Cin;
  int NTL = Ndim <= 32 ? 1 : 2, BN = 32 * NTL;
  int KC = kc_of(d->dtype);
  int n_chunks = (Kdim + KC - 1) / KC;
  int taps = d->kD * d->kH * d->kW;
  int64_t total = (int64_t)(cbim_conv3d_packed_bytes(d, mode) / elem_size(d->dtype));
  int64_t blocks = (total + NT - 1) / NT;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CBIM_BF16)
    CBIM_LAUNCH((k_pack_weights<bf16_tag>), dim3((unsigned)blocks), dim3(NT), 0, st, w, packed, d->Cout, d->Cin,
                taps, mode, BN, n_chunks, total);
  else
    CBIM_LAUNCH((k_pack_weights<float>), dim3((unsigned)blocks), dim3(NT), 0, st, w, packed, d->Cout, d->Cin, taps,
                mode, BN, n_chunks, total);
  return hipGetLastError() == hipSuccess ? CBIM_OK : CBIM_ELAUNCH;
}

extern "C" int cbim_conv3d_tile_config(const cbim_conv_desc* d, int out[4]) {
  CBIM_CHECK(d && out, CBIM_EINVAL, "null argument");
  TileCfg c = pick_cfg(d);
  out[0] = c.MT; out[1] = c.NTL; out[2] = c.tD; out[3] = c.tH;
  return CBIM_OK;
}

extern "C" int cbim_conv3d_num_tiles(const cbim_conv_desc* d) {
  if (!d) return 0;
  TileCfg c = pick_cfg(d);
  return ((d->Do + c.tD - 1) / c.tD) * ((d->Ho + c.tH - 1) / c.tH) * ((d->Wo + 7) / 8);
}

template <typename T, int MT, int NTL, int ACT>
static int launch_igemm(const IgemmParams& p, dim3 grid, size_t smem, hipStream_t st) {
#ifndef CBIM_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv_igemm<T, MT, NTL, ACT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
#endif
  CBIM_LAUNCH((k_conv_igemm<T, MT, NTL, ACT>), grid, dim3(NT), smem, st, p);
  hipError_t e = hipGetLastError();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "conv igemm launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

template <typename T, int ACT>
static int dispatch_tiles(const TileCfg& c, const IgemmParams& p, dim3 grid, size_t smem, hipStream_t st) {
  if (c.MT == 2 && c.NTL == 1) return launch_igemm<T, 2, 1, ACT>(p, grid, smem, st);
  if (c.MT == 2 && c.NTL == 2) return launch_igemm<T, 2, 2, ACT>(p, grid, smem, st);
  if (c.MT == 1 && c.NTL == 1) return launch_igemm<T, 1, 1, ACT>(p, grid, smem, st);
  return launch_igemm<T, 1, 2, ACT>(p, grid, smem, st);
}

extern "C" int cbim_conv3d_igemm(const cbim_conv_desc* d, const void* x, int64_t x_stride,
                                 const float* in_stats, const void* w_packed, const void* res,
                                 int64_t res_stride, const void* mask_x, int64_t mask_stride,
                                 const float* mask_stats, void* y, int64_t y_stride, float* partials,
                                 void* stream) {
  if (int e = validate(d)) return e;
  CBIM_CHECK(x && w_packed && y, CBIM_EINVAL, "null tensor");
  CBIM_CHECK(!mask_x || mask_stats, CBIM_EINVAL, "mask_x needs mask_stats");
  TileCfg c = pick_cfg(d);
  IgemmParams p;
  p.x = x; p.x_stride = x_stride; p.in_stats = in_stats; p.w = w_packed;
  p.res = res; p.res_stride = res_stride; p.mx = mask_x; p.mx_stride = mask_stride; p.m_stats = mask_stats;
  p.y = y; p.y_stride = y_stride; p.partials = partials;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin;
  p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.kD = d->kD; p.kH = d->kH; p.kW = d->kW; p.pD = d->pD; p.pH = d->pH; p.pW = d->pW; p.act = d->act;
  p.tD = c.tD; p.tH = c.tH; p.lgH = c.lgH;
  p.tiles_d = (d->Do + c.tD - 1) / c.tD; p.tiles_h = (d->Ho + c.tH - 1) / c.tH; p.tiles_w = (d->Wo + 7) / 8;
  p.hD = c.tD + d->kD - 1; p.hH = c.tH + d->kH - 1; p.hW = 8 + d->kW - 1;
  CBIM_CHECK(p.hD * p.hH * p.hW <= 2048, CBIM_EUNSUPPORTED, "halo too large");
  p.mHW = ((1u << 20) + (unsigned)(p.hH * p.hW) - 1) / (unsigned)(p.hH * p.hW);
  p.mW = ((1u << 20) + (unsigned)p.hW - 1) / (unsigned)p.hW;
  int KC = kc_of(d->dtype);
  p.n_chunks = (d->Cin + KC - 1) / KC;
  p.taps = d->kD * d->kH * d->kW;
  int BN = 32 * c.NTL;
  size_t smem = (size_t)p.hD * p.hH * p.hW * RB + (size_t)d->kH * d->kW * KG * 2 * BN * 16;
  size_t red = (size_t)4 * c.MT * BN * 3 * sizeof(float);
  if (smem < red) smem = red;
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "conv tile needs %zu B of LDS", smem);
  int64_t nblk = (int64_t)d->N * p.tiles_d * p.tiles_h * p.tiles_w;
  CBIM_CHECK(nblk < (1ll << 31), CBIM_EUNSUPPORTED, "too many tiles");
  dim3 grid((unsigned)nblk, (unsigned)((d->Cout + BN - 1) / BN));
  hipStream_t st = (hipStream_t)stream;
  const bool relu = d->act == CBIM_ACT_RELU;
  if (d->dtype == CBIM_BF16)
    return relu ? dispatch_tiles<bf16_tag, CBIM_ACT_RELU>(c, p, grid, smem, st)
                : dispatch_tiles<bf16_tag, -1>(c, p, grid, smem, st);
  return relu ? dispatch_tiles<float, CBIM_ACT_RELU>(c, p, grid, smem, st)
              : dispatch_tiles<float, -1>(c, p, grid, smem, st);
}
