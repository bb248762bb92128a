// swin_mfma.hip — SwinUNETR's shifted-window attention on the matrix cores (bf16, d_head 16, windows of <= 352
// tokens: every stage of the shipped SwinUNETR configurations, /root/reference/model/dim3/swin_unetr.py:467-490).
//
// Same contract, same index arithmetic (pad, roll, partition, relative-position bias, region mask, reverse, crop) and
// same outputs as the vector-ALU kernels of swin_kernels.hip; the two matrix products of the forward and the six of the
// backward run as v_mfma_f32_32x32x16_bf16.
//
// Orientation.  Scores are produced TRANSPOSED, S^T = K . Q^T, so that an accumulator holds, per lane, 16 keys of ONE
// query (lane & 31 = query, register r + lane-half = key (r&3) + 8(r>>2) + 4*half of the 32-key block): bias, mask and the
// whole softmax are lane-local (one cross-half exchange per block for the running maximum).  A product that contracts
// over the index held in registers takes the accumulator registers 8t..8t+7 directly as its MFMA B operand: k-slot j of
// lane-half h is key kappa(t, j, h) = (j&3) + 8(2t + (j>>2)) + 4h, and the other operand (V^T, K^T ...) is staged in LDS
// in exactly that slot order — no shuffle, no LDS round trip between the GEMMs.  d_head is 16, the MFMA M is 32: the
// upper 16 rows of those products are not used.
//
// One workgroup = one (window, head); a wave owns query blocks of 32.  K, Q (pre-scaled: d_head^-0.5 = 1/4 is exact in
// bf16), V^T, the bias-table column of the head (times log2 e: the softmax runs on v_exp_f32 = 2^x), and per-token
// bias coordinates / region labels live in LDS, the per-token words twice: in token order for the index a lane owns
// and in accumulator-register order ("slot order": 16 consecutive words per block and lane-half) for the index the
// registers run over, so that a block's 16 words arrive as four ds_read_b128.
//
// The kernels are bound by the vector ALU (a wave64 instruction occupies its SIMD for 4 cycles; the MFMAs of a 32x32
// block take 96-192 cycles), so the per-element work is kept minimal: one subtraction for the table address, one FMA
// for score * log2e + bias, subtract, v_exp_f32; the region mask only in windows that hold more than one region
// (the last window along a shifted dimension), the key-exists test only in a window's last block, and the softmax
// denominator as row 16 of the P.V product (a row of ones appended to V^T).
#include "swin_common.h"
#include <stdlib.h>
#include <type_traits>

namespace cbim {

static constexpr int WM_NT = 256;      // 4 waves
static constexpr int WM_NW = WM_NT / 64;
static constexpr int WM_PAD = 352;     // 11 blocks of 32 tokens
static constexpr int WM_DH = 16;
static constexpr float WM_L2E = 1.4426950408889634f, WM_LN2 = 0.6931471805599453f;
static constexpr unsigned WM_A16 = WM_PAD * WM_DH * 2;   // one [352][16] bf16 array
static constexpr unsigned WM_W = WM_PAD * 4;             // one [352] word array

// Table gathers and histogram atomics take ABSOLUTE 32-bit LDS addresses (the workgroup's LDS base folded into the
// per-token words at staging): one v_sub per access, no generic-pointer arithmetic in the inner loops.
#ifdef CBIM_EMU
#define WM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
__device__ __forceinline__ float wm_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float wm_log2(float x) { return log2f(x); }
__device__ __forceinline__ unsigned wm_lds_addr(const unsigned char* smem, unsigned off) { (void)smem; return off; }
__device__ __forceinline__ float wm_lds_f32(const unsigned char* smem, unsigned a) { return *(const float*)(smem + a); }
__device__ __forceinline__ void wm_lds_add64(unsigned char* smem, unsigned a, unsigned long long v) {
  atomicAdd((unsigned long long*)(smem + a), v);
}
__device__ __forceinline__ float wm_fract(float x) { return x - floorf(x); }
#else
#define WM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
__device__ __forceinline__ float wm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float wm_log2(float x) { return __builtin_amdgcn_logf(x); }
typedef __attribute__((address_space(3))) float wm_lds_float;
typedef __attribute__((address_space(3))) unsigned char wm_lds_u8;
__device__ __forceinline__ unsigned wm_lds_addr(const unsigned char* smem, unsigned off) {
  return (unsigned)(uintptr_t)(const wm_lds_u8*)smem + off;
}
__device__ __forceinline__ float wm_lds_f32(const unsigned char*, unsigned a) { return *(const wm_lds_float*)(uintptr_t)a; }
typedef __attribute__((address_space(3))) unsigned long long wm_lds_u64;
__device__ __forceinline__ void wm_lds_add64(unsigned char*, unsigned a, unsigned long long v) {
  __hip_atomic_fetch_add((wm_lds_u64*)(uintptr_t)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u64
}
__device__ __forceinline__ float wm_fract(float x) { return __builtin_amdgcn_fractf(x); }
#endif

// byte offset of element (token tok, channel d) inside a k-slot-ordered transposed array [kb][t][h][d][j]
__device__ __forceinline__ unsigned wm_slot_off(int tok, int d) {
  const int kb = tok >> 5, kk = tok & 31;
  const int t = kk >> 4, h = (kk >> 2) & 1, j = (kk & 3) + 4 * ((kk >> 3) & 1);
  return (unsigned)((((kb * 2 + t) * 2 + h) * WM_DH + d) * 8 + j) * 2u;
}
// word index of token tok in accumulator-register order: block, lane-half, register r with
// token-in-block = (r & 3) + 8 (r >> 2) + 4 half
__device__ __forceinline__ int wm_slot_word(int tok) {
  const int kk = tok & 31;
  return (tok & ~31) + ((kk >> 2) & 1) * 16 + (kk & 3) + 4 * (kk >> 3);
}
// a window holds more than one mask region iff it is the last one along a shifted dimension (win_token's labels)
__device__ __forceinline__ bool wm_multi_region(const WinGeom& g, int win) {
  const int ww = win % g.nw2, wh = (win / g.nw2) % g.nw1, wd = (win / (g.nw2 * g.nw1)) % g.nw0;
  return g.masked && ((g.s0 > 0 && wd == g.nw0 - 1) || (g.s1 > 0 && wh == g.nw1 - 1) || (g.s2 > 0 && ww == g.nw2 - 1));
}
__device__ __forceinline__ float wm_xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }

// 16 words of (block b, lane-half) of a slot-ordered word array
template <typename T>
__device__ __forceinline__ void wm_load16(const unsigned char* smem, unsigned base, int b, int half, T (&v)[16]) {
  static_assert(sizeof(T) == 4, "word arrays");
  const unsigned char* p = smem + base + (unsigned)(b * 32 + half * 16) * 4u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32x4 x = *(const u32x4*)(p + j * 16);
    // (scalar copies first: __builtin_bit_cast applied directly to a vector ELEMENT reads element 0 with this clang)
    const unsigned x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
    v[4 * j] = __builtin_bit_cast(T, x0); v[4 * j + 1] = __builtin_bit_cast(T, x1);
    v[4 * j + 2] = __builtin_bit_cast(T, x2); v[4 * j + 3] = __builtin_bit_cast(T, x3);
  }
}
__device__ __forceinline__ u32x4 wm_pack8(const f32x16& x, int t) {
  return u32x4{pk_bf16(x[8 * t], x[8 * t + 1]), pk_bf16(x[8 * t + 2], x[8 * t + 3]),
               pk_bf16(x[8 * t + 4], x[8 * t + 5]), pk_bf16(x[8 * t + 6], x[8 * t + 7])};
}

// LDS layout of the forward kernel (bytes)
struct WmFwdLds {
  static constexpr unsigned Q = 0;                      // [352][16] bf16, scaled
  static constexpr unsigned K = Q + WM_A16;             // [352][16] bf16
  static constexpr unsigned VT = K + WM_A16;            // [11][2][2][16][8] bf16: V^T in k-slot order
  static constexpr unsigned ONES = VT + WM_A16;         // [16][8] bf16: rows 16..31 of V^T (row 16 = ones, rest 0)
  static constexpr unsigned NBQ = ONES + 256;           // token order: LDS address of table entry (B_q + off0)
  static constexpr unsigned NLAB = NBQ + WM_W;          // token order: region label
  static constexpr unsigned SBK = NLAB + WM_W;          // slot order: 4 * B_k
  static constexpr unsigned SLAB = SBK + WM_W;          // slot order: region label
  static constexpr unsigned TBL = SLAB + WM_W;          // [TS] float, times log2 e
};
static size_t wm_fwd_smem(int TS) { return (size_t)WmFwdLds::TBL + (size_t)TS * 4; }

template <bool MASKED>
__device__ __forceinline__ void wm_fwd_blocks(const WinGeom& g, const unsigned char* smem, int n, int win, int hd, int wave,
                                              int li, int half, void* __restrict__ out, float* __restrict__ lse_out) {
  const int C = g.C;
  const int nkb = (n + 31) >> 5;
  const unsigned vt_lane = li < 16 ? WmFwdLds::VT + (unsigned)(half * WM_DH + li) * 16u : WmFwdLds::ONES + (unsigned)(li - 16) * 16u;
  const unsigned vt_inc = li < 16 ? 2 * WM_DH * 16u : 0u;          // per (block, t)
  for (int qb = wave; qb < nkb; qb += WM_NW) {
    const int q = qb * 32 + li;                         // this lane's query
    const u32x4 qf = *(const u32x4*)(smem + WmFwdLds::Q + q * 32 + half * 16);
    const int bq = *(const int*)(smem + WmFwdLds::NBQ + q * 4), lq = *(const int*)(smem + WmFwdLds::NLAB + q * 4);
    float m = -INFINITY;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    unsigned vtp = vt_lane;
    auto block = [&](int kb, auto tailc) {
      const u32x4 kf = *(const u32x4*)(smem + WmFwdLds::K + (kb * 32 + li) * 32 + half * 16);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf), s, 0, 0, 0);
      int bk[16];
      wm_load16(smem, WmFwdLds::SBK, kb, half, bk);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], WM_L2E, wm_lds_f32(smem, (unsigned)(bq - bk[r])));
      if (MASKED) {
        int lk[16];
        wm_load16(smem, WmFwdLds::SLAB, kb, half, lk);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = lk[r] != lq ? s[r] - 100.f * WM_L2E : s[r];
      }
      if (decltype(tailc)::value) {                     // the window's last block: keys past the window do not exist
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < n ? s[r] : -INFINITY;
      }
      float bm = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) bm = fmaxf(bm, s[r]);
      const float mn = fmaxf(m, wm_xhalf_max(bm));      // same for both halves of the query
      const float corr = wm_exp2(m - mn);               // 0 on the first block (m = -inf; key 0 always exists)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = wm_exp2(s[r] - mn);
      m = mn;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x4 vf = *(const u32x4*)(smem + vtp);
        vtp += vt_inc;
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, wm_pack8(s, t)), o, 0, 0, 0);
      }
    };
    const int nfull = n >> 5;                           // blocks whose 32 keys all exist
    for (int kb = 0; kb < nfull; ++kb) block(kb, std::false_type{});
    if (n & 31) block(nfull, std::true_type{});
    // row 16 of O^T (register 8 of lane-half 0) = sum of the (bf16) probabilities: the softmax denominator
    const float lx = __shfl_xor(o[8], 32, 64);
    const float lt = half ? lx : o[8];
    if (q < n) {
      if (half == 0) lse_out[((size_t)win * g.heads + hd) * WMAX + q] = (m + wm_log2(lt)) * WM_LN2;
      int64_t row;
      int lb, bc;
      win_token(g, win, q, row, lb, bc);
      if (row >= 0) {
        const float inv = 1.f / lt;
        // rows (d) of O^T held by this lane: registers 0-3 -> d = 4*half + 0..3, registers 4-7 -> d = 8 + 4*half + 0..3
        bf16_t* dst = (bf16_t*)out + (size_t)row * C + hd * WM_DH;
        u32x2 a, b;
        a.x = pk_bf16(o[0] * inv, o[1] * inv); a.y = pk_bf16(o[2] * inv, o[3] * inv);
        b.x = pk_bf16(o[4] * inv, o[5] * inv); b.y = pk_bf16(o[6] * inv, o[7] * inv);
        *(u32x2*)(dst + 4 * half) = a;
        *(u32x2*)(dst + 8 + 4 * half) = b;
      }
    }
  }
}

__global__ void __launch_bounds__(WM_NT, 2) k_winattn_fwd_mfma(WinGeom g, const void* __restrict__ qkv,
                                                               const float* __restrict__ qkv_bias,
                                                               const float* __restrict__ table, void* __restrict__ out,
                                                               float* __restrict__ lse_out) {
  WM_DYN_SMEM(smem);
  const int n = g.w0 * g.w1 * g.w2, TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  const int off0 = ((g.tw0 - 1) * (2 * g.tw1 - 1) + (g.tw1 - 1)) * (2 * g.tw2 - 1) + (g.tw2 - 1);
  const int win = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const int C = g.C;
  float* tbl = (float*)(smem + WmFwdLds::TBL);
  for (int i = tid; i < TS; i += WM_NT) tbl[i] = table[(size_t)i * g.heads + hd] * WM_L2E;
  if (tid < 16 * 4) ((unsigned*)(smem + WmFwdLds::ONES))[tid] = tid < 4 ? 0x3F803F80u : 0u;
  const bool has_bias = qkv_bias != nullptr;
  // ---- stage the window: token -> K row, scaled Q row, V^T in slot order; tokens n..351 are zeros ------------------
  for (int tok = tid; tok < WM_PAD; tok += WM_NT) {
    float q[WM_DH], k[WM_DH], v[WM_DH];
    int64_t row = -1;
    int lb = -1, bc = 0;
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) { q[d] = 0.f; k[d] = 0.f; v[d] = 0.f; }
    if (tok < n) {
      win_token(g, win, tok, row, lb, bc);
      if (row >= 0) {
        const bf16_t* base = (const bf16_t*)qkv + (size_t)row * 3 * C + hd * WM_DH;
        const u32x4 q0 = *(const u32x4*)base, q1 = *(const u32x4*)(base + 8);
        const u32x4 k0 = *(const u32x4*)(base + C), k1 = *(const u32x4*)(base + C + 8);
        const u32x4 v0 = *(const u32x4*)(base + 2 * C), v1 = *(const u32x4*)(base + 2 * C + 8);
        Elem<bf16_tag>::unpack(q0, q); Elem<bf16_tag>::unpack(q1, q + 8);
        Elem<bf16_tag>::unpack(k0, k); Elem<bf16_tag>::unpack(k1, k + 8);
        Elem<bf16_tag>::unpack(v0, v); Elem<bf16_tag>::unpack(v1, v + 8);
      } else if (has_bias) {   // window padding: the token entered qkv as zeros, so q = k = v = qkv.bias (swin_unetr.py:561-566)
#pragma unroll
        for (int d = 0; d < WM_DH; ++d) {
          q[d] = qkv_bias[hd * WM_DH + d];
          k[d] = qkv_bias[C + hd * WM_DH + d];
          v[d] = qkv_bias[2 * C + hd * WM_DH + d];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) q[d] *= g.scale;
    *(u32x4*)(smem + WmFwdLds::Q + tok * 32) = Elem<bf16_tag>::pack(q);
    *(u32x4*)(smem + WmFwdLds::Q + tok * 32 + 16) = Elem<bf16_tag>::pack(q + 8);
    *(u32x4*)(smem + WmFwdLds::K + tok * 32) = Elem<bf16_tag>::pack(k);
    *(u32x4*)(smem + WmFwdLds::K + tok * 32 + 16) = Elem<bf16_tag>::pack(k + 8);
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) *(bf16_t*)(smem + WmFwdLds::VT + wm_slot_off(tok, d)) = (bf16_t)pk_bf16(v[d], 0.f);
    const int sw = wm_slot_word(tok);
    *(int*)(smem + WmFwdLds::NBQ + tok * 4) = (int)wm_lds_addr(smem, WmFwdLds::TBL + (bc + off0) * 4);
    *(int*)(smem + WmFwdLds::NLAB + tok * 4) = lb;
    *(int*)(smem + WmFwdLds::SBK + sw * 4) = bc * 4;
    *(int*)(smem + WmFwdLds::SLAB + sw * 4) = lb;
  }
  __syncthreads();
  if (wm_multi_region(g, win)) wm_fwd_blocks<true>(g, smem, n, win, hd, wave, li, half, out, lse_out);
  else wm_fwd_blocks<false>(g, smem, n, win, hd, wave, li, half, out, lse_out);
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// 512 threads.  Waves 0-3 ("pass A", scores transposed: lane = query, registers = keys) produce dQ and the bias-table
// histogram; waves 4-7 ("pass B", lane = key, registers = queries) produce dK and dV; the two passes run concurrently.
// P is recomputed from the saved log-sum-exp, dS = P o (dP - D) with D_q = dO_q . O_q.  Contractions over the index
// held in registers take the accumulator registers as the MFMA B operand (see the file header); the other operands
// (K^T for dQ; Qs^T, dO^T for dK, dV) are staged in LDS in k-slot order.
// Queries past the window carry lse = +inf (P = 0 without a test); keys past the window are tested in the last block of
// pass A only (in pass B they are lanes whose results are never stored).
// d(bias table): entry (B_q - B_k) receives dS[q][k], a scatter with collisions -> LDS atomics.  ds_add_f32 costs 193
// clocks per wave instruction on gfx950 (measured, tools/ubench/lds_atomic.hip: lane-serial), ds_add_u64 6.4: the
// histogram is kept in 64-bit FIXED POINT.  dS is bounded by M = max|dO_q| max|V_k| + max|D_q| (Cauchy-Schwarz, P <= 1),
// found while staging; with S = 2^(20 - ceil log2 M) every term |dS S| < 2^21 and the <= 343 terms of an entry stay
// below 2^31 in the integer word, the fraction word resolves M 2^-52: each term enters with its full fp32 mantissa and
// the sum itself is exact — integer addition is associative, so ONE histogram serves the four pass-A waves and the
// result is bit-reproducible whatever the order.  S is a power of two: dS S (not dS) also feeds the dQ product and 1/S
// is folded into the final scale, bit-identical.  The compiler keeps LDS loads and LDS atomics in program order (they
// may alias), so pass A is software-pipelined by hand: the next block's bias values are requested BEFORE this block's
// 16 atomics are issued.
static constexpr int WB_NT = 512;
static constexpr unsigned WB_HS = 8800;                         // bytes per table / histogram (>= 13^3 floats)
struct WmBwdLds {
  static constexpr unsigned Q = 0, K = Q + WM_A16, V = K + WM_A16, DO = V + WM_A16;          // row-major
  static constexpr unsigned KT = DO + WM_A16, QT = KT + WM_A16, DOT = QT + WM_A16;           // k-slot order
  // token order (the index a lane owns)
  static constexpr unsigned NBQ = DOT + WM_A16;                 // LDS address of table entry (B + off0)
  static constexpr unsigned NBK = NBQ + WM_W;                   // 4 * B
  static constexpr unsigned NLAB = NBK + WM_W;
  static constexpr unsigned NLSE = NLAB + WM_W;                 // lse * log2 e (+inf: token does not exist)
  static constexpr unsigned NDS = NLSE + WM_W;                  // D_q
  // slot order (the index the accumulator registers run over)
  static constexpr unsigned SBQ = NDS + WM_W, SBK = SBQ + WM_W, SLAB = SBK + WM_W, SLSE = SLAB + WM_W, SDS = SLSE + WM_W;
  static constexpr unsigned PADS = SDS + WM_W;                  // [4 waves][2][16] float: padded-key sums of dk, dv
  static constexpr unsigned MAXS = PADS + 4 * 2 * WM_DH * 4;    // float bits: max |dO_q|^2, max |V_k|^2, max |D_q|
  static constexpr unsigned TBL = MAXS + 16;                    // [<= 2200] float, times log2 e
  static constexpr unsigned HIST = TBL + WB_HS;                 // [<= 2200] int64: fixed-point d(bias table)
  static constexpr unsigned END = HIST + 2 * WB_HS;
};

template <bool MASKED>
__device__ __forceinline__ void wm_bwd_pass_a(const WinGeom& g, unsigned char* smem, int n, int win, int hd, int wave, int li,
                                              int half, void* __restrict__ dqkv, float S, int dbg) {
  const int C = g.C;
  const int nkb = (n + 31) >> 5;
  // table entry at LDS address a  ->  histogram entry at 2 a + hrel
  const int hrel = (int)wm_lds_addr(smem, WmBwdLds::HIST) - 2 * (int)wm_lds_addr(smem, WmBwdLds::TBL);
  auto row_frag = [&](unsigned base, int tok) -> u32x4 { return *(const u32x4*)(smem + base + tok * 32 + half * 16); };
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  for (int qb = wave; qb < nkb; qb += 4) {
    const int q = qb * 32 + li;
    const u32x4 qf = row_frag(WmBwdLds::Q, q), gf = row_frag(WmBwdLds::DO, q);
    const int bq = *(const int*)(smem + WmBwdLds::NBQ + q * 4), lq = *(const int*)(smem + WmBwdLds::NLAB + q * 4);
    const float Lq = *(const float*)(smem + WmBwdLds::NLSE + q * 4), DqS = *(const float*)(smem + WmBwdLds::NDS + q * 4) * S;
    f32x16 dq = zero;
    unsigned ta[16];                                    // LDS addresses of the block's 16 table entries
    float tb[16];                                       // ... and their values
    {
      int bk[16];
      wm_load16(smem, WmBwdLds::SBK, 0, half, bk);
#pragma unroll
      for (int r = 0; r < 16; ++r) { ta[r] = (unsigned)(bq - bk[r]); tb[r] = wm_lds_f32(smem, ta[r]); }
    }
    auto block = [&](int kb, auto tailc) {
      const u32x4 kf = row_frag(WmBwdLds::K, kb * 32 + li), vf = row_frag(WmBwdLds::V, kb * 32 + li);
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf), zero, 0, 0, 0);
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, gf), zero, 0, 0, 0);
      unsigned ah[16];                                  // histogram entries (8 bytes each) of this block
#pragma unroll
      for (int r = 0; r < 16; ++r) ah[r] = 2u * ta[r] + (unsigned)hrel;
      // next block's coordinates (the last block re-reads its own)
      int bk[16];
      wm_load16(smem, WmBwdLds::SBK, decltype(tailc)::value || kb + 1 >= nkb ? kb : kb + 1, half, bk);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], WM_L2E, tb[r]);
      if (MASKED) {
        int lk[16];
        wm_load16(smem, WmBwdLds::SLAB, kb, half, lk);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = lk[r] != lq ? s[r] - 100.f * WM_L2E : s[r];
      }
      if (decltype(tailc)::value) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < n ? s[r] : -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = wm_exp2(s[r] - Lq) * fmaf(dp[r], S, -DqS);        // dS * S
#pragma unroll
      for (int r = 0; r < 16; ++r) { ta[r] = (unsigned)(bq - bk[r]); tb[r] = wm_lds_f32(smem, ta[r]); }
      // one ds_add_u64 per register: integer word floor(y), fraction word (y - floor(y)) 2^32
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned hi = (unsigned)(int)floorf(s[r]), lo = (unsigned)(wm_fract(s[r]) * 4294967296.f);
        if (!(dbg & 1)) wm_lds_add64(smem, ah[r], ((unsigned long long)hi << 32) | lo);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x4 ktf = *(const u32x4*)(smem + WmBwdLds::KT + (unsigned)((((kb * 2 + t) * 2 + half) * WM_DH + (li & 15)) * 16));
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ktf), __builtin_bit_cast(bf16x8, wm_pack8(s, t)), dq, 0, 0, 0);
      }
    };
    const int nfull = n >> 5;
    for (int kb = 0; kb < nfull; ++kb) block(kb, std::false_type{});
    if (n & 31) block(nfull, std::true_type{});
    if (q < n) {
      int64_t row;
      int lb, bc;
      win_token(g, win, q, row, lb, bc);
      if (row >= 0) {
        bf16_t* dst = (bf16_t*)dqkv + (size_t)row * 3 * C + hd * WM_DH;
        const float sc = g.scale / S;
        u32x2 a, b;
        a.x = pk_bf16(dq[0] * sc, dq[1] * sc); a.y = pk_bf16(dq[2] * sc, dq[3] * sc);
        b.x = pk_bf16(dq[4] * sc, dq[5] * sc); b.y = pk_bf16(dq[6] * sc, dq[7] * sc);
        *(u32x2*)(dst + 4 * half) = a;
        *(u32x2*)(dst + 8 + 4 * half) = b;
      }
    }
  }
}

template <bool MASKED>
__device__ __forceinline__ void wm_bwd_pass_b(const WinGeom& g, unsigned char* smem, int n, int win, int hd, int wb, int li,
                                              int half, void* __restrict__ dqkv) {
  const int C = g.C;
  const int nkb = (n + 31) >> 5;
  auto slot_frag = [&](unsigned base, int blk, int t) -> u32x4 {
    return *(const u32x4*)(smem + base + (unsigned)((((blk * 2 + t) * 2 + half) * WM_DH + (li & 15)) * 16));
  };
  auto row_frag = [&](unsigned base, int tok) -> u32x4 { return *(const u32x4*)(smem + base + tok * 32 + half * 16); };
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  float padk[8], padv[8];     // sums over the wave's window-padding keys (their rows do not exist in dqkv)
#pragma unroll
  for (int i = 0; i < 8; ++i) { padk[i] = 0.f; padv[i] = 0.f; }
  for (int kb = wb; kb < nkb; kb += 4) {
    const int k = kb * 32 + li;
    const u32x4 kf = row_frag(WmBwdLds::K, k), vf = row_frag(WmBwdLds::V, k);
    const int bk = *(const int*)(smem + WmBwdLds::NBK + k * 4), lk = *(const int*)(smem + WmBwdLds::NLAB + k * 4);
    f32x16 dk = zero, dv = zero;
    for (int qb = 0; qb < nkb; ++qb) {
      const u32x4 qf = row_frag(WmBwdLds::Q, qb * 32 + li), gf = row_frag(WmBwdLds::DO, qb * 32 + li);
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qf), __builtin_bit_cast(bf16x8, kf), zero, 0, 0, 0);
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, gf), __builtin_bit_cast(bf16x8, vf), zero, 0, 0, 0);
      int bq[16];
      float ls[16], dd[16];
      wm_load16(smem, WmBwdLds::SBQ, qb, half, bq);
      wm_load16(smem, WmBwdLds::SLSE, qb, half, ls);
      wm_load16(smem, WmBwdLds::SDS, qb, half, dd);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], WM_L2E, wm_lds_f32(smem, (unsigned)(bq[r] - bk)));
      if (MASKED) {
        int lq[16];
        wm_load16(smem, WmBwdLds::SLAB, qb, half, lq);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = lq[r] != lk ? s[r] - 100.f * WM_L2E : s[r];
      }
      f32x16 pr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = wm_exp2(s[r] - ls[r]);                  // queries past the window: lse = +inf -> 0
        s[r] = pr[r] * (dp[r] - dd[r]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, slot_frag(WmBwdLds::DOT, qb, t)),
                                                     __builtin_bit_cast(bf16x8, wm_pack8(pr, t)), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, slot_frag(WmBwdLds::QT, qb, t)),
                                                     __builtin_bit_cast(bf16x8, wm_pack8(s, t)), dk, 0, 0, 0);
      }
    }
    int64_t row = -1;
    int lb, bc;
    if (k < n) win_token(g, win, k, row, lb, bc);
    if (k < n && row >= 0) {
      bf16_t* dst = (bf16_t*)dqkv + (size_t)row * 3 * C + hd * WM_DH;
      u32x2 a, b;
      a.x = pk_bf16(dk[0], dk[1]); a.y = pk_bf16(dk[2], dk[3]); b.x = pk_bf16(dk[4], dk[5]); b.y = pk_bf16(dk[6], dk[7]);
      *(u32x2*)(dst + C + 4 * half) = a;
      *(u32x2*)(dst + C + 8 + 4 * half) = b;
      a.x = pk_bf16(dv[0], dv[1]); a.y = pk_bf16(dv[2], dv[3]); b.x = pk_bf16(dv[4], dv[5]); b.y = pk_bf16(dv[6], dv[7]);
      *(u32x2*)(dst + 2 * C + 4 * half) = a;
      *(u32x2*)(dst + 2 * C + 8 + 4 * half) = b;
    }
    // window-padding keys: q = k = v = qkv.bias, their dk / dv flow into the bias gradient (fixed-order butterfly
    // over the 32 keys of the block, halves apart: they hold different channels)
    const bool padded = k < n && row < 0;
    if (__any(padded)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float xk = padded ? dk[i] : 0.f, xv = padded ? dv[i] : 0.f;
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) { xk += __shfl_xor(xk, msk, 64); xv += __shfl_xor(xv, msk, 64); }
        padk[i] += xk;
        padv[i] += xv;
      }
    }
  }
  if (li == 0) {
    float* ps = (float*)(smem + WmBwdLds::PADS) + wb * 2 * WM_DH;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = (i & 3) + 8 * (i >> 2) + 4 * half;
      ps[d] = padk[i];
      ps[WM_DH + d] = padv[i];
    }
  }
}

__global__ void __launch_bounds__(WB_NT) k_winattn_bwd_mfma(WinGeom g, const void* __restrict__ qkv,
                                                            const float* __restrict__ qkv_bias,
                                                            const float* __restrict__ table,
                                                            const void* __restrict__ out, const void* __restrict__ dout,
                                                            const float* __restrict__ lse_in, void* __restrict__ dqkv,
                                                            float* __restrict__ part_tbl, float* __restrict__ part_pad,
                                                            int dbg) {
  WM_DYN_SMEM(smem);
  const int n = g.w0 * g.w1 * g.w2, TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  const int off0 = ((g.tw0 - 1) * (2 * g.tw1 - 1) + (g.tw1 - 1)) * (2 * g.tw2 - 1) + (g.tw2 - 1);
  const int win = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const int C = g.C;
  float* tbl = (float*)(smem + WmBwdLds::TBL);
  for (int i = tid; i < TS; i += WB_NT) tbl[i] = table[(size_t)i * g.heads + hd] * WM_L2E;
  for (int i = tid; i < (int)(2 * WB_HS / 4); i += WB_NT) ((unsigned*)(smem + WmBwdLds::HIST))[i] = 0u;
  unsigned* maxs = (unsigned*)(smem + WmBwdLds::MAXS);
  if (tid < 4) maxs[tid] = 0u;
  __syncthreads();
  const bool has_bias = qkv_bias != nullptr;
  if (tid < WM_PAD) {
    const int tok = tid;
    float q[WM_DH], k[WM_DH], v[WM_DH], go[WM_DH];
    float D = 0.f, L = INFINITY;
    int64_t row = -1;
    int lb = -1, bc = 0;
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) { q[d] = 0.f; k[d] = 0.f; v[d] = 0.f; go[d] = 0.f; }
    if (tok < n) {
      win_token(g, win, tok, row, lb, bc);
      L = lse_in[((size_t)win * g.heads + hd) * WMAX + tok] * WM_L2E;
      if (row >= 0) {
        const bf16_t* base = (const bf16_t*)qkv + (size_t)row * 3 * C + hd * WM_DH;
        Elem<bf16_tag>::unpack(*(const u32x4*)base, q); Elem<bf16_tag>::unpack(*(const u32x4*)(base + 8), q + 8);
        Elem<bf16_tag>::unpack(*(const u32x4*)(base + C), k); Elem<bf16_tag>::unpack(*(const u32x4*)(base + C + 8), k + 8);
        Elem<bf16_tag>::unpack(*(const u32x4*)(base + 2 * C), v); Elem<bf16_tag>::unpack(*(const u32x4*)(base + 2 * C + 8), v + 8);
        float o[WM_DH];
        const bf16_t* ob = (const bf16_t*)out + (size_t)row * C + hd * WM_DH;
        const bf16_t* gb = (const bf16_t*)dout + (size_t)row * C + hd * WM_DH;
        Elem<bf16_tag>::unpack(*(const u32x4*)ob, o); Elem<bf16_tag>::unpack(*(const u32x4*)(ob + 8), o + 8);
        Elem<bf16_tag>::unpack(*(const u32x4*)gb, go); Elem<bf16_tag>::unpack(*(const u32x4*)(gb + 8), go + 8);
#pragma unroll
        for (int d = 0; d < WM_DH; ++d) D = fmaf(go[d], o[d], D);     // D_q = dO_q . O_q (rows of padded queries: 0)
      } else if (has_bias) {
#pragma unroll
        for (int d = 0; d < WM_DH; ++d) {
          q[d] = qkv_bias[hd * WM_DH + d];
          k[d] = qkv_bias[C + hd * WM_DH + d];
          v[d] = qkv_bias[2 * C + hd * WM_DH + d];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) q[d] *= g.scale;
    *(u32x4*)(smem + WmBwdLds::Q + tok * 32) = Elem<bf16_tag>::pack(q);
    *(u32x4*)(smem + WmBwdLds::Q + tok * 32 + 16) = Elem<bf16_tag>::pack(q + 8);
    *(u32x4*)(smem + WmBwdLds::K + tok * 32) = Elem<bf16_tag>::pack(k);
    *(u32x4*)(smem + WmBwdLds::K + tok * 32 + 16) = Elem<bf16_tag>::pack(k + 8);
    *(u32x4*)(smem + WmBwdLds::V + tok * 32) = Elem<bf16_tag>::pack(v);
    *(u32x4*)(smem + WmBwdLds::V + tok * 32 + 16) = Elem<bf16_tag>::pack(v + 8);
    *(u32x4*)(smem + WmBwdLds::DO + tok * 32) = Elem<bf16_tag>::pack(go);
    *(u32x4*)(smem + WmBwdLds::DO + tok * 32 + 16) = Elem<bf16_tag>::pack(go + 8);
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) {
      const unsigned so = wm_slot_off(tok, d);
      *(bf16_t*)(smem + WmBwdLds::KT + so) = (bf16_t)pk_bf16(k[d], 0.f);
      *(bf16_t*)(smem + WmBwdLds::QT + so) = (bf16_t)pk_bf16(q[d], 0.f);
      *(bf16_t*)(smem + WmBwdLds::DOT + so) = (bf16_t)pk_bf16(go[d], 0.f);
    }
    const int sw = wm_slot_word(tok) * 4, nbq = (int)wm_lds_addr(smem, WmBwdLds::TBL + (bc + off0) * 4);
    *(int*)(smem + WmBwdLds::NBQ + tok * 4) = nbq;     *(int*)(smem + WmBwdLds::SBQ + sw) = nbq;
    *(int*)(smem + WmBwdLds::NBK + tok * 4) = bc * 4;  *(int*)(smem + WmBwdLds::SBK + sw) = bc * 4;
    *(int*)(smem + WmBwdLds::NLAB + tok * 4) = lb;     *(int*)(smem + WmBwdLds::SLAB + sw) = lb;
    *(float*)(smem + WmBwdLds::NLSE + tok * 4) = L;    *(float*)(smem + WmBwdLds::SLSE + sw) = L;
    *(float*)(smem + WmBwdLds::NDS + tok * 4) = D;     *(float*)(smem + WmBwdLds::SDS + sw) = D;
    // bound of |dS| (non-negative floats order like their bit patterns)
    float gn = 0.f, vn = 0.f;
#pragma unroll
    for (int d = 0; d < WM_DH; ++d) { gn = fmaf(go[d], go[d], gn); vn = fmaf(v[d], v[d], vn); }
    atomicMax(maxs, __float_as_uint(gn));
    atomicMax(maxs + 1, __float_as_uint(vn));
    atomicMax(maxs + 2, __float_as_uint(fabsf(D)));
  }
  __syncthreads();
  // S = 2^(20 - e) with M < 2^(e-1) (one binade of slack for the rounding of the norms and P slightly above 1)
  const float M = sqrtf(__uint_as_float(maxs[0])) * sqrtf(__uint_as_float(maxs[1])) + __uint_as_float(maxs[2]);
  int sexp = 0;
  if (M > 0.f && M < INFINITY) {
    int e;
    (void)frexpf(M, &e);
    sexp = 20 - (e + 1);
    sexp = sexp > 100 ? 100 : (sexp < -100 ? -100 : sexp);
  }
  const float S = ldexpf(1.f, sexp);
  const bool multi = wm_multi_region(g, win);
  if (wave < 4) {
    if (dbg & 2) {
    } else if (multi) wm_bwd_pass_a<true>(g, smem, n, win, hd, wave, li, half, dqkv, S, dbg);
    else wm_bwd_pass_a<false>(g, smem, n, win, hd, wave, li, half, dqkv, S, dbg);
  } else if (dbg & 4) {
  } else {
    if (multi) wm_bwd_pass_b<true>(g, smem, n, win, hd, wave - 4, li, half, dqkv);
    else wm_bwd_pass_b<false>(g, smem, n, win, hd, wave - 4, li, half, dqkv);
  }
  __syncthreads();
  const long long* hist = (const long long*)(smem + WmBwdLds::HIST);
  const double unfix = ldexp(1.0, -32 - sexp);
  for (int i = tid; i < TS; i += WB_NT) part_tbl[((size_t)win * g.heads + hd) * TS + i] = (float)((double)hist[i] * unfix);
  if (tid < 2 * WM_DH) {
    const float* ps = (const float*)(smem + WmBwdLds::PADS);
    part_pad[((size_t)win * g.heads + hd) * 2 * WM_DH + tid] =
        (ps[tid] + ps[2 * WM_DH + tid]) + (ps[4 * WM_DH + tid] + ps[6 * WM_DH + tid]);
  }
}

}  // namespace cbim

using namespace cbim;

bool cbim_winattn_mfma_eligible(int dtype, const WinGeom& g) {
  static const int on = 1;
  return on && dtype == CBIM_BF16 && g.dh == WM_DH && g.w0 * g.w1 * g.w2 <= WMAX && g.C % 8 == 0;
}

bool cbim_winattn_mfma_bwd_eligible(int dtype, const WinGeom& g) {
  static const int on = 1;
  const int TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  return on && cbim_winattn_mfma_eligible(dtype, g) && (size_t)TS * 4 <= WB_HS;
}

int cbim_winattn_mfma_bwd(const WinGeom& g, const void* qkv, const float* qkv_bias, const float* table, const void* out,
                          const void* dout, const float* lse, void* dqkv, float* part_tbl, float* part_pad, void* stream) {
  const int TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  const size_t smem = WmBwdLds::END;
  CBIM_CHECK((size_t)TS * 4 <= WB_HS, CBIM_EUNSUPPORTED, "window attention backward (mfma): bias table of %d entries", TS);
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "window attention backward (mfma) needs %zu B of LDS", smem);
#ifndef CBIM_EMU
  static bool once = false;
  if (!once) {
    hipError_t e = hipFuncSetAttribute((const void*)k_winattn_bwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    once = true;
  }
#endif
  dim3 grid(g.B * g.nw0 * g.nw1 * g.nw2, g.heads);
  // timing ablations (wrong results): CBIM_WM_DBG bit 0 no histogram atomics, bit 1 no pass A, bit 2 no pass B
  const int dbg = 0;
  CBIM_LAUNCH(k_winattn_bwd_mfma, grid, dim3(WB_NT), smem, (hipStream_t)stream, g, qkv, qkv_bias, table, out, dout, lse, dqkv,
              part_tbl, part_pad, dbg);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "window_attn3d_bwd (mfma) launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

int cbim_winattn_mfma_fwd(const WinGeom& g, const void* qkv, const float* qkv_bias, const float* table, void* out,
                          float* lse, void* stream) {
  const int TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  const size_t smem = wm_fwd_smem(TS);
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "window attention (mfma) needs %zu B of LDS", smem);
#ifndef CBIM_EMU
  static bool once = false;
  if (!once) {
    hipError_t e = hipFuncSetAttribute((const void*)k_winattn_fwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    once = true;
  }
#endif
  dim3 grid(g.B * g.nw0 * g.nw1 * g.nw2, g.heads);
  CBIM_LAUNCH(k_winattn_fwd_mfma, grid, dim3(WM_NT), smem, (hipStream_t)stream, g, qkv, qkv_bias, table, out, lse);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "window_attn3d_fwd (mfma) launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(swin_mfma)
