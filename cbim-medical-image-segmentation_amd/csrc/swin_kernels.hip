// swin_kernels.hip — SwinUNETR's shifted-window 3-D attention.
//
// Replaces, per SwinTransformerBlock (/root/reference/model/dim3/swin_unetr.py:554-606 forward_part1 and
// :467-490 WindowAttention.forward), the chain  F.pad -> torch.roll(-shift) -> window_partition ->
// q@k^T*scale + relative_position_bias_table[index] + shift mask -> softmax -> @v -> window_reverse ->
// torch.roll(+shift) -> crop:  the gather/scatter of all of these is index arithmetic inside ONE kernel, so
// the six full-tensor copies of the eager path never happen.
//
//   qkv   [B][D][H][W][3*C]  (token rows; within a row [3][heads][dh], :469)      T = float | bf16
//   out   [B][D][H][W][C]    ((attn@v).transpose(1,2).reshape(b,n,c), :486)
//   padded tokens (window padding, :561-566): the reference pads AFTER norm1 and BEFORE qkv, so a padded
//   token has q=k=v = qkv.bias; it takes part as a key; its own output row is cropped away.
//   bias index: relative_position_index[:n,:n] (:474) — for windows smaller than 7^3 the reference slices the
//   7^3 table by TOKEN NUMBER, i.e. token t is given the coordinates (t/49, t/7%7, t%7); reproduced here.
//   mask: compute_mask (:737-773): -100 between tokens of different roll regions, per dimension
//   [0,-w) / [-w,-s) / [-s,end) of the padded, shifted frame.
//
// One workgroup = one (window, head); thread = one query (forward, backward pass A) / one key (pass B);
// K,V (then Q,dO) of the window live in LDS, read with broadcast; online softmax in registers.  The work is
// ~61 GFLOP per forward at 128^3 (d_head 16): latency/LDS-bound, not MFMA-shaped at 343x343x16.
// Every reduction has a fixed order: d(bias table) is accumulated per workgroup in an LDS histogram — at
// key step j all queries i address distinct entries (i -> B_i - B_j is injective), one barrier per two keys
// (even / odd keys own separate histograms) —
// and the per-workgroup histograms are summed over windows in launch order.
#include "swin_common.h"
#include <stdlib.h>

namespace cbim {

#ifdef CBIM_EMU
#define CBIM_DYN_SMEM(name) unsigned char* name = cbim_emu::dyn_smem()
#else
#define CBIM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

struct WinSmem {
  float* A;       // [n][DH]  K   (pass B: scaled Q)
  float* Bv;      // [n][DH]  V   (pass B: dO)
  float* tbl;     // [TS]     bias table column of this head
  float* hist;    // [2][TS]  backward only: even / odd key steps
  float* lse;     // [n]
  float* dsum;    // [n]
  int* bco;       // [n]
  int* lab;       // [n]
  int64_t* row;   // [n]
};

template <int DH>
__device__ __forceinline__ WinSmem win_smem(unsigned char* smem, int TS) {
  WinSmem s;
  unsigned o = 0;
  s.A = (float*)(smem + o); o += WMAX * DH * 4;
  s.Bv = (float*)(smem + o); o += WMAX * DH * 4;
  s.tbl = (float*)(smem + o); o += TS * 4;
  s.hist = (float*)(smem + o); o += 2 * TS * 4;
  s.lse = (float*)(smem + o); o += WMAX * 4;
  s.dsum = (float*)(smem + o); o += WMAX * 4;
  s.bco = (int*)(smem + o); o += WMAX * 4;
  s.lab = (int*)(smem + o); o += WMAX * 4;
  o = (o + 7u) & ~7u;
  s.row = (int64_t*)(smem + o);
  return s;
}
static size_t win_smem_bytes(int dh, int TS) { return (size_t)2 * WMAX * dh * 4 + 3 * (size_t)TS * 4 + 4 * WMAX * 4 + 8 + WMAX * 8; }

template <typename T, int DH>
__global__ void __launch_bounds__(WT_THREADS) k_winattn_fwd(WinGeom g, const void* __restrict__ qkv,
                                                            const float* __restrict__ qkv_bias,
                                                            const float* __restrict__ table, void* __restrict__ out,
                                                            float* __restrict__ lse_out) {
  CBIM_DYN_SMEM(smem);
  const int n = g.w0 * g.w1 * g.w2, TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  const int off0 = ((g.tw0 - 1) * (2 * g.tw1 - 1) + (g.tw1 - 1)) * (2 * g.tw2 - 1) + (g.tw2 - 1);
  WinSmem s = win_smem<DH>(smem, TS);
  const int win = blockIdx.x, h = blockIdx.y, t = threadIdx.x;
  const int C = g.C;
  for (int i = t; i < TS; i += WT_THREADS) s.tbl[i] = table[(size_t)i * g.heads + h];
  float q[DH];
  int64_t myrow = -1;
  int mylab = 0, myb = 0;
  if (t < n) {
    win_token(g, win, t, myrow, mylab, myb);
    s.bco[t] = myb; s.lab[t] = mylab;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      float qv, kv, vv;
      if (myrow >= 0) {
        size_t base = (size_t)myrow * 3 * C + h * DH + d;
        qv = Elem<T>::load1(qkv, base); kv = Elem<T>::load1(qkv, base + C); vv = Elem<T>::load1(qkv, base + 2 * C);
      } else {
        qv = qkv_bias ? qkv_bias[h * DH + d] : 0.f;
        kv = qkv_bias ? qkv_bias[C + h * DH + d] : 0.f;
        vv = qkv_bias ? qkv_bias[2 * C + h * DH + d] : 0.f;
      }
      q[d] = qv * g.scale;
      s.A[t * DH + d] = kv;
      s.Bv[t * DH + d] = vv;
    }
  }
  __syncthreads();
  if (t >= n) return;
  float m = -INFINITY, l = 0.f, o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
  // online softmax over tiles of KT keys: one running-maximum update and one rescale of (l, o) per tile instead of per
  // key (the per-key form spent DH multiplies and a second expf on the rescale alone)
  constexpr int KT = 4;
  for (int j0 = 0; j0 < n; j0 += KT) {
    float sc[KT];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int j = j0 + u < n ? j0 + u : n - 1;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) a = fmaf(q[d], s.A[j * DH + d], a);   // explicit: the build has -ffp-contract=off
      a += s.tbl[myb - s.bco[j] + off0];
      if (g.masked && s.lab[j] != mylab) a += -100.f;
      sc[u] = j0 + u < n ? a : -INFINITY;
    }
    float mn = m;
#pragma unroll
    for (int u = 0; u < KT; ++u) mn = fmaxf(mn, sc[u]);
    const float corr = expf(m - mn);      // 0 on the first tile (m = -inf, mn finite: key j0 always exists)
    l *= corr;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] *= corr;
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int j = j0 + u < n ? j0 + u : n - 1;
      const float p = expf(sc[u] - mn);   // 0 past the last key
      l += p;
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(p, s.Bv[j * DH + d], o[d]);
    }
    m = mn;
  }
  lse_out[((size_t)win * g.heads + h) * WMAX + t] = m + logf(l);
  if (myrow >= 0) {
    float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < DH; ++d) Elem<T>::store1(out, (size_t)myrow * C + h * DH + d, o[d] * inv);
  }
}

// backward.  part_tbl [nwin][heads][TS], part_pad [nwin][heads][2][DH] (sum of dk, dv over padded keys)
template <typename T, int DH>
__global__ void __launch_bounds__(WT_THREADS) k_winattn_bwd(WinGeom g, const void* __restrict__ qkv,
                                                            const float* __restrict__ qkv_bias,
                                                            const float* __restrict__ table,
                                                            const void* __restrict__ out, const void* __restrict__ dout,
                                                            const float* __restrict__ lse_in, void* __restrict__ dqkv,
                                                            float* __restrict__ part_tbl, float* __restrict__ part_pad) {
  CBIM_DYN_SMEM(smem);
  const int n = g.w0 * g.w1 * g.w2, TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  const int off0 = ((g.tw0 - 1) * (2 * g.tw1 - 1) + (g.tw1 - 1)) * (2 * g.tw2 - 1) + (g.tw2 - 1);
  WinSmem s = win_smem<DH>(smem, TS);
  const int win = blockIdx.x, h = blockIdx.y, t = threadIdx.x;
  const int C = g.C;
  const bool act = t < n;
  for (int i = t; i < TS; i += WT_THREADS) { s.tbl[i] = table[(size_t)i * g.heads + h]; s.hist[i] = 0.f; s.hist[TS + i] = 0.f; }
  float q[DH], kk[DH], vv[DH], go[DH];
  int64_t myrow = -1;
  int mylab = 0, myb = 0;
  float mylse = 0.f, myD = 0.f;
  if (act) {
    win_token(g, win, t, myrow, mylab, myb);
    s.bco[t] = myb; s.lab[t] = mylab; s.row[t] = myrow;
    mylse = lse_in[((size_t)win * g.heads + h) * WMAX + t];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      float qv, ov = 0.f, gv = 0.f;
      if (myrow >= 0) {
        size_t base = (size_t)myrow * 3 * C + h * DH + d;
        qv = Elem<T>::load1(qkv, base); kk[d] = Elem<T>::load1(qkv, base + C); vv[d] = Elem<T>::load1(qkv, base + 2 * C);
        ov = Elem<T>::load1(out, (size_t)myrow * C + h * DH + d);
        gv = Elem<T>::load1(dout, (size_t)myrow * C + h * DH + d);
      } else {
        qv = qkv_bias ? qkv_bias[h * DH + d] : 0.f;
        kk[d] = qkv_bias ? qkv_bias[C + h * DH + d] : 0.f;
        vv[d] = qkv_bias ? qkv_bias[2 * C + h * DH + d] : 0.f;
      }
      q[d] = qv * g.scale;
      go[d] = gv;                       // rows of padded queries are cropped: zero upstream gradient
      myD += gv * ov;                   // D_i = sum_j P_ij dP_ij = dO_i . O_i
      s.A[t * DH + d] = kk[d];
      s.Bv[t * DH + d] = vv[d];
    }
    s.lse[t] = mylse; s.dsum[t] = myD;
  }
  __syncthreads();
  // ---- pass A: thread = query i.  dq_i, and the bias-table histogram.  At one key j all queries address distinct
  // entries; two keys per step go to two histograms (even / odd), so there is one barrier per TWO keys and still no
  // two threads ever touch the same word between barriers.  hist[0] + hist[1] at the end: fixed order.
  float dq[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) dq[d] = 0.f;
  for (int j0 = 0; j0 < n; j0 += 2) {
    if (act) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = j0 + u;
        if (j < n) {
          float sc = 0.f, dp = 0.f;
#pragma unroll
          for (int d = 0; d < DH; ++d) { sc = fmaf(q[d], s.A[j * DH + d], sc); dp = fmaf(go[d], s.Bv[j * DH + d], dp); }
          int idx = myb - s.bco[j] + off0;
          sc += s.tbl[idx];
          if (g.masked && s.lab[j] != mylab) sc += -100.f;
          float p = expf(sc - mylse);
          float ds = p * (dp - myD);
#pragma unroll
          for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, s.A[j * DH + d], dq[d]);
          s.hist[u * TS + idx] += ds;     // distinct idx for distinct queries at a fixed j
        }
      }
    }
    __syncthreads();
  }
  if (act && myrow >= 0) {
#pragma unroll
    for (int d = 0; d < DH; ++d) Elem<T>::store1(dqkv, (size_t)myrow * 3 * C + h * DH + d, dq[d] * g.scale);
  }
  for (int i = t; i < TS; i += WT_THREADS) part_tbl[((size_t)win * g.heads + h) * TS + i] = s.hist[i] + s.hist[TS + i];
  __syncthreads();
  // ---- pass B: thread = key j.  LDS now holds scaled Q and dO
  if (act) {
#pragma unroll
    for (int d = 0; d < DH; ++d) { s.A[t * DH + d] = q[d]; s.Bv[t * DH + d] = go[d]; }
  }
  __syncthreads();
  float dk[DH], dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  if (act) {
    for (int i = 0; i < n; ++i) {
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) { sc = fmaf(s.A[i * DH + d], kk[d], sc); dp = fmaf(s.Bv[i * DH + d], vv[d], dp); }
      sc += s.tbl[s.bco[i] - myb + off0];
      if (g.masked && s.lab[i] != mylab) sc += -100.f;
      float p = expf(sc - s.lse[i]);
      float ds = p * (dp - s.dsum[i]);
#pragma unroll
      for (int d = 0; d < DH; ++d) { dv[d] = fmaf(p, s.Bv[i * DH + d], dv[d]); dk[d] = fmaf(ds, s.A[i * DH + d], dk[d]); }
    }
    if (myrow >= 0) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        Elem<T>::store1(dqkv, (size_t)myrow * 3 * C + C + h * DH + d, dk[d]);
        Elem<T>::store1(dqkv, (size_t)myrow * 3 * C + 2 * C + h * DH + d, dv[d]);
      }
    }
  }
  __syncthreads();
  // gradient of qkv.bias through the padded keys: fixed-order sum over this window's padded tokens
  if (act) {
#pragma unroll
    for (int d = 0; d < DH; ++d) { s.A[t * DH + d] = myrow < 0 ? dk[d] : 0.f; s.Bv[t * DH + d] = myrow < 0 ? dv[d] : 0.f; }
  }
  __syncthreads();
  if (t < 2 * DH) {
    const float* src = t < DH ? s.A : s.Bv;
    int d = t % DH;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += src[i * DH + d];
    part_pad[(((size_t)win * g.heads + h) * 2 + t / DH) * DH + d] = acc;
  }
}

// dtable[i][h] = sum_win part_tbl[win][h][i];  dbias_pad[(1+kv)*C + h*DH + d] = sum_win part_pad[win][h][kv][d]
__global__ void __launch_bounds__(256) k_winattn_reduce(const float* __restrict__ part_tbl, const float* __restrict__ part_pad,
                                                        float* __restrict__ dtable, float* __restrict__ dbias, int nwin,
                                                        int heads, int TS, int DH, int C) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < TS * heads) {
    int h = i % heads, e = i / heads;
    // 8 independent chains in fixed association (a single chain pays one memory latency per window)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = 0;
    for (; w + 7 < nwin; w += 8)
#pragma unroll
      for (int q = 0; q < 8; ++q) a8[q] += part_tbl[((size_t)(w + q) * heads + h) * TS + e];
    for (; w < nwin; ++w) a8[0] += part_tbl[((size_t)w * heads + h) * TS + e];
    dtable[i] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  if (i < 3 * C) {
    float acc = 0.f;
    if (i >= C) {
      int kv = i / C - 1, h = (i % C) / DH, d = i % DH;
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
      int w = 0;
      for (; w + 3 < nwin; w += 4)
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] += part_pad[(((size_t)(w + q) * heads + h) * 2 + kv) * DH + d];
      for (; w < nwin; ++w) a4[0] += part_pad[(((size_t)w * heads + h) * 2 + kv) * DH + d];
      acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    dbias[i] = acc;
  }
}

}  // namespace cbim

using namespace cbim;

static int fill_geom(WinGeom& g, int B, int D, int H, int W, int C, int heads, const int* window, const int* shift,
                     const int* table_window) {
  CBIM_CHECK(B >= 1 && D >= 1 && H >= 1 && W >= 1 && heads >= 1 && C % heads == 0, CBIM_EINVAL, "bad window-attention extents");
  g.B = B; g.D = D; g.H = H; g.W = W; g.C = C; g.heads = heads; g.dh = C / heads;
  CBIM_CHECK(g.dh == 8 || g.dh == 16 || g.dh == 32, CBIM_EUNSUPPORTED, "window attention head dim %d (supported 8/16/32)", g.dh);
  g.w0 = window[0]; g.w1 = window[1]; g.w2 = window[2];
  g.s0 = shift[0]; g.s1 = shift[1]; g.s2 = shift[2];
  g.tw0 = table_window[0]; g.tw1 = table_window[1]; g.tw2 = table_window[2];
  CBIM_CHECK(g.w0 >= 1 && g.w1 >= 1 && g.w2 >= 1 && g.w0 * g.w1 * g.w2 <= WMAX, CBIM_EUNSUPPORTED, "window of %d tokens > %d",
             g.w0 * g.w1 * g.w2, WMAX);
  CBIM_CHECK(g.w0 * g.w1 * g.w2 <= g.tw0 * g.tw1 * g.tw2, CBIM_EINVAL, "window larger than the bias-table window");
  CBIM_CHECK(g.s0 >= 0 && g.s0 < g.w0 && g.s1 >= 0 && g.s1 < g.w1 && g.s2 >= 0 && g.s2 < g.w2, CBIM_EINVAL, "bad shift");
  g.nw0 = (D + g.w0 - 1) / g.w0; g.nw1 = (H + g.w1 - 1) / g.w1; g.nw2 = (W + g.w2 - 1) / g.w2;
  g.Dp = g.nw0 * g.w0; g.Hp = g.nw1 * g.w1; g.Wp = g.nw2 * g.w2;
  g.masked = (g.s0 | g.s1 | g.s2) != 0;
  g.scale = 1.0f / sqrtf((float)g.dh);
  return 0;
}

extern "C" int cbim_window_attn3d_num_windows(int B, int D, int H, int W, const int* window) {
  if (!window || window[0] < 1 || window[1] < 1 || window[2] < 1) return 0;
  return B * ((D + window[0] - 1) / window[0]) * ((H + window[1] - 1) / window[1]) * ((W + window[2] - 1) / window[2]);
}

extern "C" size_t cbim_window_attn3d_workspace(int B, int D, int H, int W, int C, int heads, const int* window,
                                               const int* table_window) {
  if (!window || !table_window) return 0;
  size_t nwin = (size_t)cbim_window_attn3d_num_windows(B, D, H, W, window);
  size_t TS = (size_t)(2 * table_window[0] - 1) * (2 * table_window[1] - 1) * (2 * table_window[2] - 1);
  return nwin * heads * (TS + 2 * (size_t)(C / heads)) * sizeof(float);
}

template <typename K>
static int set_smem(K kernel, size_t bytes) {
#ifndef CBIM_EMU
  // set once per kernel: ask for the CU maximum, the first caller's window need not be the largest one
  (void)bytes;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
#else
  (void)kernel; (void)bytes;
#endif
  return 0;
}

#define WIN_DISPATCH(KERNEL, ...)                                                                          \
  do {                                                                                                     \
    if (dtype == CBIM_BF16) {                                                                              \
      if (g.dh == 8) { static bool once = false; if (!once) { if (int e = set_smem(KERNEL<bf16_tag, 8>, smem)) return e; once = true; } CBIM_LAUNCH((KERNEL<bf16_tag, 8>), grid, dim3(WT_THREADS), smem, st, __VA_ARGS__); } \
      else if (g.dh == 16) { static bool once = false; if (!once) { if (int e = set_smem(KERNEL<bf16_tag, 16>, smem)) return e; once = true; } CBIM_LAUNCH((KERNEL<bf16_tag, 16>), grid, dim3(WT_THREADS), smem, st, __VA_ARGS__); } \
      else { static bool once = false; if (!once) { if (int e = set_smem(KERNEL<bf16_tag, 32>, smem)) return e; once = true; } CBIM_LAUNCH((KERNEL<bf16_tag, 32>), grid, dim3(WT_THREADS), smem, st, __VA_ARGS__); } \
    } else {                                                                                               \
      if (g.dh == 8) { static bool once = false; if (!once) { if (int e = set_smem(KERNEL<float, 8>, smem)) return e; once = true; } CBIM_LAUNCH((KERNEL<float, 8>), grid, dim3(WT_THREADS), smem, st, __VA_ARGS__); } \
      else if (g.dh == 16) { static bool once = false; if (!once) { if (int e = set_smem(KERNEL<float, 16>, smem)) return e; once = true; } CBIM_LAUNCH((KERNEL<float, 16>), grid, dim3(WT_THREADS), smem, st, __VA_ARGS__); } \
      else { static bool once = false; if (!once) { if (int e = set_smem(KERNEL<float, 32>, smem)) return e; once = true; } CBIM_LAUNCH((KERNEL<float, 32>), grid, dim3(WT_THREADS), smem, st, __VA_ARGS__); } \
    }                                                                                                      \
  } while (0)

extern "C" int cbim_window_attn3d_fwd(int dtype, const void* qkv, const float* qkv_bias, const float* table, void* out,
                                      float* lse, int B, int D, int H, int W, int C, int heads, const int* window,
                                      const int* shift, const int* table_window, void* stream) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(qkv && table && out && lse && window && shift && table_window, CBIM_EINVAL, "null argument");
  WinGeom g;
  if (int e = fill_geom(g, B, D, H, W, C, heads, window, shift, table_window)) return e;
  int TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  size_t smem = win_smem_bytes(g.dh, TS);
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "window attention needs %zu B of LDS", smem);
  dim3 grid(B * g.nw0 * g.nw1 * g.nw2, heads);
  hipStream_t st = (hipStream_t)stream;
  if (cbim_winattn_mfma_eligible(dtype, g))   // bf16, d_head 16: QK^T and PV on the matrix cores (swin_mfma.hip)
    return cbim_winattn_mfma_fwd(g, qkv, qkv_bias, table, out, lse, stream);
  WIN_DISPATCH(k_winattn_fwd, g, qkv, qkv_bias, table, out, lse);
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "window_attn3d_fwd launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

extern "C" int cbim_window_attn3d_bwd(int dtype, const void* qkv, const float* qkv_bias, const float* table,
                                      const void* out, const void* dout, const float* lse, void* dqkv, float* dtable,
                                      float* dbias_pad, int B, int D, int H, int W, int C, int heads, const int* window,
                                      const int* shift, const int* table_window, void* workspace, size_t ws_bytes,
                                      void* stream) {
  CBIM_CHECK(dtype == CBIM_F32 || dtype == CBIM_BF16, CBIM_EINVAL, "bad dtype %d", dtype);
  CBIM_CHECK(qkv && table && out && dout && lse && dqkv && dtable && dbias_pad && window && shift && table_window, CBIM_EINVAL,
             "null argument");
  WinGeom g;
  if (int e = fill_geom(g, B, D, H, W, C, heads, window, shift, table_window)) return e;
  CBIM_CHECK(workspace && ws_bytes >= cbim_window_attn3d_workspace(B, D, H, W, C, heads, window, table_window), CBIM_EWORKSPACE,
             "window attention workspace too small");
  int TS = (2 * g.tw0 - 1) * (2 * g.tw1 - 1) * (2 * g.tw2 - 1);
  size_t smem = win_smem_bytes(g.dh, TS);
  CBIM_CHECK(smem <= 160 * 1024, CBIM_EUNSUPPORTED, "window attention needs %zu B of LDS", smem);
  int nwin = B * g.nw0 * g.nw1 * g.nw2;
  dim3 grid(nwin, heads);
  hipStream_t st = (hipStream_t)stream;
  float* part_tbl = (float*)workspace;
  float* part_pad = part_tbl + (size_t)nwin * heads * TS;
  if (cbim_winattn_mfma_bwd_eligible(dtype, g)) {
    if (int e = cbim_winattn_mfma_bwd(g, qkv, qkv_bias, table, out, dout, lse, dqkv, part_tbl, part_pad, stream)) return e;
  } else {
    WIN_DISPATCH(k_winattn_bwd, g, qkv, qkv_bias, table, out, dout, lse, dqkv, part_tbl, part_pad);
  }
  hipError_t e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "window_attn3d_bwd launch: %s", hipGetErrorString(e));
  int items = TS * heads > 3 * C ? TS * heads : 3 * C;
  CBIM_LAUNCH(k_winattn_reduce, dim3((items + 255) / 256), dim3(256), 0, st, (const float*)part_tbl, (const float*)part_pad,
              dtable, dbias_pad, nwin, heads, TS, g.dh, C);
  e = CBIM_LAST_LAUNCH();
  CBIM_CHECK(e == hipSuccess, CBIM_ELAUNCH, "window_attn3d reduce launch: %s", hipGetErrorString(e));
  return CBIM_OK;
}

CBIM_DEFINE_WARM(swin)
