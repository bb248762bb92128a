// swin_common.h — window geometry shared by the shifted-window attention kernels (swin_kernels.hip: vector-ALU
// kernels for fp32 / any head size; swin_mfma.hip: matrix-core kernels for bf16, d_head 16).
#pragma once
#include "cbim_common.h"

namespace cbim {

static constexpr int WT_THREADS = 384;   // >= 343 = 7^3 tokens
static constexpr int WMAX = 343;

struct WinGeom {
  int B, D, H, W, C, heads, dh;
  int w0, w1, w2;        // window extents actually used (get_window_size, :358-381)
  int s0, s1, s2;        // shift (0 where the window covers the dimension)
  int Dp, Hp, Wp;        // padded extents
  int nw0, nw1, nw2;     // windows per dimension
  int tw0, tw1, tw2;     // extents of the module's bias table window (7,7,7)
  int masked;            // any shift > 0
  float scale;
};

// token t of window `win` -> source row in the [B][D][H][W] tensor (or -1 for a padded token), region label,
// bias-table coordinate B_t
__device__ __forceinline__ void win_token(const WinGeom& g, int win, int t, int64_t& row, int& label, int& bcoord) {
  int ww = win % g.nw2, wh = (win / g.nw2) % g.nw1, wd = (win / (g.nw2 * g.nw1)) % g.nw0, b = win / (g.nw2 * g.nw1 * g.nw0);
  int c = t % g.w2, bb = (t / g.w2) % g.w1, a = t / (g.w2 * g.w1);
  int pd = wd * g.w0 + a, ph = wh * g.w1 + bb, pw = ww * g.w2 + c;          // shifted frame
  int sd = pd + g.s0, sh = ph + g.s1, sw = pw + g.s2;                         // torch.roll(x, -shift)
  if (sd >= g.Dp) sd -= g.Dp;
  if (sh >= g.Hp) sh -= g.Hp;
  if (sw >= g.Wp) sw -= g.Wp;
  row = (sd < g.D && sh < g.H && sw < g.W) ? (((int64_t)b * g.D + sd) * g.H + sh) * g.W + sw : -1;
  int ld = g.s0 == 0 ? 2 : (pd < g.Dp - g.w0 ? 0 : (pd < g.Dp - g.s0 ? 1 : 2));
  int lh = g.s1 == 0 ? 2 : (ph < g.Hp - g.w1 ? 0 : (ph < g.Hp - g.s1 ? 1 : 2));
  int lw = g.s2 == 0 ? 2 : (pw < g.Wp - g.w2 ? 0 : (pw < g.Wp - g.s2 ? 1 : 2));
  label = (ld * 3 + lh) * 3 + lw;
  int t0 = t / (g.tw1 * g.tw2), t1 = (t / g.tw2) % g.tw1, t2 = t % g.tw2;    // coordinates by token number
  bcoord = (t0 * (2 * g.tw1 - 1) + t1) * (2 * g.tw2 - 1) + t2;
}

}  // namespace cbim

// matrix-core path (swin_mfma.hip): bf16, d_head 16, windows of <= 352 tokens
bool cbim_winattn_mfma_eligible(int dtype, const cbim::WinGeom& g);
bool cbim_winattn_mfma_bwd_eligible(int dtype, const cbim::WinGeom& g);
int cbim_winattn_mfma_fwd(const cbim::WinGeom& g, const void* qkv, const float* qkv_bias, const float* table, void* out,
                          float* lse, void* stream);
int cbim_winattn_mfma_bwd(const cbim::WinGeom& g, const void* qkv, const float* qkv_bias, const float* table,
                          const void* out, const void* dout, const float* lse, void* dqkv, float* part_tbl,
                          float* part_pad, void* stream);
