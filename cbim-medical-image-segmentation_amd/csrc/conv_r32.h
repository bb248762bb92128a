// conv_r32.h — internal interface of the "weights in registers" 3x3x3 kernel (conv_r32.hip), used by the
// cbim_conv3d_igemm launcher (conv_igemm.hip).  Not part of the C ABI.
#pragma once
#include <stdint.h>
#include "../../include/cbim_hip.h"

namespace cbim {
struct R32Params {
  const void* x; int64_t x_stride; const float* in_stats;
  const void* x2; int64_t x2_stride; int c_split;   // Cin chunks >= c_split come from x2 (virtual concatenation)
  int NC;                                           // 32-channel chunks of Cin
  int BN;                                           // cout block of the packed weight image: 32 (Cout <= 32) or 64
  const void* w;
  const void* res; int64_t res_stride;
  const void* mx; int64_t mx_stride; const float* m_stats;
  void* y; int64_t y_stride;
  float* partials;
  int N, Di, Hi, Wi, Do, Ho, Wo, Cout;
  int pD, pH, pW, act;
  int tiles_d, tiles_h, tiles_w;
  int dbg; // timing ablations for tools/ (env CBIM_R32_DBG); 0 in production
  int P;   // records per image of `partials` (cbim_conv3d_num_tiles)
  // k_conv3_rw split-K (low-resolution layers): blockIdx.z owns a slice of the Cin chunks and writes raw fp32 partial sums
  // ws[ksplit][N*Do*Ho*Wo][Cout] for k_splitk_finish (conv_igemm.hip); 1 / nullptr otherwise
  int ksplit; float* ws;
  int cin_bytes;  // 2 Cin: k_conv3_rw zero-fills the 16-byte slots of a last chunk past it (Cin = 48: chunk 1 holds 16 channels)
};
}  // namespace cbim

// bf16, 3x3x3, Cin a multiple of 32 (a second input tensor splits it at a multiple of 32), Cout <= 32 or a multiple of
// 32, enough 8x8x8 output tiles to give every CU a (tile strip, Cout chunk) pair
bool cbim_conv_r32_eligible(const cbim_conv_desc* d, const void* x2, int cin_split, const float* in_stats, const void* res,
                            const void* mask_x);
// persistent grid.x = records per image of the statistics partials; grid.y = 32-channel chunks of Cout
int64_t cbim_conv_r32_grid(const cbim_conv_desc* d);
int cbim_conv_r32_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                         int cin_split, const float* in_stats,
                         const void* w_packed, const void* res, int64_t res_stride, const void* mask_x,
                         int64_t mask_stride, const float* mask_stats, void* y, int64_t y_stride, float* partials,
                         void* stream);

// conv_rw.hip (round 4): the same convolutions with plane-major buffer-addressed LDS-DMA pieces, statistics sums in LDS and
// an optional WIDE form (64 output channels per workgroup).  Takes the calls whose input is used as it is (no in_stats) and
// whose epilogue is the forward one or the activated-mask dgrad one; asked only after cbim_conv_r32_eligible said yes.
bool cbim_conv_rw_eligible(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                           const float* in_stats, const void* mask_x, const float* mask_stats);
int64_t cbim_conv_rw_grid(const cbim_conv_desc* d);
// round 6: Cout in multiples of 48 (Cin any multiple of 8) where the 32-channel kernels do not apply — k_conv3_rw48, launched
// through cbim_conv_rw_launch; the activated-mask dgrad also with a LeakyReLU(0.01) mask tensor
bool cbim_conv_rw48_eligible(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride, int cin_split,
                             const float* in_stats, const void* mask_x, const float* mask_stats);
int cbim_conv_rw_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                        int cin_split, const void* w_packed, const void* res, int64_t res_stride, const void* mask_x,
                        int64_t mask_stride, void* y, int64_t y_stride, float* partials, void* stream);
// split-K form for layers with too few (tile, Cout block) pairs to fill the chip: the Cin chunks are shared out over
// blockIdx.z, raw fp32 partials go to `ws` (cbim_conv_rw_ksplit(d) slabs), finished by k_splitk_finish (residual, mask,
// statistics, store).  cbim_conv_rw_ksplit: 0 / 1 = not taken.
int cbim_conv_rw_ksplit(const cbim_conv_desc* d);
int cbim_conv_rw_split_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                              int cin_split, const void* w_packed, float* ws, void* stream);

// conv_pw.hip (round 4): bf16 1x1x1 convolutions as a row GEMM without operand staging (forward with InstanceNorm + activation
// on load, residual, statistics; activated-mask dgrad with the two backward sums).  cbim_conv_pw_records: statistics records per
// image it writes (0: the descriptor is not a pointwise layer it takes).
int cbim_conv_pw_records(const cbim_conv_desc* d);
bool cbim_conv_pw_eligible(const cbim_conv_desc* d, int64_t x_stride, const void* x2, int64_t res_stride, int64_t mask_stride,
                           int64_t y_stride, const void* res, const void* mask_x, const float* mask_stats);
int cbim_conv_pw_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const float* in_stats, const void* w_packed,
                        const void* res, int64_t res_stride, const void* mask_x, int64_t mask_stride, const float* mask_stats,
                        void* y, int64_t y_stride, float* partials, int P, void* stream);
// weight gradient of those layers: (64 co x 64 ci) blocks, 128 rows per stage, transposed LDS reads; writes slabs
// ws[slab][Cout_pad][Cin_pad] for conv_wgrad.hip's k_wgrad_reduce
bool cbim_conv_pw_wgrad_eligible(const cbim_conv_desc* d, int64_t x_stride, const void* x2, int64_t dy_stride, const void* dy2);
size_t cbim_conv_pw_wgrad_workspace(const cbim_conv_desc* d);
int cbim_conv_pw_wgrad_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const float* in_stats, const void* dy,
                              int64_t dy_stride, float* ws, int* n_slabs, int* Cout_pad, int* Cin_pad, void* stream);
