// conv_wgrad_r32.h — internal interface of the "accumulators in registers" 3x3x3 weight-gradient kernel
// (conv_wgrad_r32.hip), used by the cbim_conv3d_wgrad launcher (conv_wgrad.hip).  Not part of the C ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/cbim_hip.h"

// bf16, 3x3x3 / stride 1 / pad 1, Cin and Cout multiples of 32, the convolution input used AS IT IS (no statistics:
// the caller materialised act(IN(x)) once), optional second input / second dy tensor split at multiples of 32
bool cbim_wgrad_r32_eligible(const cbim_conv_desc* d, const float* in_stats, const void* x2, int cin_split, const void* dy2,
                             int cout_split);
// strips (= fp32 slabs [Cout][Cin][27] in the workspace) the launch writes; the caller reduces them in fixed order
// (cbim_wgrad_r32_reduce; a single strip writes dw directly)
int cbim_wgrad_r32_strips(const cbim_conv_desc* d);
size_t cbim_wgrad_r32_workspace(const cbim_conv_desc* d);
int cbim_wgrad_r32_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* x2, int64_t x2_stride,
                          int cin_split, const void* dy, int64_t dy_stride, const void* dy2, int64_t dy2_stride,
                          int cout_split, float* workspace, float* dw, void* stream);
// fixed-order sum of the strips' slabs into dw[co][ci][tap] (fp32, natural nn.Conv3d layout)
int cbim_wgrad_r32_reduce(const cbim_conv_desc* d, const float* workspace, float* dw, void* stream);

// Depthwise 3x3x3 weight gradient on the same kernel (round 3): dw[c][tap] = sum_v dy[v][c] x[v + tap][c] is the DIAGONAL of the
// dense 32 x 32 block of a 32-channel group — 32x more multiply-adds than needed, on matrix cores that are 16-32x faster than
// the vector ALU form and fed by LDS-DMA (k_dwconv3_wgrad_lds: 331 us on a 32^3 x 512-channel tensor that moves 67 MB).
// d->Cin == d->Cout == C; slabs [strips][C][27] into `workspace` (>= strips * C * 27 floats), summed by the caller.
bool cbim_wgrad_r32_dw_eligible(const cbim_conv_desc* d);
int cbim_wgrad_r32_dw_strips(const cbim_conv_desc* d);
int cbim_wgrad_r32_dw_launch(const cbim_conv_desc* d, const void* x, int64_t x_stride, const void* dy, int64_t dy_stride,
                             float* workspace, void* stream);
