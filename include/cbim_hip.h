/* cbim_hip.h — C ABI of libcbim_hip.so: the MI355X (gfx950) kernels behind the
 * model/dim3 forward/backward hot path of yhygao/CBIM-Medical-Image-Segmentation.
 *
 * The reference has NO native code and no FFI: its hot path is stock torch.nn modules
 * (SURVEY.md §0.1).  The boundary a maintainer binds is therefore ours to define; each entry
 * point below names the reference call site (file:line under /root/reference) whose ATen
 * work it replaces.  INTEGRATION.md shows the ctypes binding and the get_model() hook.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every pointer is DEVICE memory owned by the caller
 *    (PyTorch caching allocator); kernels never allocate, free or retain pointers.
 *  - Launch-only and asynchronous on `stream` (a hipStream_t passed as void*); re-entrant.
 *  - Activations are channels-last NDHWC: element (n,d,h,w,c) of a tensor view lives at
 *    ptr[((n*D+d)*H+h)*W+w) * row_stride + c]; `row_stride` (in elements) lets a view address
 *    a channel slice of a wider tensor.  dtype: CBIM_F32 or CBIM_BF16.  Channel counts must be
 *    multiples of 8 (bf16) / 4 (f32) so rows are whole 16-byte chunks.
 *  - Instance-norm statistics are float [N][C][2] = (mean, rstd) pairs.
 *  - Return 0 on success, a negative CBIM_E* code otherwise (never throws);
 *    cbim_last_error_string() gives the thread-local reason.
 */
#ifndef CBIM_HIP_H
#define CBIM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libcbim_hip.so is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define CBIM_F32 0
#define CBIM_BF16 1

#define CBIM_ACT_NONE 0
#define CBIM_ACT_RELU 1   /* nn.ReLU      (model/dim3/utils.py:25) */
#define CBIM_ACT_LRELU 2  /* nn.LeakyReLU (model/dim3/utils.py:26), slope 0.01 */
#define CBIM_ACT_GELU 3   /* nn.GELU      (model/dim3/utils.py:27) */
#define CBIM_ACT_SILU 4   /* nn.SiLU      (model/dim3/utils.py:28) */
#define CBIM_ACT_ELU 5    /* nn.ELU(alpha = 1) (model/dim3/vnet.py:12-16); streaming norm / activation kernels only */

#define CBIM_OK 0
#define CBIM_EINVAL (-1)
#define CBIM_EUNSUPPORTED (-2)
#define CBIM_EWORKSPACE (-3)
#define CBIM_ELAUNCH (-4)

int cbim_version(void);
const char* cbim_backend(void);            /* "hip-gfx950" (product) or "emu" (tests/emu) */
const char* cbim_last_error_string(void);
/* One no-op launch from every code object of the library (they load lazily), with a device re-select + retry
 * if a first launch fails; call once per process before the first real launch. */
int cbim_runtime_warmup(void* stream);

/* ------------------------------------------------------------------------------------------
 * InstanceNorm3d statistics — replaces the statistics half of aten::native_batch_norm reached
 * from nn.InstanceNorm3d(C, eps=1e-4) (model/dim3/conv_layers.py:40-42, model/dim3/utils.py:17).
 * Two launches: per-slab partial (sum, sum of squares) then a fixed-order fp64 finalize, so the
 * result is deterministic.  partials: float [N][P][C][3] records, P = cbim_stats_parts(S, C).
 * ------------------------------------------------------------------------------------------ */
int cbim_stats_parts(int64_t S, int C);
int cbim_instnorm_stats(int dtype, const void* x, int64_t x_stride, int N, int64_t S, int C,
                        float eps, float* partials, int P, float* stats, void* stream);
/* mode 0: (mean, rstd) from (sum x, sum x^2) with `count` elements and eps;
 * mode 1: (sum0/count, sum1/count) — the two means the InstanceNorm backward needs. */
int cbim_stats_finalize(const float* partials, int N, int P, int C, double count, float eps,
                        int mode, float* out, void* stream);

/* Per-(n, c) statistics arithmetic of the MedFormer blocks, n = N*C records, fp64 inside:
 * restat: (mean, rstd) computed with eps_from -> the same moments under eps_to (InstanceNorm3d(eps=1e-4) of
 *   ConvNormAct vs the 1e-5 default of BidirectionAttentionBlock.norm1 / PatchMerging.norm,
 *   conv_layers.py:40 vs medformer_utils.py:112,158);
 * se_fold_fwd: the SEBlock gate s (conv_layers.py:159-175) folded into the next normalisation,
 *   IN(x*s) = (x-mean) * s*rsqrt(var*s^2+eps): out_stats = (mean, s*sqrt(rz2)), rz2 = 1/(var*s^2+eps) (double [n]);
 * se_fold_bwd: ds = S*eps*m2*rz2/s from the dgrad epilogue's second InstanceNorm-backward mean (sums float [n][2]). */
int cbim_stats_restat(const float* stats, float eps_from, float eps_to, float* out, int n, void* stream);
int cbim_se_fold_fwd(const float* stats, const float* se, float eps, float* out_stats, double* rz2, int n,
                     void* stream);
int cbim_se_fold_bwd(const float* sums, const float* se, const double* rz2, float eps, double S, float* ds,
                     int n, void* stream);

/* y = act((x-mean)*rstd) — post-activation ConvNormAct tail (conv_layers.py:51). */
int cbim_norm_act_fwd(int dtype, const void* x, int64_t x_stride, const float* stats, void* y,
                      int64_t y_stride, int N, int64_t S, int C, int act, void* stream);
/* Partial sums for the InstanceNorm backward: with xh=(x-mean)*rstd and
 * g' = masked ? g*act'(xh) : g :  partials[...][c] = (sum g', sum g'*xh). */
int cbim_norm_bwd_reduce(int dtype, const void* g, int64_t g_stride, const void* x, int64_t x_stride,
                         const float* stats, int N, int64_t S, int C, int act, int masked,
                         float* partials, int P, void* stream);
/* dx = rstd*(g' - m1 - xh*m2) [+ add]   (sums = float [N][C][2] = (m1, m2)). */
int cbim_norm_bwd_apply(int dtype, const void* g, int64_t g_stride, const void* x, int64_t x_stride,
                        const float* stats, const float* sums, const void* add, int64_t add_stride,
                        void* dx, int64_t dx_stride, int N, int64_t S, int C, int act, int masked,
                        void* stream);

/* The same three streaming passes with a per-channel AFFINE between normalisation and activation (round 5):
 *     z = gamma_c * (x - mean) * rstd + beta_c,  y = act(z)            affine float [C][2] = (gamma, beta)
 * — nn.BatchNorm3d(affine) of the `norm: bn` constructor branch (model/dim3/utils.py:15-21) and VNet's ContBatchNorm3d
 * (model/dim3/vnet.py:22-33); `stats` then holds the BATCH statistics, repeated per image.
 * bwd_reduce: partial sums of g' = g * act'(z) and g' * xh per (n, c) (their batch totals are d beta, d gamma);
 * bwd_apply:  dx = rstd * (gamma g' - m1 - xh m2) with sums = (m1, m2) = gamma * (mean g', mean g' xh) given by the
 *             caller — nothing is divided by gamma, so a zero or tiny gamma is exact. */
int cbim_norm_affine_act_fwd(int dtype, const void* x, int64_t x_stride, const float* stats, const float* affine,
                             void* y, int64_t y_stride, int N, int64_t S, int C, int act, void* stream);
int cbim_norm_affine_bwd_reduce(int dtype, const void* g, int64_t g_stride, const void* x, int64_t x_stride,
                                const float* stats, const float* affine, int N, int64_t S, int C, int act, int masked,
                                float* partials, int P, void* stream);
int cbim_norm_affine_bwd_apply(int dtype, const void* g, int64_t g_stride, const void* x, int64_t x_stride,
                               const float* stats, const float* affine, const float* sums, void* dx, int64_t dx_stride,
                               int N, int64_t S, int C, int act, int masked, void* stream);

/* The per-channel arithmetic around them (round 6: one launch each instead of ~25 / ~12 float64 ATen launches on [C] vectors per
 * BatchNorm — 660 launches of the shipped VNet step):
 * bn_finish_fwd: per-image statistics st float32 [N][C][2] = (mean, rstd(eps)) -> the batch statistics of F.batch_norm(training=True)
 *                (mean over N of the means; biased variance pooled over N S values, in float64), the running-statistics update
 *                running = (1 - momentum) running + momentum (mean | var n/(n-1)) when the pointers are given, `stats_out`
 *                [N][C][2] = (mean_b, rstd_b) repeated per image and affine_out [C][2] = (gamma | 1, beta | 0);
 *                use_batch = 0: the statistics are the running ones (nn.BatchNorm3d in eval()).
 * bn_finish_bwd: sums float32 [N][C][2] = per-image means of g' and g' xh -> d gamma = cnt mean(g' xh), d beta = cnt mean(g'),
 *                sums_out [N][C][2] = gamma (mean g', mean g' xh) (zeros under running statistics) for bwd_apply. */
int cbim_bn_finish_fwd(const float* st, int N, int C, double S, float eps, float momentum, float* running_mean, float* running_var,
                       int use_batch, const float* gamma, const float* beta, float* stats_out, float* affine_out, void* stream);
int cbim_bn_finish_bwd(const float* sums, int N, int C, double cnt, const float* affine, int use_batch, float* dgamma, float* dbeta,
                       float* sums_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.MaxPool3d(scale) — unet_utils.py:36 (kernel = stride = scale, floor mode).
 * idx: uint8 [N][Do][Ho][Wo][C] window position of the first maximum (scan order d,h,w).
 * ------------------------------------------------------------------------------------------ */
int cbim_maxpool3d_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int D, int H, int W,
                       int C, int sD, int sH, int sW, void* stream);
int cbim_maxpool3d_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int D, int H,
                       int W, int C, int sD, int sH, int sW, void* stream);

/* ------------------------------------------------------------------------------------------
 * up_block head — unet_utils.py:68-71: F.interpolate(low, size=skip.shape[2:], 'trilinear',
 * align_corners=True) then torch.cat([skip, up], 1), written in one pass.
 * out: [N][D][H][W][Cs+Cl] with channels [0,Cs) = skip, [Cs,Cs+Cl) = upsampled low
 * (skip_first=1, UNet) or the opposite order (skip_first=0, MedFormer medformer_utils.py:358).
 * bwd: dskip = slice copy, dlow = exact transpose of the interpolation (gather, no atomics).
 * ------------------------------------------------------------------------------------------ */
int cbim_upcat_fwd(int dtype, const void* low, const void* skip, void* out, int N, int Dl, int Hl,
                   int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream);
/* cbim_upcat_fwd + the InstanceNorm statistics (mean, rstd; eps) of its output in the same pass: partials float
 * [N][P][Cs+Cl][3] with P = cbim_stats_parts(D*H*W, Cs+Cl), stats float [N][Cs+Cl][2]. */
int cbim_upcat_fwd_stats(int dtype, const void* low, const void* skip, void* out, int N, int Dl, int Hl, int Wl,
                         int Cl, int D, int H, int W, int Cs, int skip_first, float eps, float* partials, int P,
                         float* stats, void* stream);
/* The decoder level's first block without the stored concatenation (round 3): statistics of the virtual
 * up-sampled tensor (cbim_up_stats: partials float [N][P][Cl][3], P = cbim_stats_parts(S, Cl)), then
 * out = act(IN([skip | up(low)])) in one pass (cbim_upcat_act_fwd; stats float [N][Cs+Cl][2] in the
 * concatenation's channel order), and in the backward dx = rstd*(g - m1 - xh*m2) of the re-formed
 * concatenation written as dskip and — through dup_scratch [N][D][H][W][Cl] and the transposed
 * trilinear gather — dlow (cbim_upcat_norm_bwd).  unet_utils.py:69-71 + conv_layers.py:40-49. */
int cbim_up_stats(int dtype, const void* low, int N, int Dl, int Hl, int Wl, int Cl, int D, int H, int W,
                  float eps, float* partials, int P, float* stats, void* stream);
/* The same statistics from the COARSE grid (sum and sum of squares of the up-sampled tensor as a weighted sum and a 27-point
 * stencil over `low`; fp32 interpolation, no rounding to the storage type): partials float [N][P][Cl][3] with
 * P = cbim_up_gram_parts(Dl, Hl, Wl). */
int cbim_up_gram_parts(int Dl, int Hl, int Wl);
int cbim_up_stats_gram(int dtype, const void* low, int N, int Dl, int Hl, int Wl, int Cl, int D, int H, int W,
                       float eps, float* partials, int P, float* stats, void* stream);
int cbim_upcat_act_fwd(int dtype, const void* low, const void* skip, const float* stats, void* out, int N,
                       int Dl, int Hl, int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, int act,
                       void* stream);
int cbim_upcat_norm_bwd(int dtype, const void* g, const void* low, const void* skip, const float* stats,
                        const float* sums, void* dskip, void* dlow, void* dup_scratch, int N, int Dl, int Hl,
                        int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream);
/* One axis of the transposed trilinear interpolation (the adjoint of F.interpolate(trilinear, align_corners=True)
 * factorises over W, H, D; unet_utils.py:69 / medformer.py:91 in the backward): src [outer][F][src_row] elements of
 * which `inner` starting at c_off are reduced -> dst [outer][L][inner] dense, L <= F.  vec 0: items of one 16-byte
 * chunk of dtype; vec 1: scalar float32 items (NCDHW planes).  cbim_upcat_bwd / cbim_upcat_norm_bwd[_tile] called with
 * dlow == NULL leave the reduction of the fine-resolution gradient to three of these passes. */
int cbim_lin_adjoint_axis(int dtype, int vec, const void* src, int64_t src_row, int64_t c_off, void* dst,
                          int64_t outer, int F, int L, int64_t inner, void* stream);
/* The same three operations on LDS tiles (up_tile_kernels.hip): a workgroup stages the coarse box of a 4x8x8
 * fine tile once and interpolates from LDS.  CBIM_EUNSUPPORTED when the box does not fit the LDS budget
 * (the caller then uses the entry points above).  cbim_up_stats_tile's partials have
 * cbim_up_tile_parts(D, H, W) records per image. */
int cbim_up_tile_parts(int D, int H, int W);
int64_t cbim_up_tile_min_tiles(int64_t v);   /* process-wide knob (tests): fewest fine tiles per image the tiled kernels take; returns the old value */
int cbim_up_stats_tile(int dtype, const void* low, int N, int Dl, int Hl, int Wl, int Cl, int D, int H, int W,
                       float eps, float* partials, int P, float* stats, void* stream);
int cbim_upcat_act_fwd_tile(int dtype, const void* low, const void* skip, const float* stats, void* out, int N,
                            int Dl, int Hl, int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, int act,
                            void* stream);
int cbim_upcat_norm_bwd_tile(int dtype, const void* g, const void* low, const void* skip, const float* stats,
                             const float* sums, void* dskip, void* dlow, void* dup_scratch, int N, int Dl,
                             int Hl, int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream);
int cbim_upcat_bwd(int dtype, const void* dout, void* dlow, void* dskip, int N, int Dl, int Hl,
                   int Wl, int Cl, int D, int H, int W, int Cs, int skip_first, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3-D convolution as implicit GEMM on the matrix cores — replaces aten::convolution /
 * convolution_backward reached from nn.Conv3d in ConvNormAct (conv_layers.py:29-38,46-53),
 * stride 1, dilation 1, groups 1, bias-free.
 * ------------------------------------------------------------------------------------------ */
typedef struct cbim_conv_desc {
  int dtype;            /* activation dtype */
  int N;
  int Di, Hi, Wi, Cin;  /* input extent / channels  */
  int Do, Ho, Wo, Cout; /* output extent / channels */
  int kD, kH, kW;       /* kernel extent            */
  int pD, pH, pW;       /* leading zero padding     */
  int act;              /* activation fused on the input load (with in_stats) / mask rule */
} cbim_conv_desc;

/* Weights are re-laid into MFMA B-fragment order once per optimizer step.
 * mode 0: forward  (w[co][ci][tap]           -> Cin  is K, Cout is N)
 * mode 1: dgrad    (w[ci][co][flipped tap]   -> Cout is K, Cin  is N); desc is the FORWARD desc. */
size_t cbim_conv3d_packed_bytes(const cbim_conv_desc* fwd_desc, int mode);
int cbim_conv3d_pack_weights(const cbim_conv_desc* fwd_desc, int mode, const float* w, void* packed,
                             void* stream);
/* Both layouts in one launch (training: the dgrad layout is needed in the backward of the same step). */
int cbim_conv3d_pack_weights_both(const cbim_conv_desc* fwd_desc, const float* w, void* packed_fwd,
                                  void* packed_dgrad, void* stream);
/* The rounding residue w - bf16(w) in the same two layouts (either pointer may be NULL; bf16 descriptors only): the second
 * weight image of a GEMM that keeps fp32 accuracy, x . w = x . w_hi + x . w_lo (cbim_token_linear's w_lo_packed). */
int cbim_conv3d_pack_weights_lo(const cbim_conv_desc* d, const float* w, void* packed_fwd, void* packed_dgrad,
                                void* stream);
/* All convolution weights of a model in one launch.  The host fills one cbim_pack_item per weight with
 * cbim_conv3d_pack_item_fill (w1 != NULL: forward output channels >= rows0 come from w1 — the Cout-concatenated
 * conv1|shortcut pair of BasicBlock, conv_layers.py:86-94, packed without a torch.cat; block_begin = running sum of the
 * previous items' n_blocks; one workgroup per 32 output x 32 (bf16) / 16 (fp32) input channels, transposed through
 * LDS), copies the array to the device and launches it once per optimizer step. */
typedef struct cbim_pack_item {
  const float* w0; const float* w1; void* p0; void* p1;
  int64_t total0, total1;
  int rows0, Cout, Cin, taps, BN0, nch0, BN1, nch1, block_begin, n_blocks, dtype, _pad;
} cbim_pack_item;
int cbim_conv3d_pack_item_fill(const cbim_conv_desc* fwd_desc, const float* w0, const float* w1, int rows0,
                               void* packed_fwd, void* packed_dgrad, int block_begin, cbim_pack_item* out);
/* max_taps: the largest `taps` of the items (sizes the kernel's LDS: taps * 2 KiB) */
int cbim_conv3d_pack_weights_table(const cbim_pack_item* items_dev, int n_items, int total_blocks, int max_taps,
                                   void* stream);
/* Tuning knob: output voxels per image from which the Cin = 32 -> Cout <= 32 3x3x3 layers (bf16) run on the
 * "weights in registers" kernel (conv_r32.hip) instead of k_conv_igemm; v < 0 only queries.  Returns the previous
 * value (default 262144 = 64^3).  The two kernels compute the same function (tests lower it to cover small shapes). */
int64_t cbim_conv_r32_min_voxels(int64_t v);
/* Tile depth of that kernel: 8 = one 512-thread workgroup per CU on 8x8x8 tiles (default), 4 = two 256-thread workgroups
 * per CU on 4x8x8 tiles; any other value only queries.  Returns the previous value. */
int cbim_conv_r32_tile_depth(int td);
/* Round-4 form of that kernel (conv_rw.hip: plane-major buffer-addressed LDS-DMA, statistics sums in LDS, optional 64 output
 * channels per workgroup): on = 0 keeps every call on k_conv3_r32 / k_conv_igemm, wide = 0 keeps 32-channel workgroups; a
 * negative value leaves a switch as it is.  Returns the previous (on | wide << 1).  Process-wide knob for tests and tools
 * (defaults 1, 1); both kernels compute the same function. */
int cbim_conv_rw_enable(int on, int wide);
/* Round 6 — 3x3x3 layers with Cout in multiples of 48 that the 32-channel kernels do not take (Cin or Cout not a multiple of
 * 32; Cin any multiple of 8): k_conv3_rw48 of conv_rw.hip, 48 output channels per workgroup with the tile's twelve (16-cout
 * block, h-pair) jobs dealt 2 + 1 to the two waves of every SIMD.  Replaces aten::convolution / convolution_backward(input) of
 * the monai UnetResBlock convolutions of SwinUNETR (/root/reference/model/dim3/swin_unetr.py:129-228, feature_size 48).
 * cbim_conv_rw48_enable: on = 0 keeps those layers on k_conv_igemm, 1 (default; env CBIM_CONV_RW48) takes them when the launch
 * has >= 128 workgroups, 2 always (tests); < 0 only queries; returns the previous value.
 * cbim_conv_rw48_takes: 1 when cbim_conv3d_igemm runs `desc` on that kernel for an input used as it is (in_stats = NULL) — the
 * caller then materialises act(IN(x)) once and may pass it as the dgrad's mask tensor with act = ReLU or LeakyReLU. */
int cbim_conv_rw48_enable(int on);
int cbim_conv_rw48_takes(const cbim_conv_desc* desc);
/* bf16 1x1x1 convolutions on the row-GEMM kernel (conv_pw.hip) instead of k_conv_igemm: on = 0 | 1, < 0 only queries; returns the
 * previous value (default 1).  The two kernels compute the same function. */
int cbim_conv_pw_enable(int on);
/* Tile configuration the launcher picks for `desc`: out = {MT, NTL, tD, tH} (m-tiles per wave,
 * n-tiles per wave, tile depth, tile height; tile width is 8).  Informational (profiling labels). */
int cbim_conv3d_tile_config(const cbim_conv_desc* desc, int out[4]);
/* Kernel launched by this thread's last cbim_conv3d_igemm call: 0 = k_conv_igemm, 1 = k_conv3_r32, 2 = k_conv3_rw, 3 = k_conv3_rw split-K + finish, 4 = k_conv_pw, 5 = k_conv3_rw48 (profiling labels). */
int cbim_conv3d_last_kernel(void);
/* Records per sample of the partial-sum buffer `partials` (one per persistent workgroup, or per finish part when the
 * launcher splits K). */
int cbim_conv3d_num_tiles(const cbim_conv_desc* desc);
/* y = conv(xform(x), w) [+ res];  xform(x) = act((x-mean)*rstd) when in_stats != NULL (zero
 * padding is applied AFTER the transform, conv_layers.py:48-49).  Optional epilogue:
 *   mask_x != NULL : y *= act'((mask_x-mean)*rstd)      (dgrad through a pre-activation)
 *   partials != NULL: per-tile (sum u, sum u*v) over the stored values, float
 *                     [N][tiles][Cout][3] records; v = u (forward: InstanceNorm statistics of y) or
 *                     v = xh of mask_x (dgrad: the two InstanceNorm-backward sums).
 * For dgrad pass the dgrad desc (input = dy extent/Cout, output = x extent/Cin, p' = k-1-p).
 * x2 != NULL: the input is the channel concatenation [x | x2] without materialising it; channels
 * >= cin_split (a multiple of the 64-byte chunk) come from x2 — the two dgrads of a BasicBlock that
 * share act(IN(x)) (conv1 and the shortcut conv, conv_layers.py:86-94) run as ONE K-concatenated GEMM. */
/* mask_x with mask_stats == NULL (act = ReLU only): the mask tensor is the ACTIVATED tensor a = relu(IN(x)) the caller
 * materialised once — act'(xh) = [a > 0] and a itself stands for xh in the second InstanceNorm-backward sum. */
int cbim_conv3d_igemm(const cbim_conv_desc* desc, const void* x, int64_t x_stride, const void* x2,
                      int64_t x2_stride, int cin_split,
                      const float* in_stats, const void* w_packed, const void* res,
                      int64_t res_stride, const void* mask_x, int64_t mask_stride,
                      const float* mask_stats, void* y, int64_t y_stride, float* partials,
                      void* workspace, size_t ws_bytes, void* stream);
/* Bytes of scratch cbim_conv3d_igemm needs for `desc` (0 unless the launcher splits K: layers with
 * too few output tiles to fill the chip share the Cin chunks over blockIdx.z and a finish kernel sums
 * the fp32 partials in fixed order before the residual / mask / statistics epilogue). */
size_t cbim_conv3d_igemm_workspace(const cbim_conv_desc* desc);
/* dw[co][ci][tap] (fp32, natural nn.Conv3d layout) = sum_v dy[v][co] * xform(x)[v+tap][ci].
 * desc is the FORWARD desc.  workspace: cbim_conv3d_wgrad_workspace(desc) bytes.
 * dy2 != NULL: output channels >= cout_split (a multiple of 32) take their gradient from dy2
 * (Cout-concatenated conv1 + shortcut conv sharing one staged input halo).
 * x2 != NULL: input channels >= cin_split (a multiple of 32) come from x2 — the virtual concatenation
 * [skip | upsampled] of up_block (unet_utils.py:69-71); bf16 3x3x3 on raw inputs only (in_stats NULL).
 * bf16 3x3x3 convolutions whose input is used as it is (in_stats NULL, channel counts in multiples of 32,
 * extents >= 8) run on k_wgrad_r32 (conv_wgrad_r32.hip: LDS-DMA double-buffered 8x8x8 tiles, the 27 tap
 * accumulators in registers); everything else on k_conv_wgrad. */
size_t cbim_conv3d_wgrad_workspace(const cbim_conv_desc* fwd_desc);
int cbim_conv3d_wgrad_last_kernel(void);   /* 0 = k_conv_wgrad, 1 = k_wgrad_r32 (profiling labels) */
int cbim_wgrad_r32_enable(int on);          /* process-wide knob (tests, tools): 0 keeps every wgrad on k_conv_wgrad; returns the old value */
int cbim_wgrad_r32_waves(int waves);        /* process-wide knob (tests, tools): 8 or 4 waves per workgroup; returns the old value */
int cbim_conv3d_wgrad(const cbim_conv_desc* fwd_desc, const void* x, int64_t x_stride, const void* x2,
                      int64_t x2_stride, int cin_split,
                      const float* in_stats, const void* dy, int64_t dy_stride, const void* dy2,
                      int64_t dy2_stride, int cout_split, float* dw,
                      void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stem: inconv.conv1 = raw nn.Conv3d(in_ch, base, k, pad k//2, bias=False) (unet_utils.py:14,19)
 * reading the model input in the caller's NCDHW fp32 layout, writing NDHWC.
 * ------------------------------------------------------------------------------------------ */
int cbim_stem_conv_fwd(int dtype_out, const float* x_ncdhw, const float* w, void* y, int N, int Cin,
                       int Di, int Hi, int Wi, int Cout, int kD, int kH, int kW, int pD, int pH,
                       int pW, int Do, int Ho, int Wo, void* stream);
size_t cbim_stem_conv_wgrad_workspace(int N, int Cin, int Cout, int kD, int kH, int kW, int Do, int Ho, int Wo);
int cbim_stem_conv_wgrad(int dtype, const float* x_ncdhw, const void* dy, float* dw, int N, int Cin,
                         int Di, int Hi, int Wi, int Cout, int kD, int kH, int kW, int pD, int pH,
                         int pW, int Do, int Ho, int Wo, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Head: outc = nn.Conv3d(base, classes, 1) with bias (unet.py:47), NDHWC in, NCDHW fp32 logits out.
 * ------------------------------------------------------------------------------------------ */
int cbim_head_fwd(int dtype, const void* x, const float* w, const float* b, float* logits, int N,
                  int64_t S, int Cin, int K, void* stream);
size_t cbim_head_bwd_workspace(int64_t S, int N, int Cin, int K);
int cbim_head_bwd(int dtype, const void* x, const float* w, const float* dlogits, void* dx, float* dw,
                  float* db, int N, int64_t S, int Cin, int K, void* workspace, size_t ws_bytes,
                  void* stream);
/* process-wide switch (tests, A/B): 0 = keep the VALU head backward where the matrix-core kernel (bf16, K <= 16,
 * Cin a multiple of 32 up to 128, S a multiple of 32) would run; returns the old value. */
int cbim_head_mfma_enable(int on);
/* the same switch for the stem (one input channel, bf16 rows of 32..128 channels: k_stem_fwd_mfma, k_stem_wgrad_mfma) */
int cbim_stem_mfma_enable(int on);

/* ------------------------------------------------------------------------------------------
 * Loss: nn.CrossEntropyLoss(weight)(logits, label) + DiceLoss()(logits, label)
 * (train.py:80-81,212; training/losses.py:18-58) in one read of the logits.
 * logits float [N][C][S] (NCDHW), labels int64 [N][S].
 * fwd: out float[4]: out[0]=CE, out[1]=Dice, out[2]=CE+Dice, out[3]=number of labels outside [0,C) (the reference
 *      raises for those in scatter_/CrossEntropyLoss; here such a voxel is excluded from CE / TP / CNT, never used
 *      as an index, and counted so the host can raise); coef float [3C + 1]: [2][C] = (dL/dTP_c, dL/dSP_c) and
 *      coef[2C] = 1/sum_i w[y_i] for the backward, coef[2C+1 .. 3C] = the per-class terms 1 - dice_c
 *      (DiceLoss(reduce=False), losses.py:48-50).  workspace: cbim_dice_ce_workspace(N,C,S).
 * bwd: dlogits = grad_out[0] * d(CE+Dice)/dlogits  (grad_out is a DEVICE scalar).
 * ------------------------------------------------------------------------------------------ */
size_t cbim_dice_ce_workspace(int N, int C, int64_t S);
int cbim_dice_ce_fwd(const float* logits, const int64_t* labels, const float* weight, int N, int C,
                     int64_t S, float* out, float* coef, void* workspace, size_t ws_bytes, void* stream);
int cbim_dice_ce_bwd(const float* logits, const int64_t* labels, const float* weight, const float* coef,
                     const float* grad_out, float* dlogits, int N, int C, int64_t S, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-device augmentation — training/augmentation.py (tensor_img [1,C,D,H,W] fp32, tensor_lab
 * [1,1,D,H,W] int8|int64), called per sample from the dataset (training/dataset/dim3/
 * dataset_amos_ct.py:121-153).  Random parameters are drawn by the HOST in the reference's order.
 * ------------------------------------------------------------------------------------------ */
/* F.affine_grid(theta[3x4], size, align_corners=True) + F.grid_sample(img, bilinear, zeros) +
 * F.grid_sample(lab, nearest, zeros).long()  (augmentation.py:283-289) in one pass; the output is
 * the window [od0:od0+Do, oh0:oh0+Ho, ow0:ow0+Wo] of the full-size result, i.e. fused with the
 * centre crop_3d that follows (:320-343).  theta12 is a HOST pointer (12 floats). lab may be NULL. */
int cbim_affine_sample3d(const float* img, const void* lab, int lab_bytes, const float* theta12,
                         float* out_img, int64_t* out_lab, int C, int Di, int Hi, int Wi, int Do, int Ho,
                         int Wo, int od0, int oh0, int ow0, void* stream);
/* crop_3d (augmentation.py:320-343) of image and label (label keeps its dtype). */
int cbim_crop3d(const float* img, const void* lab, int lab_bytes, float* out_img, void* out_lab, int C,
                int Di, int Hi, int Wi, int Do, int Ho, int Wo, int d0, int h0, int w0, void* stream);
/* per-channel (min, max, mean, unbiased std): float [C][4]  (gamma :117-124, contrast :150-155). */
size_t cbim_chan_stats_workspace(int C, int64_t S);
int cbim_chan_stats(const float* x, int C, int64_t S, float* stats, void* workspace, size_t ws_bytes,
                    void* stream);
/* point-wise intensity transforms; prm/st/st2/noise are DEVICE pointers, prm = 2 floats per channel:
 * mode 0 y=x*a+b (brightness_multiply :84-101 / brightness_additive :67-82)
 * mode 1 y=pow((x-min)/rng,g)*rng+min (gamma :126)      mode 2 y=(x-mean_y)/std_y*std_x+mean_x (:128-130)
 * mode 3 y=clamp((x-mean)*f+mean,min,max) (contrast :158-161)   mode 4 y=x+noise*std+mean (gaussian_noise :17) */
int cbim_intensity(const float* x, float* y, int C, int64_t S, int mode, const float* prm,
                   int prm_per_channel, const float* st, const float* st2, int st_per_channel,
                   const float* noise, void* stream);
/* gaussian_blur (augmentation.py:46-64): the dense normalised k^3 Gaussian with zero padding, done as
 * three 1-D passes with the normalised 1-D kernel g_host[k] (HOST pointer); tmp = one scratch volume. */
int cbim_gaussian_blur3d(const float* x, float* y, float* tmp, int C, int D, int H, int W,
                         const float* g_host, int k, void* stream);

/* ------------------------------------------------------------------------------------------
 * MedFormer pieces — /root/reference/model/dim3/medformer_utils.py.
 * ------------------------------------------------------------------------------------------ */
/* Depthwise k^3 convolution, stride 1, padding k//2 — nn.Conv3d(C, C, k, groups=C, bias=False) of
 * DepthwiseSeparableConv.depthwise (conv_layers.py:137-145) and the MBConv depthwise ConvNormAct
 * (conv_layers.py:211).  y = sum_t w[c][t] * a(x)[l+off(t)], a = act((x-mean)*rstd) when in_stats
 * (float [N][C][2]) is given, zero outside the volume.  bias (float [N][C], optional) is added to every
 * in-range input; flip=1 indexes the taps in reverse (the data gradient).  w: float [C][kD*kH*kW]. */
int cbim_dwconv3d(int dtype, const void* x, int64_t x_stride, const float* in_stats, int act,
                  const float* bias, const float* w, int flip, void* y, int64_t y_stride, int N, int D,
                  int H, int W, int C, int kD, int kH, int kW, void* stream);
/* 3x3x3-class depthwise convolutions on the LDS-tiled kernel (round 4) instead of the streaming one: on = 0 | 1, < 0 only queries;
 * returns the previous value (default 1).  Same function; in bf16 the transformed input is rounded to bf16. */
int cbim_dwconv_lds_enable(int on);
/* dw[c][t] = sum_{n,l} a(x)[n,l+off(t),c] * (dy[n,l,c] + dy_bias[n][c]); deterministic two-stage sum. */
size_t cbim_dwconv3d_wgrad_workspace(int N, int D, int H, int W, int C, int kD, int kH, int kW);
int cbim_dwconv3d_wgrad(int dtype, const void* x, int64_t x_stride, const float* in_stats, int act,
                        const void* dy, int64_t dy_stride, const float* dy_bias, float* dw, int N, int D,
                        int H, int W, int C, int kD, int kH, int kW, void* workspace, size_t ws_bytes,
                        void* stream);
/* PatchMerging's strided slices + concat (medformer_utils.py:163-171):
 * dst[n,d',h',w',((i*sH+j)*sW+k)*C+c] = src[n,d'*sD+i,h'*sH+j,w'*sW+k,c]; inverse=1 is the adjoint copy
 * (src = merged tensor, dst = [N,D,H,W,C]).  D,H,W are always the UNMERGED extents. */
int cbim_space_to_depth(int dtype, const void* src, void* dst, int N, int D, int H, int W, int C, int sD,
                        int sH, int sW, int inverse, void* stream);
/* The same gather / scatter with the FINE-grid tensor (the source for inverse = 0, the destination for inverse = 1) given with a
 * row stride `fine_stride` >= C elements: round 6 — monai's UnetrUpBlock in SwinUNETR (/root/reference/model/dim3/swin_unetr.py:
 * 176-228: ConvTranspose3d(k = s = 2) -> torch.cat((up, skip))) scatters the transposed convolution's GEMM output straight into the
 * first C channels of the concatenated tensor, and its backward gathers from that channel slice of the gradient. */
int cbim_space_to_depth_strided(int dtype, const void* src, void* dst, int N, int D, int H, int W, int C, int sD,
                                int sH, int sW, int inverse, int64_t fine_stride, void* stream);
/* BidirectionAttention core (medformer_utils.py:63-97) on qv = [q | v] rows ([N][L][2*inner], row stride
 * qv_stride), inner = heads*dh, channel c = d*heads + h ("(dim_head heads)", :43-51):
 *   attn = q_f q_m^T * scale [L x M]; feat_out = softmax_M(attn) v_m; map_out = softmax_L(attn)^T v_f.
 * mq, mv, map_out, d_*: float [N][M][inner].  colstat: float [N][heads][M][2] column (max, sum) kept
 * for the backward.  Any dh >= 1 and M <= cbim_attn_wide_max_codes() (= 128): dh in {8,16,32} with M <= 64 runs the
 * register-resident kernels (bf16, dh 32, M 64: the MFMA kernel), every other size — e.g. config/acdc/medformer_3d.yaml
 * (72 codes, dh 64|80) and config/lits/medformer_3d.yaml (num_heads 1: dh up to 320) — the LDS-tiled kernels of
 * attn_wide.hip. */
int cbim_attn_wide_max_codes(void);
size_t cbim_bidir_attn_workspace(int N, int L, int heads, int dh, int M);
int cbim_bidir_attn_fwd(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                        void* feat_out, float* map_out, float* colstat, int N, int L, int heads, int dh,
                        int M, float scale, void* workspace, size_t ws_bytes, void* stream);
int cbim_bidir_attn_bwd(int dtype, const void* qv, int64_t qv_stride, const float* mq, const float* mv,
                        const float* colstat, const float* map_out, const void* d_feat_out,
                        const float* d_map_out, void* d_qv, float* d_mq, float* d_mv, int N, int L,
                        int heads, int dh, int M, float scale, void* workspace, size_t ws_bytes,
                        void* stream);
/* Round 6 — the semantic-map side of BidirectionAttentionBlock (medformer_utils.py:63-97, 102-138): norm2 (nn.InstanceNorm3d over
 * the <= 128 map positions), the map_qv / map_out 1x1x1 projections and the `map_out + semantic_map` residual, forward and
 * backward, as launches of ONE small float32 GEMM (map_kernels.hip) instead of the aten::mm / layer_norm / add / slice launches
 * of rounds 1-5:   OUT[o][n] = sum_k A[o][k] X[k][n]  (+ R[o][n]),  batched.
 *   a_t = 0: A[o*lda + k];  1: A[k*lda + o], and the rows o >= a_split of A come from A2[k*lda + o - a_split] when A2 != NULL
 *   x_t = 0: X[k*ldx + n];  1: X[n*ldx + k], and k >= x_split from X2[n*ldx + k - x_split] when X2 != NULL
 *   o_t = 0: OUT[o*ldo + n]; 1: OUT[n*ldo + o], and o >= o_split to OUT2[n*ldo + o - o_split] when OUT2 != NULL
 *   (the attention kernels take and return q and v as two [M, inner] tensors)
 *   R: residual in OUT's [o][n] layout (o_t = 0).  *_batch: element strides between the images (0 = shared, e.g. a weight).
 *   reduce_batch = 1: the products of all images are summed into one OUT (a weight gradient).
 *   ln_mode = 1 (x_t = 0, Nn <= 128): the rows of X are normalised over n on load — (x - mean) * rsqrt(var + eps), biased
 *     variance; the normalised rows are also written to Xn (X's layout) and their rstd to rstd_out[batch][K].
 *   XH != NULL (o_t = 0, Nn <= 128): the epilogue applies the normalisation's backward to each row,
 *     OUT = rstd_in[o] * (acc - mean_n(acc) - XH[o][n] * mean_n(acc * XH[o][n])) (+ R), XH in OUT's layout.
 * Fixed summation order (bit-reproducible); launch-only on `stream`; caller-owned memory. */
typedef struct {
  const float* A; const float* A2; int64_t lda, a_batch; int a_t, a_split;
  const float* X; const float* X2; int64_t ldx, x_batch; int x_t, x_split;
  float* OUT; float* OUT2; int64_t ldo, o_batch; int o_t, o_split;
  const float* R; int64_t ldr, r_batch;
  int O, K, Nn, batch, reduce_batch, ln_mode;
  float eps; int _pad;
  float* Xn; float* rstd_out;
  const float* XH; const float* rstd_in;
} cbim_map_gemm_desc;
int cbim_map_gemm(const cbim_map_gemm_desc* desc, void* stream);
/* Round 6 — SEBlock.excitation on the [N, C] channel means (conv_layers.py:159-175: 1x1x1 conv -> ReLU -> 1x1x1 conv -> Sigmoid; one
 * per MBConv of MedFormer): gate = sigmoid(W2 relu(W1 mean + b1) + b2) with W1 [H][C], W2 [C][H], biases optional, all float32.
 * fwd writes z1 = W1 mean + b1 [N][H] (kept for the backward) and gate [N][C] (two launches).  bwd (three launches) takes d gate and
 * writes dW1 [H][C], db1 [H], dW2 [C][H], db2 [C] (sums over the N images; db* optional) and d mean [N][C] (optional); dz1_ws: float
 * [N][H] scratch.  Replaces the aten::linear / relu / sigmoid / mm / sigmoid_backward / threshold_backward launches of rounds 1-5. */
int cbim_se_gate_fwd(const float* mean, const float* W1, const float* b1, const float* W2, const float* b2, float* z1, float* gate,
                     int N, int C, int H, void* stream);
int cbim_se_gate_bwd(const float* dgate, const float* gate, const float* z1, const float* mean, const float* W1, const float* W2,
                     float* dz1_ws, float* dW1, float* db1, float* dW2, float* db2, float* dmean, int N, int C, int H, void* stream);
/* BidirectionAttention core as matrix products (round 6) for the head / map sizes the register-resident kernels do not take — one wide
 * head (config/lits/medformer_3d.yaml: num_heads = 1, d_head = 128 / 256 / 320) or a few heads with more than 64 codes (config/acdc:
 * 4 heads, 72 codes) — the softmax side between the row GEMMs (cbim_token_linear / cbim_token_linear_wgrad), medformer_utils.py:77-90.
 * S float32 [L][H][M] = Q MQ^T of all heads from ONE row GEMM (weight = the block matrix of the heads), M codes in multiples of 8 up
 * to 128, H = 1 | 2 | 4 | 8 heads with H <= 64 / lanes-per-row (lanes-per-row = 4 | 8 | 16 for M <= 32 | 64 | 128).
 *   cbim_awg_rows: P = softmax over the codes of each (voxel, head) row of scale*S (bf16 [L][H][M], :80) + the column records
 *                  rec float32 [ceil(L H / CBIM_AWG_ROWS)][H][M][2] = (max, sum exp) of every (head, code) over the workgroup's rows
 *   cbim_awg_cols: lse[h][m] from the records (fixed order), C = exp(scale*S - lse) (bf16 [L][H][M]): the softmax over the voxels (:82)
 *   cbim_awg_ds:   dS = scale (P o (dP - rowsum(dP o P)) + C o (dC - colsum[h][m])) (bf16 [L][H][M]): the backward of both softmaxes;
 *                  dP = dfeat_out MV^T and dC = FV dmap_out^T float32 [L][H][M] from the row GEMM; colsum[h][m] = the sum over the head's
 *                  channels c = d H + h of dmo[m][c] mo[m][c], dmo / mo float32 [M][inner]; colsum_ws: H M floats of scratch.
 * Replaces torch.softmax(dim=-1) / torch.softmax(dim=-2) and their backward inside BidirectionAttention.forward. */
#define CBIM_AWG_ROWS 256
int cbim_awg_rows(const float* S, int64_t L, int heads, int M, float scale, void* P, float* rec, void* stream);
int cbim_awg_cols(const float* S, int64_t L, int heads, int M, float scale, const float* rec, float* lse, void* C, void* stream);
int cbim_awg_ds(const float* dP, const void* P, const float* dC, const void* C, const float* dmo, const float* mo, int inner, int64_t L,
                int heads, int M, float scale, float* colsum_ws, void* dS, void* stream);
/* SemanticMapGeneration tail (medformer_utils.py:218-228) on fw rows = [feat (C) | weight logits (M)]:
 * map[n][c][j] = sum_l feat[l,c] * softmax_L(logit[:,j])[l].  colstat: float [N][M][2].
 * M <= cbim_attn_wide_max_codes(). */
size_t cbim_colsoftmax_pool_workspace(int N, int L, int C, int M);
int cbim_colsoftmax_pool_fwd(int dtype, const void* fw, int64_t fw_stride, float* map, float* colstat,
                             int N, int L, int C, int M, void* workspace, size_t ws_bytes, void* stream);
int cbim_colsoftmax_pool_bwd(int dtype, const void* fw, int64_t fw_stride, const float* map,
                             const float* colstat, const float* dmap, void* dfw, int64_t dfw_stride, int N,
                             int L, int C, int M, void* stream);
/* F.interpolate(aux_out, size, 'trilinear', align_corners=True) on float32 NCDHW planes
 * (medformer.py:91) and its adjoint. */
int cbim_trilinear_planes_fwd(const float* x, float* y, int planes, int Di, int Hi, int Wi, int Do, int Ho,
                              int Wo, void* stream);
int cbim_trilinear_planes_bwd(const float* dy, float* dx, int planes, int Di, int Hi, int Wi, int Do, int Ho,
                              int Wo, void* stream);

/* Post-norm residual tail of monai's UnetResBlock (SwinUNETR encoder/decoder blocks, swin_unetr.py:129-226):
 * y = act(IN(a; stats_a) + (stats_b ? IN(b; stats_b) : b)).  Backward: g = dy*act'(pre) (pre recomputed);
 * reduce -> partial records [N][P][C][3] for cbim_stats_finalize(mode 1) -> sums (mean g, mean g*xhat);
 * apply -> da = rstd_a*(g - m1 - xhat_a*m2), db likewise or g. */
int cbim_resnorm_fwd(int dtype, const void* a, int64_t a_stride, const float* stats_a, const void* b,
                     int64_t b_stride, const float* stats_b, void* y, int64_t y_stride, int N, int64_t S, int C,
                     int act, void* stream);
int cbim_resnorm_bwd_reduce(int dtype, const void* dy, int64_t dy_stride, const void* a, int64_t a_stride,
                            const float* stats_a, const void* b, int64_t b_stride, const float* stats_b, int N,
                            int64_t S, int C, int act, float* partials_a, float* partials_b, int P, void* stream);
int cbim_resnorm_bwd_apply(int dtype, const void* dy, int64_t dy_stride, const void* a, int64_t a_stride,
                           const float* stats_a, const float* sums_a, const void* b, int64_t b_stride,
                           const float* stats_b, const float* sums_b, void* da, void* db, int N, int64_t S,
                           int C, int act, void* stream);

/* ------------------------------------------------------------------------------------------
 * SwinUNETR shifted-window attention — /root/reference/model/dim3/swin_unetr.py.
 * One call replaces F.pad + torch.roll + window_partition + WindowAttention's q@k^T*scale +
 * relative_position_bias_table[relative_position_index[:n,:n]] + shift mask + softmax + @v +
 * window_reverse + roll back + crop (:467-490, :554-606, :737-773).
 *   qkv  [B][D][H][W][3*C] token rows ([3][heads][dh] inside a row), out/dout [B][D][H][W][C]
 *   qkv_bias float [3*C] or NULL: the q/k/v of window-padding tokens (the reference pads after norm1)
 *   table float [(2*tw0-1)*(2*tw1-1)*(2*tw2-1)][heads]; window/shift/table_window: int[3] HOST arrays;
 *   window/shift are the EFFECTIVE values of get_window_size (:358-381); table_window is the module's (7,7,7)
 *   lse float [num_windows][heads][343] (log-sum-exp per query, kept for the backward)
 *   backward also returns d(table) and the gradient reaching qkv.bias through padded keys (float [3*C]).
 * ------------------------------------------------------------------------------------------ */
int cbim_window_attn3d_num_windows(int B, int D, int H, int W, const int* window);
size_t cbim_window_attn3d_workspace(int B, int D, int H, int W, int C, int heads, const int* window,
                                    const int* table_window);
int cbim_window_attn3d_fwd(int dtype, const void* qkv, const float* qkv_bias, const float* table, void* out,
                           float* lse, int B, int D, int H, int W, int C, int heads, const int* window,
                           const int* shift, const int* table_window, void* stream);
int cbim_window_attn3d_bwd(int dtype, const void* qkv, const float* qkv_bias, const float* table,
                           const void* out, const void* dout, const float* lse, void* dqkv, float* dtable,
                           float* dbias_pad, int B, int D, int H, int W, int C, int heads, const int* window,
                           const int* shift, const int* table_window, void* workspace, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer + EMA step — optim.AdamW(..., eps=1e-5).step() and update_ema_variables
 * (/root/reference/training/utils.py:14,98-105; train.py:216-218) over ALL parameters in one launch.
 *   tensors   : DEVICE array of records (fp32 parameter, gradient, exp_avg, exp_avg_sq, ema copy or NULL)
 *   blk_tensor/blk_chunk : DEVICE int32[nblocks]: workgroup b updates elements [chunk*C, chunk*C + C) of tensor
 *               blk_tensor[b], C = cbim_optim_chunk()
 *   hyper     : DEVICE float[10] = {step, lr, beta1, beta2, eps, weight_decay, ema_alpha, 0, 1-beta1, 1-beta2};
 *               the call first
 *               increments hyper[0] on the device (graph-capturable), then applies torch's AdamW update with
 *               bias corrections for that step, then ema = a*ema + (1-a)*p with a = min(1 - 1/step, ema_alpha).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  void* p;
  const void* g;
  void* m;
  void* v;
  void* ema;
  int64_t numel;
} cbim_optim_tensor;
int cbim_optim_chunk(void);
int cbim_adamw_ema_step(const cbim_optim_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk,
                        int nblocks, float* hyper, void* stream);
/* update_ema_variables alone (/root/reference/training/utils.py:98-102, called at train.py:218) for callers that keep
 * another optimizer: ema = alpha*ema + (1-alpha)*p over every record (.p, .ema, .numel used) in one launch; alpha is
 * the already clamped min(1 - 1/(global_step+1), ema_alpha), one_minus_alpha = (float)(1 - (double)alpha) as torch
 * rounds the scalar of add_(param, alpha=1-alpha); the result is bit-identical to the reference's loop. */
int cbim_ema_step(const cbim_optim_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk, int nblocks,
                  float alpha, float one_minus_alpha, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sliding-window inference + evaluation Dice.
 *   cbim_softmax_accumulate : prob_sum[window] += softmax(logits, dim 1); counter[window] += 1
 *                             (/root/reference/inference/inference3d.py:80-86); NCDHW float32
 *   cbim_prob_finalize      : prob_sum /= counter (:88) (counter NULL: keep), optional argmax labels (int64,
 *                             first maximum — torch.max(pred, dim=1) of training/validation.py:44)
 *   cbim_dice_counts        : calculate_dice's mask sums (/root/reference/metric/utils.py:62-82) as exact integers
 *                             per block of `block` voxels: counts int32 [ceil(N/block)][C][3] =
 *                             (#pred==c & target==c, #pred==c, #target==c); labels int8 or int64
 * ------------------------------------------------------------------------------------------ */
int cbim_softmax_accumulate(const float* logits, float* prob_sum, float* counter, int B, int K, int wd, int wh,
                            int ww, int D, int H, int W, int d0, int h0, int w0, void* stream);
int cbim_prob_finalize(float* prob_sum, const float* counter, int64_t* labels, int B, int K, int64_t S,
                       void* stream);
int cbim_dice_counts(const void* pred, int pred_bytes, const void* target, int target_bytes, int64_t N,
                     int64_t block, int C, int32_t* counts, void* stream);

/* Attention gate of AttentionUNet (/root/reference/model/dim3/attention_unet_utils.py:28-35): y = x * psi with one psi
 * per voxel (float [rows]); backward dx = dy*psi, dpsi[row] = sum_c dy*x. */
int cbim_gate_fwd(int dtype, const void* x, const float* psi, void* y, int64_t rows, int C, void* stream);
int cbim_gate_bwd(int dtype, const void* dy, const void* x, const float* psi, void* dx, float* dpsi, int64_t rows,
                  int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the channel axis of channels-last token rows — SwinUNETR trunk:
 * SwinTransformerBlock.norm1 / norm2 (model/dim3/swin_unetr.py:539,550), PatchMerging.norm (:679),
 * SwinTransformer.proj_out = F.layer_norm(x, [C]) without affine parameters (:970-983).
 * x float [rows][C] (C a multiple of 4, <= 3072); gamma / beta float [C] or NULL; y [rows][C] in
 * out_dtype (CBIM_F32 | CBIM_BF16); rowstats float [rows][2] = (mean, rstd) for the backward.
 * Backward: dx float [rows][C] = rstd*(g - mean(g) - xh*mean(g*xh)), g = dy*gamma; dgamma / dbeta
 * (float [C], NULL to skip) are summed in fixed order through
 * cbim_layernorm_bwd_workspace(rows, C) bytes of scratch.  add (float [rows][C] or NULL, round 5) is added to dx: the
 * gradient of the residual stream x + f(LN(x)) that bypasses the norm (swin_unetr.py:539-552) — no separate add pass.
 * ------------------------------------------------------------------------------------------ */
int cbim_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int out_dtype,
                       void* y, float* rowstats, int64_t rows, int C, void* stream);
size_t cbim_layernorm_bwd_workspace(int64_t rows, int C);
int cbim_layernorm_bwd(int dy_dtype, const void* dy, const float* x, const float* gamma,
                       const float* rowstats, const float* add, float* dx, float* dgamma, float* dbeta,
                       void* workspace, size_t ws_bytes, int64_t rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Token Linear (round 5) — nn.Linear over channels-last token rows on the matrix cores, the SwinUNETR trunk's
 * WindowAttention.qkv / .proj (model/dim3/swin_unetr.py:467-490), MLPBlock linear1 -> GELU -> linear2 (:552, monai
 * MLPBlock), PatchMerging.reduction (:707-731) and PatchEmbed.proj; replaces aten::linear (hipBLASLt), aten::gelu /
 * gelu_backward, the bias and residual adds and the fp32 <-> bf16 casts around them:
 *
 *     y[r][co] = ( sum_ci act_in(x[r][ci]) * w[co][ci] + bias[co] ) * act'(mask[r][co]) + res[r][co]
 *
 *   x        [rows][Cin], x_dtype CBIM_BF16 or CBIM_F32 (fp32 rows are split into bf16 hi + lo fragments in registers
 *            and keep their fp32 accuracy: the patch rows and the fp32 residual-stream gradient feed the GEMM as they
 *            are, no cast pass); act_in (CBIM_ACT_*, bf16 rows only) is
 *            applied on load — linear2 reads the stored pre-activation h, GELU(h) is never written;
 *   w_packed cbim_conv3d_pack_weights image of the weight seen as a 1x1x1 convolution [Cout][Cin] (mode 0; mode 1 of the
 *            FORWARD layer's weight for its input gradient, with Cin / Cout exchanged here);
 *   w_lo_packed  NULL, or (fp32 rows only) the cbim_conv3d_pack_weights_lo image of the same weight: three MFMAs per
 *            fragment pair (x_hi w_hi + x_lo w_hi + x_hi w_lo) give the product fp32 accuracy — the Linears the reference
 *            runs outside any reduced-precision region of ours (PatchEmbed.proj, PatchMerging.reduction);
 *   bias     float [Cout] or NULL;  mask: bf16 [rows][Cout] or NULL, mask_act the activation whose derivative is taken
 *            (input gradient of linear2 at the stored h);  res: float [rows][Cout] or NULL (the residual stream);
 *   y        [rows][Cout] in y_dtype.  Cin, Cout multiples of 8, Cin <= 4096; strides in elements.
 * Weight gradient dw[co][ci] = sum_r dy[r][co] * act_in(x[r][ci]) (float [Cout][Cin], fixed summation order), either
 * operand in bf16 or fp32; workspace cbim_token_linear_wgrad_workspace(rows, Cin, Cout) bytes.  The bias gradient is
 * cbim_colsum of dy.
 * ------------------------------------------------------------------------------------------ */
int cbim_token_linear(const void* x, int x_dtype, int64_t x_stride, int act_in, const void* w_packed, const void* w_lo_packed,
                      const float* bias,
                      const float* res, int64_t res_stride, const void* mask, int64_t mask_stride, int mask_act,
                      void* y, int y_dtype, int64_t y_stride, int64_t rows, int Cin, int Cout, void* stream);
size_t cbim_token_linear_wgrad_workspace(int64_t rows, int Cin, int Cout);
int cbim_token_linear_wgrad(const void* x, int x_dtype, int64_t x_stride, int act_in, const void* dy, int dy_dtype,
                            int64_t dy_stride, float* dw, void* workspace, size_t ws_bytes, int64_t rows, int Cin,
                            int Cout, void* stream);

/* Column sums of token rows, out[c] = sum_r x[r][c] (fp32, fixed order): the bias gradient of the trunk's token Linears
 * (swin_unetr.py:467-490,640-643).  x [rows][C] in dtype; workspace: cbim_colsum_workspace(rows, C) bytes. */
size_t cbim_colsum_workspace(int64_t rows, int C);
int cbim_colsum(int dtype, const void* x, int64_t rows, int C, float* out, void* workspace, size_t ws_bytes,
                void* stream);

/* Layout helpers (caller-facing NCDHW fp32 <-> internal NDHWC). */
int cbim_ncdhw_to_ndhwc(int dtype_out, const float* x, void* y, int N, int C, int64_t S, void* stream);
int cbim_ndhwc_to_ncdhw(int dtype_in, const void* x, float* y, int N, int C, int64_t S, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CBIM_HIP_H */
