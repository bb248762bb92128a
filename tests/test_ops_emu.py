"""CPU run of the per-kernel parity checks: the gfx950 kernel sources executed by the host-side
executor in tests/emu (see tests/emu/hip_emu.h).  The same checks run on silicon in
tests/test_gpu_ops.py."""
import pytest
import torch

from tests import op_checks as oc

F32, BF16 = torch.float32, torch.bfloat16


@pytest.fixture(autouse=True)
def _need_emu(dev):
    if dev != "cpu":
        pytest.skip("host-side executor tests run only in the CPU suite")


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_instnorm(dev, dtype):
    oc.check_instnorm(dev, dtype)
    oc.check_instnorm(dev, dtype, N=1, C=72, dhw=(2, 3, 3))
    for act in ("lrelu", "gelu", "swish", "none"):     # every code of the reference's get_act (model/dim3/utils.py:23-30)
        oc.check_instnorm(dev, dtype, N=1, C=16, dhw=(3, 5, 6), act=act)
    oc.check_instnorm(dev, dtype, N=1, C=2056, dhw=(2, 2, 3))   # > 256 chunks: channel groups


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_maxpool(dev, dtype):
    oc.check_maxpool(dev, dtype)
    oc.check_maxpool(dev, dtype, dhw=(4, 9, 7), scale=(1, 2, 2))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_upcat(dev, dtype):
    oc.check_upcat(dev, dtype)
    oc.check_upcat(dev, dtype, low=(9, 4, 4), hi=(18, 8, 8), skip_first=False)
    oc.check_upcat(dev, dtype, low=(1, 3, 2), hi=(2, 6, 4))
    oc.check_upcat_fused(dev, dtype)
    oc.check_upcat_fused(dev, dtype, tiles=False)
    oc.check_upcat_fused(dev, dtype, N=1, Cl=8, Cs=8, low=(4, 7, 5), hi=(4, 14, 10))      # anisotropic scale [1, 2, 2]
    oc.check_upcat_fused(dev, dtype, N=1, low=(5, 4, 4), hi=(10, 8, 8), skip_first=False)
    oc.check_up_gram_stats(dev, dtype)
    oc.check_up_gram_stats(dev, dtype, N=1, Cl=40, low=(4, 7, 5), hi=(4, 14, 10), offset=50.0)   # [1, 2, 2], a 40-channel tail group, large mean
    oc.check_up_gram_stats(dev, dtype, N=1, Cl=8, low=(5, 9, 10), hi=(17, 20, 31))              # ragged tiles, odd factors
    oc.check_up_adjoint(dev, dtype)
    oc.check_up_adjoint(dev, dtype, N=1, Cl=8, Cs=8, low=(4, 7, 5), hi=(4, 14, 10), skip_first=False)   # [1, 2, 2]
    oc.check_up_adjoint(dev, dtype, N=1, low=(1, 3, 2), hi=(2, 6, 4))


@pytest.mark.parametrize("dtype,N,Cin,Cout,dhw,k", [
    (F32, 1, 8, 8, (4, 8, 8), (3, 3, 3)),
    (F32, 2, 20, 40, (5, 9, 11), (3, 3, 3)),
    (BF16, 2, 40, 72, (6, 9, 10), (3, 3, 3)),
    (F32, 1, 8, 16, (5, 8, 8), (2, 3, 3)),      # even kernel: output D grows by 1 (ACDC yaml)
    (BF16, 1, 16, 8, (3, 12, 8), (1, 3, 3)),
    (BF16, 1, 8, 8, (33, 16, 16), (3, 3, 3)),   # MT=2 tiles
    (F32, 1, 4, 4, (2, 2, 2), (3, 3, 3)),       # bottom of a 32^3 pyramid
    (F32, 2, 24, 40, (5, 6, 7), (1, 1, 1)),     # pointwise (MedFormer 1x1 convs)
    (BF16, 1, 64, 32, (4, 8, 8), (1, 1, 1)),
])
def test_conv(dev, dtype, N, Cin, Cout, dhw, k):
    oc.check_conv(dev, dtype, N, Cin, Cout, dhw, k)


def test_conv_norm_no_act(dev):
    oc.check_conv(dev, F32, 1, 16, 24, (4, 6, 8), (1, 1, 1), act="none")
    oc.check_conv(dev, BF16, 1, 16, 16, (4, 8, 8), (3, 3, 3), act="none")
    # norm -> act fused into the conv's input transform / dgrad mask / wgrad for the other activation codes
    oc.check_conv(dev, F32, 1, 8, 16, (4, 6, 8), (3, 3, 3), act="gelu")
    oc.check_conv(dev, F32, 1, 8, 16, (4, 6, 8), (3, 3, 3), act="swish")
    oc.check_conv(dev, BF16, 1, 16, 16, (4, 8, 8), (3, 3, 3), act="lrelu")


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_dwconv(dev, dtype):
    oc.check_dwconv(dev, dtype)
    oc.check_dwconv(dev, dtype, N=1, C=40, dhw=(2, 2, 2), act="none")
    oc.check_dwconv(dev, dtype, N=1, C=264, dhw=(3, 4, 5))             # two channel-chunk groups
    oc.check_dwconv(dev, dtype, N=1, C=16, dhw=(3, 6, 9), k=(1, 3, 3))
    oc.check_dwconv(dev, dtype, N=1, C=16, dhw=(4, 5, 6), k=(3, 3, 1))  # generic path


@pytest.mark.parametrize("dtype,W", [(BF16, 64), (BF16, 78), (BF16, 80), (BF16, 112), (F32, 160)])
def test_dwconv_wide_rows(dev, dtype, W):
    """rows too wide for the LDS-tiled depthwise wgrad must fall back to the streaming kernel (the executor refuses
    launches over 160 KiB of LDS like the hardware does; round-1 crash at W = 112, amos_mr/medformer_3d.yaml)."""
    oc.check_dwconv(dev, dtype, N=1, C=16, dhw=(2, 3, W))


def test_dwconv_wgrad_on_matrix_cores(dev):
    oc.check_dwconv_wgrad_mfma(dev, with_stats=False, with_bias=False)
    oc.check_dwconv_wgrad_mfma(dev)
    oc.check_dwconv_wgrad_mfma(dev, N=2, C=32, dhw=(9, 8, 11), act="none", with_bias=False)   # ragged tiles, two images


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_space_to_depth(dev, dtype):
    oc.check_space_to_depth(dev, dtype)
    oc.check_space_to_depth(dev, dtype, C=16, dhw=(2, 4, 6), scale=(1, 2, 2))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attn(dev, dtype):
    oc.check_attn(dev, dtype)
    oc.check_attn(dev, dtype, N=1, heads=1, dh=16, dhw=(2, 2, 2), M=8)
    oc.check_attn(dev, dtype, N=1, heads=2, dh=32, dhw=(4, 6, 6), M=64)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_mappool(dev, dtype):
    oc.check_mappool(dev, dtype)
    oc.check_mappool(dev, dtype, N=1, C=72, M=64, dhw=(6, 6, 5))
    oc.check_mappool(dev, dtype, N=1, C=40, M=27, dhw=(4, 5, 6))     # bcv map_size [3,3,3]: element-wise (one-wave) backward


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attn_wide_heads_and_maps(dev, dtype):
    """d_head / map sizes of the shipped ACDC (72 codes, d_head 64 | 80) and LiTS (one head: d_head = channels)
    MedFormer configurations — attn_wide.hip."""
    oc.check_attn(dev, dtype, N=1, heads=4, dh=80, dhw=(2, 6, 6), M=72)       # acdc down4: 320 channels, 4 heads
    oc.check_attn(dev, dtype, N=2, heads=1, dh=64, dhw=(3, 5, 7), M=64)       # lits down1-sized head
    oc.check_attn(dev, dtype, N=1, heads=2, dh=24, dhw=(4, 4, 5), M=27)       # bcv map [3,3,3], odd d_head
    oc.check_attn(dev, dtype, N=1, heads=1, dh=40, dhw=(1, 1, 3), M=128)      # more codes than voxels, code limit


def test_wgrad_of_1x3x3_kernels_as_the_centre_plane_of_3x3x3(dev):
    oc.check_wgrad_133(dev)
    oc.check_wgrad_133(dev, N=2, Cin=64, Cout=32, dhw=(8, 8, 24))


def test_attn_core_as_matrix_products(dev):
    """round 6: the BidirectionAttention core of ONE wide head (config/lits) on the row-GEMM kernels + csrc/attn_gemm_kernels.hip"""
    oc.check_attn_gemm(dev, N=2, dh=64, dhw=(8, 8, 10), M=64)
    oc.check_attn_gemm(dev, N=1, dh=160, dhw=(8, 8, 8), M=32)         # d_head not a power of two (lits: 320), 32 codes
    oc.check_attn_gemm(dev, N=1, dh=64, dhw=(5, 9, 13), M=128)
    oc.check_attn_gemm(dev, N=1, dh=32, dhw=(8, 12, 12), M=72, heads=4)     # acdc down2: 4 heads x 72 codes
    oc.check_attn_gemm(dev, N=2, dh=80, dhw=(4, 12, 12), M=72, heads=4)     # acdc down4
    oc.check_attn_gemm(dev, N=1, dh=40, dhw=(8, 8, 8), M=24, heads=2)       # 24 codes: lanes past M idle        # ragged last record (585 rows), 128 codes


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_mappool_more_than_64_codes(dev, dtype):
    oc.check_mappool(dev, dtype, N=1, C=40, M=72, dhw=(4, 6, 6))              # acdc map_size [2,6,6]
    oc.check_mappool(dev, dtype, N=2, C=24, M=128, dhw=(3, 7, 5))


def test_trilinear_planes(dev):
    oc.check_trilinear_planes(dev)
    oc.check_trilinear_planes(dev, lo=(1, 2, 2), hi=(4, 4, 4))
    oc.check_trilinear_planes(dev, N=2, C=3, lo=(2, 3, 4), hi=(5, 6, 8))      # four outputs per thread (Wo % 4 == 0), several rows and slabs


def test_conv_r32_weights_in_registers(dev):
    oc.check_conv_r32(dev)                                          # ragged tiles, two images
    oc.check_conv_r32(dev, N=1, Cout=16, dhw=(8, 8, 16), act="none")
    oc.check_conv_r32(dev, tile_depth=4)                            # two 256-thread workgroups per CU, 4x8x8 tiles
    oc.check_conv_r32(dev, N=1, Cout=32, dhw=(13, 8, 24), tile_depth=4, act="none")
    # several 32-channel chunks: Cin chunks walked with streamed weights, Cout chunks on blockIdx.y
    oc.check_conv_r32(dev, N=1, Cin=64, Cout=32, dhw=(9, 8, 16))
    oc.check_conv_r32(dev, N=2, Cin=96, Cout=64, dhw=(8, 9, 8), dy_split=32)     # 64-cout weight blocks; dgrad over [dy1 | dout]
    oc.check_conv_r32(dev, N=1, Cin=32, Cout=96, dhw=(8, 8, 8), act="none")     # Cout 96: the last 64-block is half empty


def test_conv_rw_buffer_addressed_halo(dev):
    """k_conv3_rw (conv_rw.hip) against k_conv3_r32 (bit for bit where the arithmetic is the same) and torch."""
    oc.check_conv_rw(dev)                                                        # ragged tiles, two images, one chunk
    oc.check_conv_rw(dev, N=3, dhw=(8, 8, 24), seed=62)                          # strips of one tile, three images
    oc.check_conv_rw(dev, N=1, Cin=64, Cout=32, dhw=(9, 8, 16))                  # streamed weights
    oc.check_conv_rw(dev, N=1, Cin=96, Cout=64, dhw=(8, 9, 8), x_split=32)       # 64-cout weight blocks, [x | x2] input, narrow
    oc.check_conv_rw(dev, N=2, Cin=96, Cout=64, dhw=(8, 9, 8), x_split=32, wide=2)   # the same on 64-cout workgroups
    oc.check_conv_rw(dev, N=1, Cin=32, Cout=64, dhw=(8, 8, 16), wide=2)          # wide, one chunk
    oc.check_conv_rw(dev, N=1, Cin=64, Cout=128, dhw=(8, 8, 8), wide=2)          # two 64-cout blocks
    oc.check_conv_rw(dev, N=1, Cin=32, Cout=96, dhw=(8, 8, 8))                   # Cout 96: the last 64-block is half empty
    # low-resolution layers: split-K over the Cin chunks + the finish pass
    oc.check_conv_rw_split(dev)                                                  # one tile, two chunks
    oc.check_conv_rw_split(dev, N=2, Cin=96, Cout=32, dhw=(8, 8, 16))            # three chunks, two images
    oc.check_conv_rw_split(dev, N=1, Cin=128, Cout=64, dhw=(9, 8, 8), seed=64)   # ragged depth


def test_conv_rw48_forty_eight_channel_workgroups(dev):
    """k_conv3_rw48 (round 6; SwinUNETR's 48-channel monai blocks) against k_conv_igemm and torch, and functional.NormConvFn's
    materialised path on it."""
    oc.check_conv_rw48(dev)                                                      # 48 -> 48, LeakyReLU mask, ragged tiles
    oc.check_conv_rw48(dev, Cin=96, Cout=48, dhw=(8, 8, 8), act="relu", seed=72)  # three chunks; its dgrad has two 48-cout workgroups
    oc.check_conv_rw48(dev, N=2, Cin=48, Cout=96, dhw=(9, 8, 8), seed=73)        # two images, ragged depth
    oc.check_conv_rw48(dev, Cin=8, Cout=48, dhw=(8, 8, 16), seed=74)             # the padded network input: one quarter-filled chunk
    oc.check_norm_conv_mat48(dev)


def test_map_branch_small_gemm(dev):
    """k_map_gemm (round 6): norm2 / map_qv / map_out / residual of MedFormer's BidirectionAttentionBlock, forward and backward."""
    oc.check_map_branch(dev)                                          # 64 positions (AMOS)
    oc.check_map_branch(dev, B=2, C=72, I=40, M=27, seed=82)          # 27 positions (BCV), two images: batch-summed weight gradients
    oc.check_map_branch(dev, B=1, C=136, I=160, M=72, seed=83)        # 72 positions (ACDC): both column halves of a lane, several tiles
    oc.check_map_branch(dev, B=1, C=32, I=32, M=128, seed=84)         # the largest map


def test_se_gate_excitation_kernels(dev):
    oc.check_se_gate(dev)                                      # 40 channels, 10 hidden
    oc.check_se_gate(dev, N=2, C=136, H=34, seed=96)           # two images (weight gradients summed), ragged column blocks
    oc.check_se_gate(dev, N=1, C=1280, H=320, seed=97)         # MedFormer's widest MBConv (4 x 320 expanded channels)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_upcat_skip_scatter_into_the_concatenation(dev, dtype):
    oc.check_upcat_skip(dev, dtype)
    oc.check_upcat_skip(dev, dtype, N=1, Cu=48, Cs=48, dhw=(4, 4, 4), seed=92)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_stem_head(dev, dtype):
    oc.check_stem_head(dev, dtype)
    oc.check_stem_head(dev, dtype, N=1, Cin=1, base=72, K=6, dhw=(5, 8, 9))     # head rows of 9 / 18 chunks: chunk groups on blockIdx.z
    oc.check_stem_head(dev, dtype, N=2, Cin=1, base=48, K=4, dhw=(4, 6, 10))    # SwinUNETR's head width
    oc.check_stem_head(dev, dtype, Cin=1, base=16, K=3, dhw=(4, 8, 8), k=(1, 3, 3))
    oc.check_stem_head(dev, dtype, Cin=5, base=8, K=3, dhw=(4, 8, 9))     # two channel groups in the stem wgrad
    oc.check_stem_head(dev, dtype, N=2, Cin=1, base=32, K=16, dhw=(5, 7, 9))   # head: 4 (bf16) / 8 (fp32) chunks per row
    oc.check_stem_head(dev, dtype, Cin=1, base=64, K=2, dhw=(3, 5, 20))        # head: two chunks per wave (bf16) / generic kernels (fp32)
    oc.check_stem_head(dev, dtype, Cin=1, base=16, K=19, dhw=(3, 4, 5))        # head: more than 16 classes -> generic kernels


def test_stem_on_matrix_cores(dev):
    oc.check_stem_mfma(dev)
    oc.check_stem_mfma(dev, N=1, base=64, dhw=(4, 8, 8), k=(1, 3, 3))
    oc.check_stem_mfma(dev, N=1, base=96, dhw=(5, 7, 9))


def test_head_backward_on_matrix_cores(dev):
    oc.check_head_mfma(dev)
    oc.check_head_mfma(dev, N=1, base=64, K=16, dhw=(8, 8, 9))              # 18 steps over 8 waves: ragged last round
    oc.check_head_mfma(dev, N=1, base=96, K=3, dhw=(2, 4, 8))
    oc.check_head_mfma(dev, N=1, base=128, K=13, dhw=(4, 4, 8), need_dx=False)
    oc.check_head_mfma(dev, N=1, base=48, K=4, dhw=(4, 8, 8))               # 16-channel tiles (round 5): SwinUNETR's 48-channel head
    oc.check_head_mfma(dev, N=1, base=16, K=2, dhw=(2, 4, 8))
    oc.check_head_mfma(dev, N=1, base=112, K=7, dhw=(2, 4, 8))


def test_loss(dev):
    oc.check_loss(dev)
    oc.check_dice_reductions(dev)
    oc.check_loss(dev, N=1, C=3, dhw=(4, 4, 4), weighted=False, seed=9)
    oc.check_loss(dev, N=2, C=20, dhw=(3, 5, 6))     # C > 16: two voxels per thread
    oc.check_loss(dev, N=2, C=20, dhw=(3, 5, 7))     # odd plane size: one voxel per thread
    oc.check_loss(dev, N=2, C=5, dhw=(3, 3, 5))


def test_errors_are_loud(dev):
    from cbim_amd import ops
    x = torch.zeros(1, 2, 2, 2, 6)           # 6 channels: not a whole 16-byte chunk
    with pytest.raises(RuntimeError, match="multiple"):
        ops.instnorm_stats(x)
    with pytest.raises(TypeError):
        ops.instnorm_stats(torch.zeros(1, 2, 2, 2, 8, dtype=torch.float16))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_fused_conv1_shortcut_block(dev, dtype):
    oc.check_fused_block(dev, dtype)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_window_attention(dev, dtype):
    oc.check_window_attn(dev, dtype)                                                    # padded + shifted
    oc.check_window_attn(dev, dtype, dhw=(7, 7, 7), shift=(0, 0, 0), C=16, heads=2)     # one full window
    oc.check_window_attn(dev, dtype, dhw=(8, 4, 4), shift=(3, 3, 3), C=16, heads=1)     # window (7,4,4), shift (3,0,0)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_resnorm(dev, dtype):
    oc.check_resnorm(dev, dtype)
    oc.check_resnorm(dev, dtype, N=1, C=24, dhw=(2, 1, 1), with_b_stats=False)
    oc.check_resnorm(dev, dtype, N=2, C=48, dhw=(9, 7, 5), seed=23)        # 6 chunks per row: threads 252..255 idle, ragged rows


def test_fused_adamw_ema(dev):
    from tests.optim_checks import check_adamw_ema, check_ema_buffers
    check_adamw_ema(dev)
    check_ema_buffers(dev)


def test_window_attention_matrix_core_path(dev):
    oc.check_window_attn_mfma(dev)                                                  # padded + shifted, 3 heads
    oc.check_window_attn_mfma(dev, dhw=(8, 4, 4), shift=(3, 3, 3), C=16, heads=1)   # window (7,4,4): 112 tokens


def test_training_utils_surface(dev):
    from tests.optim_checks import check_training_utils_surface
    check_training_utils_surface(dev)


def test_inference_and_dice(dev):
    from tests import infer_checks as ic
    ic.check_dice_exact(dev)
    ic.check_sliding_window(dev, full=False)      # one 32^3 window on the executor; the 12-window run is a GPU test


def test_wgrad_unrolled_plane_path_at_32k_voxels(dev):
    oc.check_wgrad_large(dev, N=1, Cin=32)                                # 8x64x64 = 2 x 8 x 8 tiles of 4x8x8
    oc.check_wgrad_large(dev, N=1, Cin=32, Cout=64, dhw=(8, 64, 64), act="none", split=32)
    oc.check_wgrad_large(dev, N=1, Cin=32, Cout=32, dhw=(8, 64, 64), raw=True)
    # three cout blocks x two Cin blocks
    oc.check_wgrad_large(dev, N=1, Cin=64, Cout=96, dhw=(8, 64, 64))


def test_wgrad_r32_accumulators_in_registers(dev):
    """k_wgrad_r32 (raw bf16 3x3x3 inputs, channel multiples of 32) on the executor: both wave layouts, ragged extents,
    several (co, ci) chunk pairs, dy and x as two tensors."""
    from cbim_amd import _lib
    L = _lib.lib()
    try:
        for wv in (8, 4):
            L.cbim_wgrad_r32_waves(wv)
            oc.check_wgrad_r32(dev)
            oc.check_wgrad_r32(dev, N=2, Cin=64, Cout=64, dhw=(9, 11, 17), split=32)
        oc.check_wgrad_r32(dev, N=1, Cin=96, Cout=32, dhw=(8, 8, 24), xsplit=32)
        oc.check_wgrad_r32(dev, N=1, Cin=64, Cout=32, dhw=(8, 8, 8))      # one tile = one strip: written straight into dw
        # round 6: channel counts in multiples of 16 (SwinUNETR's 48 / 96): a zero-filled last block, cropped gradient rows
        oc.check_wgrad_r32(dev, Cin=48, Cout=48, dhw=(8, 16, 8), seed=11)
        oc.check_wgrad_r32(dev, N=2, Cin=96, Cout=48, dhw=(9, 8, 8), seed=12)
        oc.check_wgrad_r32(dev, Cin=48, Cout=96, dhw=(8, 8, 16), seed=13)
    finally:
        L.cbim_wgrad_r32_waves(8)


def test_layernorm_token_rows(dev):
    """nn.LayerNorm of the SwinUNETR trunk: every channel count of the shipped model (48 ... 3072), with and without affine
    parameters, fp32 and bf16 outputs, row counts that do not fill the last workgroup."""
    oc.check_layernorm(dev)
    oc.check_layernorm(dev, rows=(2, 3, 11), C=96, out_bf16=True)
    oc.check_layernorm(dev, rows=(1, 9, 4), C=192, affine=False)
    oc.check_layernorm(dev, rows=(37,), C=384)
    oc.check_layernorm(dev, rows=(5, 3), C=768, out_bf16=True)
    oc.check_layernorm(dev, rows=(9,), C=1536)
    oc.check_layernorm(dev, rows=(4,), C=3072, affine=True)
    import torch
    oc.check_colsum(dev, torch.bfloat16)
    oc.check_colsum(dev, torch.bfloat16, rows=77, C=576)
    oc.check_colsum(dev, torch.float32, rows=513, C=144)


def test_dgrad_masked_by_the_activated_tensor(dev):
    import torch
    oc.check_dgrad_mask_by_activated(dev, torch.bfloat16)                                  # k_conv3_r32, single chunk
    oc.check_dgrad_mask_by_activated(dev, torch.bfloat16, Cin=64, Cout=32, dhw=(8, 8, 16))   # k_conv3_r32, several chunks
    oc.check_dgrad_mask_by_activated(dev, torch.float32, Cin=8, Cout=12, dhw=(5, 6, 7))      # k_conv_igemm


@pytest.mark.parametrize("cfg", [dict(), dict(N=1, Cin=48, Cout=24, dhw=(3, 5, 9), act="none"), dict(N=1, Cin=128, Cout=320, dhw=(4, 8, 8)),
                                 dict(N=2, Cin=32, Cout=32, dhw=(2, 3, 5), act="gelu"), dict(N=1, Cin=8, Cout=136, dhw=(1, 1, 130)), dict(N=1, Cin=160, Cout=64, dhw=(2, 4, 8), act="relu")])
def test_pointwise_conv_row_gemm(dev, cfg):
    oc.check_conv_pw(dev, **cfg)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("lds", [0, 1])
def test_dwconv_lds_tiled_and_streaming_kernels(dev, dtype, lds):
    """k_dwconv3_lds (round 4: halo staged once, transformed once) and the streaming k_dwconv3 on the same cases, ragged tiles
    and several tiles per axis included"""
    from cbim_amd import _lib
    old = _lib.lib().cbim_dwconv_lds_enable(lds)
    try:
        oc.check_dwconv(dev, dtype)
        oc.check_dwconv(dev, dtype, N=1, C=40, dhw=(2, 2, 2), act="none")
        oc.check_dwconv(dev, dtype, N=1, C=72, dhw=(9, 10, 19))
        oc.check_dwconv(dev, dtype, N=1, C=16, dhw=(3, 6, 9), k=(1, 3, 3))
        oc.check_dwconv(dev, dtype, N=1, C=24, dhw=(5, 9, 8), k=(3, 1, 3), act="none")
    finally:
        _lib.lib().cbim_dwconv_lds_enable(old)


def test_token_linear(dev):
    """the SwinUNETR trunk's token Linears on the engine's row GEMM (round 5)"""
    oc.check_token_linear(dev)
    oc.check_token_linear(dev, rows=64, Cin=384, Cout=96)        # few rows, long K: the four waves split K (KS = 4)
    oc.check_token_linear(dev, rows=130, Cin=32, Cout=48)        # patch embedding width; ragged last row tile
    oc.check_token_linear(dev, rows=257, Cin=96, Cout=384)       # 128 output channels per wave


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_batchnorm_affine(dev, dtype):
    oc.check_batchnorm_affine(dev, dtype)
    oc.check_batchnorm_affine(dev, dtype, N=1, C=8, dhw=(3, 5, 7), act="relu")
    oc.check_batchnorm_affine(dev, dtype, N=3, C=24, dhw=(2, 4, 4), act="none")


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_dual_raw_conv(dev, dtype):
    oc.check_dual_raw_conv(dev, dtype)
