"""Plugin registry with the reference's signature: ``get_model(args, pretrain=False)``
(/root/reference/model/utils.py:6, 3D branches :70-122).  ``args`` is the argparse Namespace
that carries every YAML key (train.py:266-270)."""


def get_model(args, pretrain=False):
    if args.dimension == "2d":
        raise NotImplementedError("cbim_amd: the 2D model zoo is outside the model/dim3 hot path")
    if args.dimension != "3d":
        raise ValueError("Invalid dimension, should be '2d' or '3d'")
    if args.model == "vnet":   # model/utils.py:70-74 of the reference (note: `downsample_scale`, not `down_scale`)
        from .dim3 import VNet
        if pretrain:
            raise ValueError("No pretrain model available")
        return VNet(args.in_chan, args.classes, scale=args.downsample_scale, baseChans=args.base_chan)
    if args.model in ("resunet", "unet"):
        from .dim3 import UNet
        if pretrain and args.model == "resunet":
            raise ValueError("No pretrain model available")
        return UNet(args.in_chan, args.base_chan, num_classes=args.classes, scale=args.down_scale,
                    norm=args.norm, kernel_size=args.kernel_size, block=args.block)
    if args.model == "unet++":   # model/utils.py:85-87 of the reference
        from .dim3 import UNetPlusPlus
        return UNetPlusPlus(args.in_chan, args.base_chan, num_classes=args.classes, scale=args.down_scale,
                            norm=args.norm, kernel_size=args.kernel_size, block=args.block)
    if args.model == "attention_unet":   # model/utils.py:88-90 of the reference
        from .dim3 import AttentionUNet
        return AttentionUNet(args.in_chan, args.base_chan, num_classes=args.classes, scale=args.down_scale,
                             norm=args.norm, kernel_size=args.kernel_size, block=args.block)
    if args.model == "medformer":   # model/utils.py:92-95 of the reference
        from .dim3 import MedFormer
        return MedFormer(args.in_chan, args.classes, args.base_chan, map_size=args.map_size,
                         conv_block=args.conv_block, conv_num=args.conv_num, trans_num=args.trans_num,
                         num_heads=args.num_heads, fusion_depth=args.fusion_depth, fusion_dim=args.fusion_dim,
                         fusion_heads=args.fusion_heads, expansion=args.expansion, attn_drop=args.attn_drop,
                         proj_drop=args.proj_drop, proj_type=args.proj_type, norm=args.norm, act=args.act,
                         kernel_size=args.kernel_size, scale=args.down_scale, aux_loss=args.aux_loss)
    if args.model == "swin_unetr":   # model/utils.py:111-119 of the reference (window_size is passed as img_size)
        from .dim3 import SwinUNETR
        model = SwinUNETR(args.window_size, args.in_chan, args.classes, feature_size=args.base_chan)
        if getattr(args, "pretrain", False):   # model/utils.py:115-117: the self-supervised Swin-ViT checkpoint
            import torch
            # the reference hard-codes its authors' cluster path; `args.swin_pretrain_path` (or CBIM_SWIN_PRETRAIN) names the file
            # here, a missing file is torch.load's FileNotFoundError exactly as there
            import os
            path = getattr(args, "swin_pretrain_path", None) or os.environ.get(
                "CBIM_SWIN_PRETRAIN", "/research/cbim/vast/yg397/ConvFormer/ConvFormer/initmodel/model_swinvit.pt")
            model.load_from(weights=torch.load(path, map_location="cpu"))
        return model
    raise NotImplementedError(f"cbim_amd: 3D model '{args.model}' is not built yet")
