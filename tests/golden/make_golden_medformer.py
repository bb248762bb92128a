"""Golden fixtures for MedFormer, produced by EXECUTING THE REAL REFERENCE on CPU (fp32).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_medformer.py

Same import shim as make_golden.py.  Cases
  medformer_tiny_32   reduced widths (dim_head 8, 8 map codes), 32^3, full state_dict + every gradient:
                      pins oracle/medformer_ref.py and the HIP path (CPU executor and GPU)
  medformer_amos_64   the shipped AMOS config (config/amos_ct/medformer_3d.yaml) at 64^3; weights are the
                      reference constructor's under torch.manual_seed(seed) (our module reproduces them
                      bit for bit — checked through sd_checksum); stores strided logits, losses and
                      per-parameter gradient norms / sums.  GPU parity test only.
  medformer_acdc_tiny the STRUCTURE of config/acdc/medformer_3d.yaml at reduced widths: anisotropic kernels/scales,
                      map_size [2,6,6] = 72 codes, 4 heads everywhere (d_head 8 | 16 | 20), transformer blocks at
                      every inner level — exercises attn_wide.hip and the >64-code map pooling
  medformer_bcv_tiny  the STRUCTURE of config/bcv/medformer_3d.yaml: 27 map codes (odd: padded code rows), 14 classes
  medformer_lits_tiny the STRUCTURE of config/lits/medformer_3d.yaml: num_heads all 1 (d_head = channels: 32, 64,
                      80), aux_loss False (forward returns one tensor)
  medformer_linear_tiny  proj_type 'linear' at TINY's widths: 1x1x1 feature projections, FusedMBConv feed-forward
  medformer_bn_tiny / medformer_ln_tiny  `norm: bn` / `norm: ln` at TINY's widths (round 6): nn.BatchNorm3d / the channels-first
                      LayerNorm in every ConvNormAct (eps 1e-4), as norm1 / norm2 of every attention block and PatchMerging.norm
                      (default eps) — train() mode, so BatchNorm runs on batch statistics and updates its running buffers (the
                      updated buffers are stored as `b:<key>`)
No reference source is copied; only tensors it produced.
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402

TINY = dict(base_chan=8, map_size=[2, 2, 2], conv_block="BasicBlock", conv_num=[2, 1, 0, 0, 0, 1, 2, 2],
            trans_num=[0, 1, 1, 2, 1, 1, 0, 0], chan_num=[16, 16, 32, 40, 32, 16, 16, 8],
            num_heads=[1, 2, 4, 5, 4, 2, 1, 1], fusion_depth=2, fusion_dim=40, fusion_heads=5, expansion=4,
            attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu", kernel_size=[[3, 3, 3]] * 5,
            scale=[[2, 2, 2]] * 4, aux_loss=True)
AMOS = dict(base_chan=32, map_size=[4, 4, 4], conv_block="BasicBlock", conv_num=[2, 1, 0, 0, 0, 1, 2, 2],
            trans_num=[0, 1, 4, 6, 4, 1, 0, 0], chan_num=[64, 128, 256, 320, 256, 128, 64, 32],
            num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4,
            attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu", kernel_size=[[3, 3, 3]] * 5,
            scale=[[2, 2, 2]] * 4, aux_loss=True)
ACDC_T = dict(base_chan=8, map_size=[2, 6, 6], conv_block="BasicBlock", conv_num=[2, 0, 0, 0, 0, 0, 2, 2],
              trans_num=[0, 2, 1, 1, 1, 1, 0, 0], chan_num=[16, 32, 64, 80, 64, 32, 16, 8],
              num_heads=[1, 4, 4, 4, 4, 4, 1, 1], fusion_depth=2, fusion_dim=32, fusion_heads=4, expansion=4,
              attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu",
              kernel_size=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
              scale=[[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]], aux_loss=True)
# config/bcv/medformer_3d.yaml's structure: map_size [3,3,3] = 27 codes (not a multiple of the 8-channel chunk: padded with
# zero-weight codes in SemanticMapGeneration), anisotropic stem, trans_num [0,2,4,6,4,2,0,0] cut to one block per level
BCV_T = dict(base_chan=8, map_size=[3, 3, 3], conv_block="BasicBlock", conv_num=[2, 0, 0, 0, 0, 0, 2, 2],
             trans_num=[0, 1, 1, 1, 1, 1, 0, 0], chan_num=[16, 32, 64, 80, 64, 32, 16, 8],
             num_heads=[1, 2, 4, 5, 4, 2, 1, 1], fusion_depth=2, fusion_dim=40, fusion_heads=5, expansion=4,
             attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu",
             kernel_size=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
             scale=[[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]], aux_loss=True)
LITS_T = dict(base_chan=8, map_size=[2, 2, 2], conv_block="BasicBlock", conv_num=[2, 0, 0, 0, 0, 0, 2, 2],
              trans_num=[0, 1, 1, 2, 1, 1, 0, 0], chan_num=[16, 32, 64, 80, 64, 32, 16, 8],
              num_heads=[1, 1, 1, 1, 1, 1, 1, 1], fusion_depth=2, fusion_dim=40, fusion_heads=5, expansion=4,
              attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu", kernel_size=[[3, 3, 3]] * 5,
              scale=[[2, 2, 2]] * 4, aux_loss=False)
# proj_type 'linear' (medformer_utils.py:26-28,121-122,153-154: 1x1x1 feature projections, FusedMBConv(kernel_size=1) feed-forward,
# 1x1x1 patch-merging reduction) at the reduced widths of TINY — no shipped yaml uses it; a constructor branch of the drop-in
LIN_T = dict(TINY, proj_type="linear")
BN_T, LN_T = dict(TINY, norm="bn"), dict(TINY, norm="ln")
CASES = {
    "medformer_bn_tiny": (1, 4, BN_T, (32, 32, 32), 1, 3061, 20000),
    "medformer_ln_tiny": (1, 4, LN_T, (32, 32, 32), 1, 3062, 20000),
    # name: (in_chan, classes, kwargs, spatial, batch, seed, full)
    "medformer_linear_tiny": (1, 4, LIN_T, (32, 32, 32), 1, 3051, 20000),
    "medformer_tiny_32": (1, 4, TINY, (32, 32, 32), 1, 3031, True),
    "medformer_amos_64": (1, 16, AMOS, (64, 64, 64), 1, 3032, False),
    # full = 20000: full gradients of every tensor with at most that many elements (all attention / map-side tensors),
    # weights come from the seed (sd_checksum)
    "medformer_acdc_tiny": (1, 4, ACDC_T, (8, 32, 32), 1, 3033, 20000),
    "medformer_lits_tiny": (1, 3, LITS_T, (32, 32, 32), 1, 3034, 20000),
    "medformer_bcv_tiny": (1, 14, BCV_T, (8, 32, 32), 1, 3041, 20000),   # seed chosen so that the reference fp32 run has no ReLU-mask flip (fp32 vs fp64 gradients 5e-3; seed 3035: 0.29)
}
AUX_WEIGHT = [0.5, 0.5]   # config/amos_ct/medformer_3d.yaml: aux_weight


def main():
    _, DiceLoss = mg.import_reference()
    MedFormer = importlib.import_module("model.dim3.medformer").MedFormer
    from oracle.unet_ref import state_dict_checksum
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (in_ch, classes, kw, shape, batch, seed, full) in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(seed)
        net = MedFormer(in_ch, classes, **kw)
        net.train()
        chk0 = state_dict_checksum(net.state_dict())      # of the constructor's state (BatchNorm buffers change in the forward)
        gen = torch.Generator().manual_seed(seed + 1)
        x = torch.randn((batch, in_ch) + shape, generator=gen).clamp_(-7.4, 2.2)
        lab = mg.make_labels(classes, shape, batch, gen)
        weight = torch.ones(classes)
        weight[0] = 0.5
        outs = net(x)
        if not kw["aux_loss"]:
            outs = [outs]                      # medformer.py:98-101: a single tensor without aux_loss
        ce_fn, dl_fn = torch.nn.CrossEntropyLoss(weight=weight), DiceLoss()
        ces = [ce_fn(o, lab.squeeze(1)) for o in outs]
        dls = [dl_fn(o, lab) for o in outs]
        if kw["aux_loss"]:
            loss = sum(w * (c + d) for w, c, d in zip(AUX_WEIGHT, ces, dls))   # train.py:207-210
        else:
            loss = ces[0] + dls[0]                                             # train.py:212-213
        loss.backward()
        sd = net.state_dict()
        grads = {k: p.grad for k, p in net.named_parameters()}
        zero = torch.zeros(())
        for k in sd:                      # buffers (BatchNorm running statistics) have no gradient: 0 in the per-key arrays
            grads.setdefault(k, zero)
        st = 1 if full else 4
        if st == 1 and full is not True:
            st = 2
        out = {
            "x": x.numpy(), "label": lab.numpy().astype(np.int64), "weight": weight.numpy(),
            "logits": outs[0].detach().numpy()[..., ::st, ::st, ::st],
            "aux_logits": outs[-1].detach().numpy()[..., ::st, ::st, ::st], "stride": np.int64(st),
            "ce": np.array([float(c) for c in ces]), "dice": np.array([float(d) for d in dls]),
            "loss": np.float64(loss.item()),
            "n_params": np.int64(sum(p.numel() for p in net.parameters())), "n_tensors": np.int64(len(sd)),
            "n_buffers": np.int64(len(list(net.buffers()))),
            "keys": np.array(list(sd.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]),
            "grad_norms": np.array([float(grads[k].double().norm()) for k in sd.keys()]),
            "grad_sums": np.array([float(grads[k].double().sum()) for k in sd.keys()]),
            "sd_checksum": np.float64(chk0), "seed": np.int64(seed),
            "g:inc.conv1.weight": grads["inc.conv1.weight"].numpy(),
            "g:outc.weight": grads["outc.weight"].numpy(),
        }
        if kw["aux_loss"]:
            out["g:aux_out.weight"] = grads["aux_out.weight"].numpy()
        pnames = {k for k, _ in net.named_parameters()}
        for k, v in sd.items():
            if k not in pnames:           # the buffers AFTER the training-mode forward (running statistics updated once)
                out["b:" + k] = v.numpy()
        if full is True:
            for k, v in sd.items():
                out["p:" + k] = v.numpy()
                out["g:" + k] = grads[k].numpy()
        elif full:
            for k, v in sd.items():
                if v.numel() <= full and k in pnames:
                    out["g:" + k] = grads[k].numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "logits", tuple(outs[0].shape), "loss", float(loss), "params", int(out["n_params"]), "tensors",
              int(out["n_tensors"]), "size", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
