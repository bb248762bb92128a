"""Parameter-layout fingerprints of EVERY shipped 3D configuration of the in-scope models, produced by calling the
REAL reference's get_model() (model/utils.py:6) on the YAML exactly as train.py:259-270 loads it.

Run in the build container only (needs /root/reference):   python tests/golden/make_shipped_configs.py

Writes tests/golden/shipped_configs.json: per config file the model-relevant YAML keys (so the GPU box, which has no
/root/reference, can rebuild the argparse Namespace), the parameter / tensor / buffer counts and an order-sensitive
digest of (name, shape) of the state_dict.  tests/test_shipped_configs.py replays it through cbim_amd's get_model().
SwinUNETR goes through the torch-only monai stand-in (tests/golden/monai_standin), as in make_golden_swin.py.
No reference source or YAML text is copied; only the values of the keys get_model reads.
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "monai_standin"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402

MODELS = ("unet", "resunet", "unet++", "attention_unet", "medformer", "swin_unetr", "vnet")
# the keys model/utils.py:70-122 reads for these models + what the step around them needs
KEYS = ("dimension", "model", "in_chan", "base_chan", "classes", "down_scale", "kernel_size", "norm", "block", "map_size",
        "conv_block", "conv_num", "trans_num", "num_heads", "fusion_depth", "fusion_dim", "fusion_heads", "expansion",
        "attn_drop", "proj_drop", "proj_type", "act", "aux_loss", "aux_weight", "window_size", "training_size", "weight",
        "pretrain", "downsample_scale")


def layout_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(f"{k}:{tuple(v.shape)};".encode())
    return h.hexdigest()


def main():
    mg.import_reference()
    dim3 = sys.modules["model.dim3"]
    for mod, cls in (("unet", "UNet"), ("unetpp", "UNetPlusPlus"), ("attention_unet", "AttentionUNet"),
                     ("medformer", "MedFormer"), ("swin_unetr", "SwinUNETR"), ("vnet", "VNet")):
        setattr(dim3, cls, getattr(importlib.import_module("model.dim3." + mod), cls))
    get_model = importlib.import_module("model.utils").get_model
    out = {}
    for path in sorted(glob.glob(os.path.join(mg.REF, "config", "*", "*_3d.yaml"))):
        with open(path) as f:
            cfg = yaml.load(f, Loader=yaml.SafeLoader)
        model = os.path.basename(path)[:-len("_3d.yaml")]      # train.py:259: config/<dataset>/<model>_<dimension>.yaml
        if model not in MODELS:
            continue
        cfg.update(model=model, dimension="3d", pretrain=False)   # the command-line arguments of train.py:239-242
        args = argparse.Namespace(**cfg)                       # train.py:267-268
        net = get_model(args)
        sd = net.state_dict()
        rel = os.path.relpath(path, os.path.join(mg.REF, "config"))
        out[rel] = {
            "args": {k: cfg[k] for k in KEYS if k in cfg},
            "n_params": sum(p.numel() for p in net.parameters()), "n_tensors": len(sd),
            "n_buffers": len(list(net.buffers())), "layout_sha256": layout_digest(sd),
        }
        print(rel, out[rel]["n_params"], out[rel]["n_tensors"], out[rel]["n_buffers"])
    with open(os.path.join(HERE, "shipped_configs.json"), "w") as f:      # one line per configuration file
        f.write("{\n" + ",\n".join(f" {json.dumps(k)}: {json.dumps(out[k], sort_keys=True)}" for k in sorted(out)) + "\n}\n")
    print(len(out), "configurations")


if __name__ == "__main__":
    main()
