"""Drop-in check of the plugin surface: EVERY shipped 3D configuration of the in-scope models
(/root/reference/config/*/{unet,resunet,unet++,attention_unet,medformer,swin_unetr}_3d.yaml), loaded the way
train.py:259-270 does, must build through cbim_amd's get_model() with the reference's exact parameter layout
(names, shapes, order, counts) — fingerprints recorded from the REAL reference by
tests/golden/make_shipped_configs.py — and every MedFormer head / map size must be one the attention entry points take.
"""
import argparse
import hashlib
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "shipped_configs.json")) as f:
    SHIPPED = json.load(f)


def _digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(f"{k}:{tuple(v.shape)};".encode())
    return h.hexdigest()


def test_fixture_covers_the_shipped_zoo():
    models = sorted({v["args"]["model"] for v in SHIPPED.values()})
    assert models == ["attention_unet", "medformer", "resunet", "swin_unetr", "unet", "unet++"]
    assert len(SHIPPED) == 23


@pytest.mark.parametrize("cfg", sorted(SHIPPED))
def test_get_model_builds_reference_layout(cfg):
    from cbim_amd.model.utils import get_model
    rec = SHIPPED[cfg]
    net = get_model(argparse.Namespace(**rec["args"]))
    sd = net.state_dict()
    assert sum(p.numel() for p in net.parameters()) == rec["n_params"], cfg
    assert len(sd) == rec["n_tensors"] and len(list(net.buffers())) == rec["n_buffers"], cfg
    assert _digest(sd) == rec["layout_sha256"], cfg


@pytest.mark.parametrize("cfg", sorted(k for k, v in SHIPPED.items() if v["args"]["model"] == "medformer"))
def test_medformer_attention_shapes_are_supported(cfg):
    """d_head = chan_num[i] // num_heads[i] with the constructor's default chan_num (medformer.py:20,38) and
    map_size -> codes: inside what cbim_bidir_attn_* / cbim_colsoftmax_pool_* accept (include/cbim_hip.h)."""
    from cbim_amd import _lib
    a = SHIPPED[cfg]["args"]
    chan = [64, 128, 256, 320, 256, 128, 64, 32]
    codes = a["map_size"][0] * a["map_size"][1] * a["map_size"][2]
    L = _lib.lib()
    assert 1 <= codes <= L.cbim_attn_wide_max_codes(), (cfg, codes)
    for i in (1, 2, 3, 4, 5):                       # the levels that hold transformer blocks
        if a["trans_num"][i]:
            assert chan[i] % a["num_heads"][i] == 0
            dh = chan[i] // a["num_heads"][i]
            assert L.cbim_bidir_attn_workspace(1, 4096, a["num_heads"][i], dh, codes) > 0
