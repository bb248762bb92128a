"""Name -> class maps with the reference's keys (/root/reference/model/dim3/utils.py:7-30)."""
from .conv_layers import BasicBlock, Bottleneck, SingleConv

_NORMS = ("in", "bn", "ln")   # every shipped 3D config uses `norm: in`; `bn` = nn.BatchNorm3d, `ln` = the channels-first
                              # LayerNorm of trans_layers.py:120-149 (round 5: composed paths of the UNet / ResUNet blocks)
_ACTS = ("relu", "lrelu", "gelu", "swish")


def get_block(name):
    blocks = {"SingleConv": SingleConv, "BasicBlock": BasicBlock, "Bottleneck": Bottleneck}
    return blocks[name]  # KeyError for unknown names, like the reference


def get_norm(name, allow=_NORMS):
    """`allow`: the norms the calling model family is built for (UNet / ResUNet: in, bn, ln; the others: in)"""
    if name in _NORMS and name not in allow:
        raise NotImplementedError(f"cbim_amd: norm '{name}' is built for the UNet / ResUNet blocks only")
    if name in _NORMS:
        return name
    raise KeyError(name)


def get_act(name):
    if name not in _ACTS:
        raise KeyError(name)
    return name
