"""Name -> class maps with the reference's keys (/root/reference/model/dim3/utils.py:7-30)."""
from .conv_layers import BasicBlock, Bottleneck, SingleConv

_NORMS = ("in",)           # every shipped 3D config uses `norm: in`
_ACTS = ("relu", "lrelu", "gelu", "swish")


def get_block(name):
    blocks = {"SingleConv": SingleConv, "BasicBlock": BasicBlock, "Bottleneck": Bottleneck}
    return blocks[name]  # KeyError for unknown names, like the reference


def get_norm(name):
    if name in _NORMS:
        return name
    if name in ("bn", "ln"):
        raise NotImplementedError(
            f"cbim_amd: norm '{name}' is not built; all shipped 3D configs use InstanceNorm ('in')")
    raise KeyError(name)


def get_act(name):
    if name not in _ACTS:
        raise KeyError(name)
    return name
