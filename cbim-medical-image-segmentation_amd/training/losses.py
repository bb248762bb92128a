"""Loss modules with the reference's call surface (/root/reference/training/losses.py:8-58,
/root/reference/train.py:80-81,212), computed by the fused HIP Dice+CE kernels."""
import torch
import torch.nn as nn

from .. import functional as Fn


class DiceLoss(nn.Module):
    """DiceLoss()(preds[B,C,...], targets[B,1,...] int64) — losses.py:8-58 (size_average, reduce
    defaults; alpha/beta constructor arguments are overwritten by the adaptive alpha, :38-42)."""

    def __init__(self, alpha=0.5, beta=0.5, size_average=True, reduce=True):
        super().__init__()
        self.size_average, self.reduce = size_average, reduce

    def forward(self, preds, targets):
        with torch.autocast(device_type=preds.device.type, enabled=False):
            if not self.reduce:                    # losses.py:48-50: the per-class vector 1 - dice_c
                return Fn.DicePerClassFn.apply(preds, targets)
            loss = Fn.DiceCEFn.apply(preds, targets, None)[1]
            # losses.py:52-56: the sum over classes, divided by C only under size_average
            return loss if self.size_average else loss * int(preds.shape[1])


class DiceCELoss(nn.Module):
    """criterion(result, label.squeeze(1)) + criterion_dl(result, label) of train.py:212 in one
    pass; ``weight`` is the class weight of nn.CrossEntropyLoss (train.py:80)."""

    def __init__(self, weight=None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32))

    def forward(self, logits, label):
        with torch.autocast(device_type=logits.device.type, enabled=False):
            return Fn.DiceCEFn.apply(logits, label, self.weight)[2]


def check_labels(reset: bool = True) -> int:
    """Call once per epoch / validation pass: raises IndexError if any label outside [0, classes) reached the loss since the last
    call (the per-step check of the first calls is not repeated every step: it is a device synchronisation)."""
    return Fn.check_labels(reset)
