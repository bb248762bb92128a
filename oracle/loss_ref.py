"""Oracle: CrossEntropy + Dice loss of the reference train step.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``/root/reference/train.py:80-81,212`` (criterion = nn.CrossEntropyLoss(weight),
criterion_dl = DiceLoss(); loss = CE(result, label.squeeze(1)) + Dice(result, label)) and
``/root/reference/training/losses.py:18-58`` (DiceLoss.forward).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SMOOTH = 1e-5  # losses.py:24


def cross_entropy(logits, labels, weight=None):
    """nn.CrossEntropyLoss(weight=w)(logits[B,C,...], labels[B,...] int64)  (train.py:80,212):
    sum_i w[y_i] * (lse(z_i) - z_i[y_i]) / sum_i w[y_i]."""
    return F.cross_entropy(logits, labels, weight=weight)


def dice_stats(logits, labels):
    """Per-class TP / FP / FN over batch AND space (losses.py:23-36)."""
    C = logits.shape[1]
    P = F.softmax(logits, dim=1)                                   # :23
    mask = torch.zeros_like(logits).scatter_(1, labels, 1.0)      # :26-27
    red = [0] + list(range(2, logits.dim()))
    TP = (P * mask).sum(red)                                       # :33
    FP = (P * (1 - mask)).sum(red)                                 # :34
    FN = ((1 - P) * mask).sum(red)                                 # :35
    return TP, FP, FN


def dice_loss(logits, labels, size_average=True, reduce=True):
    """DiceLoss(size_average, reduce)(preds[B,C,...], targets[B,1,...] int64)  (losses.py:18-58).

    alpha_c = clamp(FP/(FP+FN+smooth), 0.2, 0.8) is NOT detached (:38-40): the gradient
    flows through alpha wherever it is not clamped."""
    C = logits.shape[1]
    TP, FP, FN = dice_stats(logits, labels)
    alpha = torch.clamp(FP / (FP + FN + SMOOTH), min=0.2, max=0.8)  # :38-40
    beta = 1 - alpha                                               # :42
    den = TP + alpha * FP + beta * FN                              # :44
    dice = TP / (den + SMOOTH)                                     # :46
    if not reduce:
        return 1 - dice                                            # :48-50  per-class vector
    loss = (1 - dice).sum()                                        # :52-53
    return loss / C if size_average else loss                      # :55-56


def ce_dice_loss(logits, labels, weight=None):
    """train.py:212  loss = criterion(result, label.squeeze(1)) + criterion_dl(result, label)."""
    return cross_entropy(logits, labels.squeeze(1), weight) + dice_loss(logits, labels)


def hard_dice(pred, target, num_classes):
    """metric/utils.py:62-82 calculate_dice on integer label maps:
    2|P∩T| / (|P|+|T|+1e-5) per class (class 0 = background included here)."""
    out = []
    for c in range(num_classes):
        p = pred == c
        t = target == c
        inter = (p & t).sum().double()
        out.append(2 * inter / (p.sum().double() + t.sum().double() + 1e-5))
    return torch.stack(out)
