"""Evaluation Dice with the reference's call surface (/root/reference/metric/utils.py:33-82).

The reference sums 0/1 masks in float32; those sums are integers, counted exactly on the GPU
(``cbim_dice_counts``) — the float32 arithmetic around them (the +1e-5 terms, including the reference's
double application in ``calculate_dice_split``) is replayed on the tiny per-class vectors, so results are
bit-identical to the reference's.  ``calculate_distance`` (ASD/HD, CPU surface metrics) is outside the hot path.
"""
import torch

from .. import _lib
from ..ops import _dev_ok, _p, _stream


def _counts(pred, target, C, block):
    pred, target = pred.contiguous().view(-1), target.contiguous().view(-1)
    if pred.dtype not in (torch.int8, torch.int64):
        pred = pred.long()
    if target.dtype not in (torch.int8, torch.int64):
        target = target.long()
    _dev_ok(pred, target)
    N = pred.numel()
    nblk = (N + block - 1) // block
    counts = torch.empty((nblk, C, 3), dtype=torch.int32, device=pred.device)
    _lib.check(_lib.lib().cbim_dice_counts(_p(pred), pred.element_size(), _p(target), target.element_size(), N, block, C,
                                           _p(counts), _stream(pred)), "dice_counts")
    return counts


def calculate_dice(pred, target, C):
    """pred, target: [N, 1] label tensors -> (dice[C], intersection[C], summ[C]) — metric/utils.py:62-82
    (note: the returned summ already includes the +1e-5, as in the reference)."""
    assert pred.shape[0] == target.shape[0]
    c = _counts(pred, target, C, max(int(pred.shape[0]), 1))[0]
    intersection = c[:, 0].to(torch.float32)
    summ = (c[:, 1] + c[:, 2]).to(torch.float32)
    summ += 1e-5
    return 2 * intersection / summ, intersection, summ


def calculate_dice_split(pred, target, C, block_size=64 * 64 * 64):
    """metric/utils.py:33-53: block-wise accumulation (each block's summ carries its own +1e-5)."""
    assert pred.shape[0] == target.shape[0]
    N = int(pred.shape[0])
    counts = _counts(pred, target, C, block_size)
    total_sum = torch.zeros(C, device=pred.device)
    total_intersection = torch.zeros(C, device=pred.device)
    for b in range(counts.shape[0]):
        total_intersection += counts[b, :, 0].to(torch.float32)
        summ = (counts[b, :, 1] + counts[b, :, 2]).to(torch.float32)
        summ += 1e-5
        total_sum += summ
    dice = 2 * total_intersection / (total_sum + 1e-5)
    return dice, total_intersection, total_sum
