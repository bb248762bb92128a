"""The names train.py / train_ddp.py import from ``training.utils`` for the step around the model
(/root/reference/train.py:16-22,94,216-218; /root/reference/training/utils.py:8-14,50-105), same signatures:

    get_optimizer(args, net)                         -> FusedAdamW for ``optimizer: adamw`` (every shipped 3D yaml)
    update_ema_variables(model, ema_model, alpha, global_step)   one multi-tensor launch (cbim_ema_step)
    exp_lr_scheduler_with_warmup / multistep_lr_scheduler_with_warmup   host arithmetic on optimizer.param_groups

Logging / checkpoint glue of the same reference file (log_evaluation_result, unwrap_model_checkpoint, ...) is outside
the hot path and stays with the reference.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch
from torch import optim

from .. import _lib
from ..ops import _p, _stream
from .optim import FusedAdamW, _Rec

__all__ = ["get_optimizer", "update_ema_variables", "exp_lr_scheduler_with_warmup", "multistep_lr_scheduler_with_warmup"]


def get_optimizer(args, net):
    """training/utils.py:8-14.  adamw (eps 1e-5) is the fused multi-tensor HIP step; sgd / adam, which no shipped 3D
    configuration selects, are torch's own optimizers on the same parameters."""
    if args.optimizer == "sgd":
        return optim.SGD(net.parameters(), lr=args.base_lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if args.optimizer == "adam":
        return optim.Adam(net.parameters(), lr=args.base_lr, betas=args.betas, weight_decay=args.weight_decay)
    if args.optimizer == "adamw":
        return FusedAdamW(net.parameters(), lr=args.base_lr, betas=tuple(args.betas), weight_decay=args.weight_decay, eps=1e-5)
    return None    # the reference falls through the same way


class _EmaTable:
    """Device table of (param, ema) records for one (model, ema_model) pair, rebuilt when a storage moves."""

    def __init__(self, model, ema_model):
        self.pairs = [(e, p) for e, p in zip(ema_model.parameters(), model.parameters())]     # utils.py:101 zip order
        for e, p in self.pairs:
            if e.dtype != torch.float32 or p.dtype != torch.float32 or not e.is_contiguous() or not p.is_contiguous():
                raise RuntimeError("cbim_amd: update_ema_variables needs contiguous fp32 parameters")
            if e.shape != p.shape or e.device != p.device:
                raise RuntimeError("cbim_amd: update_ema_variables: model / ema_model parameters differ in shape or device")
        dev = self.pairs[0][1].device
        chunk = _lib.lib().cbim_optim_chunk()
        bt, bc = [], []
        for i, (_, p) in enumerate(self.pairs):
            n = (p.numel() + chunk - 1) // chunk
            bt += [i] * n
            bc += list(range(n))
        self.blk_tensor = torch.tensor(bt, dtype=torch.int32, device=dev)
        self.blk_chunk = torch.tensor(bc, dtype=torch.int32, device=dev)
        self.nblocks = len(bt)
        self.table = torch.empty((len(self.pairs) * C.sizeof(_Rec),), dtype=torch.uint8, device=dev)
        self.ptrs = None

    def fill(self):
        ptrs = tuple((p.data_ptr(), e.data_ptr()) for e, p in self.pairs)
        if ptrs == self.ptrs:
            return
        host = torch.empty((len(self.pairs) * C.sizeof(_Rec),), dtype=torch.uint8)
        recs = (_Rec * len(self.pairs)).from_address(host.data_ptr())
        for i, (e, p) in enumerate(self.pairs):
            recs[i] = _Rec(p.data_ptr(), None, None, None, e.data_ptr(), p.numel())
        self.table.copy_(host)
        self.ptrs = ptrs


_TABLES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


@torch.no_grad()
def update_ema_variables(model, ema_model, alpha, global_step):
    """training/utils.py:98-105: alpha = min(1 - 1/(global_step+1), alpha); ema = alpha*ema + (1-alpha)*param for every
    parameter (one launch over all tensors), then the buffers are copied (none in the shipped 3D nets)."""
    alpha = min((1 - 1 / (global_step + 1)), alpha)
    key = _TABLES.get(ema_model)
    if key is None or key[0]() is not model:
        tab = _EmaTable(model, ema_model)
        _TABLES[ema_model] = (weakref.ref(model), tab)
    else:
        tab = key[1]
    if tab.pairs:
        tab.fill()
        p0 = tab.pairs[0][1]
        _lib.check(_lib.lib().cbim_ema_step(_p(tab.table), _p(tab.blk_tensor), _p(tab.blk_chunk), tab.nblocks, float(alpha),
                                            float(1 - alpha), _stream(p0)), "ema_step")
        # the kernel wrote the EMA parameters through raw pointers: move their version counters, which is what the
        # packed-weight cache (ops.PackedWeights) keys on — an ema_net forward after this must re-pack
        torch.autograd.graph.increment_version([e for e, _ in tab.pairs])
    for ema_buffer, m_buffer in zip(ema_model.buffers(), model.buffers()):
        ema_buffer.copy_(m_buffer)


def multistep_lr_scheduler_with_warmup(optimizer, init_lr, epoch, warmup_epoch, lr_decay_epoch, max_epoch, gamma=0.1):
    """training/utils.py:50-74."""
    if 0 <= epoch <= warmup_epoch:
        lr = init_lr if epoch == warmup_epoch else init_lr * 2.718 ** (10 * (float(epoch) / float(warmup_epoch) - 1.))
        for group in optimizer.param_groups:
            group["lr"] = lr
        return lr
    for i, e in enumerate(lr_decay_epoch):
        if epoch == e:
            lr = init_lr * gamma ** (i + 1)
            for group in optimizer.param_groups:
                group["lr"] = lr
            return lr
    return optimizer.param_groups[0]["lr"]


def exp_lr_scheduler_with_warmup(optimizer, init_lr, epoch, warmup_epoch, max_epoch):
    """training/utils.py:77-94 (train.py:94 calls it with warmup_epoch=5 whatever the yaml says)."""
    if 0 <= epoch <= warmup_epoch and warmup_epoch != 0:
        lr = init_lr if epoch == warmup_epoch else init_lr * 2.718 ** (10 * (float(epoch) / float(warmup_epoch) - 1.))
    else:
        lr = init_lr * (1 - epoch / max_epoch) ** 0.9
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr
