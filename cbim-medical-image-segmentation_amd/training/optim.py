"""Fused AdamW (+ EMA of the weights): the reference's ``get_optimizer(args, net)`` for ``optimizer: adamw``
(/root/reference/training/utils.py:8-14) and ``update_ema_variables`` (:98-105) as ONE kernel launch per
iteration (``csrc/optim_kernels.hip``).

``FusedAdamW`` is a ``torch.optim.Optimizer``: ``param_groups`` (the lr schedulers of training/utils.py mutate
``param_group['lr']``), ``zero_grad``, ``state_dict``/``load_state_dict`` with torch's AdamW state keys
(``step``, ``exp_avg``, ``exp_avg_sq``) so checkpoints interchange (train.py:103-107).  Pass ``ema_model`` to fold
``update_ema_variables(model, ema_model, alpha, global_step)`` into the same launch; its buffers are copied as
the reference does (the shipped nets have none).  One parameter group, fp32 parameters on the GPU; anything
else raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..ops import _p, _stream


class _Rec(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("numel", C.c_int64)]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ema_model=None, ema_alpha=0.99):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("cbim_amd: FusedAdamW supports one parameter group")
        self._params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        for p in self._params:
            if p.dtype != torch.float32:
                raise NotImplementedError("cbim_amd: FusedAdamW needs fp32 parameters")
        self.ema_model, self.ema_alpha = ema_model, float(ema_alpha)
        self._ema_params = None
        if ema_model is not None:
            self._ema_params = [p for p in ema_model.parameters()]
            if len(self._ema_params) != len(list(self.param_groups[0]["params"])):
                raise ValueError("ema_model must have the same parameters as the optimised model")
            self._ema_params = [e for e, p in zip(self._ema_params, self.param_groups[0]["params"]) if p.requires_grad]
            self._ema_all = list(self._ema_params)
            self._ema_active = list(self._ema_params)
        self._all_params = list(self._params)
        self._active = None
        self._table = None
        self._ptrs = None
        self._hyper = None
        self._lr_pushed = None

    # -- state (torch.optim.AdamW layout) ---------------------------------------------------------------
    def _init_state(self):
        for p in self._params:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _build(self):
        dev = self._params[0].device
        chunk = _lib.lib().cbim_optim_chunk()
        bt, bc = [], []
        for i, p in enumerate(self._params):
            n = (p.numel() + chunk - 1) // chunk
            bt += [i] * n
            bc += list(range(n))
        self._blk_tensor = torch.tensor(bt, dtype=torch.int32, device=dev)
        self._blk_chunk = torch.tensor(bc, dtype=torch.int32, device=dev)
        self._nblocks = len(bt)
        g = self.param_groups[0]
        step0 = float(self.state[self._params[0]]["step"])
        self._hyper = torch.tensor([step0, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                                    self.ema_alpha, 0.0, 1.0 - g["betas"][0], 1.0 - g["betas"][1]], dtype=torch.float32,
                                   device=dev)
        self._lr_pushed = self._hyper_key(g)
        # pinned staging buffers, allocated up front (no host allocation may happen while a hipGraph is being
        # captured) and used round-robin: a captured graph re-reads the buffer it was recorded with on replay
        nb = len(self._params) * C.sizeof(_Rec)
        self._hosts = [torch.empty((nb,), dtype=torch.uint8).pin_memory() if dev.type == "cuda"
                       else torch.empty((nb,), dtype=torch.uint8) for _ in range(8)]
        self._host_i = 0
        self._table = torch.empty((len(self._params) * C.sizeof(_Rec),), dtype=torch.uint8, device=dev)

    def _fill_table(self):
        ptrs = tuple((p.data_ptr(), p.grad.data_ptr()) for p in self._params)
        if ptrs == self._ptrs:
            return
        host = self._hosts[self._host_i % len(self._hosts)]
        self._host_i += 1
        recs = (_Rec * len(self._params)).from_address(host.data_ptr())
        for i, p in enumerate(self._params):
            st = self.state[p]
            e = self._ema_active[i] if self._ema_params is not None else None
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or not p.is_contiguous():
                raise RuntimeError("cbim_amd: FusedAdamW needs contiguous fp32 parameters and gradients")
            recs[i] = _Rec(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                           e.data_ptr() if e is not None else None, p.numel())
        self._table.copy_(host, non_blocking=True)
        self._ptrs = ptrs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # parameters that received no gradient are skipped, as torch.optim does (e.g. AttentionUNet's unused conv_ch)
        active = tuple(i for i, p in enumerate(self._all_params) if p.grad is not None)
        if not active:
            return loss
        if active != self._active:
            step_now = float(self._hyper[0]) if self._hyper is not None else None
            self._params = [self._all_params[i] for i in active]
            if self._ema_params is not None:
                self._ema_active = [self._ema_all[i] for i in active]
            self._active, self._table, self._ptrs = active, None, None
            self._init_state()
            self._build()
            if step_now is not None:
                self._hyper[0] = step_now
        self._init_state()
        if self._table is None:
            self._build()
        g = self.param_groups[0]
        key = self._hyper_key(g)
        if key != self._lr_pushed:              # the scheduler changed the learning rate (training/utils.py:51-95), or the
            # user edited betas / eps / weight_decay in param_groups: push the whole tuple (one small H2D copy; under a
            # captured graph the step is replayed with the values captured — edit them between captures)
            vals = torch.tensor([key[0], key[1], key[2], key[3], key[4], 1.0 - key[1], 1.0 - key[2]], dtype=torch.float32)
            self._hyper[1:6] = vals[:5].to(self._hyper.device)
            self._hyper[8:10] = vals[5:].to(self._hyper.device)
            self._lr_pushed = key
        self._fill_table()
        _lib.check(_lib.lib().cbim_adamw_ema_step(_p(self._table), _p(self._blk_tensor), _p(self._blk_chunk), self._nblocks,
                                                  _p(self._hyper), _stream(self._params[0])), "adamw_ema_step")
        # the kernel wrote the parameters (and the EMA copies) through raw pointers: tell autograd / every cache keyed on
        # Tensor._version (ops.PackedWeights: packed convolution weights) that they changed
        torch.autograd.graph.increment_version(self._params)
        if self._ema_params is not None:
            torch.autograd.graph.increment_version([e for e in self._ema_active if e is not None])
        if self.ema_model is not None:          # training/utils.py:104-105 (no-op for the shipped nets: 0 buffers)
            for eb, mb in zip(self.ema_model.buffers(), self._model_buffers()):
                eb.copy_(mb)
        return loss

    @staticmethod
    def _hyper_key(g):
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))

    def _model_buffers(self):
        src = getattr(self, "_buffers_src", None)
        if src is None:
            # update_ema_variables copies every buffer of the model into the EMA model (training/utils.py:104-105); an EMA
            # model WITH buffers (SwinUNETR: relative_position_index) needs the source model: fail loudly, not silently
            if any(True for _ in self.ema_model.buffers()):
                raise RuntimeError("cbim_amd: FusedAdamW(ema_model=...) mirrors buffers; call attach_buffers(model) "
                                   "(get_optimizer does) so that the EMA model's buffers follow the model's")
            return ()
        return src

    def attach_buffers(self, model):
        """Give the model whose buffers the EMA model mirrors (only needed for nets that have buffers)."""
        self._buffers_src = tuple(model.buffers())

    def state_dict(self):
        if self._hyper is not None:            # publish the device step counter in torch's per-parameter form
            step = self._hyper[0].detach().clone()
            for p in self._params:
                self.state[p]["step"] = step.clone()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = None
        self._ptrs = None


def get_optimizer(args, net, ema_net=None):
    """training/utils.py:8-14 with the EMA update (training/utils.py:98-105) folded into the same launch for ``optimizer: adamw``;
    ``sgd`` / ``adam`` (no shipped 3-D configuration selects them) are torch's own optimizers on the engine's parameters, exactly
    as the reference builds them — their EMA stays with ``training.utils.update_ema_variables``."""
    if args.optimizer == "sgd":
        return torch.optim.SGD(net.parameters(), lr=args.base_lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if args.optimizer == "adam":
        return torch.optim.Adam(net.parameters(), lr=args.base_lr, betas=args.betas, weight_decay=args.weight_decay)
    if args.optimizer != "adamw":
        return None    # the reference falls through the same way
    opt = FusedAdamW(net.parameters(), lr=args.base_lr, betas=args.betas, weight_decay=args.weight_decay, eps=1e-5,
                     ema_model=ema_net, ema_alpha=getattr(args, "ema_alpha", 0.99))
    if ema_net is not None:
        opt.attach_buffers(net)
    return opt
