"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL/xGMI
overlapped with the backward pass.

The reference's only parallelism is ``torch.nn.parallel.DistributedDataParallel`` in
``/root/reference/train_ddp.py:353`` (one replica per GPU, gradient all-reduce(mean) per step,
``find_unused_parameters=True``).  The models here have no unused parameters, so
the exchange is static: parameters are grouped into fixed buckets in REVERSE registration order
(the order the backward produces them: decoder first); every bucket owns ONE flat fp32 buffer and
``param.grad`` IS a view into it: the engine's weight-gradient kernels write their result straight into the slot
(``ops.grad_slot``: no copy at all — all conv / stem / head weights of the UNet family), any other gradient that arrives from
autograd is copied once into its slot (no ``torch.cat``, no copy back), and when the last gradient of a bucket has landed the flat buffer
is all-reduced IN PLACE, asynchronously, on the process group's own stream while the backward of
the earlier layers keeps the compute stream busy.  The optimizer then reads the averaged
gradients straight from the views.  ``backend="nccl"`` is RCCL on ROCm; the same code runs on
``gloo`` for the CPU tests.  Everything here is stream-ordered device work (copies, RCCL
collectives, stream waits), so a whole training step including the exchange can be captured in
one hipGraph (bench.py does that for N > 1 as it does for N = 1).

Sizing for xGMI (7 links x ~153 GB/s per GPU, point to point): a 162 MB fp32 gradient set in
~6 buckets of 32 MB keeps each ring transfer per-link bound at >= 4 MB per step of the ring,
large enough to run at link rate and small enough that the last bucket (encoder stem) is the
only one not hidden behind compute.
"""
from __future__ import annotations

import contextlib
from typing import List

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "pending", "flat", "work", "fired")

    def __init__(self):
        self.params: List[torch.nn.Parameter] = []
        self.offsets: List[int] = []
        self.numel = 0
        self.pending = 0
        self.flat = None
        self.work = None
        self.fired = False


class _GradSlot:
    """Handle attached to a parameter (`param._cbim_grad_slot`): lets the engine's weight-gradient kernels write straight into
    the parameter's slot of the flat bucket (ops.slot_of / ops.grad_slot)."""
    __slots__ = ("owner", "param", "view", "flat", "off", "__weakref__")

    def __init__(self, owner, param, view, flat, off):
        self.owner, self.param, self.view, self.flat, self.off = owner, param, view, flat, off

    def _free(self):
        o = self.owner
        return o._direct and self.param.grad is None and self.param not in o._claimed

    def claim(self):
        if not self._free():
            return None                       # accumulated gradient present / slot already written in this pass
        self.owner._claimed.add(self.param)
        self.owner.direct_writes += 1
        return self.view.detach()             # a fresh alias: autograd adopts it as param.grad without a copy

    def claim_with(self, other):
        """Both slots as ONE tensor [rows(self) + rows(other), ...] when `other` lies directly behind `self` in the same flat
        buffer (GradAllReduce lays out the pairs a module names in `cbim_grad_pairs()` that way): the Cout-concatenated
        weight gradient of conv1 | shortcut is then written in place.  -> (cat, alias_self, alias_other) or None."""
        n = self.param.numel()
        if (other is None or other.owner is not self.owner or other.flat is not self.flat or other.off != self.off + n
                or tuple(other.param.shape[1:]) != tuple(self.param.shape[1:]) or not self._free() or not other._free()):
            return None
        o = self.owner
        o._claimed.update((self.param, other.param))
        o.direct_writes += 2
        rows = int(self.param.shape[0]) + int(other.param.shape[0])
        cat = self.flat[self.off:self.off + n + other.param.numel()].view((rows,) + tuple(self.param.shape[1:]))
        return cat, self.view.detach(), other.view.detach()


class GradAllReduce:
    """Attach to a replica; gradients are averaged over the process group.

        ddp = GradAllReduce(net)              # after net.to(device); broadcasts rank-0 weights
        loss.backward()                       # buckets fire as their gradients complete
        ddp.synchronize()                     # wait; param.grad (views of the flat buffers) now hold the averages
                                              # (parameters that got no gradient keep grad None, like
                                              # find_unused_parameters=True)
        optimizer.step()

    Gradient accumulation over several backward passes (DistributedDataParallel.no_sync semantics)::

        with ddp.no_sync():
            loss_a.backward()                 # accumulates locally, nothing is exchanged
        loss_b.backward()                     # the exchange happens in the last backward
        ddp.synchronize()

    Every backward outside ``no_sync`` must be followed by ``synchronize()`` before the next one.
    """

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 32.0, process_group=None,
                 broadcast_parameters: bool = True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._exchange = dist.is_initialized()      # a 1-rank group still runs the collective (tests, graph capture)
        self._sync = True
        params = [p for p in module.parameters() if p.requires_grad]
        if broadcast_parameters and self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                # t.detach() shares the version counter with t (t.data does not): the in-place broadcast invalidates any
                # packed copy of a weight made by a forward pass that ran before the wrap (ops.PackedWeights)
                with torch.no_grad():
                    dist.broadcast(t.detach(), src=0, group=process_group)
        cap = int(bucket_mb * 1024 * 1024 / 4)
        # parameter pairs whose gradients one kernel writes as a single tensor (conv1 | shortcut of a BasicBlock): the second
        # is laid out directly behind the first
        follower, paired = {}, set()                     # (Parameters hash by identity)
        for m in module.modules():
            for a, b in (m.cbim_grad_pairs() if hasattr(m, "cbim_grad_pairs") else ()):
                if a.requires_grad and b.requires_grad and a.numel() % 4 == 0 and a not in paired and b not in paired:
                    follower[a] = b
                    paired.update((a, b))
        second = set(follower.values())
        self.buckets: List[_Bucket] = []
        cur = _Bucket()
        for p in reversed(params):
            if p in second:
                continue
            group = [p, follower[p]] if p in follower else [p]
            if any(q.dtype != torch.float32 for q in group):
                raise TypeError("cbim_amd: GradAllReduce expects float32 master parameters")
            if cur.params and cur.numel + sum(q.numel() for q in group) > cap:
                self.buckets.append(cur)
                cur = _Bucket()
            for q in group:
                cur.params.append(q)
                cur.offsets.append(cur.numel)
                cur.numel += (q.numel() + 3) // 4 * 4      # 16-byte aligned slots
        if cur.params:
            self.buckets.append(cur)
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None    # gloo has no AVG: SUM, then scale
        self._owner = {}
        self._views = {}
        self._handles = []
        self._direct = True          # weight-gradient kernels may write into the bucket slots (False: always copy)
        self._claimed = set()
        self.direct_writes = 0       # counters (tests, diagnostics): gradients written in place / copied into their slot
        self.copies = 0
        for b in self.buckets:
            dev = b.params[0].device
            b.flat = torch.zeros((b.numel,), dtype=torch.float32, device=dev)
            b.pending = len(b.params)
            for p, off in zip(b.params, b.offsets):
                self._owner[p] = b
                self._views[p] = b.flat[off:off + p.numel()].view_as(p)
                p._cbim_grad_slot = _GradSlot(self, p, self._views[p], b.flat, off)
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- hooks ------------------------------------------------------------------------------------------
    def _on_grad(self, p: torch.nn.Parameter):
        b = self._owner[p]
        view = self._views[p]
        g = p.grad
        if g is not view:
            if g.data_ptr() != view.data_ptr():  # fresh tensor from autograd (grad was None): move it into its slot
                view.copy_(g)                    # (a None grad means nothing was accumulated for p so far in this step)
                self.copies += 1
            # else: the weight-gradient kernel wrote the slot itself (_GradSlot.claim) and autograd adopted the alias
            p.grad = view
        # else: autograd accumulated in place into the view (grad was already the view)
        if not self._sync:
            return
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.fired = True
        b.pending = len(b.params)
        if not self._exchange:
            return
        op = self._avg_op if self._avg_op is not None else dist.ReduceOp.SUM
        b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate gradients locally; the first backward after the block exchanges the sums."""
        old = self._sync
        self._sync = False
        try:
            yield
        finally:
            self._sync = old

    def synchronize(self):
        """Wait for every bucket; ``param.grad`` (a view of the bucket's flat buffer) then holds the average."""
        for b in self.buckets:
            if not b.fired:
                # some parameters of this bucket received no gradient in this step (the reference wraps with
                # DistributedDataParallel(find_unused_parameters=True), train_ddp.py:353; e.g. AttentionUNet's unused
                # conv_ch): the bucket never fired from the hooks — exchange it now, zeros standing in for the
                # missing gradients (every rank sees the same graph, so every rank takes this branch)
                for p in b.params:
                    if p.grad is None:
                        self._views[p].zero_()
                self._launch(b)
            if b.work is not None:
                b.work.wait()
                b.work = None
                if self._avg_op is None and self.world > 1:
                    b.flat.div_(self.world)
            b.fired = False
            b.pending = len(b.params)
        self._claimed.clear()

    def reset(self):
        """Forget a step that was abandoned half-way (an exception inside forward / backward, a failed hipGraph capture): pending
        collectives are waited for where they can be, counters re-armed.  Every rank must abandon the same step."""
        for b in self.buckets:
            if b.work is not None:
                try:
                    b.work.wait()
                except Exception:
                    pass
                b.work = None
            b.fired = False
            b.pending = len(b.params)
        self._claimed.clear()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self._owner:
            if getattr(p, "_cbim_grad_slot", None) is not None and p._cbim_grad_slot.owner is self:
                del p._cbim_grad_slot
