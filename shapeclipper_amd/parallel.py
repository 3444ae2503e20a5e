"""Single-node data parallelism for MI355X: one process per GPU, ONE flat RCCL all-reduce per step.

The reference wraps the graph in DistributedDataParallel (25 MB buckets, find_unused_parameters
graph walk, per-forward buffer broadcast; model/runner.py:121).  On an 8-GPU xGMI node the whole
gradient is 36.8 M fp32 = 147 MB, small next to the step, so this build keeps every gradient in one
contiguous buffer (parameters' .grad are views into it -- autograd accumulates in place, nothing is
copied) and issues a single all_reduce(SUM) over RCCL, then scales by 1/world.  Parameters that
received no gradient simply contribute zeros (the semantics of find_unused_parameters=True).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


class ModuleHolder(nn.Module):
    """Gives `.module` like DataParallel/DDP do (the runner reaches into graph.module.*)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class FlatGradAllReduce:
    """Gradients are produced by autograd as usual (`.grad` starts as None each step, so the first accumulation of a
    parameter is a pointer move, not an add kernel), then packed with one multi-tensor copy into the flat buffer,
    all-reduced with ONE collective, averaged, and handed back to the optimizer as views of the flat buffer."""

    def __init__(self, module: nn.Module, world_size: int, broadcast_buffers: bool = True, always_communicate: bool = False):
        self.module = module
        self.world = world_size
        self.comm = world_size > 1 or always_communicate      # always_communicate: run the collectives on 1 rank too (tests)
        self.params = [p for p in module.parameters() if p.requires_grad]
        # shared parameters (renderer.sdf_network is sdf_network) appear once in .parameters()
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buffers = [b for b in module.buffers() if b.dtype.is_floating_point] if broadcast_buffers else []
        if self.comm:
            self.broadcast_parameters()

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        """Copy every produced gradient into its slot of the flat buffer (parameters that received none contribute
        zeros, the semantics of find_unused_parameters=True) and point `.grad` at the slots."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if len(have) != len(self.params):
            self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(self.views, self.params):
            p.grad = v

    def broadcast_parameters(self, src=0):
        flat = torch.cat([p.data.reshape(-1) for p in self.params])
        dist.broadcast(flat, src)
        off = 0
        for p in self.params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def broadcast_buffers(self, src=0):
        """BN running statistics follow rank 0, as DDP's broadcast_buffers=True does each forward."""
        if not self.comm or not self.buffers:
            return
        flat = torch.cat([b.reshape(-1) for b in self.buffers])
        dist.broadcast(flat, src)
        views, off = [], 0
        for b in self.buffers:
            views.append(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        torch._foreach_copy_(self.buffers, views)

    def all_reduce(self):
        """Call after backward(): mean of the gradients over all ranks, one collective."""
        self.pack()
        if self.comm:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)


def gather_eval_records(records: torch.Tensor, world_size: int) -> torch.Tensor:
    """Sharded evaluation (BASELINE config[4]): every rank evaluated the samples idx % world == rank and holds
    `records` [n_local, 10] = (idx, cd_acc, cd_comp, f_score[6], category).  Returns on every rank the records of
    all ranks sorted by sample index ([N, 10]); rank 0 writes chamfer.txt / cd_cat.txt / f_score.txt from them.
    One small all_gather (ranks may hold different counts: padded with idx = -1)."""
    if world_size <= 1:
        return records[records[:, 0].argsort()]
    n_local = torch.tensor([records.shape[0]], device=records.device)
    counts = [torch.zeros_like(n_local) for _ in range(world_size)]
    dist.all_gather(counts, n_local)
    n_max = int(max(c.item() for c in counts))
    padded = records.new_full((n_max, records.shape[1]), -1.0)
    padded[:records.shape[0]] = records
    out = [torch.empty_like(padded) for _ in range(world_size)]
    dist.all_gather(out, padded)
    allr = torch.cat(out, 0)
    allr = allr[allr[:, 0] >= 0]
    return allr[allr[:, 0].argsort()]
