"""Single-node data parallelism for MI355X: one process per GPU, the gradients of a step in ONE flat buffer, exchanged over
RCCL / xGMI with one all-reduce -- or, overlapped with the backward pass, as its two contiguous halves (see FlatGradAllReduce).

The reference wraps the graph in DistributedDataParallel (25 MB buckets, find_unused_parameters
graph walk, per-forward buffer broadcast; model/runner.py:121).  On an 8-GPU xGMI node the whole
gradient is 36.8 M fp32 = 147 MB, small next to the step, so this build keeps every gradient in one
contiguous buffer (parameters' .grad end up as views into it) and all-reduces (SUM) that buffer, then scales by 1/world.
Parameters that received no gradient simply contribute zeros (the semantics of find_unused_parameters=True).
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


def _is_late(name: str) -> bool:
    """Parameters whose gradients are produced LAST by the backward pass: the stems and layer1 / layer2 of the two ResNet trunks
    (5.5 % of the bytes but about half of the trunks' backward time: the feature maps are largest there; everything else -- layer3 /
    layer4 with 90 % of the parameters, the heads, the MLPs, the projections -- is ready while those still run)."""
    parts = name.split(".")
    for trunk in ("encoder", "feature_extractor"):
        if trunk in parts:
            k = parts.index(trunk)
            return len(parts) > k + 1 and parts[k + 1] in ("conv1", "bn1", "layer1", "layer2")
    return False


class FlatGradAllReduce:
    """All gradients of a step in ONE flat buffer, exchanged over RCCL as the reference's DDP would (mean over ranks), without DDP's
    25 MB buckets, graph walk and per-forward buffer broadcast (model/runner.py:121).

    Gradients are produced by autograd as usual (`.grad` starts as None each step, so the first accumulation of a parameter is a
    pointer move, not an add kernel), packed with multi-tensor copies into the flat buffer, all-reduced (SUM) and averaged; the
    optimizer sees `.grad` as views of the flat buffer.

    Exchange schedule.  `overlap=False`: one all_reduce of the whole buffer after backward (the north-star form).  `overlap=True`
    (default on GPUs): the buffer is laid out [early | late] -- `late` = the trunks' stems, layer1 and layer2, whose gradients the
    backward pass produces last -- and the all-reduce of the EARLY segment (94.5 % of the 147 MB) is issued from a
    post-accumulate-grad hook on a side stream as soon as its last gradient exists, so it travels over xGMI while the early trunk
    layers are still being differentiated; after backward only the 8 MB late segment remains: two collectives per step instead of one,
    the same bytes, the same result (each element is summed over the ranks exactly once).  The schedule is STATIC: with overlap on,
    every rank issues exactly [early segment, late segment] in that order every step -- if an early parameter received no gradient
    on this rank (or a hook was missed), all_reduce() issues the early collective itself before the late one, so ranks can never
    disagree on the number or the sizes of the collectives of a step.
    BatchNorm running statistics follow rank 0 (DDP's broadcast_buffers=True): the floating-point buffers are re-homed once into a
    persistent flat tensor (they become views of it), so the per-step broadcast is ONE collective on that tensor, no cat / copy-back."""

    def __init__(self, module: nn.Module, world_size: int, broadcast_buffers: bool = True, always_communicate: bool = False,
                 overlap=None):
        self.module = module
        self.world = world_size
        self.comm = world_size > 1 or always_communicate      # always_communicate: run the collectives on 1 rank too (tests)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]       # shared parameters appear once
        early = [(n, p) for n, p in named if not _is_late(n)]
        late = [(n, p) for n, p in named if _is_late(n)]
        dev = named[0][1].device
        self.overlap = (dev.type == "cuda" and self.comm) if overlap is None else bool(overlap)
        if not self.overlap:
            early, late = named, []                            # module order, one segment
        self.params = [p for _, p in early + late]
        self.n_early = len(early)
        self.early_numel = sum(p.numel() for _, p in early)
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.collectives = 0                                   # data-path collectives issued so far (tests, bench)
        self._early_work, self._early_done, self._seen = None, False, set()
        self._comm_stream = torch.cuda.Stream(device=dev) if (self.overlap and dev.type == "cuda") else None
        self._main_stream = None
        self.split = bool(self.overlap and late)               # two collectives per step (static), else one
        self._hooks = []
        if self.split:
            self._early_ids = {id(p) for p in self.params[:self.n_early]}
            for p in self.params[:self.n_early]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        # persistent flat home of the floating-point buffers
        self.buffers, self.buf_flat = [], None
        if broadcast_buffers:
            bufs = [(m, k, b) for m in module.modules() for k, b in m._buffers.items() if b is not None and b.dtype.is_floating_point]
            if bufs:
                self.buf_flat = torch.cat([b.detach().reshape(-1).float() for _, _, b in bufs])
                off = 0
                for m, k, b in bufs:
                    view = self.buf_flat[off:off + b.numel()].view_as(b)
                    b.data = view                              # same Tensor object (state_dict, BN kernels), storage inside buf_flat
                    self.buffers.append(b)
                    off += b.numel()
        if self.comm:
            self.broadcast_parameters()

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def close(self):
        """Remove the gradient hooks (a second reducer on the same module must not trigger this one's collectives)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def _reset_step(self):
        self._early_work, self._early_done, self._seen = None, False, set()

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self._reset_step()
        if self._comm_stream is not None:
            self._main_stream = torch.cuda.current_stream()    # the stream the step is enqueued on (see _launch_early)

    # ---- packing ---------------------------------------------------------------------------------------------------------
    def _pack_range(self, lo, hi):
        """Copy the produced gradients of params[lo:hi] into their slots (parameters that received none contribute zeros, the
        semantics of find_unused_parameters=True) and point `.grad` at the slots."""
        views, params = self.views[lo:hi], self.params[lo:hi]
        have = [(v, p.grad) for v, p in zip(views, params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if len(have) != len(params) and hi > lo:
            a = self.views[lo].data_ptr() - self.flat.data_ptr()
            self.flat[a // 4: a // 4 + sum(p.numel() for p in params)].zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, params):
            p.grad = v
        return have

    def pack(self):
        self._pack_range(0, len(self.params))

    # ---- early segment: issued from inside backward ---------------------------------------------------------------------------
    def _on_grad(self, param):
        if not self._hooks:                                    # closed
            return
        self._seen.add(id(param))                              # distinct parameters, not hook firings
        if len(self._seen) == self.n_early and not self._early_done and self.comm:
            self._launch_early()

    def _launch_early(self):
        seg = self.flat[:self.early_numel]
        self._early_done = True
        if self._comm_stream is None:                          # CPU / gloo: no streams, same schedule
            self._pack_range(0, self.n_early)
            self._early_work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
            self.collectives += 1
            return
        grads = [p.grad for p in self.params[:self.n_early] if p.grad is not None]
        # The hook runs under the stream of the LAST early gradient's node -- the estimator trunk's side stream as often as the main
        # one -- while the other early gradients were produced on every stream the step uses.  All of them have been enqueued by now
        # (one autograd worker per device), so the side stream waits for the work queued so far on each: the hook's stream, the stream
        # the step was started on, the device's default stream and the trunks' side stream.  (Waiting only for the hook's stream and
        # the side stream let the pack read unfinished main-stream gradients whenever the hook fired under the side stream:
        # tests/test_gpu_two_ranks.py.)
        waited = set()
        for s in [torch.cuda.current_stream(), self._main_stream, torch.cuda.default_stream(seg.device)] + _known_streams(seg.device):
            if s is not None and s.cuda_stream not in waited and s.cuda_stream != self._comm_stream.cuda_stream:
                waited.add(s.cuda_stream)
                self._comm_stream.wait_stream(s)
        with torch.cuda.stream(self._comm_stream):
            for g in grads:
                g.record_stream(self._comm_stream)             # produced on another stream, read here
            self._pack_range(0, self.n_early)
            self._early_work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
        self.collectives += 1

    def broadcast_parameters(self, src=0):
        flat = torch.cat([p.data.reshape(-1) for p in self.params])
        dist.broadcast(flat, src)
        off = 0
        for p in self.params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def broadcast_buffers(self, src=0):
        """BN running statistics follow rank 0, as DDP's broadcast_buffers=True does each forward: ONE collective on the persistent
        flat tensor the buffers live in."""
        if not self.comm or self.buf_flat is None:
            return
        dist.broadcast(self.buf_flat, src)

    def all_reduce(self):
        """Call after backward(): mean of the gradients over all ranks."""
        if not self.split:                                     # one segment, one collective
            self.pack()
            if self.comm:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self.collectives += 1
                self.flat.mul_(1.0 / self.world)
            self._reset_step()
            return
        if not self._early_done:
            # an early parameter received no gradient on this rank (its slot contributes zeros): the SAME two collectives as on
            # every other rank, the early one simply starts here instead of inside the backward pass
            self._pack_range(0, self.n_early)
            if self.comm:
                dist.all_reduce(self.flat[:self.early_numel], op=dist.ReduceOp.SUM)
                self.collectives += 1
            self._early_done = True
        self._pack_range(self.n_early, len(self.params))
        if self.comm:
            dist.all_reduce(self.flat[self.early_numel:], op=dist.ReduceOp.SUM)
            self.collectives += 1
            if self._early_work is not None:
                self._early_work.wait()                        # the current stream waits for the side-stream collective
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
            self.flat.mul_(1.0 / self.world)
        self._reset_step()

    def reduce_flag(self, bad: torch.Tensor):
        """MAX over the ranks of a NaN/Inf flag (one element): every rank takes the SAME decision before optim.step() -- with a local
        flag only the poisoned rank raised and the others stepped on the all-reduced (poisoned) gradients, then hung in the next
        collective.  A control message, not part of the gradient exchange: issued right after the forward pass (on the side stream
        when there is one), so it is long complete when the host asks for it.  Returns a device tensor (float, > 0 = bad)."""
        f = bad.detach().reshape(1).to(torch.float32)
        if not self.comm:
            return f
        if self._comm_stream is None:
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            return f
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            f.record_stream(self._comm_stream)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
        return f


def _known_streams(device):
    """Streams the backward pass may be running on besides the current one (the estimator trunk's side stream, model/graph.py)."""
    try:
        from .model.graph import _SIDE_STREAMS
        s = _SIDE_STREAMS.get(str(device))
        return [s] if s is not None else []
    except Exception:       # noqa: BLE001
        return []


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
