"""Data-parallel training over the GPUs of one MI355X node: one process per GPU, whole scans sharded
across ranks, gradients averaged with bucketed RCCL all-reduce overlapped with backward.

This replaces what the reference gets implicitly from Lightning's ``strategy='ddp'``
(/root/reference: train_mask_bev.py:92-96; collective call sites C1-C6 of SURVEY.md §2b):

* C1 gradient all-reduce → buckets filled in the order gradients become ready (post-accumulate hooks),
  each launched asynchronously on RCCL's own stream as soon as it is full, so the large early buckets
  (decoder, pixel decoder, Swin stage 3/4) overlap the rest of backward.  The two 134 MB (C, ny, nx)
  LayerNorm-affine gradients are produced LAST (first op after the scatter) and get buckets of their
  own; they are the exposed tail.
* C2 ``reduce_mean(avg_factor)`` and C3 per-scalar ``sync_dist`` logging → elided: with equal per-rank
  batches (``drop_last=True``) the value is the same constant B·Q on every rank, and logged scalars are
  reduced in one stacked all-reduce by :func:`reduce_scalars`.
* C5 buffer broadcast (PFN BatchNorm running stats, < 2 KB) → one flat broadcast per step.
* C6 parameter broadcast at construction.

xGMI is point-to-point (7 links x ~153 GB/s): bucket size defaults to 64 MB so that each collective is
long enough to be link-bound rather than launch-bound.  Works with the ``nccl`` (= RCCL) backend on GPUs
and with ``gloo`` on CPU (tests).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn


class _Bucket:
    __slots__ = ('params', 'numel', 'flat', 'seen', 'handle', 'offsets')

    def __init__(self):
        self.params: List[nn.Parameter] = []
        self.numel = 0
        self.flat: Optional[torch.Tensor] = None
        self.seen = set()                   # ids of the parameters whose gradient has been announced this step
        self.handle = None
        self.offsets: List[int] = []


class _WireHandle:
    """An arena chunk reduced in a 16-bit wire type: ``wait()`` orders the current stream behind the collective and
    writes the sum back into the f32 arena view."""
    __slots__ = ('work', 'wire', 'view')

    def __init__(self, work, wire, view):
        self.work, self.wire, self.view = work, wire, view

    def wait(self):
        self.work.wait()
        self.view.copy_(self.wire)


class GradientAllReducer:
    """Bucketed, backward-overlapped gradient averaging for ``module``'s parameters."""

    def __init__(self, module: nn.Module, bucket_mb: float = 64.0, process_group=None,
                 grad_dtype: Optional[torch.dtype] = None, broadcast_buffers: bool = True):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.grad_dtype = grad_dtype          # e.g. torch.bfloat16 halves the bytes on the wire
        self.broadcast_buffers = broadcast_buffers
        cap = int(bucket_mb * 1024 * 1024)
        # Gradients become ready roughly in reverse order of registration; fill buckets in that order.
        params = [p for p in module.parameters() if p.requires_grad]
        self.buckets: List[_Bucket] = []
        cur = _Bucket()
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur.params and (cur.numel * 4 + nbytes > cap):
                self.buckets.append(cur)
                cur = _Bucket()
            cur.offsets.append(cur.numel)
            cur.params.append(p)
            cur.numel += p.numel()
        if cur.params:
            self.buckets.append(cur)
        self._bucket_of: Dict[int, _Bucket] = {}
        self._hooks = []
        # Parameters that live in a ParameterArena (arena.py) receive their gradients by in-place accumulation, once
        # per USE of the parameter (a packed in_proj weight announces itself three times): a per-parameter ready-hook
        # cannot tell the last use from the first, so the hook-driven buckets are refused for them — the arena's
        # gradient is reduced in contiguous ranges after the backward instead (finish() -> reduce_arena()).
        self.arena = getattr(module, '_arena', None)
        if self.arena is None and any(getattr(p, '_mbv_arena', False) for p in params):
            raise RuntimeError('parameters live in a ParameterArena the module does not expose as `_arena`: '
                               'reduce it with reduce_arena() / start_ranges()')
        if self.arena is None:
            for b in self.buckets:
                for p in b.params:
                    self._bucket_of[id(p)] = b
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        self._active = self.arena is None
        self._next = 0                      # buckets are launched in index order on every rank
        self.debug_pending = None           # tests: [(RangeReady, (lo, hi)), ...] checked by start_ranges
        self.sync_parameters()
        self._reset()

    # -- construction-time collectives (C6) and per-step buffer broadcast (C5)
    @torch.no_grad()
    def sync_parameters(self):
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src=0, group=self.group)
        arena = getattr(self.module, '_arena', None)
        if arena is not None:               # the broadcast wrote the f32 arena: the bf16 shadow the GEMMs read follows
            arena.refresh_shadow()

    @torch.no_grad()
    def sync_buffers_begin(self):
        """Launch the broadcast of rank 0's floating-point buffers (the PFN's BatchNorm running statistics: < 2 KB) as ONE
        asynchronous collective on the communication stream and return a handle for :meth:`sync_buffers_end` — nothing
        here waits on the host, so the step's kernels behind it are issued at once."""
        if not self.broadcast_buffers:
            return None
        bufs = [b for b in self.module.buffers() if b.is_floating_point()]
        if not bufs:
            return None
        flat = torch.cat([b.reshape(-1).float() for b in bufs])
        work = dist.broadcast(flat, src=0, group=self.group, async_op=True)
        return work, flat, bufs

    @torch.no_grad()
    def sync_buffers_end(self, handle) -> None:
        """Order the current stream behind the broadcast and write rank 0's values into the buffers."""
        if handle is None:
            return
        work, flat, bufs = handle
        work.wait()
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()

    def sync_buffers(self):
        """Lightning's ``strategy='ddp'`` default (``broadcast_buffers=True``): every forward starts from rank 0's
        buffers (/root/reference: train_mask_bev.py:92-96).  Eager step: begin + end back to back."""
        self.sync_buffers_end(self.sync_buffers_begin())

    def _reset(self):
        for b in self.buckets:
            b.seen = set()
            b.handle = None
        self._next = 0

    def _on_grad_ready(self, p: nn.Parameter):
        if not self._active:
            return
        b = self._bucket_of[id(p)]
        b.seen.add(id(p))                   # a set, not a counter: a second announcement of p changes nothing
        self._launch_ready()

    def _launch_ready(self):
        """Launch complete buckets strictly in index order: the sequence of collectives is then the same on every
        rank whatever order (or subset) of gradients each rank produced."""
        while self._next < len(self.buckets) and len(self.buckets[self._next].seen) == len(self.buckets[self._next].params):
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b: _Bucket):
        dtype = self.grad_dtype or b.params[0].grad.dtype
        if b.flat is None or b.flat.dtype != dtype or b.flat.device != b.params[0].grad.device:
            b.flat = torch.empty(b.numel, dtype=dtype, device=b.params[0].grad.device)
        for p, off in zip(b.params, b.offsets):
            b.flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        b.flat.div_(self.world)
        b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, optimizer=None):
        """Call after ``loss.backward()``: waits for every bucket and scatters the averaged gradients back.
        Parameters that received no gradient this step (unused) are treated as zero, like DDP's
        ``find_unused_parameters=True`` default under Lightning 1.9; the remaining buckets
        are launched in index order, so ranks whose sets of unused parameters differ still pair their collectives.
        With a parameter arena the arena gradient is reduced in place, in contiguous ranges (no bucket copies)."""
        if self.arena is not None:
            self.reduce_arena(self.arena, optimizer)
            return
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            for p in b.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            self._launch(b)
            self._next += 1
        for b in self.buckets:
            b.handle.wait()
            for p, off in zip(b.params, b.offsets):
                p.grad.copy_(b.flat[off:off + p.numel()].view_as(p.grad))
        self._reset()

    def reduce_all(self):
        """Average every gradient now (all buckets launched back to back, then awaited).  Used when the backward
        ran inside a HIP graph (mask_bev_amd/graph.py), where the per-parameter hooks are switched off with
        ``no_sync(True)`` because collectives must not be issued during capture / replay."""
        self.finish()

    def _reduce_chunk(self, arena, lo: int, hi: int):
        """One asynchronous SUM all-reduce over arena.grad[lo:hi].  ``grad_dtype`` (bf16 / fp16) set: the chunk travels in
        that type — half the bytes on every xGMI link — through a staging copy that :meth:`finish_arena` writes back
        (the sum is formed in the wire type: an opt-in trade of gradient bits for link time; default = f32, in place)."""
        view = arena.grad[lo:hi]
        if self.grad_dtype is None or self.grad_dtype == view.dtype:
            return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        wire = view.to(self.grad_dtype)
        work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return _WireHandle(work, wire, view)

    def start_arena(self, arena, names, chunk_mb: float = 256.0) -> list:
        """Launch SUM all-reduces over the named arena segments' gradients (arena.py) in contiguous chunks — no
        bucket copies; asynchronous on RCCL's stream, ordered after the work already queued on this stream."""
        return self.start_ranges(arena, [arena.segments[name] for name in names], chunk_mb)

    def start_ranges(self, arena, ranges, chunk_mb: float = 256.0) -> list:
        """Like :meth:`start_arena` for explicit element ranges [(lo, hi), ...] of the arena gradient — the parts
        of a segment whose backward has already finished (graph.py launches them while the rest still runs)."""
        handles = []
        step = max(1, int(chunk_mb * 1024 * 1024) // 4)
        if self.debug_pending is not None:
            # debug flag (tests): a range may only go on the wire when no RangeReady guard still waits for one of its
            # parameters' gradients in this backward pass
            for guard, (ga, gb) in self.debug_pending:
                if guard.pending() and not guard.fired and any(a < gb and ga < b for a, b in ranges):
                    raise RuntimeError(f'start_ranges: range ({ga}, {gb}) launched while {len(guard.pending())} of its '
                                       f'parameters have not accumulated their gradient')
        for a, b in ranges:
            for lo in range(a, b, step):
                handles.append(self._reduce_chunk(arena, lo, min(b, lo + step)))
        return handles

    def finish_arena(self, arena, handles: list, optimizer=None):
        """Wait for :meth:`start_arena`; the mean is applied by the optimizer kernel (``grad_scale = 1/world``)
        when it supports it, else by one in-place division."""
        for h in handles:
            h.wait()
        if optimizer is not None and hasattr(optimizer, 'grad_scale'):
            optimizer.grad_scale = 1.0 / self.world
        else:
            arena.grad.div_(self.world)

    def reduce_arena(self, arena, optimizer=None, order=('head', 'backbone', 'encoder')):
        """Segments are reduced in the order their gradients complete (head first, encoder last).  With a
        :class:`ParameterArena` use this (after ``no_sync(True)``) instead of the hook-driven buckets also in the eager
        step: arena gradients are accumulated once per use of a parameter, so a per-parameter ready-hook would fire
        before the last use has added its part."""
        self.finish_arena(arena, self.start_arena(arena, order), optimizer)

    def no_sync(self, flag: bool = True):
        """Switch the hook-driven buckets off (gradient accumulation steps, HIP-graph capture / replay).  With a
        parameter arena they are never on."""
        self._active = (not flag) and self.arena is None

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class RangeReady:
    """Fires ``callback()`` ONCE per backward pass, when EVERY parameter of a contiguous arena range has announced its
    gradient (post-accumulate hooks; the arena's direct-accumulation kernels announce through ``ops._fire_grad_hooks``).
    A range holds several parameters (a LayerNorm's weight and bias, a Linear's weight and bias): launching its in-place
    all-reduce from ONE parameter's hook races with the accumulation of the others — autograd accumulates a layer's
    parameters in no promised order (1 failure in 8 runs of the gloo test).  A parameter that is
    announced twice in one pass (K3's backward is) counts once — so the guard is for parameters with ONE use per pass
    (a packed in_proj weight is accumulated three times: its range goes out after the pass).  ``arm()`` before each backward, ``remove()`` after."""

    def __init__(self, params, callback):
        self.params = [p for p in params if p.requires_grad]
        self.callback = callback
        self._seen, self._fired = set(), False
        self._hooks = [p.register_post_accumulate_grad_hook(self._announce) for p in self.params]

    def arm(self):
        self._seen, self._fired = set(), False
        return self

    @property
    def fired(self) -> bool:
        return self._fired

    def pending(self):
        """Parameters of the range that have not announced a gradient in this pass."""
        return [p for p in self.params if id(p) not in self._seen]

    def _announce(self, p):
        self._seen.add(id(p))
        if not self._fired and len(self._seen) == len(self.params):
            self._fired = True
            self.callback()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def reduce_scalars(values: Dict[str, torch.Tensor], group=None) -> Dict[str, float]:
    """Mean over ranks of a dict of scalars in ONE collective (replaces ≈46 ``sync_dist`` all-reduces,
    /root/reference: mask_bev/mask_bev_module.py:197-207,277-279)."""
    keys = sorted(values.keys())
    dev = next((v.device for v in values.values() if torch.is_tensor(v)), torch.device('cpu'))
    flat = torch.stack([torch.as_tensor(values[k], dtype=torch.float32, device=dev).detach().reshape(())
                        for k in keys])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, group=group)
        flat = flat / dist.get_world_size(group)
    return dict(zip(keys, flat.cpu().tolist()))


def shard_scans(items: Iterable, rank: int, world: int) -> List:
    """Whole scans are the unit of data parallelism: rank r takes items r, r+world, … (DistributedSampler
    order without shuffling); callers drop the ragged tail so that every rank has equal work
    (/root/reference: mask_bev/datasets/semantic_kitti/semantic_kitti_mask_data_module.py:124 drop_last)."""
    items = list(items)
    usable = len(items) - len(items) % world
    return items[rank:usable:world]
