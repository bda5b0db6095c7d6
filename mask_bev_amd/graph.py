"""HIP-graph training step.

A MaskBEV training step at batch 4 issues ≈ 5 000 kernel launches; measured on MI355X the step costs ≈ 50 ms of
fixed, launch-bound time plus only ≈ 6 ms of GPU work per scan (step time vs batch size: 55 / 61 / 72 / 101 ms at
B = 1 / 2 / 4 / 8).  Everything downstream of the BEV pseudo-image has static shapes, so it is captured ONCE into
a HIP graph (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) — Swin backbone, pixel decoder, masked-attention
decoder, the sync-free loss (K8/K9/K10) and their whole backward — and replayed every step:

    eager   K1 voxelise → K2 PFN → K3 scatter+LayerNorm          (pillar counts differ per batch: dynamic shapes)
    replay  backbone → head → loss → backward of head + last Swin stage        (graph 1)
    replay  backward of Swin stages 1-3 and the patch projection               (graph 2, same memory pool)
    eager   backward of K3 / K2 from the graph's d(loss)/d(pseudo-image), optimizer step

Data-parallel runs launch the gradient all-reduce in four pieces as their gradients complete — head + last stage
after graph 1 (underneath graph 2), the rest of the backbone after graph 2 and the encoder's LayerNorm affine from
its gradient hook (both underneath the encoder backward), the PFN last — so that only the tail is exposed.

With a parameter arena (``module.flatten_parameters()``, arena.py) the graph accumulates into the arena's static
gradient buffer, the optimizer step is one K11 launch that also clears the gradient and refreshes the bf16 weight
shadow, and the data-parallel all-reduce runs over contiguous arena chunks.

The graph's inputs (pseudo-image, labels, GT masks) and outputs (loss, parameter gradients, gradient of the
pseudo-image) live in static buffers.  This is the "HIP streams and graphs instead of a tracing compiler" part of
the design: no kernel is changed, only how they are launched.

Construct :class:`GraphedTrainStep` before running any eager backward of the module on the default stream (or run
such eager code inside ``torch.cuda.stream(side)``): like PyTorch's whole-network capture, the warm-up iterations
here run on a side stream because autograd state created on the legacy default stream cannot be captured.  Do not
keep the loss (or anything else that holds the autograd graph) of an earlier eager step alive across the
construction: its AccumulateGrad nodes stay bound to the stream they were created on and would accumulate outside
the capture (PyTorch emits its "AccumulateGrad node's stream does not match" warning in that case).
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

from . import ops, switches
from .ddp import RangeReady
from .mask_bev_module import MaskBevModule


def arena_reduce_plan(module: MaskBevModule, arena):
    """The four pieces of the data-parallel gradient all-reduce of a graph step, in launch order, as element ranges of
    the arena gradient: [(name, [(lo, hi), ...]), ...].  Each piece is launched as soon as its gradients are complete
    (``GraphedTrainStep.step``): (1) after graph 1, underneath graph 2; (2) after graph 2 and (3) from the gradient
    hook of K3's backward, both underneath the eager encoder backward; (4) after it — the only exposed part.
    Together they cover the arena exactly once (tests/test_ddp_gloo.py).  Replaces Lightning's DDP bucket order
    (/root/reference: train_mask_bev.py:92-96, ``strategy='ddp'``)."""
    swin = module._backbone._backbone
    last = len(swin.stages) - 1
    a, b = arena.segments['backbone']
    lo, hi = arena.range_of(swin.stages[last])
    ln = arena.range_of(module._encoder._layer_norm)
    ea, eb = arena.segments['encoder']
    nonempty = lambda rs: [r for r in rs if r[1] > r[0]]
    return [('head + last backbone stage', nonempty([arena.segments['head'], (lo, hi)])),
            ('earlier backbone stages + patch projection', nonempty([(a, lo), (hi, b)])),
            ('encoder LayerNorm affine', [ln]),
            ('pillar feature net', nonempty([(ea, ln[0]), (ln[1], eb)]))]


class GraphedTrainStep:
    """``step(batch) -> loss`` with the static-shape part of the step replayed from a HIP graph."""

    def __init__(self, module: MaskBevModule, optimizer: torch.optim.Optimizer, example_batch, warmup_iters: int = 3,
                 reducer=None, overlap_matcher: bool = True):
        self.m = module
        self.opt = optimizer
        self.reducer = reducer
        self.trace = None       # a list -> every step appends [(mark, host time, bytes), ...] (bench.py --dry-run-collectives)
        scans, labels, masks, _ = module._unpack(example_batch)
        dev = labels.device
        head = module._panoptic_head._panoptic_head
        self._overlap_prev = head.overlap_matcher
        # the matcher's side-stream fork / join is captured as a parallel branch of the graph
        head.overlap_matcher = overlap_matcher
        # the encoder writes its result (the (B, C, ny, nx) map, or the backbone's bf16 patch rows) straight into the
        # graph's static input buffer
        self._patch = module._patch_handoff()
        with torch.no_grad(), module._autocast():
            x = module._encoder(scans, patch=self._patch)
        self.x_static = torch.zeros_like(self._rows(x)).requires_grad_()
        self._x_in = (ops.PatchTokens(self.x_static, x.channels, x.patch) if isinstance(x, ops.PatchTokens)
                      else self.x_static)
        if not isinstance(x, ops.PatchTokens) and self.x_static.dtype == torch.float32 and ops.static_amax_wanted():
            # fp32 compute: K3 leaves the map's absmax record for the captured patch projection — one record at a fixed address
            static_rec = ops.static_amax_register(self.x_static)
        else:
            static_rec = None
        x = self._rows(x)
        self.labels = labels.clone()
        # dense (B, Q, ny, nx) masks or the bit-packed targets of batch.instance_targets (K14)
        # {0, 1} dense masks (the batch contract) are bit-packed by an eager launch of each step straight into the
        # graph's static words (one 420 MB read at 512², B = 4) instead of a 420 MB copy into a static dense tensor
        # plus the same packing pass inside the graph
        self._pack_dense = (not isinstance(masks, ops.PackedMasks) and masks.is_cuda and masks.dim() == 4
                            and getattr(head, 'binary_gt_masks', False) and masks.shape[2] * masks.shape[3] <= 1024 * 1024)
        if isinstance(masks, ops.PackedMasks):
            self.masks = ops.PackedMasks(masks.words.clone(), masks.h, masks.w)
        elif self._pack_dense:
            self.masks = ops.pack_binary_masks(masks.flatten(0, 1))
        else:
            self.masks = masks.clone()
        # One GPU: the AdamW update of the encoder's (C, ny, nx) LayerNorm affine happens inside K3's backward (one backward
        # per step here by construction); a data-parallel step needs the all-reduced gradient and keeps the plain form
        self._k3_fused = False
        if (reducer is None and switches.get('k3_adam') and getattr(module, '_arena', None) is not None
                and hasattr(optimizer, 'fuse_layernorm_affine')):
            ln = module._encoder._layer_norm
            self._k3_fused = optimizer.fuse_layernorm_affine(ln.weight, ln.bias)
        self._graph_params = list(module._backbone.parameters()) + list(module._panoptic_head.parameters())
        # warm-up on a side stream (allocator / library workspaces / autotuning settle before capture)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.x_static.data.copy_(x)
            if static_rec is not None:
                # the warm-up passes read the static map through its registered record, which only K3's launch of a real
                # step writes: give it this copy's maximum (an all-zero record reads as "unscaled")
                static_rec.copy_(ops.f32_absmax([self.x_static.detach().view(-1, self.x_static.shape[-1])]))
            for _ in range(warmup_iters):
                self._forward_backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.arena = getattr(module, '_arena', None)
        if self.arena is not None:
            # gradients are static arena views: accumulated in place by the graph, cleared by the optimizer kernel
            self.arena.zero_grad()
        else:
            for p in self._graph_params:
                p.grad = None
        self.x_static.grad = None
        # Two graphs sharing one memory pool: (1) forward, loss, backward of the head and of the LAST backbone stage;
        # (2) backward of the earlier stages.  Between the two replays a data-parallel step launches the all-reduce
        # of the gradients graph 1 completed (head + last stage = 38 % of the bytes), which then runs on RCCL's
        # stream underneath graph 2; on one GPU the two replays are simply back to back.
        ops.amax_new_capture()        # records / hints of an earlier captured step in this process are not this step's
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss_static = self._forward_backward_head()
        # A/B switch `switches.tn_overlap` (one GPU only): the grouped weight-gradient launch of the early stages (≈ 0.65 ms,
        # nobody's input inside the graph) is taken out of the capture and issued after the replay on a side stream,
        # beside the eager encoder backward; the captured operands stay alive in the graph's pool (held here), so their
        # addresses are the replay's.  MEASURED SLOWER — 28.3 vs 27.7 ms: the MFMA / L2-heavy launch slows the
        # latency-bound per-pillar walks it runs beside by more than it hides (as the optimizer pass did).
        self._late_tn = [] if (reducer is None and switches.get('tn_overlap')) else None
        self._tn_stream = torch.cuda.Stream(device=dev) if self._late_tn is not None else None
        self.graph_late = torch.cuda.CUDAGraph()
        ops.set_tn_sink(self._late_tn)
        try:
            with torch.cuda.graph(self.graph_late, pool=self.graph.pool()):
                self._backward_early_stages()
        finally:
            ops.set_tn_sink(None)
        torch.cuda.synchronize()
        self._ranges_head, self._ranges_late = self._arena_ranges()

    @staticmethod
    def _rows(x):
        return x.rows if isinstance(x, ops.PatchTokens) else x

    def _forward_backward_head(self) -> torch.Tensor:
        """Forward of backbone + head + loss, then the backward of the head and of the last backbone stage (the
        autograd graph is cut in front of that stage and at the three earlier feature maps)."""
        m = self.m
        dt = m._compute_dtype
        swin = m._backbone._backbone
        cut = {'stage': len(swin.stages) - 1}
        with torch.autocast('cuda', dtype=dt or torch.bfloat16, enabled=dt is not None, cache_enabled=False):
            feats = m._backbone(self._x_in, cut=cut)
            early = list(feats[:cut['stage']])
            leaves = [f.detach().requires_grad_() for f in early]
            m._panoptic_head._panoptic_head.announce_targets(self.labels, self.masks)      # static buffers, filled before the replay
            cls, masks, heights = m._panoptic_head(leaves + list(feats[cut['stage']:]))
        loss = m.loss(m.compute_loss(cls, masks, self.labels, self.masks, heights, None))
        m.scale_loss(loss).backward()          # fp16: times the device-side loss scale (a captured multiply)
        self._late_roots = early + [cut['x_in']]
        self._late_grads = [l.grad for l in leaves] + [cut['x_leaf'].grad]
        return loss.detach()

    def _backward_early_stages(self):
        torch.autograd.backward(self._late_roots, self._late_grads)
        self._late_roots = self._late_grads = None

    def _forward_backward(self) -> torch.Tensor:
        loss = self._forward_backward_head()
        self._backward_early_stages()
        return loss

    def _arena_ranges(self):
        """Arena ranges whose gradients are complete after graph 1 / only after graph 2."""
        if self.arena is None:
            return None, None
        plan = dict(arena_reduce_plan(self.m, self.arena))
        return plan['head + last backbone stage'], plan['earlier backbone stages + patch projection']

    def step(self, batch) -> torch.Tensor:
        m = self.m
        scans, labels, masks, _ = m._unpack(batch)
        with m._autocast():                                # eager: K1 → K2 → K3 (into the static buffer)
            x = self._rows(m._encoder(scans, patch=self._patch, out=self.x_static.detach()))
        # Buffer broadcast (DDP's broadcast_buffers: the PFN's BatchNorm running statistics, < 2 KB).  In training mode the
        # forward only WRITES them, so the exchange does not have to sit in front of the eager encoder as a blocking
        # collective: rank 0's statistics of THIS step go out asynchronously right behind the encoder
        # forward and are written into every rank's buffers after both graphs are queued — the replicas' buffers are equal
        # when the step ends (DDP: when the next one starts).
        buffers = self.reducer.sync_buffers_begin() if self.reducer is not None else None
        if labels.data_ptr() != self.labels.data_ptr():
            self.labels.copy_(labels)
        if isinstance(masks, ops.PackedMasks):
            if masks.words.data_ptr() != self.masks.words.data_ptr():
                self.masks.words.copy_(masks.words)
        elif self._pack_dense:
            ops.pack_binary_masks(masks.flatten(0, 1), out=self.masks)
        elif masks.data_ptr() != self.masks.data_ptr():
            self.masks.copy_(masks)
        overlap = self.reducer is not None and self.arena is not None
        tr = self.trace
        if tr is not None:
            tr.append([])
        mark = (lambda name, nbytes=0: tr[-1].append((name, time.perf_counter(), int(nbytes)))) if tr is not None \
            else (lambda *a: None)
        nb = lambda ranges: 4 * sum(b - a for a, b in ranges)
        mark('graph 1 replay')
        self.graph.replay()                                # backbone, head, loss; backward of head + last stage
        handles = None
        if overlap:                                        # their all-reduce runs underneath the second graph
            mark('all-reduce: head + last backbone stage', nb(self._ranges_head))
            handles = self.reducer.start_ranges(self.arena, self._ranges_head)
        mark('graph 2 replay')
        self.graph_late.replay()                           # backward of the earlier backbone stages
        if buffers is not None:
            self.reducer.sync_buffers_end(buffers)
        if overlap:
            mark('all-reduce: earlier backbone stages + patch projection', nb(self._ranges_late))
            handles += self.reducer.start_ranges(self.arena, self._ranges_late)
            # K3's backward is the FIRST kernel of the encoder backward and produces the two largest gradients of
            # the model (the (C, ny, nx) LayerNorm affine, 268 MB): their all-reduce starts from the gradient hook,
            # underneath the PFN backward that follows
            ln = self.m._encoder._layer_norm
            ln_range = self.arena.range_of(ln)
            early = []

            def launch_ln():            # once per backward, when BOTH affine gradients have been announced
                mark('all-reduce: encoder LayerNorm affine (from its gradient hooks)', nb([ln_range]))
                early.extend(self.reducer.start_ranges(self.arena, [ln_range]))

            # the range holds weight AND bias: it goes on the wire when every parameter in it has announced itself
            # (RangeReady), not from one parameter's hook — K3's backward writes both in one launch before it fires the
            # hooks (ops_encoder.py, _ScatterLayerNorm.backward), and this no longer depends on that
            guard = RangeReady(list(ln.parameters()), launch_ln).arm()
            try:
                mark('encoder backward (eager)')
                x.backward(self.x_static.grad)             # eager: backward of K3 / K2
            finally:
                guard.remove()
            a, b = self.arena.segments['encoder']
            rest = [(a, b)] if not early else [(a, ln_range[0]), (ln_range[1], b)]
            rest = [r for r in rest if r[1] > r[0]]
            mark('all-reduce: pillar feature net' + ('' if early else ' + LayerNorm affine'), nb(rest))
            handles += early + self.reducer.start_ranges(self.arena, rest)
            mark('wait for the collectives')
            self.reducer.finish_arena(self.arena, handles, self.opt)
            mark('optimizer')
        else:
            if self._late_tn:                              # the early stages' weight gradients, beside the encoder backward
                main = torch.cuda.current_stream()
                self._tn_stream.wait_stream(main)
                with torch.cuda.stream(self._tn_stream):
                    ops.launch_tn_group(self._late_tn)
            x.backward(self.x_static.grad)                 # eager: backward of K3 / K2
            if self._late_tn:
                torch.cuda.current_stream().wait_stream(self._tn_stream)
            if self.reducer is not None:
                self.reducer.reduce_all()
        self.opt.step()
        if self.arena is None:
            for p in self.m._encoder.parameters():         # graph-owned gradients are overwritten by the replay
                p.grad = None
        elif not getattr(self.opt, 'zero_grad_in_step', False):
            self.arena.zero_grad()
        return self.loss_static

    def close(self):
        self.m._panoptic_head._panoptic_head.overlap_matcher = self._overlap_prev
        if self._k3_fused:
            self.opt.fuse_layernorm_affine(None)
            self._k3_fused = False
