"""Mask2Former-style panoptic head of MaskBEV on gfx950: pixel decoder, masked-attention decoder, loss.

Interface and checkpoint keys of ``Mask2FormerHead``
(/root/reference: mask_bev/models/networks/mask2former_head/mask2former_head.py:20-562) and of the mmdet
3.0.0 ``MSDeformAttnPixelDecoder`` / ``Mask2FormerTransformerDecoder`` it builds from the config at
mask_bev/models/head/mask_bev_panoptic_head.py:98-215.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, switches
from .layers import ConvGN, FFN, conv1x1, LayerNorm, Linear, MultiScaleDeformableAttention, MultiheadAttention, sine_positional_encoding


# --------------------------------------------------------------------------------------
# pixel decoder
# --------------------------------------------------------------------------------------
class _DeformEncoderLayer(nn.Module):
    def __init__(self, embed_dims, num_heads, num_levels, num_points, ffn_channels):
        super().__init__()
        self.self_attn = MultiScaleDeformableAttention(embed_dims, num_heads, num_levels, num_points)
        self.ffn = FFN(embed_dims, ffn_channels, act='relu')
        self.norms = nn.ModuleList([LayerNorm(embed_dims), LayerNorm(embed_dims)])

    def forward(self, q, pos, ref, shapes, shapes_t, level_start, q_branch=None, fanout=False, pos_share=None,
                pos_share_index=0, wcat=None):
        """post-LN: LN(q + branch(q)), the add fused into the LayerNorm kernel (K12).  A LayerNorm output has two
        consumers — the next residual add and the next branch: it is handed on as a PAIR of tensors over one buffer
        (``q`` for the residual, ``q_branch`` for the branch) so that K12's backward receives the two gradients
        separately and adds them on load.  ``fanout``: return such a pair (every layer but the last)."""
        c = q.shape[-1]
        ob, fb = self.self_attn.output_proj.bias, self.ffn.layers[1].bias
        d1, d2 = ops.bias_grad_deferrable(ob, c), ops.bias_grad_deferrable(fb, c)
        q_branch = q if q_branch is None else q_branch
        q, qb = self.norms[0](q, self.self_attn(q_branch, pos, ref, shapes, shapes_t, level_start, add_identity=False,
                                                defer_out_bias=d1, pos_share=pos_share, pos_share_index=pos_share_index,
                                                wcat=wcat),
                              residual_bias=ob if d1 else None, fanout=True, branch_gemm=True)
        return self.norms[1](q, self.ffn(qb, add_identity=False, defer_out_bias=d2), residual_bias=fb if d2 else None,
                             fanout=fanout)


class _DeformEncoder(nn.Module):
    def __init__(self, num_layers, **kw):
        super().__init__()
        self.layers = nn.ModuleList([_DeformEncoderLayer(**kw) for _ in range(num_layers)])


class MSDeformAttnPixelDecoder(nn.Module):
    """6-layer multi-scale deformable-attention encoder over the 3 coarsest maps + FPN step to stride 4.
    Keys: input_convs.i.{conv,gn}, encoder.layers.l.{self_attn,ffn,norms}, level_encoding,
    lateral_convs.i, output_convs.i, mask_feature."""

    def __init__(self, in_channels: Sequence[int], feat_channels: int, out_channels: int, strides=(4, 8, 16, 32),
                 num_outs: int = 3, num_layers: int = 6, num_heads: int = 8, num_levels: int = 3, num_points: int = 4,
                 ffn_channels: int = 1024):
        super().__init__()
        self.strides = list(strides)
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = num_levels
        self.feat_channels = feat_channels
        n = self.num_input_levels
        self.input_convs = nn.ModuleList([
            ConvGN(in_channels[i], feat_channels, 1, bias=True, relu=False) for i in range(n - 1, n - num_levels - 1, -1)])
        self.encoder = _DeformEncoder(num_layers, embed_dims=feat_channels, num_heads=num_heads, num_levels=num_levels,
                                      num_points=num_points, ffn_channels=ffn_channels)
        self.level_encoding = nn.Embedding(num_levels, feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        for i in range(n - num_levels - 1, -1, -1):
            self.lateral_convs.append(ConvGN(in_channels[i], feat_channels, 1, bias=False, relu=False))
            self.output_convs.append(ConvGN(feat_channels, feat_channels, 3, bias=False, relu=True))
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, kernel_size=1)
        self.num_outs = num_outs
        self._geom_cache: Dict[Tuple, Tuple] = {}
        self._tail_streams: Dict = {}

    def init_weights(self):
        for m in self.input_convs:
            nn.init.xavier_uniform_(m.conv.weight, gain=1)
            nn.init.zeros_(m.conv.bias)
        for m in list(self.lateral_convs) + list(self.output_convs):
            nn.init.kaiming_uniform_(m.conv.weight, a=1)
        nn.init.kaiming_uniform_(self.mask_feature.weight, a=1)
        nn.init.zeros_(self.mask_feature.bias)
        nn.init.normal_(self.level_encoding.weight, mean=0, std=1)
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        for layer in self.encoder.layers:
            layer.self_attn.init_weights()

    def _geometry(self, shapes: Tuple[Tuple[int, int], ...], device):
        key = (shapes, str(device))
        g = self._geom_cache.get(key)
        if g is None:
            refs, pos = [], []
            for (h, w) in shapes:
                xs = (torch.arange(w, dtype=torch.float32, device=device) + 0.5) / w
                ys = (torch.arange(h, dtype=torch.float32, device=device) + 0.5) / h
                refs.append(torch.stack([xs.repeat(h), ys.view(-1, 1).repeat(1, w).view(-1)], -1))
                pos.append(sine_positional_encoding(h, w, self.feat_channels // 2, device).flatten(2).transpose(1, 2))
            shapes_t = torch.tensor(shapes, dtype=torch.int64, device=device)
            starts = [0]
            for (h, w) in shapes[:-1]:
                starts.append(starts[-1] + h * w)
            g = (torch.cat(refs, 0), pos, shapes_t, torch.tensor(starts, dtype=torch.int64, device=device))
            self._geom_cache[key] = g
        return g

    def forward(self, feats: List[torch.Tensor]):
        bs = feats[0].shape[0]
        n, nl = self.num_input_levels, self.num_encoder_levels
        shapes = tuple((int(feats[n - i - 1].shape[2]), int(feats[n - i - 1].shape[3])) for i in range(nl))
        ref, pos, shapes_t, level_start = self._geometry(shapes, feats[0].device)
        tokens = []
        for i in range(nl):
            proj = self.input_convs[i](feats[n - i - 1])
            tokens.append(proj.flatten(2).transpose(1, 2))
        q = torch.cat(tokens, 1)
        if q.is_cuda:
            qpos = ops.level_positions(self.level_encoding.weight, pos[:nl])
        else:
            qpos = torch.cat([pos[i] + self.level_encoding.weight[i].view(1, 1, -1) for i in range(nl)], 1)
        q_branch = None
        nlay = len(self.encoder.layers)
        # one d(pos) product for the chain of layers instead of one per layer (ops.PosGradShare)
        share = (ops.PosGradShare(nlay) if (q.is_cuda and torch.is_grad_enabled() and qpos.requires_grad and nlay > 1
                                            and switches.get('pos_share')) else None)
        # the layers' stacked projection weights [Wv; Wo; Wa] in the compute dtype: 18 pieces, one launch
        wcats = None
        if q.is_cuda and q.dtype == torch.float32 and switches.get('msda_fused'):
            cdt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else q.dtype
            wcats = ops.msda_weight_stacks([l.self_attn for l in self.encoder.layers], cdt)
        for li, layer in enumerate(self.encoder.layers):
            kw = dict(pos_share=share, pos_share_index=li, wcat=None if wcats is None else wcats[li])
            if li + 1 < nlay:
                q, q_branch = layer(q, qpos, ref, shapes, shapes_t, level_start, q_branch=q_branch, fanout=True, **kw)
            else:
                q = layer(q, qpos, ref, shapes, shapes_t, level_start, q_branch=q_branch, **kw)
        outs = [t.transpose(1, 2).reshape(bs, -1, h, w) for t, (h, w) in
                zip(torch.split(q, [h * w for h, w in shapes], dim=1), shapes)]
        # The FPN tail (lateral / output convolutions, mask-feature projection) runs on a second stream and is joined
        # at once.  Forward gains nothing from it; BACKWARD does: autograd replays a node on the stream its forward ran
        # on, so the tail's backward — which needs only d(mask_features), the first gradient the head produces — becomes
        # a parallel branch underneath the decoder layers' backward (400-row kernels that leave the chip idle).
        side = self._tail_stream(feats[0].device)
        if side is not None:
            main = torch.cuda.current_stream(feats[0].device)
            side.wait_stream(main)
            ctx = torch.cuda.stream(side)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            tail = list(outs)
            for j, i in enumerate(range(n - nl - 1, -1, -1)):
                if side is not None and feats[i].dtype in ops._LO_DTYPES:
                    # a 16-bit map would be saved as it is by the convolution node below, whose backward runs on this stream:
                    # give that node a tensor of this stream's pool (see swin.CustomSwinTransformer.forward)
                    feats = list(feats)
                    feats[i] = feats[i].clone()
                # K18: GroupNorm + up-sampled add of the coarser level in one pass, stored for the 3 x 3 convolution
                y = self.lateral_convs[j](feats[i], add_upsampled=tail[-1], conv_input=True)
                # outputs beyond num_outs only feed the next FPN step / the mask-feature projection (convolutions)
                tail.append(self.output_convs[j](y, conv_input=len(tail) >= self.num_outs))
            mask_feature = (conv1x1(self.mask_feature, tail[-1]) if tail[-1].is_cuda
                            else self.mask_feature(tail[-1]))
        if side is not None:
            main.wait_stream(side)
        return mask_feature, tail[:self.num_outs]

    def _tail_stream(self, device):
        if device.type != 'cuda' or not torch.is_grad_enabled() or not switches.get('tail_stream'):
            return None
        st = self._tail_streams.get(device)
        if st is None:
            st = self._tail_streams[device] = torch.cuda.Stream(device=device)
        return st


# --------------------------------------------------------------------------------------
# transformer decoder
# --------------------------------------------------------------------------------------
class _DecoderLayer(nn.Module):
    """masked cross-attention → LN → self-attention → LN → FFN → LN (post-norm)."""

    def __init__(self, embed_dims, num_heads, ffn_channels):
        super().__init__()
        self.self_attn = MultiheadAttention(embed_dims, num_heads)
        self.cross_attn = MultiheadAttention(embed_dims, num_heads)
        self.ffn = FFN(embed_dims, ffn_channels, act='relu')
        self.norms = nn.ModuleList([LayerNorm(embed_dims) for _ in range(3)])

    def forward(self, query, memory, query_pos, memory_pos, blocked, memory_key=None, shared_kv=None,
                query_branch=None, fanout=False):
        """``memory_key`` = memory + memory_pos when the caller has it (it is the same for every layer that reads
        this level); ``shared_kv``: this layer's slot of the level's batched key / value projection.
        ``query_branch``: the previous layer's output as handed to this layer's branch (``query`` then only feeds the
        residual add); ``fanout``: return the output as such a pair ``(residual copy, branch copy)``."""
        qb0 = query if query_branch is None else query_branch
        if shared_kv is not None:
            r = self.cross_attn(qb0, None, None, query_pos, None, blocked, add_identity=False, shared_kv=shared_kv)
        elif memory_key is None:
            r = self.cross_attn(qb0, memory, memory, query_pos, memory_pos, blocked, add_identity=False)
        else:
            r = self.cross_attn(qb0, memory_key, memory, query_pos, None, blocked, add_identity=False)
        # post-LN, residual add inside the LayerNorm kernel (K12); each output leaves as a pair (residual / branch
        # consumer) whose gradients K12's backward adds on load
        q, qb = self.norms[0](query, r, fanout=True)
        q, qb = self.norms[1](q, self.self_attn(qb, qb, qb, query_pos, query_pos, None, add_identity=False), fanout=True)
        return self.norms[2](q, self.ffn(qb, add_identity=False), fanout=fanout)


class Mask2FormerTransformerDecoder(nn.Module):
    def __init__(self, num_layers, embed_dims, num_heads, ffn_channels):
        super().__init__()
        self.embed_dims = embed_dims
        self.layers = nn.ModuleList([_DecoderLayer(embed_dims, num_heads, ffn_channels) for _ in range(num_layers)])
        self.post_norm = LayerNorm(embed_dims)


# --------------------------------------------------------------------------------------
# head
# --------------------------------------------------------------------------------------
class LossDict(dict):
    """The reference's loss dictionary (40 scalars, same keys) plus ``total``: the sum of all its ``*loss*`` entries
    computed vectorised.  ``MaskBevModule.loss`` (mask_bev_module.py:193-195 in the reference) returns ``total``
    when it is handed this object, and falls back to summing the entries otherwise."""
    total: Optional[torch.Tensor] = None


class PointSource:
    """Uniform sampling points for the loss.  ``seed=None`` → device RNG (training); an int → a CPU
    generator drawn in the reference's order (mask2former_head.py:191, then mmdet's
    ``get_uncertain_point_coords_with_randomness``) so tests can share the points with the oracle."""

    def __init__(self, device, seed: Optional[int] = None):
        self.device = device
        self.gen = None if seed is None else torch.Generator().manual_seed(seed)

    def rand(self, *shape) -> torch.Tensor:
        if self.gen is None:
            return torch.rand(*shape, device=self.device)
        return torch.rand(*shape, generator=self.gen).to(self.device)


class _DeferredHeads(torch.autograd.Function):
    """The prediction heads of all D = layers + 1 decoder outputs, backward in ONE batched pass.

    The forward values (class scores, mask logits in the stacked buffer, the next layer's attention mask) are needed
    layer by layer and are computed there without a graph.  Their backward does not have that dependency: the loss
    gradient of all D outputs is available before the decoder's backward starts.  Run per layer it costs ≈ 16 launches
    on B*Q = 400 rows each, ten times (two batched GEMMs against the mask features, nine accumulations of the 16 MB
    feature gradient, the three-layer MLP, the class head, the LayerNorm); here it is one cast + permute of the
    stacked logit gradient, two GEMMs with D*Q rows per sample, and one re-evaluation + backward of the small head
    on D*B*Q rows (the re-evaluation uses the precision the per-layer forward used).  mask2former_head.py:428-472."""

    @staticmethod
    def forward(ctx, head, cls_all, mask_feature, *query_feats):
        ctx.head = head
        ctx.nd = len(query_feats)
        ctx.save_for_backward(mask_feature, *query_feats)
        # the D mask predictions leave as the D slices of the stacked buffer the per-layer passes filled (views: the
        # loss re-assembles them without a copy, ops.stack_slices); the buffer itself is not an autograd input
        stack = head._mask_stack
        return (cls_all.view_as(cls_all),) + tuple(stack[i] for i in range(ctx.nd))

    @staticmethod
    def backward(ctx, d_cls, *d_masks):
        head, nd = ctx.head, ctx.nd
        mask_feature, *query_feats = ctx.saved_tensors
        b, q, c = query_feats[0].shape
        hh, ww = mask_feature.shape[-2:]
        hw = hh * ww
        dt = mask_feature.dtype
        # K8's backward may have left the gradient sample-major in the operand type already (ops.StackGradSink): then what
        # arrives here are views of the sink's token; anything else means other gradients were added — ordinary path, plus
        # the sink's share
        sink = getattr(head, '_stack_sink', None)
        sunk = None
        if sink is not None and sink.grad is not None:
            sunk, sink.grad = sink.grad, None
            if sunk.dtype != dt or tuple(sunk.shape) != (b, nd, q, hw):
                raise ops.MaskBevHipError('deferred heads: the stacked gradient sink holds another layout')
        if sunk is not None and all(sink.is_token(g) for g in d_masks):
            return _DeferredHeads._finish(ctx, d_cls, sunk.view(b, nd * q, hw))
        # stacked logit gradient (D, B, Q, H, W): the slices handed back by ops.stack_slices are views of one buffer
        g0 = d_masks[0]
        full = None
        if g0 is not None and all(g is not None and g.is_contiguous() and g.dtype == g0.dtype for g in d_masks):
            step = g0.numel() * g0.element_size()
            if all(g.data_ptr() == g0.data_ptr() + i * step for i, g in enumerate(d_masks)):
                full = torch.as_strided(g0, (nd, b, q, hw), (b * q * hw, q * hw, hw, 1))
        if full is None:
            zero = None
            parts = []
            for g in d_masks:
                if g is None:
                    zero = zero if zero is not None else torch.zeros((b, q, hh, ww), dtype=torch.float32,
                                                                     device=mask_feature.device)
                    g = zero
                parts.append(g.float())
            full = torch.stack(parts, 0).view(nd, b, q, hw)
        dl = torch.empty((b, nd, q, hw), dtype=dt, device=full.device)       # per sample: D*Q rows, one cast+permute
        dl.copy_(full.permute(1, 0, 2, 3))
        if sunk is not None:
            dl += sunk
        return _DeferredHeads._finish(ctx, d_cls, dl.view(b, nd * q, hw))

    @staticmethod
    def _finish(ctx, d_cls, dl):
        """dl (B, D*Q, H*W): the stacked logit gradient, sample-major, in the mask features' dtype."""
        head, nd = ctx.head, ctx.nd
        mask_feature, *query_feats = ctx.saved_tensors
        b, q, c = query_feats[0].shape
        hh, ww = mask_feature.shape[-2:]
        hw = hh * ww
        dt = mask_feature.dtype
        # re-evaluate the small head on all D*B*Q rows with a graph, in the per-layer forward's precision
        # (the row-chain decoder evaluates its heads with 16-bit operands whatever the row count: so does this)
        small = b * q * c <= ops._SMALL_F32_ROWS * c and not getattr(head, '_deferred_lowp_heads', False)
        adt = head._deferred_autocast
        with torch.enable_grad(), torch.autocast('cuda', dtype=adt or torch.bfloat16,
                                                 enabled=(adt is not None) and not small, cache_enabled=False):
            q_all = torch.stack([t.detach() for t in query_feats], 0).requires_grad_()      # (D, B, Q, C)
            y = head.transformer_decoder.post_norm(q_all)
            cls_re = head.cls_embed(y)
            e_re = head.mask_embed(y)                                                      # (D, B, Q, C)
        e_b = torch.empty((b, nd, q, c), dtype=dt, device=dl.device)
        e_b.copy_(e_re.detach().permute(1, 0, 2, 3))
        e_b = e_b.view(b, nd * q, c)
        ff = mask_feature.reshape(b, c, hw)
        d_e, d_f = ops.mask_logits_backward(dl, e_b, ff)        # (B, D*Q, C), (B, C, HW): K17 for 16-bit operands
        d_f = d_f.view(b, c, hh, ww)
        d_e = d_e.view(b, nd, q, c).permute(1, 0, 2, 3).to(e_re.dtype)
        roots, grads = [e_re], [d_e]
        if d_cls is not None:
            roots.append(cls_re)
            grads.append(d_cls.to(cls_re.dtype))
        torch.autograd.backward(roots, grads)                # parameter gradients accumulate where they always do
        gq = q_all.grad
        # Row-chain decoder: the gradient of decoder output i (i = 1 .. D - 2) also reaches its producer from the next layer;
        # handed over through the layer context, the producer's backward program adds it while it loads that gradient — not
        # autograd with a launch per layer.  Output 0 is the query parameter, output D - 1 has no other consumer.
        stash = getattr(head, '_gq_stash', None)
        if stash is not None and nd > 2 and switches.get('gq_stash'):
            box = stash.setdefault('gq', {})
            for i in range(1, nd - 1):
                box[i] = gq[i]
            return (None, None, d_f, gq[0]) + (None,) * (nd - 2) + (gq[nd - 1],)
        return (None, None, d_f) + tuple(gq[i] for i in range(nd))


class Mask2FormerHead(nn.Module):
    def __init__(self, in_channels, feat_channels, out_channels, num_things_classes=80, num_stuff_classes=53,
                 num_queries=100, num_transformer_feat_level=3, pixel_decoder=None, enforce_decoder_input_project=False,
                 transformer_decoder=None, positional_encoding=None, loss_cls=None, loss_mask=None, loss_dice=None,
                 train_cfg=None, test_cfg=None, init_cfg=None, predict_height: bool = False, loss_height=None, **kwargs):
        super().__init__()
        if predict_height:
            raise NotImplementedError('predict_heights is a dead path in the reference '
                                      '(mask2former_head.py:231-244,384; SURVEY.md Appendix B)')
        pixel_decoder = pixel_decoder or {}
        transformer_decoder = transformer_decoder or {}
        train_cfg = train_cfg or {}
        loss_cls = loss_cls or {}
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.num_queries = num_queries
        self.num_transformer_feat_level = num_transformer_feat_level
        enc = pixel_decoder.get('encoder', {})
        lcfg = enc.get('layer_cfg', {})
        sa = lcfg.get('self_attn_cfg', {})
        self.pixel_decoder = MSDeformAttnPixelDecoder(
            in_channels, feat_channels, out_channels, strides=kwargs.get('strides', (4, 8, 16, 32)),
            num_outs=pixel_decoder.get('num_outs', 3), num_layers=enc.get('num_layers', 6),
            num_heads=sa.get('num_heads', 8), num_levels=sa.get('num_levels', 3), num_points=sa.get('num_points', 4),
            ffn_channels=lcfg.get('ffn_cfg', {}).get('feedforward_channels', 1024))
        assert self.pixel_decoder.num_encoder_levels == num_transformer_feat_level
        dl = transformer_decoder.get('layer_cfg', {})
        self.num_heads = dl.get('cross_attn_cfg', {}).get('num_heads', 8)
        self.num_transformer_decoder_layers = transformer_decoder.get('num_layers', 9)
        self.transformer_decoder = Mask2FormerTransformerDecoder(
            self.num_transformer_decoder_layers, feat_channels, self.num_heads,
            dl.get('ffn_cfg', {}).get('feedforward_channels', 2048))
        self.decoder_embed_dims = feat_channels
        self.decoder_input_projs = nn.ModuleList([nn.Identity() for _ in range(num_transformer_feat_level)])
        if enforce_decoder_input_project:
            raise NotImplementedError('MaskBEV never enables enforce_decoder_input_project '
                                      '(mask_bev_panoptic_head.py:147)')
        self.query_embed = nn.Embedding(num_queries, feat_channels)
        self.query_feat = nn.Embedding(num_queries, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.height_embed = None
        self.cls_embed = Linear(feat_channels, self.num_classes + 1)
        self.mask_embed = nn.Sequential(Linear(feat_channels, feat_channels), nn.ReLU(),
                                        Linear(feat_channels, feat_channels), nn.ReLU(),
                                        Linear(feat_channels, out_channels))
        self.num_points = train_cfg.get('num_points', 12544)
        self.oversample_ratio = train_cfg.get('oversample_ratio', 3.0)
        self.importance_sample_ratio = train_cfg.get('importance_sample_ratio', 0.75)
        self.class_weight = list(loss_cls.get('class_weight', [1.0] * self.num_classes + [0.1]))
        self.loss_cls_weight = float(loss_cls.get('loss_weight', 2.0))
        self.loss_mask_weight = float((loss_mask or {}).get('loss_weight', 5.0))
        self.loss_dice_weight = float((loss_dice or {}).get('loss_weight', 5.0))
        self.point_seed: Optional[int] = None          # tests set this to share points with the oracle
        self._iota_cache: Dict = {}
        self._side_streams: Dict = {}
        self.overlap_matcher = switches.get('overlap_matcher')
        # GT masks are float32 {0, 1} by the reference's batch contract (semantic_kitti_transforms.py:77-81);
        # set False to sample arbitrary-valued maps through the generic f32 path
        self.binary_gt_masks = True
        self.world_size_fn = None                      # set by the DDP wrapper: () -> (world, all_reduce_fn)

    def init_weights(self):
        self.pixel_decoder.init_weights()
        for p in self.transformer_decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    # ------------------------------------------------------------------ forward
    def _forward_head(self, decoder_out, mask_feature, target_size, out=None):
        """mask2former_head.py:428-472 → cls (B,Q,K+1), mask logits (B,Q,H,W), blocked (B,1,Q,h*w) bool.
        The boolean mask is kept once per query (broadcast over heads) instead of being repeated 8x, and
        the 'row fully blocked → unblock the row' rule of :538-539 is applied here."""
        x = self.transformer_decoder.post_norm(decoder_out)
        cls_pred = self.cls_embed(x)
        mask_embed = self.mask_embed(x)
        mask_pred, blocked = ops.mask_logits(mask_embed, mask_feature, target_size, out)
        return cls_pred, mask_pred, blocked

    @staticmethod
    def _target_key(t):
        if isinstance(t, ops.PackedMasks):
            t = t.words
        if not torch.is_tensor(t) or not t.is_cuda:
            return None
        return (t.data_ptr(), tuple(t.shape), t.dtype)

    def announce_targets(self, gt_labels, gt_masks) -> None:
        """The caller's promise that these device tensors — the ones it will hand to `loss()` — are complete in the
        current stream's order NOW, before `forward()` starts.  Only then may the loss's target preparation fork from
        the start of the head's forward (see loss()): targets that are stacked from a list inside `loss()`, moved to the
        device after `forward()`, or produced by anything launched later would be read by the side stream before their
        producers ran.  `MaskBevModule._step` and the HIP-graph step announce; a bare `forward()` +
        `compute_loss()` pair does not, and then the preparation stays on the caller's stream."""
        keys = (self._target_key(gt_labels), self._target_key(gt_masks))
        self._announced_targets = keys if None not in keys else None

    def forward(self, x: List[torch.Tensor], batch_data_samples=None):
        bs = x[0].shape[0]
        # fork point of the loss's target preparation (see loss()): everything it reads — the batch's labels and masks —
        # exists before the head starts, so it may run underneath the pixel decoder and the decoder layers
        self._early_event = None
        announced, self._announced_targets = getattr(self, '_announced_targets', None), None
        if (announced is not None and x[0].is_cuda and self.training and torch.is_grad_enabled()
                and switches.get('early_targets')):
            self._early_event = (torch.cuda.Event(), announced)
            self._early_event[0].record(torch.cuda.current_stream(x[0].device))
        mask_features, memories = self.pixel_decoder(x)
        dec_in, dec_pos, dec_key = [], [], []
        # key = memory + pos, shared by the 3 layers of a level; under autocast one cast per level instead of one per
        # (layer, projection): the K / V GEMMs read these copies
        adt = (torch.get_autocast_dtype('cuda') if (torch.is_autocast_enabled('cuda') and memories[0].is_cuda)
               else memories[0].dtype)
        for i in range(self.num_transformer_feat_level):
            m = memories[i]
            h, w = m.shape[-2:]
            dec_pos.append(sine_positional_encoding(h, w, self.decoder_embed_dims // 2, m.device)
                           .flatten(2).transpose(1, 2))
            a, k = ops.level_inputs(m, self.level_embed.weight, i, dec_pos[-1], adt)
            dec_in.append(a)
            dec_key.append(k)
        query_feat = self.query_feat.weight.unsqueeze(0).expand(bs, -1, -1)
        query_embed = self.query_embed.weight.unsqueeze(0).expand(bs, -1, -1)
        cls_list, mask_list = [], []
        # the 10 mask predictions are written as f32 straight into one stacked buffer: the loss reads it as is
        nd = len(self.transformer_decoder.layers) + 1
        stack = torch.empty((nd, bs, self.num_queries) + tuple(mask_features.shape[-2:]), dtype=torch.float32,
                            device=mask_features.device) if mask_features.is_cuda else None
        self._mask_stack = stack
        self._stack_sink = None
        self._gq_stash = None
        # training on the GPU: the heads run layer by layer WITHOUT a graph and get one batched backward (_DeferredHeads)
        deferred = (stack is not None and self.training and torch.is_grad_enabled()
                    and switches.get('deferred_heads'))
        if deferred and switches.get('stack_grad_sink'):
            # K8's backward writes the stacked logit gradient where and how _DeferredHeads reads it (ops.StackGradSink)
            self._stack_sink = ops.StackGradSink(nd, bs, self.num_queries, mask_features.dtype, mask_features.device)
        feats_q = [query_feat]

        def heads(qf, size, slot):
            if not deferred:
                return self._forward_head(qf, mask_features, size, slot)
            with torch.no_grad():
                return self._forward_head(qf, mask_features, size, slot)

        cls_pred, mask_pred, blocked = heads(query_feat, memories[0].shape[-2:], None if stack is None else stack[0])
        cls_list.append(cls_pred)
        mask_list.append(mask_pred)
        nl = self.num_transformer_feat_level
        # the three layers that read a memory level share ONE key and ONE value GEMM (and one data-gradient GEMM each
        # in backward): ops.SharedKV
        layers = self.transformer_decoder.layers
        shared = [None] * len(layers)
        if (ops.shared_kv_supported(self.num_queries, mask_features.device)
                and switches.get('shared_kv')):
            for lvl in range(nl):
                idx = [i for i in range(len(layers)) if i % nl == lvl]
                holder, token = ops.shared_kv_project(
                    dec_key[lvl], dec_in[lvl],
                    [(layers[i].cross_attn.attn.in_proj_weight, layers[i].cross_attn.attn.in_proj_bias) for i in idx])
                for slot, i in enumerate(idx):
                    shared[i] = (holder, token, slot)
        if self._fused_decoder_ok(mask_features, shared, deferred):
            return self._forward_decoder_fused(query_feat, query_embed, mask_features, memories, shared, stack, deferred,
                                               cls_list, mask_list, blocked, feats_q)
        query_branch = None
        for i, layer in enumerate(layers):
            lvl = i % nl
            if i + 1 < len(layers):       # the output feeds the next residual add (query_feat) and, as its twin over the
                #                           same buffer, the next branch and the prediction head (K12's fan-out)
                query_feat, query_branch = layer(query_feat, dec_in[lvl], query_embed, dec_pos[lvl], blocked,
                                                 dec_key[lvl], shared[i], query_branch=query_branch, fanout=True)
            else:
                query_feat = layer(query_feat, dec_in[lvl], query_embed, dec_pos[lvl], blocked, dec_key[lvl], shared[i],
                                   query_branch=query_branch)
                query_branch = query_feat
            feats_q.append(query_branch)
            cls_pred, mask_pred, blocked = heads(query_branch, memories[(i + 1) % nl].shape[-2:],
                                                 None if stack is None else stack[i + 1])
            cls_list.append(cls_pred)
            mask_list.append(mask_pred)
        if deferred:
            self._deferred_lowp_heads = False
            self._deferred_autocast = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else None
            outs = _DeferredHeads.apply(self, torch.stack(cls_list, 0), mask_features, *feats_q)
            cls_list, mask_list = list(outs[0].unbind(0)), list(outs[1:])
        return cls_list, mask_list, [None for _ in cls_list]

    # ------------------------------------------------------------------ fused query side (K19)
    def _fused_decoder_ok(self, mask_features, shared, deferred) -> bool:
        """The row-chain form of the decoder layers (decoder_fused.py): GPU, shared key / value projections for every
        layer (<= 128 queries), embed / head widths the 256-column LDS slots hold, and either no autograd graph or the
        deferred (batched-backward) prediction heads — the chain's own heads carry no gradient."""
        from . import decoder_fused as DF
        e = self.decoder_embed_dims
        layers = self.transformer_decoder.layers
        f = layers[0].ffn.layers[0][0].out_features
        oc = self.mask_embed[4].out_features
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else torch.float32
        # What a program can hold (mbv_rowchain_max_stages() = 64 stages, 256-column products): the class head is one
        # GEMM stage (<= 256 classes); the MLP is ONE stage in its fused form (16-bit weights, f % 256 == 0, at most 64
        # slices) and f / 256 chunks of 3 forward / 5 backward stages otherwise — beyond 2048 hidden units that form
        # does not fit beside the layer's other ~20 stages, and such a head takes the per-op path.
        fused_ffn = dt != torch.float32 and f % 256 == 0 and e % 32 == 0 and switches.get('rc_ffn')
        if self.cls_embed.out_features > 256 or (f > 16384 if fused_ffn else f > 2048):
            return False
        return (DF.enabled(dt) and mask_features.is_cuda and all(s is not None for s in shared)
                and (deferred or not torch.is_grad_enabled())
                and e % 32 == 0 and e <= 256 and f % 32 == 0 and oc % 4 == 0 and oc <= 256
                and self.mask_embed[0].out_features <= 256 and self.mask_embed[2].out_features <= 256
                and self.mask_embed[0].out_features % 32 == 0 and self.mask_embed[2].out_features % 32 == 0
                and dt in ops._ACT_DTYPES and mask_features.dtype in ops._ACT_DTYPES)

    def _forward_decoder_fused(self, query_feat, query_embed, mask_features, memories, shared, stack, deferred,
                               cls_list, mask_list, blocked, feats_q):
        """The decoder loop of mask2former_head.py:535-560 with the query side of every layer as two row-chain launches
        (+ the attention kernels): decoder_fused._DecA / _DecB."""
        from . import decoder_fused as DF
        layers = self.transformer_decoder.layers
        nl = self.num_transformer_feat_level
        bs, q, e = query_feat.shape
        dev = mask_features.device
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else torch.float32
        need_grad = torch.is_grad_enabled()
        tw = getattr(self, '_dec_weights', None)
        if tw is None:
            tw = self._dec_weights = DF.WeightCopies()
        # the chains' GEMM operands (fragment-major 16-bit copies; with a graph also the transposed ones for the data
        # gradients): one grouped launch per step
        ents = [(self.cls_embed.weight, None, False)] + [(self.mask_embed[j].weight, None, False) for j in (0, 2, 4)]
        # (the MLP's second weight — forward W2, backward W1^T — in k-major block order when its stage is split over CUs)
        km = DF.ffn_split(layers[0].ffn.layers[0][0].out_features, e, dt) > 1
        for i, layer in enumerate(layers):
            ca, sa = layer.cross_attn.attn, layer.self_attn.attn
            fc1, fc2 = layer.ffn.layers[0][0], layer.ffn.layers[1]
            ents += [(ca.out_proj.weight, None, False), (sa.in_proj_weight, None, False), (sa.out_proj.weight, None, False),
                     (fc1.weight, None, False), (fc2.weight, None, False, km)]
            if i > 0:
                ents.append((ca.in_proj_weight, None, False))
            if need_grad:
                ents += [(ca.out_proj.weight, None, True), (sa.in_proj_weight, (0, e), True),
                         (sa.in_proj_weight, (e, 2 * e), True), (sa.in_proj_weight, (2 * e, 3 * e), True),
                         (sa.out_proj.weight, None, True), (fc1.weight, None, True, km), (fc2.weight, None, True)]
                if i > 0:
                    ents.append((ca.in_proj_weight, (0, e), True))
        tw.refresh(ents, dt)
        holder = {}
        self._gq_stash = holder if deferred else None
        qpos = DF.QueryPositions.apply(self.query_embed.weight, holder)
        f = layers[0].ffn.layers[0][0].out_features
        lc = DF.LayerCtx(bs, q, e, self.num_heads, f, layers[0].norms[0].eps, dt, tw, holder, qpos)
        post = self.transformer_decoder.post_norm
        mlp = [(self.mask_embed[j].weight, self.mask_embed[j].bias) for j in (0, 2, 4)]
        with torch.autocast('cuda', enabled=False):
            # layer 0's masked cross-attention (its query projection has no producer chain)
            l0 = layers[0].cross_attn.attn
            q_in = query_feat + qpos.unsqueeze(0)
            qc = ops.linear(q_in, l0.in_proj_weight, l0.in_proj_bias, rows=(0, e))
            h0, tok0, slot0 = shared[0]
            o1 = ops.attention_shared_kv(qc, tok0, blocked, self.num_heads, h0, slot0)
            x = query_feat
            for i, layer in enumerate(layers):
                ca, sa = layer.cross_attn.attn, layer.self_attn.attn
                x1, o2 = DF._DecA.apply(lc, x, o1, ca.out_proj.weight, ca.out_proj.bias, layer.norms[0].weight,
                                        layer.norms[0].bias, sa.in_proj_weight, sa.in_proj_bias)
                head = DF.HeadSpec(post.weight, post.bias, self.cls_embed.weight, self.cls_embed.bias, mlp, mask_features,
                                   None if stack is None else stack[i + 1], index=i + 1)
                nxt, token, nw, nb = None, None, None, None
                if i + 1 < len(layers):
                    hn, token, slotn = shared[i + 1]
                    nca = layers[i + 1].cross_attn.attn
                    nw, nb = nca.in_proj_weight, nca.in_proj_bias
                    nxt = DF.NextCross(nw, nb, hn, slotn, memories[(i + 1) % nl].shape[-2:])
                fc1, fc2 = layer.ffn.layers[0][0], layer.ffn.layers[1]
                outs = DF._DecB.apply(lc, head, nxt, x1, o2, token, sa.out_proj.weight, sa.out_proj.bias,
                                      layer.norms[1].weight, layer.norms[1].bias, fc1.weight, fc1.bias, fc2.weight,
                                      fc2.bias, layer.norms[2].weight, layer.norms[2].bias, nw, nb)
                x, cls_pred, mask_pred = outs[0], outs[1], outs[2]
                if nxt is not None:
                    o1 = outs[3]
                feats_q.append(x)
                cls_list.append(cls_pred)
                mask_list.append(mask_pred)
        if deferred:
            self._deferred_lowp_heads = dt != torch.float32
            self._deferred_autocast = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else None
            outs = _DeferredHeads.apply(self, torch.stack(cls_list, 0), mask_features, *feats_q)
            cls_list, mask_list = list(outs[0].unbind(0)), list(outs[1:])
        return cls_list, mask_list, [None for _ in cls_list]

    # ------------------------------------------------------------------ loss
    def _stream_for(self, name: str, device):
        key = (name, str(device))
        st = self._side_streams.get(key)
        if st is None:
            st = self._side_streams[key] = torch.cuda.Stream(device=device)
        return st

    @torch.no_grad()
    def _real_cols(self, labels_gt, gt_flat, d, b, nq, ng):
        """Per problem, the count K of real ground-truth columns for K9's padded mode (None: solve the square problem).  The
        dataset pads the instance list to num_queries with all-zero masks of label 0; a column counts as padding only if it
        has label 0 AND an empty mask AND every later column is padding too."""
        if not (nq == ng <= 320 and isinstance(gt_flat, ops.PackedMasks) and switches.get('k9_padded')):
            return None
        dev = labels_gt.device
        # "the packed mask holds a set bit" in TWO reduction stages of <= 128 words per output: a one-stage reduction of 8 192
        # words per mask is split by ATen over several workgroups that meet through a memset-cleared semaphore — not
        # replay-safe inside a captured HIP graph on this stack (scratch/dbg_graph_reduce.py)
        words = gt_flat.words.view(b, ng, -1)
        nw = words.shape[-1]
        inner = next((k for k in range(min(256, nw), 15, -1) if nw % k == 0), 0) if nw > 256 else 0     # 512 x 512: 8 192 words = 32 x 256
        any_bit = words.view(b, ng, nw // inner, inner).amax(-1).amax(-1) if inner else words.amax(-1)
        real = (labels_gt != 0) | (any_bit != 0)                                                     # (B, G)
        last = (real.to(torch.int32) * (self._iota(ng, dev).view(1, ng) + 1)).amax(-1)                # (B,) = K
        return last.to(torch.int32).view(1, b).expand(d, b).reshape(-1).contiguous()

    @staticmethod
    def _cached_ready(device):
        """A tensor that enters one of this head's caches is produced on whatever stream first asks for it — the matcher's,
        the target preparation's — and then read by all of them for the rest of the run.  Its first readers on OTHER streams
        would otherwise race with the kernels that fill it (seen as a loss that differed in the 6th digit one run in four,
        through a half-written row-index vector): wait for it once, here.  (Not during a HIP-graph capture: the eager
        warm-up steps in front of a capture have filled the caches.)"""
        if device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(device).synchronize()

    def _iota(self, n: int, device, div: int = 1, mod: int = 0, mul: int = 1) -> torch.Tensor:
        """Cached int32 index vector ((i // div) % mod) * mul, i < n."""
        key = (n, div, mod, str(device)) if mul == 1 else (n, div, mod, str(device), mul)
        t = self._iota_cache.get(key)
        if t is None:
            t = torch.arange(n, device=device)
            t = t // div if div > 1 else t
            t = t % mod if mod > 0 else t
            t = t * mul if mul != 1 else t
            t = t.to(torch.int32)
            self._cached_ready(t.device)
            self._iota_cache[key] = t
        return t

    def _const(self, device, values) -> torch.Tensor:
        """Small f32 constants as cached device tensors (a host→device copy is not capturable in a HIP graph)."""
        key = ('const', str(device), tuple(float(v) for v in values))
        t = self._iota_cache.get(key)
        if t is None:
            t = torch.tensor(list(key[2]), dtype=torch.float32, device=device)
            self._cached_ready(t.device)
            self._iota_cache[key] = t
        return t

    def _next_point_seed(self, device) -> torch.Tensor:
        """Device-resident 64-bit seed of the in-kernel point generator, advanced in place once per loss evaluation
        (an ordinary kernel: it is captured and replayed with a HIP graph).  Initialised from torch's generator, so
        ``torch.manual_seed`` makes runs repeatable; ranks are decorrelated by their rank."""
        key = ('point_seed', str(device))
        t = self._iota_cache.get(key)
        if t is None:
            rank = torch.distributed.get_rank() if (torch.distributed.is_available()
                                                    and torch.distributed.is_initialized()) else 0
            t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=device) + rank * 1_000_003
            self._cached_ready(t.device)
            self._iota_cache[key] = t
        t.add_(0x2545F4914F6CDD1D)            # wraps modulo 2^64
        return t

    def _draw_points(self, pts: PointSource, num_layers: int, batch: int, g: int):
        """All uniform points of one loss evaluation, drawn in the reference's order (per decoder output:
        B x rand(1, P, 2) for the matcher at mask2former_head.py:191, then rand(g, 3P, 2) and
        rand(g, P - int(0.75 P), 2) inside get_uncertain_point_coords_with_randomness)."""
        p = self.num_points
        n_samp = int(p * self.oversample_ratio)
        n_rand = p - int(self.importance_sample_ratio * p)
        if pts.gen is None:                       # device RNG: the draw order is irrelevant, draw in bulk
            # the 3P over-sampled candidates (1.2 GB per step) are not drawn at all: the fused importance sampling
            # generates them in-kernel from the device seed (None here)
            return (pts.rand(num_layers * batch, p, 2), None,
                    pts.rand(num_layers * g, n_rand, 2) if n_rand > 0 else None)
        mc, oc, rc = [], [], []
        for _ in range(num_layers):
            mc.append(torch.cat([pts.rand(1, p, 2) for _ in range(batch)], 0))
            oc.append(pts.rand(g, n_samp, 2))
            if n_rand > 0:
                rc.append(pts.rand(g, n_rand, 2))
        return torch.cat(mc, 0), torch.cat(oc, 0), (torch.cat(rc, 0) if rc else None)

    def _sample_gt(self, gt, src_index, coords, coord_index):
        if isinstance(gt, ops.PackedMasks):
            return ops.point_sample_packed(gt, src_index, coords, coord_index)
        return ops.point_sample(gt, src_index, coords, coord_index)

    @torch.no_grad()
    def _match_cost(self, cls, masks_flat, labels_gt, gp, match_coords):
        """Matching costs of ALL decoder outputs and images at once (mask2former_head.py:154-210).
        cls (D, B, Q, K+1); masks_flat (D*B*Q, H, W); gp (D*B*G, P) the ground truth sampled at match_coords (D*B, P, 2)
        (:meth:`_matcher_targets`).  Costs 2*(-softmax cls) +
        5*BCE + 5*dice on the sampled points; the D*B assignment problems are then solved on the device by one
        launch of K9 — no device→host copy, no scipy."""
        d, b, nq = cls.shape[:3]
        ng = labels_gt.shape[1]
        dev = cls.device
        mp = ops.point_sample(masks_flat, self._iota(d * b * nq, dev), match_coords,
                              self._iota(d * b * nq, dev, div=nq)).view(d, b, nq, -1)            # (D, B, Q, P)
        gp = gp.view(d, b, ng, -1)                                                               # (D, B, G, P)
        if (cls.is_cuda and switches.get('loss_glue') and switches.get('match_fused')
                and ops.match_products_supported(nq, ng, mp.shape[-1])):
            # K13c: the products x·t, sigmoid(x)·t and every row sum from one kernel (terms evaluated per 32-point chunk,
            # split into half pairs, contracted on MFMA) — the (DB, 3Q + 1, P) term planes and the f32 GEMM are gone
            prod, neg = ops.match_products(mp.reshape(d * b, nq, -1), gp.reshape(d * b, ng, -1))
            return ops.match_cost_split(cls, labels_gt, prod, neg, self.num_points)
        gpt = gp.reshape(d * b, ng, -1).transpose(1, 2)                                       # (DB, P, G)
        if cls.is_cuda and switches.get('loss_glue'):
            # K13: the cost terms with an all-ones row behind them — ONE batched GEMM against the sampled ground truth then
            # gives the three cost matrices AND the targets' row sums — and one launch that turns the products, the class
            # logits and the labels into the cost matrices (was: softmax, gather, a 200 MB row reduction, ~ 20 ATen launches)
            terms, sums = ops.match_cost_terms(mp.reshape(d * b, nq, -1), ones_row=True)      # (DB, 3Q + 1, P), (DB, Q, 2)
            return ops.match_cost(cls, labels_gt, torch.matmul(terms, gpt), sums, self.num_points)
        prob = cls.softmax(-1)
        lab = labels_gt.view(1, b, 1, ng).expand(d, b, nq, ng)
        cls_cost = -torch.gather(prob, 3, lab) * 2.0                                            # (D, B, Q, G)
        # K13: softplus(-x), softplus(x), sigmoid(x) and two row sums in one pass over the sampled logits; the three
        # cost matrices come from ONE batched GEMM against the sampled ground truth
        terms, sums = ops.match_cost_terms(mp.reshape(d * b, nq, -1))                         # (DB, 3Q, P), (DB, Q, 2)
        prod = torch.matmul(terms, gpt).view(d, b, 3, nq, ng)
        pos_gp, neg_gp, sig_gp = prod[:, :, 0], prod[:, :, 1], prod[:, :, 2]
        sums = sums.view(d, b, nq, 2)
        # BCE against 1 on the GT pixels + against 0 elsewhere: neg·(1 - gp) = Σ neg - neg·gp
        bce = (pos_gp + sums[..., 0:1] - neg_gp) / self.num_points
        den = sums[..., 1:2] + gp.sum(-1)[..., None, :]
        dice = 1 - (2 * sig_gp + 1.0) / (den + 1.0)
        return (cls_cost + 5.0 * bce + 5.0 * dice).flatten(0, 1)                               # (D*B, Q, G)

    def loss(self, all_cls_scores, all_mask_preds, gt_labels_list, gt_masks_list, img_metas=None, heights_pred=None,
             heights_gt=None) -> Dict[str, torch.Tensor]:
        """Mask2FormerHead.loss (mask2former_head.py:246-298 → _loss_by_feat_single :326-426) evaluated for all
        decoder outputs in one batched pass; labels (B, G) int64, masks (B, G, ny, nx) {0,1}.
        Differences in *mechanism* only: points are sampled by K8 straight from the (B, Q, H, W) logits and
        the (B, G, ny, nx) GT masks (no gathered copies), the assignments come from K9 on the device, and
        nothing synchronises with the host."""
        labels_gt = gt_labels_list if torch.is_tensor(gt_labels_list) else torch.stack(list(gt_labels_list), 0)
        packed_gt = gt_masks_list if isinstance(gt_masks_list, ops.PackedMasks) else None
        masks_gt = None
        if packed_gt is None:
            masks_gt = gt_masks_list if torch.is_tensor(gt_masks_list) else torch.stack(list(gt_masks_list), 0)
        dev = all_cls_scores[0].device
        d = len(all_cls_scores)
        b, nq = all_cls_scores[0].shape[:2]
        ng = labels_gt.shape[1]
        m = min(nq, ng)                                   # matched pairs per image (LSA matches min(Q, G))
        g = b * m
        p = self.num_points
        eps = torch.finfo(torch.float32).eps
        cls = torch.stack([c.float() for c in all_cls_scores], 0)                                # (D, B, Q, K+1)
        stacked = ops.stack_slices(getattr(self, '_mask_stack', None), list(all_mask_preds))
        stack_is_buffer = stacked is not None
        if stacked is None:
            stacked = torch.stack([mk.float() for mk in all_mask_preds], 0)
        masks_flat = stacked.flatten(0, 2)                                                       # (D*B*Q, H, W)
        # Everything of the loss that depends on the BATCH only — the packed targets, the uniform points, the targets sampled
        # at the matcher's points, the count of real columns — is issued on a stream that forks where the head's forward began
        # (the event forward() recorded): in the captured step it is a parallel branch underneath the pixel decoder and the
        # decoder layers instead of ≈ 0.25 ms in front of the matcher.
        early = getattr(self, '_early_event', None)
        self._early_event = None
        if early is not None:
            # the fork is only taken for the very tensors announced before forward() (announce_targets): anything else —
            # list targets stacked above, tensors that reached the device after the head started — is ordered by the
            # caller's stream, so its preparation stays there
            early, announced = early
            if announced != (self._target_key(labels_gt), self._target_key(packed_gt if packed_gt is not None else masks_gt)):
                early = None
        prep = self._stream_for('prep', dev) if (early is not None and cls.is_cuda) else None
        main = torch.cuda.current_stream() if cls.is_cuda else None
        if prep is not None:
            prep.wait_event(early)
        with (torch.cuda.stream(prep) if prep is not None else contextlib.nullcontext()):
            if packed_gt is not None:                     # K14 produced the bit-packed targets directly (batch.py)
                gt_flat = packed_gt
            else:
                gt_flat = masks_gt.float().flatten(0, 1)                                         # (B*G, ny, nx)
                if self.binary_gt_masks and gt_flat.shape[1] * gt_flat.shape[2] <= 1024 * 1024:
                    gt_flat = ops.pack_binary_masks(gt_flat)  # {0,1} by the batch contract: 32 KB per 512x512 mask
            pts = PointSource(dev, self.point_seed)
            match_c, over_c, rand_c = self._draw_points(pts, d, b, g)
            with torch.no_grad():
                gp_match = self._sample_gt(gt_flat, self._iota(d * b * ng, dev, mod=b * ng), match_c,
                                           self._iota(d * b * ng, dev, div=ng))                   # (D*B*G, P)
            real_k = self._real_cols(labels_gt, gt_flat, d, b, nq, ng)
        if prep is not None:
            main.wait_stream(prep)
        # K9 is latency-bound (one wavefront per problem) and the cost kernels in front of it are short.  When every query
        # gets matched (G >= Q, the dataset's padding convention) nothing of the importance sampling below depends on the
        # assignment, so the whole matcher — sampling the logits at its points, K13c, K9 — runs on a side stream underneath it.
        overlap = m == nq and self.overlap_matcher and cls.is_cuda
        if overlap:
            side = self._stream_for('matcher', dev)
            # every buffer the side stream's result lives in is allocated on the main stream and outlives the join below,
            # so no record_stream bookkeeping is needed (it also upsets a later HIP-graph capture)
            assigned = torch.empty((d * b, nq), dtype=torch.int32, device=dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cost = self._match_cost(cls, masks_flat.detach(), labels_gt, gp_match, match_c)
                ops.hungarian(cost.float().contiguous(), out=assigned, real_cols=real_k)
            assigned = assigned.view(d, b, nq)
        else:
            cost = self._match_cost(cls, masks_flat.detach(), labels_gt, gp_match, match_c)
            assigned = ops.hungarian(cost, real_cols=real_k).view(d, b, nq)                       # (D, B, Q) i32
            matched = assigned >= 0
            safe = assigned.clamp(min=0).long()

        # matched (decoder output, image, query) rows in (d, b, q) order, without a host sync
        if m == nq:
            qsel = self._iota(nq, dev).long().view(1, 1, nq).expand(d, b, nq)
        else:
            qsel = torch.sort((~matched).to(torch.uint8), dim=-1, stable=True).indices[..., :m]
        if m == nq:
            pred_index = self._iota(d * b * nq, dev)                                             # every row of masks_flat, in order
        else:
            db = self._iota(d * b, dev).long().view(d, b, 1)
            pred_index = (db * nq + qsel).flatten().to(torch.int32)                              # rows of masks_flat
        rows = self._iota(d * g, dev)
        with torch.no_grad():
            n_unc = int(self.importance_sample_ratio * p)
            # most uncertain = smallest |logit| among the over-sampled candidates, then the uniform tail
            if over_c is None:
                coords = ops.sample_select_uncertain(masks_flat.detach(), pred_index, None, n_unc, rand_c,
                                                     seed=self._next_point_seed(dev),
                                                     num_candidates=int(p * self.oversample_ratio))
            else:
                coords = ops.sample_select_uncertain(masks_flat.detach(), pred_index, over_c, n_unc, rand_c)
        # (the sink applies when the stack IS the heads' buffer and every map is sampled: all queries matched)
        sink = getattr(self, '_stack_sink', None) if (stack_is_buffer and m == nq) else None
        pred = ops.point_sample(masks_flat, pred_index, coords, rows, grad_sink=sink)            # (D*g, P), grads
        if overlap:                                     # join the matcher
            main.wait_stream(side)
        # kept for the metrics path (mask_bev_amd/metrics.py), which the reference feeds by running the matcher again
        self.last_assignment = assigned.detach()
        self.last_gt_packed = gt_flat if isinstance(gt_flat, ops.PackedMasks) else None
        glue = cls.is_cuda and switches.get('loss_glue')
        with torch.no_grad():
            if m == nq and glue:
                # every query is matched and `qsel` is the identity: the ground-truth row of (d, b, q) is b * G + assigned —
                # two launches on the int32 assignment instead of compare / clamp / cast / gather / multiply / add / cast
                boff = self._iota(d * b * nq, dev, div=nq, mod=b, mul=ng).view(d, b, nq)
                gt_index = (assigned.clamp(min=0) + boff).flatten()
            else:
                if overlap:
                    matched = assigned >= 0
                    safe = assigned.clamp(min=0).long()
                bsel = self._iota(b, dev).long().view(1, b, 1)
                gt_index = (bsel * ng + torch.gather(safe, 2, qsel)).flatten().to(torch.int32)   # rows of gt_flat
            tgt = self._sample_gt(gt_flat, gt_index, coords, rows)                               # (D*g, P)

        # classification loss (class-weighted CE, avg_factor = sum of the class weights of the targets)
        class_weight = self._const(dev, self.class_weight)
        if cls.is_cuda and switches.get('loss_glue'):
            # K13: labels from the assignment, weighted cross entropy and its normaliser per decoder output in one launch
            # (and one for the gradient) instead of where / gather / log_softmax / nll_loss / index / sums
            loss_cls = ops.cls_loss(cls, assigned, labels_gt, class_weight, self.loss_cls_weight, eps)
        else:
            labels = torch.where(matched, torch.gather(labels_gt.view(1, b, ng).expand(d, b, ng), 2, safe),
                                 torch.full_like(safe, self.num_classes))
            ce = F.cross_entropy(cls.flatten(0, 2), labels.flatten(), weight=class_weight, reduction='none').view(d, -1)
            loss_cls = self.loss_cls_weight * ce.sum(1) / (class_weight[labels].view(d, -1).sum(1) + eps)

        # MaskPseudoSampler: avg_factor = num_pos + num_neg = Q per image; reduce_mean over ranks (:388) is the
        # identity for equal per-rank batches (drop_last=True), see ddp.py
        if self.world_size_fn is None:
            # avg_factor = B * Q is a host constant (reduce_mean over equal per-rank batches is the identity): the loss
            # weights over it are plain floats — no device arithmetic, and K13's fused reduce takes them as arguments
            num_total_masks = max(float(b * nq), 1.0)
        else:
            num_total_masks = self.world_size_fn(self._const(dev, [float(b * nq)])).clamp(min=1.0)[0]

        if pred.is_cuda and switches.get('loss_node'):
            # K13's row sums (Σ σ·t, Σ σ, Σ t, Σ bce in one pass) and the dice / BCE algebra on them as one autograd node
            with torch.no_grad():
                c_dice = self.loss_dice_weight / (num_total_masks + eps)
                c_mask = self.loss_mask_weight / (num_total_masks * p + eps)
            loss_dice, loss_mask = ops.mask_dice_bce(pred, tgt, d, c_dice, c_mask)
        else:
            sums = ops.mask_loss_rows(pred, tgt)              # K13: (D*g, 4) = Σ σ·t, Σ σ, Σ t, Σ bce in one pass
            dice = (2 * sums[:, 0] + 1.0) / (sums[:, 1] + sums[:, 2] + 1.0)
            loss_dice = self.loss_dice_weight * (1 - dice).view(d, g).sum(1) / (num_total_masks + eps)
            loss_mask = self.loss_mask_weight * sums[:, 3].view(d, g).sum(1) / (num_total_masks * p + eps)

        out = LossDict(loss_cls=loss_cls[-1], loss_mask=loss_mask[-1], loss_dice=loss_dice[-1], loss_height=0)
        for i in range(d - 1):
            out[f'd{i}.loss_cls'], out[f'd{i}.loss_mask'], out[f'd{i}.loss_dice'] = loss_cls[i], loss_mask[i], loss_dice[i]
            out[f'd{i}.loss_height'] = 0
        # the sum of every entry, taken on the (D,) vectors: 3 reductions instead of 40 scalar adds forward and
        # ≈ 120 slice-gradient kernels backward (MaskBevModule.loss returns it when handed this dict)
        out.total = torch.cat([loss_cls, loss_mask, loss_dice]).sum()
        return out
