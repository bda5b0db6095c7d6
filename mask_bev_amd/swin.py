"""Swin backbone of MaskBEV, organised around channels-last (B, H, W, C) token maps.

Mirrors the interface and the checkpoint keys of ``CustomSwinTransformer``
(/root/reference: mask_bev/models/networks/swin/swin.py:465-774); the shifted-window attention
(swin.py:80-118,179-284) is one fused op, :func:`mask_bev_amd.ops.window_attention`, that folds
padding, cyclic shift, window partition/reverse, relative-position bias and the shift mask into its
addressing instead of materialising five layout copies per block.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, switches
from .layers import FFN, LayerNorm, Linear, PatchEmbed, PatchMerging, trunc_normal_


def relative_position_index(ws: int) -> torch.Tensor:
    """(ws², ws²) index into the (2ws-1)² bias table: (dy + ws - 1) * (2ws - 1) + (dx + ws - 1) with
    d = query - key.  Same values as the buffer built at swin.py:64-68."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing='ij')).flatten(1)   # (2, ws²)
    rel = coords[:, :, None] - coords[:, None, :] + (ws - 1)
    return (rel[0] * (2 * ws - 1) + rel[1]).contiguous()


class WindowMSA(nn.Module):
    """Parameter holder of one window attention (keys as swin.py:59-73)."""

    def __init__(self, embed_dims: int, num_heads: int, window_size: int):
        super().__init__()
        self.embed_dims, self.num_heads, self.window_size = embed_dims, num_heads, window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer('relative_position_index', relative_position_index(window_size))
        self.qkv = Linear(embed_dims, embed_dims * 3)
        self.proj = Linear(embed_dims, embed_dims)
        trunc_normal_(self.relative_position_bias_table, std=0.02)


class ShiftWindowMSA(nn.Module):
    def __init__(self, embed_dims: int, num_heads: int, window_size: int, shift_size: int):
        super().__init__()
        assert 0 <= shift_size < window_size
        self.window_size, self.shift_size = window_size, shift_size
        self.w_msa = WindowMSA(embed_dims, num_heads, window_size)

    def forward(self, x: torch.Tensor, defer_out_bias: bool = False) -> torch.Tensor:   # (B, H, W, C) → (B, H, W, C)
        m = self.w_msa
        # the bias gradient of the qkv projection is the column sums of d(qkv): K4's backward has those tiles in
        # registers anyway (and already collects the padded tokens' share), so it accumulates them and the Linear skips
        fb = torch.is_grad_enabled() and x.is_cuda and m.qkv.bias is not None
        qkv = m.qkv(x, skip_bias_grad=fb)
        o = ops.window_attention(qkv, m.qkv.bias, m.relative_position_bias_table, m.num_heads, self.window_size,
                                 self.shift_size, full_bias_grad=fb)
        return m.proj(o, skip_bias_grad=defer_out_bias)


class SwinBlock(nn.Module):
    def __init__(self, embed_dims: int, num_heads: int, feedforward_channels: int, window_size: int, shift: bool):
        super().__init__()
        self.norm1 = LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0)
        self.norm2 = LayerNorm(embed_dims)
        self.ffn = FFN(embed_dims, feedforward_channels, act='gelu')

    def forward(self, x: torch.Tensor, pending: Optional[torch.Tensor] = None,
                pending_bias: Optional[torch.Tensor] = None, defer_ffn_bias: bool = True):
        """Pre-LN block (swin.py:357-377): x ← x + attn(LN1(x)); x ← x + ffn(LN2(x)).  The residual adds are fused
        into the LayerNorm that reads their result (K12): the block takes the not-yet-added output ``pending`` of
        the previous block's FFN and returns ``(x, pending, pending_bias)`` with the stream's value being
        ``x + pending``.  The bias gradients of the two output projections (proj, fc2) are column sums of exactly
        the gradient K12 computes for its residual input, so K12 accumulates them (``pending_bias`` names the bias
        whose gradient the consumer of ``pending`` owes; ``defer_ffn_bias=False`` when that consumer is not K12)."""
        c = x.shape[-1]
        y, x = self.norm1(x, pending, gemm_input=True, return_sum=True, residual_bias=pending_bias)
        proj_b = self.attn.w_msa.proj.bias
        d1 = ops.bias_grad_deferrable(proj_b, c)
        y, x = self.norm2(x, self.attn(y, defer_out_bias=d1), gemm_input=True, return_sum=True,
                          residual_bias=proj_b if d1 else None)
        fc2_b = self.ffn.layers[1].bias
        d2 = defer_ffn_bias and ops.bias_grad_deferrable(fc2_b, c)
        return x, self.ffn(y, add_identity=False, defer_out_bias=d2), (fc2_b if d2 else None)


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims: int, num_heads: int, feedforward_channels: int, depth: int, window_size: int,
                 downsample: nn.Module = None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, shift=(i % 2 == 1))
            for i in range(depth)])
        self.downsample = downsample

    def forward(self, x: torch.Tensor, out_norm: Optional[nn.Module] = None, pending: Optional[torch.Tensor] = None,
                out_gemm_input: bool = False):
        """→ (input of the next stage, this stage's output — normalised by ``out_norm`` when given; the last
        residual add of the stage is fused into that LayerNorm).  ``pending``: a term still to be added to ``x`` (the
        absolute position embedding in front of the first stage); the first block's LayerNorm launch adds it."""
        pending_bias = None
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):
            x, pending, pending_bias = blk(x, pending, pending_bias, defer_ffn_bias=(i < last or out_norm is not None))
        if out_norm is not None:
            # ``out_gemm_input``: the stage output only feeds 1 x 1 convolutions (GEMMs) — under 16-bit autocast K12 stores it
            # in that type, what those GEMMs would cast it to anyway
            out, x = out_norm(x, pending, return_sum=True, residual_bias=pending_bias, gemm_input=out_gemm_input)
        else:
            x = x if pending is None else x + pending
            out = x
        return (self.downsample(x) if self.downsample is not None else x), out


class CustomSwinTransformer(nn.Module):
    """Same constructor keywords as the reference class (swin.py:523-548) for the subset MaskBEV uses
    (mask_bev/models/backbones/mask_bev_backbone.py:39-64); dropout / drop-path are 0 there."""

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 use_abs_pos_embed=False, act_cfg=None, norm_cfg=None, with_cp=False, init_cfg=None,
                 swap_dims=False, **_unused):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError('MaskBEV builds its backbone with all dropout rates at 0 '
                                      '(mask_bev_backbone.py:55-57)')
        if not qkv_bias or qk_scale is not None or not patch_norm:
            raise NotImplementedError('only the MaskBEV configuration (qkv_bias, default scale, patch_norm)')
        if isinstance(pretrain_img_size, int):
            pretrain_img_size = (pretrain_img_size, pretrain_img_size)
        assert strides[0] == patch_size, 'Use non-overlapping patch embed.'
        self.out_indices = tuple(out_indices)
        self.use_abs_pos_embed = use_abs_pos_embed
        self.patch_embed = PatchEmbed(in_channels, embed_dims, patch_size)
        if use_abs_pos_embed:
            rows, cols = pretrain_img_size[0] // patch_size, pretrain_img_size[1] // patch_size
            if swap_dims:
                rows, cols = cols, rows
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dims, rows, cols))
        self.stages = nn.ModuleList()
        c = embed_dims
        for i, depth in enumerate(depths):
            down = PatchMerging(c, 2 * c, strides[i + 1]) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(c, num_heads[i], mlp_ratio * c, depth, window_size, down))
            if down is not None:
                c *= 2
        self.num_features = [int(embed_dims * 2 ** i) for i in range(len(depths))]
        for i in self.out_indices:
            self.add_module(f'norm{i}', LayerNorm(self.num_features[i]))

    def init_weights(self):
        """swin.py:674-682: trunc-normal(0.02) linears / abs-pos-embed, unit LayerNorms."""
        if self.use_abs_pos_embed:
            trunc_normal_(self.absolute_pos_embed, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor, cut: Optional[dict] = None) -> List[torch.Tensor]:
        """``cut = {'stage': i}`` severs the autograd graph in front of stage ``i`` (graph.py runs the backward of
        the head and the last stage, then of the earlier stages, as two HIP graphs with a gradient all-reduce
        launched in between): on return ``cut['x_in']`` is the attached input of that stage and ``cut['x_leaf']``
        the detached leaf the stage actually consumed (its ``.grad`` is the gradient to continue with)."""
        x = self.patch_embed(x)                                    # (B, H, W, E)
        b, h, w, e = x.shape
        pos_pending = None
        if self.use_abs_pos_embed:
            ape = self.absolute_pos_embed
            w_, h_ = ape.shape[2:4]                                # swin.py:750 — (w, h) on purpose
            if h != h_ or w != w_:
                ape = F.interpolate(ape, size=(h, w), mode='bicubic', align_corners=False)
            # the reference flattens the (rows, cols) map row-major into the token axis, whatever h, w are
            # (made contiguous first: the broadcast add of the transposed view ran at 1.8 TB/s, 84 us per step)
            if (x.is_cuda and ape is self.absolute_pos_embed and switches.get('pos_fused') and len(self.stages[0].blocks) > 0
                    and (cut is None or cut['stage'] > 0) and ops.add_layernorm_supported(e)):
                # the add rides on the first block's LayerNorm launch; the gradient takes one transposing pass (ops.pos_tokens)
                pos_pending = ops.pos_tokens(ape, b, h, w)
            else:
                x = x + ape.flatten(2).transpose(1, 2).reshape(1, h, w, e).contiguous()
        outs = []
        for i, stage in enumerate(self.stages):
            if cut is not None and i == cut['stage']:
                cut['x_in'] = x
                x = cut['x_leaf'] = x.detach().requires_grad_()
            # Stage outputs 1.. feed the head's input convolutions — GEMMs on the stream that produced them — and leave in the
            # autocast dtype (three cast launches per step).  Stage 0's feeds the FPN tail, which the pixel decoder runs on a
            # second stream: a 16-bit map produced HERE would be saved by a node whose backward runs THERE, and its block — freed
            # to this stream's pool when that node releases it — could be handed out again while the other stream still reads
            # it (seen as NaNs in lateral_convs.0's weight gradient).  It stays f32; its cast is the tail stream's own tensor.
            x, out = stage(x, getattr(self, f'norm{i}') if i in self.out_indices else None,
                           pending=pos_pending if i == 0 else None,
                           out_gemm_input=bool(i > 0 and switches.get('stage_out_lowp')))
            if i in self.out_indices:
                # (B, C, H, W) as the reference returns it, but as a VIEW of the channels-last map: the head's 1 x 1
                # convolutions read it as tokens (layers.conv1x1), so no NCHW copy is made — forward or backward
                outs.append(out.permute(0, 3, 1, 2) if out.is_cuda and switches.get('conv1x1_tokens')
                            else out.permute(0, 3, 1, 2).contiguous())
        return outs
