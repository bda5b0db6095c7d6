"""Building blocks shared by the backbone and the head.

Parameter names follow the checkpoints of the reference stack (mmcv 2.0.0 / mmdet 3.0.0 leaf names, see
SURVEY.md §8f-4) so that ``state_dict()`` keys are interchangeable; the implementations are our own and
are organised around channels-last (B, H, W, C) token maps, which is what the gfx950 kernels consume.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, switches


def trunc_normal_(t: torch.Tensor, std: float = 0.02) -> torch.Tensor:
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std)


class Linear(nn.Linear):
    """``nn.Linear`` (same parameters / checkpoint keys) routed through :func:`ops.linear`.
    ``skip_bias_grad=True``: the caller guarantees that the op consuming the output accumulates the bias gradient
    (K12 with ``branch_bias``)."""

    def forward(self, x: torch.Tensor, skip_bias_grad: bool = False) -> torch.Tensor:
        return ops.linear(x, self.weight, self.bias, skip_bias_grad=skip_bias_grad)


class LayerNorm(nn.LayerNorm):
    """``nn.LayerNorm(C)`` (same parameters / checkpoint keys) on K12: optionally fused with the residual add that
    feeds it, writing straight in the consumer's dtype.

    ``forward(x, residual=None, gemm_input=False, return_sum=False)``:
    ``y = LN(x + residual)``; ``gemm_input=True`` says y only feeds GEMM layers, so under autocast it is stored in
    the autocast dtype (what the GEMM would cast it to anyway); ``return_sum`` also returns the f32 sum; ``fanout``
    returns ``(y, y')``, the same values for two consumers (post-LN: the next residual add and the next branch)."""

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, gemm_input: bool = False,
                return_sum: bool = False, residual_bias: Optional[torch.Tensor] = None, fanout: bool = False,
                branch_gemm: bool = False):
        """``residual_bias``: the bias Parameter of the Linear that produced ``residual`` when that layer was run with
        ``skip_bias_grad=True`` — its gradient (the column sums of d(residual)) is accumulated by this op.
        ``branch_gemm`` (with ``fanout``): the second tensor only feeds GEMM layers — under 16-bit autocast it is written in
        the autocast dtype by the same launch."""
        c = x.shape[-1]
        if (x.is_cuda and len(self.normalized_shape) == 1 and self.weight is not None and self.bias is not None
                and ops.add_layernorm_supported(c) and x.dtype in ops._ACT_DTYPES
                and (residual is None or residual.dtype in ops._ACT_DTYPES)):
            out_dtype = torch.float32
            if (gemm_input and torch.is_autocast_enabled('cuda')
                    and torch.get_autocast_dtype('cuda') in ops._LO_DTYPES):
                out_dtype = torch.get_autocast_dtype('cuda')
            if fanout and not switches.get('ln_fanout'):
                y = ops.add_layernorm(x, residual, self.weight, self.bias, self.eps, out_dtype, branch_bias=residual_bias)
                return y, y
            branch_dtype = None
            if (fanout and branch_gemm and torch.is_autocast_enabled('cuda')
                    and torch.get_autocast_dtype('cuda') in ops._LO_DTYPES and switches.get('ln_branch_lowp')):
                branch_dtype = torch.get_autocast_dtype('cuda')
            return ops.add_layernorm(x, residual, self.weight, self.bias, self.eps, out_dtype, return_sum,
                                     branch_bias=residual_bias, fanout=fanout, branch_dtype=branch_dtype)
        if residual_bias is not None:
            residual = ops.accumulate_bias_grad(residual, residual_bias)      # the deferred gradient must not be lost
        s = x if residual is None else x + residual
        y = super().forward(s.float() if s.dtype != torch.float32 and not torch.is_autocast_enabled('cuda') else s)
        if fanout:
            return y, y
        return (y, s) if return_sum else y


class FFN(nn.Module):
    """Two-layer MLP with residual; keys ``layers.0.0.*`` / ``layers.1.*`` (mmcv ``FFN`` layout, used at
    mask_bev/models/networks/swin/swin.py:347-355 and mask_bev_panoptic_head.py:137-142,168-175)."""

    def __init__(self, embed_dims: int, feedforward_channels: int, act: str = 'gelu'):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(Linear(embed_dims, feedforward_channels), nn.GELU() if act == 'gelu' else nn.ReLU()),
            Linear(feedforward_channels, embed_dims))

    def forward(self, x: torch.Tensor, identity: Optional[torch.Tensor] = None, add_identity: bool = True,
                defer_out_bias: bool = False) -> torch.Tensor:
        """``add_identity=False`` returns the branch alone: the caller fuses the residual add into the LayerNorm
        that follows (K12); with ``defer_out_bias`` that LayerNorm also accumulates the output layer's bias gradient
        (pass ``self.layers[1].bias`` as its ``residual_bias``)."""
        fc1, fc2 = self.layers[0][0], self.layers[1]
        kind = 'gelu' if isinstance(self.layers[0][1], nn.GELU) else 'relu'
        if ops.ffn_fused_ok(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias):
            # K17: fc1 + bias + activation, and fc2's data gradient + activation backward + d bias, as single launches
            y = ops.ffn(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, kind,
                        defer_out_bias=defer_out_bias and not add_identity)
            return y if not add_identity else (x if identity is None else identity) + y
        if ops.ffn32_ok(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias):
            # fp32 compute, K20: the same two fusions on f32 products formed from IEEE-half pairs
            y = ops.ffn32(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, kind,
                          defer_out_bias=defer_out_bias and not add_identity)
            return y if not add_identity else (x if identity is None else identity) + y
        if not add_identity:
            return self.layers[1](self._hidden(x), skip_bias_grad=defer_out_bias)
        return (x if identity is None else identity) + self.layers[1](self._hidden(x))

    def _hidden(self, x: torch.Tensor) -> torch.Tensor:
        """act(fc1(x)); for token-major activations beyond the small-token path the bias gradient of fc1 is taken
        inside the activation's backward kernel (ops.bias_act) instead of by a separate pass over d(fc1 output)."""
        fc1, act = self.layers[0][0], self.layers[0][1]
        kind = 'gelu' if isinstance(act, nn.GELU) else 'relu'
        rows = x.numel() // max(1, x.shape[-1])
        d = (rows > 2048 and fc1.out_features % 4 == 0 and ops.bias_grad_deferrable(fc1.bias, 4))
        return ops.bias_act(fc1(x, skip_bias_grad=d), fc1.bias if d else None, kind)


def corner_pad(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """Pad bottom/right of an NCHW map so a (k, stride s) kernel tiles it ('corner' adaptive padding)."""
    h, w = x.shape[-2:]
    ph = max((math.ceil(h / s) - 1) * s + k - h, 0)
    pw = max((math.ceil(w / s) - 1) * s + k - w, 0)
    return F.pad(x, [0, pw, 0, ph]) if (ph or pw) else x


class PatchEmbed(nn.Module):
    """Non-overlapping patch projection + LayerNorm → channels-last tokens (B, H', W', E).
    Keys ``projection.*`` / ``norm.*`` (mmdet ``PatchEmbed``, built at swin.py:579-586)."""

    def __init__(self, in_channels: int, embed_dims: int, patch: int):
        super().__init__()
        self.patch = patch
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=patch, stride=patch)
        self.norm = LayerNorm(embed_dims)

    def forward(self, x) -> torch.Tensor:
        if isinstance(x, ops.PatchTokens):
            # K3 already wrote the projection's input rows (bf16, (dy, c, dx) order): one GEMM, no layout pass
            if x.patch != self.patch or x.channels != self.projection.in_channels:
                raise ValueError('PatchTokens do not match this projection')
            w = self.projection.weight                                   # (E, C, p, p) → (E, p, C, p)
            w2 = w.permute(0, 2, 1, 3).reshape(w.shape[0], -1)
            return self.norm(ops.linear(x.rows, w2, self.projection.bias))
        x = corner_pad(x, self.patch, self.patch)
        if self.patch == 4 and ops.patch_embed32_ok(x, self.projection.weight, self.projection.bias):
            # fp32 compute: the projection as K20 products that gather the NCHW image (no MIOpen convolution, tokens out)
            return self.norm(ops.patch_embed32(x, self.projection.weight, self.projection.bias))
        x = self.projection(x)
        return self.norm(x.permute(0, 2, 3, 1))


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat (channel order c*4 + kh*2 + kw, as ``nn.Unfold``) → LN(4C) → Linear(4C→2C).
    Keys ``norm.*`` / ``reduction.weight`` (mmdet ``PatchMerging``, built at swin.py:611-616)."""

    def __init__(self, in_channels: int, out_channels: int, stride: int = 2):
        super().__init__()
        self.stride = stride
        self.norm = LayerNorm(4 * in_channels)
        self.reduction = Linear(4 * in_channels, out_channels, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:          # (B, H, W, C) → (B, H', W', 2C)
        b, h, w, c = x.shape
        s = self.stride
        ph = max((math.ceil(h / s) - 1) * s + 2 - h, 0)
        pw = max((math.ceil(w / s) - 1) * s + 2 - w, 0)
        if ph or pw:
            x = F.pad(x, (0, 0, 0, pw, 0, ph))
            h, w = h + ph, w + pw
        oh, ow = (h - 2) // s + 1, (w - 2) // s + 1
        n = self.norm
        if (s == 2 and not (ph or pw) and ops.merge_layernorm_supported(x) and n.weight is not None
                and n.bias is not None):
            # unfold + LayerNorm as one gather pass (K12): no (B, H/2, W/2, 4C) copy with 4-byte scattered elements
            lo = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in ops._LO_DTYPES
            y = ops.merge_layernorm(x, n.weight, n.bias, n.eps, torch.get_autocast_dtype('cuda') if lo else torch.float32)
            return self.reduction(y)
        if s == 2:
            x = x[:, :oh * 2, :ow * 2].reshape(b, oh, 2, ow, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(b, oh, ow, 4 * c)
        else:
            x = F.unfold(x.permute(0, 3, 1, 2), kernel_size=2, stride=s).transpose(1, 2).reshape(b, oh, ow, 4 * c)
        return self.reduction(self.norm(x, gemm_input=True))


def conv1x1(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A 1 x 1 ``nn.Conv2d`` (its parameters, its state-dict keys) evaluated as ONE strided-batched GEMM
    ``W (Cout, Cin) @ x[b] (Cin, H*W)`` instead of a MIOpen implicit-GEMM convolution: no NCHW<->NHWC transposes around
    it (0.11 ms per step of `batched_transpose` kernels in profiles/r02/b_kernel_stats.csv), and bit-reproducible in
    fp16 (MIOpen's fp16 solver for the mask-feature projection was measured to change its result from call to call,
    which made every decoder layer after it — and the gradients — irreproducible)."""
    b, c, h, w = x.shape
    if (not x.is_contiguous() and x.is_cuda and x.permute(0, 2, 3, 1).is_contiguous()
            and switches.get('conv1x1_tokens')):
        # a channels-last map (the backbone's stage outputs are (B, H, W, C) tokens seen through a permute): the GEMM
        # reads the token matrix as its transposed operand and writes NCHW; its backward returns token-major gradients
        y = ops.conv1x1_tokens(x.permute(0, 2, 3, 1).reshape(b, h * w, c), conv.weight.view(conv.weight.shape[0], c),
                               conv.bias)
        return y.view(b, -1, h, w)
    if x.is_cuda:
        # one node whose bias gradient is a product with a ones block (autograd's own would be a multi-workgroup ATen
        # reduction of 65 536 elements per channel: not replay-safe inside a captured HIP graph on this stack)
        return ops.conv1x1_rows(x.flatten(2), conv.weight.view(conv.weight.shape[0], c), conv.bias).view(b, -1, h, w)
    w2 = conv.weight.view(conv.weight.shape[0], c).unsqueeze(0).expand(b, -1, -1)
    x2 = x.flatten(2)
    if conv.bias is None:
        y = torch.bmm(w2, x2)
    else:
        y = torch.baddbmm(conv.bias.view(1, -1, 1), w2, x2)
    return y.view(b, -1, h, w)


class ConvGN(nn.Module):
    """conv → GroupNorm(32) [→ ReLU]; keys ``conv.*`` / ``gn.*`` (mmcv ``ConvModule`` layout used by the
    pixel decoder configured at mask_bev_panoptic_head.py:119-123)."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool, relu: bool, groups: int = 32):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias)
        self.gn = nn.GroupNorm(groups, cout)
        self.relu = relu

    def forward(self, x: torch.Tensor, add_upsampled: Optional[torch.Tensor] = None, conv_input: bool = False) -> torch.Tensor:
        """``add_upsampled``: a coarser (B, C, h, w) map whose bilinear up-sampling is added to the normalised result
        (the FPN step of the pixel decoder: ``lateral + F.interpolate(previous)``); ``conv_input``: the result only
        feeds convolutions, so under autocast it is stored in the autocast dtype (what they would cast it to)."""
        if self.conv.kernel_size == (1, 1) and x.is_cuda:
            y = conv1x1(self.conv, x)
        elif ops.conv3x3_32_ok(x, self.conv):
            y = ops.conv3x3_32(x, self.conv.weight)          # fp32 compute: K20 products instead of MIOpen
        elif ops.conv3x3_16_ok(x, self.conv):
            y = ops.conv3x3_16(x, self.conv.weight)          # 16-bit compute: K17 products instead of MIOpen
        else:
            y = self.conv(x)
        if ops.group_norm_supported(y, self.gn.num_groups) and self.gn.weight is not None and \
                (add_upsampled is None or (y.shape[-1] % 4 == 0 and add_upsampled.dtype in ops._ACT_DTYPES)):
            lo = (conv_input and torch.is_autocast_enabled('cuda')
                  and torch.get_autocast_dtype('cuda') in ops._LO_DTYPES)
            return ops.group_norm(y, self.gn.weight, self.gn.bias, self.gn.num_groups, self.gn.eps, self.relu,
                                  add_upsampled, torch.get_autocast_dtype('cuda') if lo else torch.float32)
        y = self.gn(y)
        if add_upsampled is not None:
            y = y + F.interpolate(add_upsampled, size=y.shape[-2:], mode='bilinear', align_corners=False)
        return F.relu(y) if self.relu else y


_SINE_CACHE: Dict[Tuple, torch.Tensor] = {}


def sine_positional_encoding(h: int, w: int, num_feats: int, device, temperature: float = 10000.0,
                             scale: float = 2 * math.pi, eps: float = 1e-6) -> torch.Tensor:
    """(1, 2*num_feats, H, W) normalised sine encoding of an un-padded map (mmdet ``SinePositionalEncoding``
    with normalize=True; mask_bev_panoptic_head.py:144-149).  Constant per shape, so cached on device."""
    key = (h, w, num_feats, str(device))
    pe = _SINE_CACHE.get(key)
    if pe is None:
        y = torch.arange(1, h + 1, dtype=torch.float32, device=device).view(h, 1).expand(h, w)
        x = torch.arange(1, w + 1, dtype=torch.float32, device=device).view(1, w).expand(h, w)
        y = y / (h + eps) * scale
        x = x / (w + eps) * scale
        dim_t = torch.arange(num_feats, dtype=torch.float32, device=device)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / num_feats)
        px = x[:, :, None] / dim_t
        py = y[:, :, None] / dim_t
        px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).view(h, w, -1)
        py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).view(h, w, -1)
        pe = torch.cat((py, px), dim=2).permute(2, 0, 1).unsqueeze(0).contiguous()
        _SINE_CACHE[key] = pe
    return pe


class _AttnParams(nn.Module):
    """Parameter holder with ``nn.MultiheadAttention``'s names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, embed_dims: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dims, embed_dims))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dims))
        self.out_proj = Linear(embed_dims, embed_dims)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class MultiheadAttention(nn.Module):
    """Residual multi-head attention with positional terms added to q / k only.
    Keys ``attn.in_proj_weight`` …  (mmcv ``MultiheadAttention`` wrapping ``nn.MultiheadAttention``;
    configured at mask_bev_panoptic_head.py:154-167, called from the decoder loop of
    mask_bev/models/networks/mask2former_head/mask2former_head.py:542-553)."""

    def __init__(self, embed_dims: int, num_heads: int):
        super().__init__()
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.attn = _AttnParams(embed_dims)

    def forward(self, query, key, value, query_pos=None, key_pos=None, blocked: Optional[torch.Tensor] = None,
                add_identity: bool = True, defer_out_bias: bool = False, shared_kv=None):
        """query (B, Q, E), key/value (B, L, E); ``blocked`` (B, 1|H, Q, L) bool, True = may NOT attend.
        ``add_identity=False`` returns the attention branch alone (residual add fused into the next LayerNorm).
        ``shared_kv`` = (holder, token, slot): keys / values were projected for this layer by
        ``ops.shared_kv_project`` together with the other layers of the same memory (key / value are ignored)."""
        e, h = self.embed_dims, self.num_heads
        w, bias = self.attn.in_proj_weight, self.attn.in_proj_bias
        q_in = query + query_pos if query_pos is not None else query
        q = ops.linear(q_in, w, bias, rows=(0, e))
        if shared_kv is not None:
            holder, token, slot = shared_kv
            o = ops.attention_shared_kv(q, token, blocked, h, holder, slot)
        else:
            if key is query and key_pos is query_pos:        # self-attention: the positioned input is shared by q and k
                k = q_in
            else:
                k = key + key_pos if key_pos is not None else key
            k = ops.linear(k, w, bias, rows=(e, 2 * e))
            v = ops.linear(value, w, bias, rows=(2 * e, 3 * e))
            o = ops.attention(q, k, v, blocked, h)                 # K6: heads split by addressing, mask per query
        o = self.attn.out_proj(o, skip_bias_grad=defer_out_bias and not add_identity)
        return query + o if add_identity else o


class MultiScaleDeformableAttention(nn.Module):
    """Multi-scale deformable self-attention of the pixel decoder (mmcv ``MultiScaleDeformableAttention``;
    configured at mask_bev_panoptic_head.py:127-136).  Keys ``sampling_offsets``, ``attention_weights``,
    ``value_proj``, ``output_proj``."""

    def __init__(self, embed_dims: int = 256, num_heads: int = 8, num_levels: int = 3, num_points: int = 4):
        super().__init__()
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self.sampling_offsets = Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.output_proj = Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.zeros_(self.sampling_offsets.weight)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, query: torch.Tensor, query_pos: torch.Tensor, reference_points: torch.Tensor,
                spatial_shapes: Sequence[Tuple[int, int]], shapes_t: torch.Tensor, level_start: torch.Tensor,
                add_identity: bool = True, defer_out_bias: bool = False, pos_share=None, pos_share_index: int = 0,
                wcat=None):
        """query (B, N, E) un-positioned (also the value source); reference_points (N, 2) in [0, 1] (x, y).
        ``pos_share``: an ``ops.PosGradShare`` common to a chain of layers that add the same ``query_pos``."""
        b, n, e = query.shape
        h, l, p = self.num_heads, self.num_levels, self.num_points
        if (spatial_shapes is not None and query.is_cuda and query.dtype == torch.float32
                and query_pos.shape[0] == 1 and sum(hh * ww for hh, ww in spatial_shapes) == n
                and ops.msda_prepare_supported(l, p) and switches.get('msda_fused')):
            out = ops.msda_query_side(query, query_pos, reference_points, self.value_proj, self.sampling_offsets,
                                      self.attention_weights, h, l, p, spatial_shapes, shapes_t, level_start,
                                      pos_share=pos_share, pos_share_index=pos_share_index, wcat=wcat)
            out = self.output_proj(out, skip_bias_grad=defer_out_bias and not add_identity)
            return out + query if add_identity else out
        q = query + query_pos
        if torch.is_autocast_enabled('cuda') and q.is_cuda:
            q = q.to(torch.get_autocast_dtype('cuda'))       # one cast for both projections of q
        value = self.value_proj(query).view(b, n, h, e // h)
        off = self.sampling_offsets(q).view(b, n, h, l, p, 2)
        aw = self.attention_weights(q).view(b, n, h, l * p)
        if spatial_shapes is not None and off.is_cuda and ops.msda_prepare_supported(l, p):
            loc, aw = ops.msda_prepare(off, aw, reference_points, spatial_shapes)              # K16: one launch
        else:
            aw = aw.softmax(-1).view(b, n, h, l, p)
            normalizer = torch.stack([shapes_t[:, 1], shapes_t[:, 0]], -1).to(off.dtype)      # (L, 2) = (w, h)
            loc = reference_points.view(1, n, 1, 1, 1, 2) + off / normalizer.view(1, 1, 1, l, 1, 2)
        out = ops.ms_deform_attn(value, spatial_shapes, shapes_t, level_start, loc, aw)
        out = self.output_proj(out, skip_bias_grad=defer_out_bias and not add_identity)
        return out + query if add_identity else out
