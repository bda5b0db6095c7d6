"""The loss path: K8 indexed bilinear point sampling (+ bit-packed targets), K9 batched Hungarian assignment, K10 importance
sampling, K13 mask-loss row sums / matching costs / class loss (mask2former_head.py:154-232,326-426)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, switches
from ._lib import MaskBevHipError, check
from .ops_core import *          # noqa: F401,F403


# --------------------------------------------------------------------------------------
# K8 indexed bilinear point sampling (loss / matcher)
# --------------------------------------------------------------------------------------
class StackGradSink:
    """Side channel for the gradient of the stacked mask logits (D, B, Q, H, W).  Autograd requires that gradient in the
    stack's own shape and type — f32, decoder-output-major — while its only consumer, the batched backward of the
    prediction heads (mask2former_head._DeferredHeads), wants it sample-major in the GEMM operand type: a 262 MB permute +
    cast pass.  With a sink armed, K8's backward stores the gradient in THAT form here and hands autograd a zero-stride
    token of the required shape; the consumer checks that what reached it is the token (nothing else contributed a
    gradient) and takes ``grad``; otherwise it finds ``grad`` unset or the token replaced and uses the ordinary tensors."""

    def __init__(self, outer: int, inner: int, rows: int, dtype: torch.dtype, device):
        self.dims = (int(outer), int(inner), int(rows))
        self.dtype = dtype
        self.token = torch.zeros((), dtype=torch.float32, device=device)
        self.grad = None            # (inner, outer, rows, H*W) in `dtype`, written by K8's backward

    def is_token(self, g) -> bool:
        return (g is not None and g.data_ptr() == self.token.data_ptr() and all(s == 0 for s in g.stride()))


class _PointSample(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, src, src_index, coords, coord_index, sink=None):
        lib = _lib.load()
        ctx.sink = sink
        _need_gpu(src, src_index, coords, coord_index)
        src, coords = src.contiguous(), coords.contiguous()
        n_src, h, w = src.shape
        g = int(src_index.shape[0])
        p = int(coords.shape[1])
        out = torch.empty((g, p), dtype=torch.float32, device=src.device)
        _lib.WORK_HINT['point_sample'] = (int(n_src), int(coords.shape[0]))
        rc = lib.mbv_point_sample_fwd(_ptr(src), _ptr(src_index), _ptr(coords), _ptr(coord_index), g, p, h, w, _ptr(out),
                                      _stream())
        check(rc, 'mbv_point_sample_fwd')
        ctx.save_for_backward(src_index, coords, coord_index)
        ctx.dims = (n_src, h, w)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_out):
        lib = _lib.load()
        src_index, coords, coord_index = ctx.saved_tensors
        n_src, h, w = ctx.dims
        grad_out = grad_out.to(torch.float32).contiguous()
        g, p = grad_out.shape
        _lib.WORK_HINT['point_sample'] = (int(n_src), int(coords.shape[0]))
        sink = ctx.sink
        if sink is not None:
            o, n, r = sink.dims
            if g == n_src == o * n * r and g <= 65535 and w <= 16384 and -(-h // max(1, 16384 // w)) <= 64:
                sink.grad = torch.empty((n, o, r, h * w), dtype=sink.dtype, device=grad_out.device)
                check(lib.mbv_point_sample_bwd_stack(_ptr(grad_out), _ptr(src_index), _ptr(coords), _ptr(coord_index), g, p,
                                                     h, w, o, n, r, _ptr(sink.grad), _dt_flag(sink.dtype), _stream()),
                      'mbv_point_sample_bwd_stack')
                return sink.token.expand(n_src, h, w), None, None, None, None
        g_src = torch.empty((n_src, h, w), dtype=torch.float32, device=grad_out.device)
        rc = lib.mbv_point_sample_bwd(_ptr(grad_out), _ptr(src_index), _ptr(coords), _ptr(coord_index), g, p, h, w, n_src,
                                      _ptr(g_src), _stream())
        check(rc, 'mbv_point_sample_bwd')
        return g_src, None, None, None, None


def point_sample(src: torch.Tensor, src_index: torch.Tensor, coords: torch.Tensor,
                 coord_index: torch.Tensor, grad_sink: Optional[StackGradSink] = None) -> torch.Tensor:
    """out[g, p] = bilinear(src[src_index[g]], coords[coord_index[g], p]) — mmcv ``point_sample`` semantics
    (grid_sample at 2p-1, align_corners=False, zero padding) without gathering the maps first (K8).
    src (N, H, W); indices int32 (G,), ``src_index`` without duplicates; coords (*, P, 2) in [0, 1] as (x, y).
    ``grad_sink``: see :class:`StackGradSink` (src is then a stack of which every map is sampled)."""
    src, coords = src.float(), coords.float()
    n = int(src_index.shape[0])
    if n <= 65535:
        return _PointSample.apply(src, src_index, coords, coord_index, grad_sink)
    return torch.cat([_PointSample.apply(src, src_index[i:i + 65535], coords, coord_index[i:i + 65535])
                      for i in range(0, n, 65535)], 0)


class PackedMasks:
    """Binary maps packed 32 pixels / word (see include/maskbev_hip.h)."""

    def __init__(self, words: torch.Tensor, h: int, w: int):
        self.words, self.h, self.w = words, h, w


@torch.no_grad()
def pack_binary_masks(masks: torch.Tensor, out: Optional[PackedMasks] = None) -> PackedMasks:
    """masks (N, H, W) with values in {0, 1} → bit-packed form for :func:`point_sample_packed` (into ``out``'s words
    when given: the HIP-graph step packs each batch's dense targets straight into its static buffer)."""
    lib = _lib.load()
    _need_gpu(masks)
    masks = masks.float().contiguous()
    n, h, w = masks.shape
    if out is not None:
        words = out.words
        if (out.h, out.w) != (h, w) or tuple(words.shape) != (n, lib.mbv_packed_mask_words(h, w)) \
                or words.dtype != torch.int32 or not words.is_contiguous() or words.device != masks.device:
            raise MaskBevHipError('pack_binary_masks: `out` does not fit these masks')
    else:
        words = torch.empty((n, lib.mbv_packed_mask_words(h, w)), dtype=torch.int32, device=masks.device)
    for i in range(0, n, 65535):
        rc = lib.mbv_pack_binary_masks(_ptr(masks[i:i + 65535]), min(65535, n - i), h, w, _ptr(words[i:i + 65535]),
                                       _stream())
        check(rc, 'mbv_pack_binary_masks')
    return out if out is not None else PackedMasks(words, h, w)


@torch.no_grad()
def point_sample_packed(pm: PackedMasks, src_index: torch.Tensor, coords: torch.Tensor,
                        coord_index: torch.Tensor) -> torch.Tensor:
    """:func:`point_sample` on bit-packed binary maps (no gradient: GT masks only)."""
    lib = _lib.load()
    _need_gpu(src_index, coords, coord_index)
    coords = coords.float().contiguous()
    g, p = int(src_index.shape[0]), int(coords.shape[1])
    out = torch.empty((g, p), dtype=torch.float32, device=coords.device)
    _lib.WORK_HINT['point_sample'] = (int(pm.words.shape[0]), int(coords.shape[0]))
    rc = lib.mbv_point_sample_packed_fwd(_ptr(pm.words), _ptr(src_index), _ptr(coords), _ptr(coord_index), g, p, pm.h,
                                         pm.w, _ptr(out), _stream())
    check(rc, 'mbv_point_sample_packed_fwd')
    return out


# --------------------------------------------------------------------------------------
# K9 batched Hungarian assignment
# --------------------------------------------------------------------------------------
@torch.no_grad()
def hungarian(cost: torch.Tensor, out: Optional[torch.Tensor] = None,
              real_cols: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cost (N, R, C) f32 on the device → (N, R) int32: column assigned to each row (min total cost), -1 for
    rows left out when R > C.  No host synchronisation (K9, include/maskbev_hip.h).
    ``real_cols`` (N,) int32 on the device: columns ``real_cols[n]`` … C-1 of problem n are identical padding (the
    dataset's zero-padded instance list) — the equivalent rectangular problem of the real columns is solved instead
    (R <= C <= 320; same optimum, same real pairs when it is unique)."""
    lib = _lib.load()
    _need_gpu(cost)
    cost = cost.to(torch.float32).contiguous()
    n, r, c = cost.shape
    if out is None:
        out = torch.empty((n, r), dtype=torch.int32, device=cost.device)
    if real_cols is not None and r <= c <= 320:
        real_cols = real_cols.to(torch.int32).contiguous()
        if real_cols.numel() != n or not real_cols.is_cuda:
            raise MaskBevHipError('hungarian: real_cols must be a device tensor with one entry per problem')
        check(lib.mbv_hungarian_padded(_ptr(cost), n, r, c, _ptr(real_cols), _ptr(out), _stream()), 'mbv_hungarian_padded')
        return out
    out.fill_(-1)
    if max(r, c) > 128 and r > c:         # wide problems are solved from global memory in (rows <= cols) orientation
        cost_t = cost.transpose(1, 2).contiguous()
        rc = lib.mbv_hungarian_wide_t(_ptr(cost_t), n, r, c, _ptr(out), _stream())
        check(rc, 'mbv_hungarian_wide_t')
        return out
    rc = lib.mbv_hungarian(_ptr(cost), n, r, c, _ptr(out), _stream())
    check(rc, 'mbv_hungarian')
    return out


# --------------------------------------------------------------------------------------
# K10 importance sampling: the k most uncertain points of each row
# --------------------------------------------------------------------------------------
@torch.no_grad()
def select_uncertain_points(logits: torch.Tensor, coords: torch.Tensor, k: int) -> torch.Tensor:
    """logits (R, n) sampled mask logits, coords (R, n, 2) → (R, k, 2): coordinates of the k points with the
    smallest |logit| per row (radix select + ordered compaction, K10); same set as ``topk(-|logits|, k)``."""
    lib = _lib.load()
    _need_gpu(logits, coords)
    logits, coords = logits.float().contiguous(), coords.float().contiguous()
    r, n = logits.shape
    out = torch.empty((r, k, 2), dtype=torch.float32, device=logits.device)
    rc = lib.mbv_select_uncertain_points(_ptr(logits), _ptr(coords), r, n, int(k), _ptr(out), _stream())
    check(rc, 'mbv_select_uncertain_points')
    return out


@torch.no_grad()
def uniform_points(seed: torch.Tensor, rows: int, n: int) -> torch.Tensor:
    """(rows, n, 2) f32 uniform points in [0, 1): the counter-based generator of the fused importance sampling,
    written out (``seed``: device int64 tensor with one element)."""
    lib = _lib.load()
    _need_gpu(seed)
    if seed.dtype != torch.int64 or seed.numel() != 1:
        raise MaskBevHipError('uniform_points: seed must be one device int64')
    out = torch.empty((rows, n, 2), dtype=torch.float32, device=seed.device)
    for r0 in range(0, rows, 65535):                      # grid.y limit
        r1 = min(rows, r0 + 65535)
        if r0 == 0 and r1 == rows:
            check(lib.mbv_uniform_points(_ptr(seed), rows, n, _ptr(out), _stream()), 'mbv_uniform_points')
        else:
            raise MaskBevHipError('uniform_points: more than 65 535 rows')
    return out


@torch.no_grad()
def sample_select_uncertain(src: torch.Tensor, src_index: torch.Tensor, coords: Optional[torch.Tensor], k: int,
                            rand_coords: Optional[torch.Tensor] = None, seed: Optional[torch.Tensor] = None,
                            num_candidates: Optional[int] = None) -> torch.Tensor:
    """Importance sampling of the mask loss in one launch (fused K8 + K10): for row r, sample n candidate points
    from the map ``src[src_index[r]]`` (H, W), keep the k with the smallest |logit|, append ``rand_coords[r]``.
    The candidates are either ``coords`` (R, n, 2) or — ``coords=None`` — generated inside the kernel from the
    device int64 ``seed`` (``num_candidates`` per row; equal to ``uniform_points(seed, R, n)``).  Returns
    (R, k + n_rand, 2).  Falls back to the two-kernel form for maps larger than the 64 KB LDS tile, more than
    40 960 candidates or more than 16 384 selected points per row."""
    lib = _lib.load()
    _need_gpu(src, src_index, coords, rand_coords, seed)
    if (coords is None) == (seed is None):
        raise MaskBevHipError('sample_select_uncertain: give either coords or seed')
    src = src.float().contiguous()
    src_index = src_index.to(torch.int32).contiguous()
    r = src_index.shape[0]
    n = coords.shape[1] if coords is not None else int(num_candidates)
    h, w = src.shape[-2:]
    n_rand = 0 if rand_coords is None else rand_coords.shape[1]
    if h * w > 16384 or n > 40960 or k > 16384:
        if coords is None:
            coords = uniform_points(seed, r, n)
        coords = coords.float().contiguous()
        rows = torch.arange(r, device=src.device, dtype=torch.int32)
        sel = select_uncertain_points(point_sample(src, src_index, coords, rows), coords, k)
        return sel if rand_coords is None else torch.cat((sel, rand_coords.float()), dim=1).contiguous()
    if coords is not None:
        coords = coords.float().contiguous()
    if rand_coords is not None:
        rand_coords = rand_coords.float().contiguous()
    out = torch.empty((r, k + n_rand, 2), dtype=torch.float32, device=src.device)
    rc = lib.mbv_sample_select_uncertain(_ptr(src), _ptr(src_index), _ptr(coords), _ptr(seed), r, n, int(k), h, w,
                                         _ptr(rand_coords), n_rand, _ptr(out), _stream())
    check(rc, 'mbv_sample_select_uncertain')
    return out


# --------------------------------------------------------------------------------------
# K13 row sums of the point-sampled dice / BCE losses
# --------------------------------------------------------------------------------------
class _MaskLossRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        lib = _lib.load()
        _need_gpu(logits, targets)
        x = logits.float().contiguous()
        t = targets.float().contiguous()
        if x.shape != t.shape or x.dim() != 2:
            raise MaskBevHipError('mask_loss_rows: logits and targets must both be (rows, points)')
        out = torch.empty((x.shape[0], 4), dtype=torch.float32, device=x.device)
        check(lib.mbv_mask_loss_rows_fwd(_ptr(x), _ptr(t), x.shape[0], x.shape[1], _ptr(out), _stream()),
              'mbv_mask_loss_rows_fwd')
        ctx.save_for_backward(x, t)
        ctx.in_dtype = logits.dtype
        return out

    @staticmethod
    def backward(ctx, grad_sums):
        lib = _lib.load()
        x, t = ctx.saved_tensors
        g = grad_sums.float().contiguous()
        dx = torch.empty_like(x)
        check(lib.mbv_mask_loss_rows_bwd(_ptr(x), _ptr(t), _ptr(g), x.shape[0], x.shape[1], _ptr(dx), _stream()),
              'mbv_mask_loss_rows_bwd')
        return dx.to(ctx.in_dtype), None


def mask_loss_rows(logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """(rows, points) logits / targets → (rows, 4) f32 [Σ σ(x)·t, Σ σ(x), Σ t, Σ bce_with_logits(x, t)] in one pass
    (K13); differentiable w.r.t. the logits."""
    return _MaskLossRows.apply(logits, targets)


class _MaskDiceBce(torch.autograd.Function):
    """Dice and BCE losses of D decoder outputs from the point-sampled logits in one node: K13's row sums, then
    ``dice = (2 Σσt + 1) / (Σσ + Σt + 1)``, ``loss_dice[i] = c_dice Σ_rows (1 - dice)``, ``loss_mask[i] = c_mask Σ_rows Σbce``
    (mmdet DiceLoss / CrossEntropyLoss(use_sigmoid) as configured at mask2former_head.py:96-110, reduced per decoder
    output).  Written as ops autograd spends ≈ 30 launches on the backward of this algebra over 4 000-element tensors
    (slice gradients materialise zeros + copies); here the gradient of the four sums is assembled analytically."""

    @staticmethod
    def forward(ctx, logits, targets, d, c_dice, c_mask):
        lib = _lib.load()
        _need_gpu(logits, targets)
        x = logits.float().contiguous()
        t = targets.float().contiguous()
        if x.shape != t.shape or x.dim() != 2 or x.shape[0] % d:
            raise MaskBevHipError('mask_dice_bce: (D * g, points) logits and targets')
        rows = x.shape[0]
        sums = torch.empty((rows, 4), dtype=torch.float32, device=x.device)
        check(lib.mbv_mask_loss_rows_fwd(_ptr(x), _ptr(t), rows, x.shape[1], _ptr(sums), _stream()),
              'mbv_mask_loss_rows_fwd')
        ctx.consts = (d, c_dice, c_mask)
        ctx.in_dtype = logits.dtype
        ctx.fused = not torch.is_tensor(c_dice) and not torch.is_tensor(c_mask)
        if ctx.fused:
            # plain-float constants (the usual case: avg_factor = B * Q is a host constant): the algebra on the sums is ONE
            # launch, which also leaves the per-row gradient coefficients the backward kernel scales on the fly
            out = torch.empty((2, d), dtype=torch.float32, device=x.device)
            coef = torch.empty((rows, 3), dtype=torch.float32, device=x.device)
            check(lib.mbv_dice_bce_reduce(_ptr(sums), rows, d, float(c_dice), float(c_mask), _ptr(out[0]), _ptr(out[1]),
                                          _ptr(coef), _stream()), 'mbv_dice_bce_reduce')
            ctx.save_for_backward(x, t, coef)
            return out[0], out[1]
        den = sums[:, 1] + sums[:, 2] + 1.0
        dice = (2.0 * sums[:, 0] + 1.0) / den
        loss_dice = (1.0 - dice).view(d, rows // d).sum(1) * c_dice
        loss_mask = sums[:, 3].reshape(d, rows // d).sum(1) * c_mask
        ctx.save_for_backward(x, t, den, dice)
        return loss_dice, loss_mask

    @staticmethod
    def backward(ctx, g_dice, g_mask):
        lib = _lib.load()
        d, c_dice, c_mask = ctx.consts
        if ctx.fused:
            x, t, coef = ctx.saved_tensors
            rows = x.shape[0]

            def vec(g):          # (pointer holder, element stride) of an upstream (D,) gradient: expanded scalars stay as they are
                if g is None:
                    return None, 0
                g = g if g.dtype == torch.float32 else g.float()
                if g.dim() != 1 or g.stride(0) not in (0, 1):
                    g = g.contiguous().view(-1)
                return g, int(g.stride(0))
            gd, sd = vec(g_dice)
            gm, sm = vec(g_mask)
            dx = torch.empty_like(x)
            check(lib.mbv_mask_loss_rows_bwd_coef(_ptr(x), _ptr(t), _ptr(coef), _ptr(gd), sd, _ptr(gm), sm, rows, d, x.shape[1],
                                                  _ptr(dx), _stream()), 'mbv_mask_loss_rows_bwd_coef')
            return dx.to(ctx.in_dtype), None, None, None, None
        x, t, den, dice = ctx.saved_tensors
        rows = x.shape[0]
        g = rows // d
        zero = None
        if g_dice is None or g_mask is None:
            zero = torch.zeros(d, dtype=torch.float32, device=x.device)
        gd = ((g_dice if g_dice is not None else zero).float() * c_dice).view(d, 1).expand(d, g).reshape(rows)
        gm = ((g_mask if g_mask is not None else zero).float() * c_mask).view(d, 1).expand(d, g).reshape(rows)
        r = gd / den
        g_s12 = r * dice
        grad_sums = torch.stack((r * -2.0, g_s12, g_s12, gm), 1).contiguous()
        dx = torch.empty_like(x)
        check(lib.mbv_mask_loss_rows_bwd(_ptr(x), _ptr(t), _ptr(grad_sums), rows, x.shape[1], _ptr(dx), _stream()),
              'mbv_mask_loss_rows_bwd')
        return dx.to(ctx.in_dtype), None, None, None, None


def mask_dice_bce(logits: torch.Tensor, targets: torch.Tensor, d: int, c_dice, c_mask):
    """(D * g, points) sampled logits / targets → (loss_dice (D,), loss_mask (D,)); ``c_dice`` / ``c_mask``: the loss
    weights over their averaging factors (floats or 0-dim device tensors that need no gradient)."""
    return _MaskDiceBce.apply(logits, targets, int(d), c_dice, c_mask)


@torch.no_grad()
def match_cost_terms(logits: torch.Tensor, ones_row: bool = False):
    """logits (G, Q, P) f32 sampled mask logits → (terms (G, 3Q [+ 1], P) f32 = [softplus(-x); softplus(x); sigmoid(x)
    [; ones]] stacked along the query axis, row_sums (G, Q, 2) = [Σ softplus(x), Σ sigmoid(x)]) in one pass (K13)."""
    lib = _lib.load()
    _need_gpu(logits)
    x = logits.float().contiguous()
    g, q, p = x.shape
    terms = torch.empty((g, 3 * q + (1 if ones_row else 0), p), dtype=torch.float32, device=x.device)
    sums = torch.empty((g, q, 2), dtype=torch.float32, device=x.device)
    check(lib.mbv_match_cost_terms(_ptr(x), g, q, p, 1 if ones_row else 0, _ptr(terms), _ptr(sums), _stream()),
          'mbv_match_cost_terms')
    return terms, sums


@torch.no_grad()
def match_cost(cls: torch.Tensor, labels_gt: torch.Tensor, prod: torch.Tensor, sums: torch.Tensor, num_points: int):
    """The (D*B, Q, G) matching costs from the products of :func:`match_cost_terms` (``ones_row=True``) with the sampled
    ground truth: cls (D, B, Q, K+1) f32 logits, labels_gt (B, G) i64, prod (D*B, 3Q + 1, G) — one launch (K13)."""
    lib = _lib.load()
    d, b, q, k1 = cls.shape
    g = int(labels_gt.shape[1])
    cls, labels_gt, prod, sums = cls.float().contiguous(), labels_gt.contiguous(), prod.contiguous(), sums.contiguous()
    _need_gpu(cls, labels_gt, prod, sums)
    if tuple(prod.shape) != (d * b, 3 * q + 1, g) or labels_gt.dtype != torch.int64:
        raise MaskBevHipError('match_cost: prod (D*B, 3Q+1, G) and int64 labels expected')
    cost = torch.empty((d * b, q, g), dtype=torch.float32, device=cls.device)
    check(lib.mbv_match_cost(_ptr(cls), _ptr(labels_gt), _ptr(prod), _ptr(sums), d * b, q, g, k1, b, int(num_points),
                             _ptr(cost), _stream()), 'mbv_match_cost')
    return cost


def match_products_supported(queries: int, targets: int, points: int) -> bool:
    return bool(_lib.load().mbv_match_products_supported(int(queries), int(targets), int(points)))


@torch.no_grad()
def match_products(logits: torch.Tensor, targets: torch.Tensor, splits: Optional[int] = None):
    """Sampled mask logits (N, Q, P) f32 and sampled ground truth (N, G, P) f32 → the sliced products (N, S, 2Q + 1, G + 1) =
    [x ; sigmoid(x) ; 1] · [t ; 1]ᵀ and softplus sums (N, S, Q) of K13c: no term planes, no library GEMM.  S slices of the
    points per group, by default ≈ two workgroups per CU over all groups."""
    lib = _lib.load()
    x, t = logits.float().contiguous(), targets.float().contiguous()
    _need_gpu(x, t)
    n, q, p = x.shape
    g = int(t.shape[1])
    if tuple(t.shape) != (n, g, p):
        raise MaskBevHipError('match_products: logits (N, Q, P) and targets (N, G, P) expected')
    chunks = (p + 31) // 32
    if splits is None:
        splits = max(1, min(chunks, 512 // max(n, 1)))
    prod = torch.empty((n, splits, 2 * q + 1, g + 1), dtype=torch.float32, device=x.device)
    neg = torch.empty((n, splits, q), dtype=torch.float32, device=x.device)
    check(lib.mbv_match_products(_ptr(x), _ptr(t), n, q, g, p, int(splits), _ptr(prod), _ptr(neg), _stream()),
          'mbv_match_products')
    return prod, neg


@torch.no_grad()
def match_cost_split(cls: torch.Tensor, labels_gt: torch.Tensor, prod: torch.Tensor, neg: torch.Tensor, num_points: int):
    """The (D*B, Q, G) matching costs from :func:`match_products`' slices: cls (D, B, Q, K+1) f32, labels_gt (B, G) i64."""
    lib = _lib.load()
    d, b, q, k1 = cls.shape
    g = int(labels_gt.shape[1])
    cls, labels_gt = cls.float().contiguous(), labels_gt.contiguous()
    _need_gpu(cls, labels_gt, prod, neg)
    splits = int(prod.shape[1])
    if (tuple(prod.shape) != (d * b, splits, 2 * q + 1, g + 1) or tuple(neg.shape) != (d * b, splits, q)
            or labels_gt.dtype != torch.int64 or not prod.is_contiguous() or not neg.is_contiguous()):
        raise MaskBevHipError('match_cost_split: prod (D*B, S, 2Q+1, G+1), neg (D*B, S, Q) and int64 labels expected')
    cost = torch.empty((d * b, q, g), dtype=torch.float32, device=cls.device)
    check(lib.mbv_match_cost_split(_ptr(cls), _ptr(labels_gt), _ptr(prod), _ptr(neg), d * b, q, g, k1, b, int(num_points),
                                   splits, _ptr(cost), _stream()), 'mbv_match_cost_split')
    return cost


class _ClsLoss(torch.autograd.Function):
    """Class-weighted cross entropy of all decoder outputs against the assignment, one launch each way (K13)."""

    @staticmethod
    def forward(ctx, cls, assigned, labels_gt, class_weight, loss_weight, eps):
        lib = _lib.load()
        d, b, q, k1 = cls.shape
        g = int(labels_gt.shape[1])
        x = cls.float().contiguous()
        assigned = assigned.to(torch.int32).contiguous()
        labels_gt, class_weight = labels_gt.contiguous(), class_weight.float().contiguous()
        _need_gpu(x, assigned, labels_gt, class_weight)
        loss = torch.empty(d, dtype=torch.float32, device=x.device)
        wsum = torch.empty(d, dtype=torch.float32, device=x.device)
        check(lib.mbv_cls_loss_fwd(_ptr(x), _ptr(assigned), _ptr(labels_gt), _ptr(class_weight), d, b, q, g, k1,
                                   float(loss_weight), float(eps), _ptr(loss), _ptr(wsum), _stream()), 'mbv_cls_loss_fwd')
        ctx.save_for_backward(x, assigned, labels_gt, class_weight, wsum)
        ctx.meta = (d, b, q, g, k1, float(loss_weight), float(eps), cls.dtype)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        lib = _lib.load()
        x, assigned, labels_gt, class_weight, wsum = ctx.saved_tensors
        d, b, q, g, k1, lw, eps, dt = ctx.meta
        dx = torch.empty_like(x)
        check(lib.mbv_cls_loss_bwd(_ptr(x), _ptr(assigned), _ptr(labels_gt), _ptr(class_weight), _ptr(wsum),
                                   _ptr(g_loss.float().contiguous()), d, b, q, g, k1, lw, eps, _ptr(dx), _stream()),
              'mbv_cls_loss_bwd')
        return dx.to(dt), None, None, None, None, None


def cls_loss(cls: torch.Tensor, assigned: torch.Tensor, labels_gt: torch.Tensor, class_weight: torch.Tensor,
             loss_weight: float, eps: float) -> torch.Tensor:
    """(D,) classification losses: cls (D, B, Q, K+1), assigned (D, B, Q) i32 (ground-truth column or -1), labels_gt
    (B, G) i64, class_weight (K+1,) — mmdet CrossEntropyLoss(class_weight) with avg_factor = Σ class weights of the targets."""
    return _ClsLoss.apply(cls, assigned, labels_gt, class_weight, loss_weight, eps)


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
