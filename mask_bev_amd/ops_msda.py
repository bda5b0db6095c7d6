"""K5 multi-scale deformable attention and K16 its query-side preparation (mmcv MultiScaleDeformableAttention as configured
at mask_bev_panoptic_head.py:119-146)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, switches
from ._lib import MaskBevHipError, check
from .ops_core import *          # noqa: F401,F403
from .ops_records import *          # noqa: F401,F403
from .ops_gemm import *          # noqa: F401,F403


# --------------------------------------------------------------------------------------
# K5 multi-scale deformable attention
# --------------------------------------------------------------------------------------
class _MSDeformAttn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, value, shapes_t, level_start, loc, attn, shapes_host):
        lib = _lib.load()
        _need_gpu(value, shapes_t, level_start, loc, attn)
        ctx.shapes_host = shapes_host
        value, loc, attn = value.contiguous(), loc.contiguous(), attn.contiguous()
        b, nv, nh, d = value.shape
        _, nq, _, nl, npnt, _ = loc.shape
        out = torch.empty((b, nq, nh * d), dtype=torch.float32, device=value.device)
        rc = lib.mbv_ms_deform_attn_fwd(_ptr(value), _ptr(shapes_t), _ptr(level_start), _ptr(loc), _ptr(attn), b, nv,
                                        nh, d, nl, nq, npnt, _ptr(out), _stream())
        check(rc, 'mbv_ms_deform_attn_fwd')
        ctx.save_for_backward(value, shapes_t, level_start, loc, attn)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_out):
        lib = _lib.load()
        value, shapes_t, level_start, loc, attn = ctx.saved_tensors
        b, nv, nh, d = value.shape
        _, nq, _, nl, npnt, _ = loc.shape
        grad_out = grad_out.to(torch.float32).contiguous()
        g_value = torch.empty_like(value)
        g_loc = torch.empty_like(loc)
        g_attn = torch.empty_like(attn)
        host = None
        if ctx.shapes_host is not None and len(ctx.shapes_host) == nl:       # banded LDS accumulation (K5)
            host = (ctypes.c_int64 * (2 * nl))(*[int(v) for hw in ctx.shapes_host for v in hw])
        _msda_backward(lib, grad_out, value, shapes_t, level_start, loc, attn, (b, nv, nh, d, nl, nq, npnt), host,
                       g_value, g_loc, g_attn)
        return g_value, None, None, g_loc, g_attn, None


_MSDA_SIDE = {}


def msda_value_packed_ok(dims, host) -> bool:
    """Whether d(value) of this shape can take the packed fixed-point form (mbv_ms_deform_attn_bwd_value_packed)."""
    b, nv, nh, d, nl, nq, npnt = dims
    return bool(host is not None and switches.get('msda_packed')
                and _lib.load().mbv_ms_deform_attn_bwd_value_packed_supported(d, nl, npnt, nq, host))


def _msda_backward(lib, g_out, value, shapes_t, level_start, loc, attn, dims, host, g_value, g_loc, g_attn,
                   packed_out=None):
    """``packed_out = (tensor, row stride in elements)``: d(value) goes there in the tensor's dtype through the packed
    fixed-point kernel (16-bit compute modes; the caller guarantees softmaxed weights) and ``g_value`` is not written."""
    if packed_out is not None:
        b, nv, nh, d, nl, nq, npnt = dims
        dst, ld = packed_out
        ws = _workspace(lib.mbv_ms_deform_attn_bwd_value_packed_workspace_bytes(b, nh, nl, nq), g_out.device)
        check(lib.mbv_ms_deform_attn_bwd_value_packed(_ptr(g_out), _ptr(loc), _ptr(attn), b, nv, nh, d, nl, nq, npnt, host,
                                                      _ptr(dst), _dt_flag(dst.dtype), int(ld), _ptr(ws), ws.numel(), _stream()),
              'mbv_ms_deform_attn_bwd_value_packed')
        # d(location), d(weight): gathers from the value map in its own dtype (16-bit in the 16-bit compute modes, f32 in the
        # fp32 mode) — no limit on a level's size, unlike part 2 of mbv_ms_deform_attn_bwd's split form
        check(lib.mbv_ms_deform_attn_bwd_locattn(_ptr(g_out), _ptr(value), _dt_flag(value.dtype), _ptr(shapes_t),
                                                 _ptr(level_start), _ptr(loc), _ptr(attn), b, nv, nh, d, nl, nq, npnt,
                                                 _ptr(g_loc), _ptr(g_attn), _stream()), 'mbv_ms_deform_attn_bwd_locattn')
        return
    _msda_backward_f64(lib, g_out, value, shapes_t, level_start, loc, attn, dims, host, g_value, g_loc, g_attn)


def _msda_backward_f64(lib, g_out, value, shapes_t, level_start, loc, attn, dims, host, g_value, g_loc, g_attn):
    """K5 backward.  The no-atomics form has two independent parts — d(value), bound by the LDS f64-atomic rate, and
    d(location) / d(weight), bound by L2 gathers.  `switches.msda_bwd_overlap` puts them on two streams; measured inside the
    HIP-graph step the fork / join edges cost more than the overlap returns (34.17 vs 33.88 ms per step), so the
    default is one stream.  The side stream only touches buffers that were allocated on the current stream and
    outlive the join."""
    import os
    b, nv, nh, d, nl, nq, npnt = dims
    args = (_ptr(g_out), _ptr(value), _ptr(shapes_t), _ptr(level_start), _ptr(loc), _ptr(attn), b, nv, nh, d, nl, nq,
            npnt, host, _ptr(g_value), _ptr(g_loc), _ptr(g_attn))
    split = host is not None and lib.mbv_ms_deform_attn_bwd_split(d, nl, host)
    if not split or not switches.get('msda_bwd_overlap'):
        check(lib.mbv_ms_deform_attn_bwd(*args, 3, _stream()), 'mbv_ms_deform_attn_bwd')
        return
    main = torch.cuda.current_stream()
    side = _MSDA_SIDE.get(g_out.device)
    if side is None:
        side = _MSDA_SIDE[g_out.device] = torch.cuda.Stream(device=g_out.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        check(lib.mbv_ms_deform_attn_bwd(*args, 2, _stream()), 'mbv_ms_deform_attn_bwd')      # d(location), d(weight)
    check(lib.mbv_ms_deform_attn_bwd(*args, 1, _stream()), 'mbv_ms_deform_attn_bwd')          # d(value)
    main.wait_stream(side)


class _MSDAPrepare(torch.autograd.Function):
    """K16: (offsets, logits) → (sampling locations, softmaxed weights), f32 out; gradients in the input dtype."""

    @staticmethod
    def forward(ctx, off, logits, ref, shapes_host):
        lib = _lib.load()
        _need_gpu(off, logits, ref)
        if off.dtype != logits.dtype or off.dtype not in _ACT_DTYPES:
            raise MaskBevHipError('msda_prepare: offsets and logits must share one of f32, bf16, fp16')
        off, logits = off.contiguous(), logits.contiguous()
        ref = ref.to(torch.float32).contiguous()
        b, n, h, l, p, _ = off.shape
        host = (ctypes.c_int64 * (2 * l))(*[int(v) for hw in shapes_host for v in hw])
        loc = torch.empty((b, n, h, l, p, 2), dtype=torch.float32, device=off.device)
        attn = torch.empty((b, n, h, l, p), dtype=torch.float32, device=off.device)
        check(lib.mbv_msda_prepare_fwd(_ptr(off), _ptr(logits), _dt_flag(off.dtype), _ptr(ref), host,
                                       b, n, h, l, p, _ptr(loc), _ptr(attn), _stream()), 'mbv_msda_prepare_fwd')
        ctx.save_for_backward(attn)
        ctx.meta = (host, off.dtype, (b, n, h, l, p), logits.shape)
        return loc, attn

    @staticmethod
    def backward(ctx, g_loc, g_attn):
        lib = _lib.load()
        attn, = ctx.saved_tensors
        host, dt, (b, n, h, l, p), lshape = ctx.meta
        g_loc = g_loc.to(torch.float32).contiguous()
        g_attn = g_attn.to(torch.float32).contiguous()
        g_off = torch.empty((b, n, h, l, p, 2), dtype=dt, device=attn.device)
        g_logit = torch.empty(lshape, dtype=dt, device=attn.device)
        check(lib.mbv_msda_prepare_bwd(_ptr(g_loc), _ptr(g_attn), _ptr(attn), host, b, n, h, l, p,
                                       _dt_flag(dt), _ptr(g_off), _ptr(g_logit), _stream()),
              'mbv_msda_prepare_bwd')
        return g_off, g_logit, None, None


def msda_prepare_supported(num_levels: int, num_points: int) -> bool:
    return bool(_lib.load().mbv_msda_prepare_supported(num_levels, num_points))


def msda_prepare(offsets: torch.Tensor, logits: torch.Tensor, reference_points: torch.Tensor, spatial_shapes):
    """offsets (B, Nq, H, L, P, 2), logits (B, Nq, H, L*P) (one dtype: f32, bf16 or fp16), reference_points (Nq, 2) in
    [0, 1], spatial_shapes [(h, w)] * L  →  sampling locations (B, Nq, H, L, P, 2) f32 and attention weights
    (B, Nq, H, L, P) f32 (softmax over L*P) — K16, include/maskbev_hip.h."""
    host = tuple((int(h), int(w)) for h, w in spatial_shapes)
    return _MSDAPrepare.apply(offsets, logits, reference_points, host)


def ms_deform_attn(value: torch.Tensor, spatial_shapes, shapes_t: torch.Tensor, level_start: torch.Tensor,
                   sampling_locations: torch.Tensor, attention_weights: torch.Tensor) -> torch.Tensor:
    """value (B, N, H, D); sampling_locations (B, Nq, H, L, P, 2) in [0,1]; weights (B, Nq, H, L, P)
    → (B, Nq, H*D) f32.  Bilinear, zero padding, align_corners=False (K5, include/maskbev_hip.h)."""
    host = None if spatial_shapes is None else tuple((int(h), int(w)) for h, w in spatial_shapes)
    return _MSDeformAttn.apply(value.float(), shapes_t, level_start, sampling_locations.float(),
                               attention_weights.float(), host)


class PosGradShare:
    """d(pos) of the pixel decoder's encoder layers, taken once.  Every layer adds the same positional map to its query, so
    d(pos) = Σ_layers (Σ_batch G_l[:, E:]) · [Wo_l; Wa_l] — per layer a batch sum, a (N, 3HLP) x (3HLP, E) product and an
    accumulation into the running gradient.  With a share, layer l only stores its batch sum into column block l of one
    (N, layers · 3HLP) matrix; the layer whose backward runs LAST (index 0: the first of the forward) multiplies the whole
    matrix with the stacked weights — one product with a 6x longer contraction — and returns it as its d(pos); the others
    return none.  Valid because the layers form a chain: every layer's backward has run when layer 0's does (checked)."""

    def __init__(self, layers: int):
        self.layers = int(layers)
        self.weights = [None] * self.layers          # [Wo; Wa] of each layer in the compute dtype, (3HLP, E)
        self.sums = None                             # (N, layers * 3HLP)
        self.written = set()

    def store(self, index: int, block: torch.Tensor, weight: torch.Tensor):
        """block (B, N, 3HLP) strided view of G → its batch sum into column block ``index``."""
        b, n, w = block.shape
        if self.sums is None:
            self.sums = torch.empty((n, self.layers * w), dtype=block.dtype if block.dtype in _LO_DTYPES else torch.float32,
                                    device=block.device)
        torch.sum(block, 0, out=self.sums[:, index * w:(index + 1) * w])
        self.weights[index] = weight
        self.written.add(index)

    def finish(self) -> torch.Tensor:
        if len(self.written) != self.layers:
            raise MaskBevHipError(f'PosGradShare: {len(self.written)} of {self.layers} layers ran their backward')
        od = {} if self.sums.dtype == torch.float32 else dict(out_dtype=torch.float32)
        out = torch.mm(self.sums, torch.cat(self.weights, 0), **od)
        self.sums, self.weights, self.written = None, [None] * self.layers, set()
        return out


class _MSDAQuerySide(torch.autograd.Function):
    """The query side of the pixel decoder's deformable self-attention as ONE autograd node:

        value = value_proj(x);  q = x + pos;  off = sampling_offsets(q);  logits = attention_weights(q)
        loc, attn = K16(off, logits);  out = K5(value, loc, attn)                      (before output_proj)

    Forward is the same sequence of launches as the composed ops.  Backward assembles the gradients of the three
    projections side by side in one (B*N, E + 2HLP + HLP) matrix G — K5's value gradient cast into the first E columns,
    K16's backward writing the other two blocks in place (row strides) — so that d(x) is ONE data-gradient GEMM
    G · [Wv; Wo; Wa] with no casts or accumulation passes, the bias gradients one column-sum pass, and d(pos) a
    batch-sum of the offset / weight columns times [Wo; Wa].  mmcv MultiScaleDeformableAttention.forward
    (mask_bev_panoptic_head.py:127-136); replaces 3 GEMMs + 3 column sums + 9 element-wise launches per layer."""

    @staticmethod
    def forward(ctx, x, pos, ref, wv, bv, wo, bo, wa, ba, heads, levels, points, shapes_host, shapes_t, level_start,
                share=None, share_index=0, wcat=None):
        lib = _lib.load()
        ctx.share = (share, int(share_index))
        _need_gpu(x, pos, ref, wv, wo, wa)
        b, n, e = x.shape
        d = e // heads
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
        if dt not in _ACT_DTYPES:
            raise MaskBevHipError('msda_query_side supports f32, bf16 and fp16 compute')
        lo, la = heads * levels * points * 2, heads * levels * points
        with torch.autocast('cuda', enabled=False):
            qb = torch.empty(x.shape, dtype=dt, device=x.device)
            # [Wv; Wo; Wa] (E + 2HLP + HLP, E) in the compute dtype, once: the backward's single data-gradient GEMM reads it
            # whole, and its last two blocks make offsets and attention logits ONE projection of q here
            if wcat is None or wcat.dtype != dt or tuple(wcat.shape) != (e + lo + la, e):
                wcat = torch.cat([_compute_copy(wv, dt), _compute_copy(wo, dt), _compute_copy(wa, dt)], 0)
            wvc = wcat[:e]
            pos_rows = pos.numel() // e
            if (dt in _LO_DTYPES and x.dtype == torch.float32 and pos.dtype == torch.float32 and x.is_contiguous()
                    and pos.is_contiguous() and e % 4 == 0 and (b * n) % pos_rows == 0):
                xb = torch.empty(x.shape, dtype=dt, device=x.device)      # both 16-bit GEMM inputs in one pass over x
                check(lib.mbv_msda_query_inputs(_ptr(x), _ptr(pos), b * n, pos_rows, e, _dt_flag(dt), _ptr(xb), _ptr(qb),
                                                _stream()), 'mbv_msda_query_inputs')
                # the value map is consumed in f32 (K5): accumulate and store it in f32, no 16-bit round trip + cast.  K17 takes
                # the f32 bias in its epilogue (the library's addmm first copies the broadcast bias into the f32 result)
                bvf = bv.float().contiguous()
                lo_value = False
                if gemm16_policy() != 'none' and _gemm16_ok(xb.view(b * n, e), wvc) and bvf.data_ptr() % 16 == 0:
                    # The value map in the compute dtype (what this Linear's output IS under the reference's autocast): K5's
                    # forward and its location / weight gradient are bound by the bytes of their bilinear taps, 128 B per
                    # (tap, head) in f32.  Needs the packed value gradient (its f64 alternative wants an f32 map), head dim 32.
                    host_b = (ctypes.c_int64 * (2 * levels))(*[int(v) for hw in shapes_host for v in hw])
                    lo_value = bool(switches.get('msda_value_lowp') and d == 32 and (e + lo + la) % 2 == 0
                                    and msda_value_packed_ok((b, n, heads, d, levels, n, points), host_b))
                    value = gemm16_nt(xb.view(b * n, e), wvc, bvf,
                                      out_dtype=None if lo_value else torch.float32).view(b, n, e)
                else:
                    value = torch.addmm(bvf, xb.view(b * n, e), wvc.t(), out_dtype=torch.float32).view(b, n, e)
            else:
                xb = x.to(dt)
                torch.add(x, pos, out=qb)                 # the sum, stored in the compute dtype by the same launch
                if dt == torch.float32:
                    value = mm32_nt(xb.reshape(b * n, e), wvc, bv.contiguous()).view(b, n, e)
                else:
                    value = torch.nn.functional.linear(xb, wvc, _compute_copy(bv, dt)).float().contiguous()
            # [offsets | logits] = q . [Wo; Wa]^T without the biases: K16 adds them in f32 on load
            ol = mm32_nt(qb.view(b * n, e), wcat[e:]) if dt == torch.float32 else torch.mm(qb.view(b * n, e), wcat[e:].t())
        host = (ctypes.c_int64 * (2 * levels))(*[int(v) for hw in shapes_host for v in hw])
        ref32 = ref.to(torch.float32).contiguous()
        loc = torch.empty((b, n, heads, levels, points, 2), dtype=torch.float32, device=x.device)
        attn = torch.empty((b, n, heads, levels, points), dtype=torch.float32, device=x.device)
        esz = ol.element_size()
        check(lib.mbv_msda_prepare_fwd_ld(_ptr(ol), lo + la, ctypes.c_void_p(ol.data_ptr() + lo * esz), lo + la,
                                          _ptr(bo.float().contiguous()), _ptr(ba.float().contiguous()), _dt_flag(dt),
                                          _ptr(ref32), host, b, n, heads, levels, points, _ptr(loc), _ptr(attn), _stream()),
              'mbv_msda_prepare_fwd_ld')
        del ol
        out = torch.empty((b, n, e), dtype=torch.float32, device=x.device)
        check(lib.mbv_ms_deform_attn_fwd_v(_ptr(value), _dt_flag(value.dtype), _ptr(shapes_t), _ptr(level_start), _ptr(loc),
                                           _ptr(attn), b, n, heads, d, levels, n, points, _ptr(out), _stream()),
              'mbv_ms_deform_attn_fwd_v')
        ctx.save_for_backward(xb, qb, value, loc, attn, shapes_t, level_start, wcat)
        ctx.params = (wv, bv, wo, bo, wa, ba)
        ctx.meta = (heads, levels, points, host, tuple(shapes_host), dt, x.dtype, pos.dtype, tuple(pos.shape))
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        xb, qb, value, loc, attn, shapes_t, level_start, wcat = ctx.saved_tensors
        wv, bv, wo, bo, wa, ba = ctx.params
        heads, levels, points, host, shapes_host, dt, x_dtype, pos_dtype, pos_shape = ctx.meta
        b, n, e = xb.shape
        d = e // heads
        lo, la = heads * levels * points * 2, heads * levels * points
        width = e + lo + la
        t = b * n
        dev = xb.device
        g_out = g_out.to(torch.float32).contiguous()
        g_loc = torch.empty_like(loc)
        g_attn = torch.empty_like(attn)
        host_b = (ctypes.c_int64 * (2 * levels))(*[int(v) for hw in shapes_host for v in hw])
        g = torch.empty((t, width), dtype=dt, device=dev)                 # [d value | d offsets | d logits]
        dims = (b, n, heads, d, levels, n, points)
        # (fp32 compute: the packed form's 2^-30-of-the-bound fixed point is ~ 1e-6 of a typical sum — `switches.msda_packed_f32`)
        packed = ((dt in _LO_DTYPES or (dt == torch.float32 and switches.get('msda_packed_f32'))) and width % 2 == 0
                  and msda_value_packed_ok(dims, host_b))
        if value.dtype in _LO_DTYPES and not packed:
            # the forward stored the value map in 16 bits on the promise of the packed gradient; the f64 form reads the
            # map as f32 — were a switch flipped between the two halves it would read out of bounds
            raise MaskBevHipError('MSDA backward: the forward kept a 16-bit value map, which only the packed value gradient '
                                  'reads, and that path is unavailable now (switch changed between forward and backward?)')
        if packed:
            # 16-bit compute: K5's value gradient is accumulated in packed fixed point (the attention weights are
            # K16's softmax outputs) and stored straight into the first E columns of G in its dtype — no f32
            # d(value) tensor, no cast pass
            _msda_backward(lib, g_out, value, shapes_t, level_start, loc, attn, dims, host_b, None, g_loc, g_attn,
                           packed_out=(g, width))
        else:
            g_value = torch.empty_like(value)
            _msda_backward(lib, g_out, value, shapes_t, level_start, loc, attn, dims, host_b, g_value, g_loc, g_attn)
            g[:, :e].copy_(g_value.view(t, e))
        esz = g.element_size()
        check(lib.mbv_msda_prepare_bwd_ld(_ptr(g_loc), _ptr(g_attn), _ptr(attn), host, b, n, heads, levels, points,
                                          _dt_flag(dt), ctypes.c_void_p(g.data_ptr() + e * esz), width,
                                          ctypes.c_void_p(g.data_ptr() + (e + lo) * esz), width, _stream()),
              'mbv_msda_prepare_bwd_ld')
        od = {} if dt == torch.float32 else dict(out_dtype=torch.float32)
        gx = gpos = None
        if ctx.needs_input_grad[0]:
            gx = (mm32_nn(g, wcat) if dt == torch.float32 else torch.mm(g, wcat, **od)).view(b, n, e).to(x_dtype)
        if ctx.needs_input_grad[1]:                                       # pos is broadcast over the batch
            # (a 16-bit sum accumulates in f32 and rounds once on the way out: the same value as an f32 sum + cast, one launch)
            share, share_index = ctx.share
            if share is not None:
                share.store(share_index, g.view(b, n, width)[:, :, e:], wcat[e:])
                gpos = share.finish().view(1, n, e) if share_index == 0 else None
            else:
                gq_sum = g.view(b, n, width)[:, :, e:].sum(0) if dt in _LO_DTYPES else g.view(b, n, width)[:, :, e:].sum(0, dtype=torch.float32)
                gpos = torch.mm(gq_sum, wcat[e:], **od).view(1, n, e)
            if gpos is not None:
                if tuple(pos_shape) != (1, n, e):
                    gpos = gpos.sum_to_size(pos_shape) if len(pos_shape) == 3 else gpos.reshape(pos_shape)
                gpos = gpos.to(pos_dtype)
        x2, q2 = xb.view(t, e), qb.view(t, e)
        cols = ((0, e, x2), (e, e + lo, q2), (e + lo, width, q2))
        grads = [None] * 6
        # bias gradients = column sums of the three blocks of G.  Arena biases take theirs through the end-of-pass grouped
        # column-sum launch (one entry per block, row stride `width`): no zero fill, no column-sum launch of its own, no
        # accumulate launch — 25 us per layer
        bias_tmp = None
        arena_bias = [bool(ctx.needs_input_grad[4 + 2 * j] and getattr(bia, '_mbv_arena', False) and bia.grad is not None
                           and bia.grad.dtype == torch.float32 and bia.grad.is_contiguous())
                      for j, bia in enumerate((bv, bo, ba))]
        deferred = [False, False, False]
        for j, (c0, c1) in enumerate(((0, e), (e, e + lo), (e + lo, width))):
            if arena_bias[j]:
                deferred[j] = _defer_colsum(g, (bv, bo, ba)[j].grad, t, c1 - c0, width, offset=c0)
        if any(ctx.needs_input_grad[4 + 2 * j] and not deferred[j] for j in range(3)):
            bias_tmp = torch.zeros(width, dtype=torch.float32, device=dev)
            colsum_accum(g, bias_tmp)
        dst, src = [], []
        for j, ((c0, c1, inp), w, bia) in enumerate(zip(cols, (wv, wo, wa), (bv, bo, ba))):
            gj = g[:, c0:c1]                                              # column block: a GEMM operand with lda = width
            if ctx.needs_input_grad[3 + 2 * j]:
                if getattr(w, '_mbv_arena', False) and w.grad is not None and w.grad.dtype == torch.float32:
                    _wgrad_into(w.grad, gj, inp, persistent=True)
                    _fire_grad_hooks(w)
                else:
                    acc = torch.zeros(w.shape, dtype=torch.float32, device=dev)
                    _wgrad_into(acc, gj, inp)
                    grads[2 * j] = acc.to(w.dtype)
            if ctx.needs_input_grad[4 + 2 * j]:
                if deferred[j]:
                    _fire_grad_hooks(bia)
                elif getattr(bia, '_mbv_arena', False) and bia.grad is not None and bia.grad.dtype == torch.float32:
                    dst.append(bia.grad)
                    src.append(bias_tmp[c0:c1])
                else:
                    grads[2 * j + 1] = bias_tmp[c0:c1].to(bia.dtype)
        if dst:
            torch._foreach_add_(dst, src)
            for j, bia in enumerate((bv, bo, ba)):
                if not deferred[j]:
                    _fire_grad_hooks(bia)
        return (gx, gpos, None) + tuple(grads) + (None,) * 9


@torch.no_grad()
def msda_weight_stacks(attns, dtype) -> Optional[List[torch.Tensor]]:
    """[Wv; Wo; Wa] (E + 2HLP + HLP, E) in ``dtype`` for every deformable-attention module of a chain, all pieces copied by
    ONE launch (mbv_copy_group) instead of one concatenation per layer; None when that does not apply."""
    if not attns or not attns[0].value_proj.weight.is_cuda:
        return None
    dev = attns[0].value_proj.weight.device
    src, dst, nb, outs, keep = [], [], [], [], []
    for a in attns:
        parts = [_compute_copy(m.weight, dtype) for m in (a.value_proj, a.sampling_offsets, a.attention_weights)]
        keep.append(parts)              # per-call casts (parameters outside an arena) must outlive the launch below
        if any(not p.is_contiguous() or p.dtype != dtype for p in parts):
            return None
        rows, e = sum(int(p.shape[0]) for p in parts), int(parts[0].shape[1])
        out = torch.empty((rows, e), dtype=dtype, device=dev)
        r0 = 0
        for p in parts:
            src.append(p.data_ptr())
            dst.append(out[r0:r0 + p.shape[0]].data_ptr())
            nb.append(p.numel() * p.element_size())
            r0 += int(p.shape[0])
        outs.append(out)
    k = len(src)
    check(_lib.load().mbv_copy_group((ctypes.c_void_p * k)(*src), (ctypes.c_void_p * k)(*dst), (ctypes.c_int64 * k)(*nb), k,
                                     _stream()), 'mbv_copy_group')
    del keep
    return outs


def msda_query_side(x, pos, ref, value_proj, sampling_offsets, attention_weights, heads, levels, points, spatial_shapes,
                    shapes_t, level_start, pos_share=None, pos_share_index=0, wcat=None):
    """x (B, N, E) f32, pos (1, N, E) → the deformable-attention output (B, N, E) f32 before ``output_proj``; the three
    ``nn.Linear`` modules supply the parameters (checkpoint keys unchanged).  See :class:`_MSDAQuerySide`."""
    host = tuple((int(h), int(w)) for h, w in spatial_shapes)
    return _MSDAQuerySide.apply(x, pos, ref, value_proj.weight, value_proj.bias, sampling_offsets.weight,
                                sampling_offsets.bias, attention_weights.weight, attention_weights.bias, heads, levels,
                                points, host, shapes_t, level_start, pos_share, pos_share_index, wcat)


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
