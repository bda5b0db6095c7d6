"""K4 shifted-window attention (swin.py:80-118,179-284), K6 the decoder's masked multi-head attention with shared key / value
projections, K7 per-query mask logits + the next layer's attention mask (mask2former_head.py:428-472,535-560)."""
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
# K4 shifted-window attention
# --------------------------------------------------------------------------------------
class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, qkv_bias, bias_table, num_heads, ws, shift, full_bias_grad=False):
        lib = _lib.load()
        _need_gpu(qkv, qkv_bias, bias_table)
        ctx.full_bias_grad = full_bias_grad
        if qkv.dtype not in _ACT_DTYPES:
            raise MaskBevHipError(f'window_attention supports f32, bf16 and fp16 qkv, got {qkv.dtype}')
        qkv = qkv.contiguous()
        b, h, w, c3 = qkv.shape
        c = c3 // 3
        bias32 = qkv_bias.detach().to(torch.float32).contiguous()
        table32 = bias_table.detach().to(torch.float32).contiguous()
        out = torch.empty((b, h, w, c), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((lib.mbv_window_attn_lse_elems(b, h, w, num_heads, ws),), dtype=torch.float32,
                          device=qkv.device)
        is_bf16 = _dt_flag(qkv.dtype)
        ctx.amax_qkv = None
        if (qkv.dtype == torch.float32 and switches.get('k4_split') and qkv.data_ptr() % 16 == 0 and c % 4 == 0
                and lib.mbv_window_attn_split_supported(c, num_heads, ws)):
            # fp32 compute: the products on the 16-bit matrix pipe from IEEE-half pairs (K20's arithmetic inside K4); the
            # tensor's scale from the record its producer left (the qkv projection's epilogue), else one pass over it
            q2 = qkv.view(-1, c3)
            rec = amax_hint_get(qkv) if switches.get('amax_hints') else None
            ctx.amax_qkv = rec if rec is not None else f32_absmax([q2])
            AMAX_VERIFY.check(qkv, ctx.amax_qkv, 'window_attn_split_fwd qkv')
            check(lib.mbv_window_attn_split_fwd(_ptr(qkv), _ptr(bias32), _ptr(table32), b, h, w, c, num_heads, ws, shift,
                                                _amax_ptr(ctx.amax_qkv, 0), _ptr(out), _ptr(lse), _stream()),
                  'mbv_window_attn_split_fwd')
        else:
            rc = lib.mbv_window_attn_fwd(_ptr(qkv), _ptr(bias32), _ptr(table32), is_bf16, b, h, w, c, num_heads, ws, shift,
                                         _ptr(out), _ptr(lse), _stream())
            check(rc, 'mbv_window_attn_fwd')
        ctx.save_for_backward(qkv, bias32, table32, out, lse)
        ctx.cfg = (num_heads, ws, shift, qkv_bias.dtype, bias_table.dtype)
        ctx.params = (qkv_bias, bias_table)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        qkv, bias32, table32, out, lse = ctx.saved_tensors
        num_heads, ws, shift, bias_dtype, table_dtype = ctx.cfg
        b, h, w, c3 = qkv.shape
        c = c3 // 3
        grad_out = grad_out.to(qkv.dtype).contiguous()
        g_qkv = torch.empty_like(qkv)
        is_bf16 = _dt_flag(qkv.dtype)
        pb, pt = ctx.params
        direct = all(getattr(p, '_mbv_arena', False) and p.grad is not None and p.grad.dtype == torch.float32
                     and p.grad.is_contiguous() for p in (pb, pt))
        if direct:          # the kernel's atomics add straight into the arena gradients: no fill, no add_ afterwards
            g_table, g_bias = pt.grad, pb.grad
        else:               # the two small f32 gradients share one allocation: the library clears them with one fill
            small = torch.empty(table32.numel() + bias32.numel(), dtype=torch.float32, device=qkv.device)
            g_table = small[:table32.numel()].view(table32.shape)
            g_bias = small[table32.numel():]
        if ctx.amax_qkv is not None and grad_out.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0:
            hints = bool(switches.get('amax_hints'))
            go2 = grad_out.view(-1, c)
            rec_do = amax_hint_get(grad_out) if hints else None
            if rec_do is None:
                rec_do = f32_absmax([go2])
            rec_out = amax_record(qkv.device) if hints else None
            AMAX_VERIFY.check(qkv, ctx.amax_qkv, 'window_attn_split_bwd qkv')
            AMAX_VERIFY.check(grad_out, rec_do, 'window_attn_split_bwd d(out)')
            check(lib.mbv_window_attn_split_bwd(_ptr(qkv), _ptr(bias32), _ptr(table32), _ptr(out), _ptr(grad_out), _ptr(lse),
                                                b, h, w, c, num_heads, ws, shift, _amax_ptr(ctx.amax_qkv, 0),
                                                _amax_ptr(rec_do, 0), _ptr(g_qkv), _ptr(g_table), _ptr(g_bias),
                                                1 if ctx.full_bias_grad else 0, 1 if direct else 0, _ptr(rec_out), _stream()),
                  'mbv_window_attn_split_bwd')
            amax_hint_set(g_qkv, rec_out)
        else:
            rc = lib.mbv_window_attn_bwd(_ptr(qkv), _ptr(bias32), _ptr(table32), _ptr(out), _ptr(grad_out), _ptr(lse),
                                         is_bf16, b, h, w, c, num_heads, ws, shift, _ptr(g_qkv), _ptr(g_table),
                                         _ptr(g_bias), 1 if ctx.full_bias_grad else 0, 1 if direct else 0, _stream())
            check(rc, 'mbv_window_attn_bwd')
        if direct:
            _fire_grad_hooks(pb)
            _fire_grad_hooks(pt)
            return g_qkv, None, None, None, None, None, None
        return g_qkv, g_bias.to(bias_dtype), g_table.to(table_dtype), None, None, None, None


def window_attention(qkv: torch.Tensor, qkv_bias: torch.Tensor, bias_table: torch.Tensor, num_heads: int, ws: int,
                     shift: int, full_bias_grad: bool = False) -> torch.Tensor:
    """Shifted-window multi-head attention on a channels-last map (K4, include/maskbev_hip.h).

    qkv (B, H, W, 3C) is the fused projection of the *un-padded* tokens; tokens that the reference pads in
    (swin.py:185-188: zeros after LayerNorm) have qkv == bias, which the kernel substitutes while staging.
    Returns (B, H, W, C) (before the output projection), same dtype as qkv (f32 or bf16).
    ``full_bias_grad``: the gradient returned for ``qkv_bias`` is the WHOLE bias gradient of the qkv projection
    (column sums of d(qkv) over all tokens) — run that Linear with ``skip_bias_grad=True``."""
    out = _WindowAttention.apply(qkv, qkv_bias, bias_table, num_heads, ws, shift, full_bias_grad)
    # every output element is a convex combination of v elements: the absmax record of qkv bounds the attention output
    amax_hint_set(out, amax_hint_get(qkv))
    return out


# --------------------------------------------------------------------------------------
# K6 decoder multi-head attention
# --------------------------------------------------------------------------------------
def _k6_split(dt, heads: int, d: int, ld: int, *tensors) -> bool:
    """fp32 compute: K6's products on the 16-bit matrix pipe from IEEE-half pairs (``switches.k6_split``) for f32 tensors whose
    shapes and alignment the split mode takes."""
    return bool(dt == torch.float32 and switches.get('k6_split')
                and all(t is None or t.data_ptr() % 16 == 0 for t in tensors)
                and _lib.load().mbv_attn_split_supported(heads, d, ld))


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, blocked, num_heads):
        lib = _lib.load()
        _need_gpu(q, k, v, blocked)
        dt = k.dtype                     # the (large) key / value side decides; q (B*Q rows) is cast to it
        if dt not in _ACT_DTYPES:
            raise MaskBevHipError(f'attention supports f32, bf16 and fp16, got {dt}')
        ctx.in_dtypes = (q.dtype, k.dtype, v.dtype)
        q, k, v = q.to(dt).contiguous(), k.contiguous(), v.to(dt).contiguous()
        b, nq, e = q.shape
        nl = k.shape[1]
        d = e // num_heads
        mask = None
        if blocked is not None:
            mask = blocked.reshape(b, nq, nl).contiguous()
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        out = torch.empty_like(q)
        lse = torch.empty((b, num_heads, nq), dtype=torch.float32, device=q.device)
        ws = _workspace(lib.mbv_attn_workspace_bytes(b, nq, nl, num_heads, d), q.device)
        if _k6_split(dt, num_heads, d, e, q, k, v, out):
            check(lib.mbv_attn_split_fwd_ld(_ptr(q), _ptr(k), _ptr(v), e, _ptr(mask), b, nq, nl, num_heads, d, _ptr(out),
                                            _ptr(lse), _ptr(ws), ws.numel(), _stream()), 'mbv_attn_split_fwd_ld')
        else:
            rc = lib.mbv_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(mask), _dt_flag(dt), b, nq, nl,
                                  num_heads, d, _ptr(out), _ptr(lse), _ptr(ws), ws.numel(), _stream())
            check(rc, 'mbv_attn_fwd')
        ctx.save_for_backward(q, k, v, mask, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        q, k, v, mask, out, lse = ctx.saved_tensors
        b, nq, e = q.shape
        nl = k.shape[1]
        h = ctx.num_heads
        grad_out = grad_out.to(q.dtype).contiguous()
        g_q = torch.empty((b, nq, e), dtype=torch.float32, device=q.device)
        g_k = torch.empty((b, nl, e), dtype=torch.float32, device=q.device)
        g_v = torch.empty((b, nl, e), dtype=torch.float32, device=q.device)
        if _k6_split(q.dtype, h, e // h, e, q, k, v, out, grad_out):
            check(lib.mbv_attn_split_bwd_ld(_ptr(q), _ptr(k), _ptr(v), e, _ptr(mask), _ptr(out), _ptr(grad_out), _ptr(lse),
                                            b, nq, nl, h, e // h, _ptr(g_q), _ptr(g_k), _ptr(g_v), e, _stream()),
                  'mbv_attn_split_bwd_ld')
        else:
            rc = lib.mbv_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(out), _ptr(grad_out), _ptr(lse),
                                  _dt_flag(q.dtype), b, nq, nl, h, e // h, _ptr(g_q), _ptr(g_k),
                                  _ptr(g_v), _stream())
            check(rc, 'mbv_attn_bwd')
        dq, dk, dv = ctx.in_dtypes
        return g_q.to(dq), g_k.to(dk), g_v.to(dv), None, None


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, blocked: Optional[torch.Tensor],
              num_heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(d) masked) v per head on MFMA (K6).  q (B, Q, E), k / v (B, L, E) projected inputs;
    ``blocked`` (B, 1|-, Q, L) bool/u8 with True = may not attend, or None.  Returns (B, Q, E)."""
    return _Attention.apply(q, k, v, blocked, num_heads)


class SharedKV:
    """Key / value projections of ONE memory level for the n decoder layers that attend to it (layers l, l+3, l+6 of
    the Mask2Former decoder read the same level, mask2former_head.py:535-560), written side by side by one GEMM each:
    ``k_cat`` / ``v_cat`` (B, L, n*E).  Layer slot j reads columns [j*E, (j+1)*E) in place (K6 with a row stride)
    and its backward writes its dK / dV into the same columns of ``dk_cat`` / ``dv_cat``, so that the gradient of the
    memory is ONE data-gradient GEMM per operand with no accumulation passes.  Plain Python object: autograd sees
    only the scalar ``token`` that orders the backward."""

    def __init__(self):
        self.k_cat = self.v_cat = self.dk_cat = self.dv_cat = None
        self.n = self.e = 0
        self.written = set()


class _SharedKVProject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, key_in, val_in, *wb):
        n = len(wb) // 2
        e = key_in.shape[-1]
        dt = key_in.dtype
        ws, bs = wb[0::2], wb[1::2]
        wc, bc = [_compute_copy(w, dt) for w in ws], [_compute_copy(b_, dt) for b_ in bs]
        if key_in.is_cuda and all(t.is_contiguous() for t in wc + bc):
            # the k / v rows of the n layers' packed parameters → (n*E, E) / (n*E) operands: 4 n pieces, ONE launch (was 4 cats)
            wk = torch.empty((n * e, e), dtype=dt, device=key_in.device)
            wv = torch.empty((n * e, e), dtype=dt, device=key_in.device)
            bk = torch.empty((n * e,), dtype=dt, device=key_in.device)
            bv = torch.empty((n * e,), dtype=dt, device=key_in.device)
            src, dst, nb = [], [], []
            for j in range(n):
                for full, out, r0 in ((wc[j], wk, e), (wc[j], wv, 2 * e), (bc[j], bk, e), (bc[j], bv, 2 * e)):
                    src.append(full[r0:r0 + e].data_ptr())
                    dst.append(out[j * e:(j + 1) * e].data_ptr())
                    nb.append(full[r0:r0 + e].numel() * full.element_size())
            k = len(src)
            check(_lib.load().mbv_copy_group((ctypes.c_void_p * k)(*src), (ctypes.c_void_p * k)(*dst),
                                             (ctypes.c_int64 * k)(*nb), k, _stream()), 'mbv_copy_group')
        else:
            wk = torch.cat([w[e:2 * e] for w in wc], 0)           # (n*E, E)
            wv = torch.cat([w[2 * e:3 * e] for w in wc], 0)
            bk = torch.cat([b_[e:2 * e] for b_ in bc], 0)
            bv = torch.cat([b_[2 * e:3 * e] for b_ in bc], 0)
        with torch.autocast('cuda', enabled=False):
            if dt == torch.float32 and key_in.is_cuda:       # fp32 compute: K20 when the token count allows (else the library)
                holder.k_cat = mm32_nt(key_in.reshape(-1, e), wk, bk).view(key_in.shape[:-1] + (n * e,))
                holder.v_cat = mm32_nt(val_in.to(dt).reshape(-1, e), wv, bv).view(val_in.shape[:-1] + (n * e,))
            else:
                holder.k_cat = torch.nn.functional.linear(key_in, wk, bk)
                holder.v_cat = torch.nn.functional.linear(val_in.to(dt), wv, bv)
        holder.n, holder.e = n, e
        holder.dk_cat = holder.dv_cat = None
        holder.written = set()
        ctx.holder, ctx.params, ctx.n, ctx.e = holder, wb, n, e
        ctx.save_for_backward(key_in, val_in, wk, wv)
        return key_in.new_zeros(())

    @staticmethod
    def backward(ctx, _g_token):
        holder, n, e = ctx.holder, ctx.n, ctx.e
        key_in, val_in, wk, wv = ctx.saved_tensors
        dk, dv = holder.dk_cat, holder.dv_cat
        holder.k_cat = holder.v_cat = holder.dk_cat = holder.dv_cat = None
        grads = [None] * (3 + 2 * n)
        if dk is None:                                   # no layer attended to this level
            return tuple(grads)
        for j in range(n):                               # a slot whose layer did not run contributes nothing
            if j not in holder.written:
                dk[..., j * e:(j + 1) * e].zero_()
                dv[..., j * e:(j + 1) * e].zero_()
        t = key_in.numel() // e
        dk2, dv2 = dk.view(t, n * e), dv.view(t, n * e)
        key2, val2 = key_in.reshape(t, e), val_in.reshape(t, e).to(dk.dtype)
        f32 = dk2.dtype == torch.float32 and dk2.is_cuda
        if ctx.needs_input_grad[1]:
            grads[1] = (mm32_nn(dk2, wk) if f32 else dk2.mm(wk)).view_as(key_in)
        if ctx.needs_input_grad[2]:
            grads[2] = (mm32_nn(dv2, wv) if f32 else dv2.mm(wv)).view_as(val_in).to(val_in.dtype)
        # weight / bias gradients.  Arena parameters: every layer's k / v rows take their product straight into the
        # gradient rows (strided column blocks of dk_cat / dv_cat; the 16-bit products and the column sums join the
        # grouped launches at the end of the pass) — per level that was 2 fills, 2 GEMMs + 2 part sums, 2 column sums
        # and a multi-tensor add: 9 launches of 5-20 us.
        def _arena(p):
            return getattr(p, '_mbv_arena', False) and p.grad is not None and p.grad.dtype == torch.float32
        if (switches.get('skv_direct')
                and all(_arena(ctx.params[i]) and ctx.needs_input_grad[3 + i] for i in range(2 * n))):
            for j in range(n):
                w, b_ = ctx.params[2 * j], ctx.params[2 * j + 1]
                for g2, x2, r0 in ((dk2, key2, e), (dv2, val2, 2 * e)):
                    _wgrad_into(w.grad[r0:r0 + e], g2[:, j * e:(j + 1) * e], x2, None, persistent=True)
                    if not _defer_colsum(g2, b_.grad[r0:r0 + e], t, e, n * e, offset=j * e):
                        _colsum_now(g2, b_.grad[r0:r0 + e], t, e, n * e, offset=j * e)
                _fire_grad_hooks(w)
                _fire_grad_hooks(b_)
            return tuple(grads)
        # ... otherwise: one f32-accumulating GEMM and one column-sum pass per operand ...
        gw = torch.zeros((2, n * e, e), dtype=torch.float32, device=dk.device)
        gb = torch.zeros((2, n * e), dtype=torch.float32, device=dk.device)
        _wgrad_into(gw[0], dk2, key2)
        _wgrad_into(gw[1], dv2, val2)
        colsum_accum(dk2, gb[0])
        colsum_accum(dv2, gb[1])
        # ... then added to the k / v rows of each layer's packed in_proj parameters in one multi-tensor launch
        dst, src = [], []
        for j in range(n):
            w, b_ = ctx.params[2 * j], ctx.params[2 * j + 1]
            rows = slice(j * e, (j + 1) * e)
            for p, g, slot in ((w, gw, 3 + 2 * j), (b_, gb, 4 + 2 * j)):
                if not ctx.needs_input_grad[slot]:
                    continue
                if getattr(p, '_mbv_arena', False) and p.grad is not None and p.grad.dtype == torch.float32:
                    dst += [p.grad[e:2 * e], p.grad[2 * e:3 * e]]
                    src += [g[0][rows], g[1][rows]]
                else:
                    full = torch.zeros_like(p)
                    full[e:2 * e] = g[0][rows]
                    full[2 * e:3 * e] = g[1][rows]
                    grads[slot] = full
        if dst:
            torch._foreach_add_(dst, src)
            for j in range(n):
                _fire_grad_hooks(ctx.params[2 * j])
                _fire_grad_hooks(ctx.params[2 * j + 1])
        return tuple(grads)


class _LevelInputs(torch.autograd.Function):
    """Decoder inputs of one memory level: ``value = tokens(memory) + level_row`` and ``key = value + pos`` in the compute
    dtype (mask2former_head.py:518-527: flatten + level_embed add, + positional encoding in the layer).  One node instead
    of add / add / cast / cast: its backward is one sum of the two 16-bit gradients and a column sum into the embedding
    row — autograd's version was 2 casts, 2 adds, a two-stage ``sum`` with a device memset, ``select_backward``'s zeros +
    copy and an ``add_`` per level, several of them blit nodes with 15-60 us of idle stream around them in a graph."""

    @staticmethod
    def forward(ctx, memory, level_weight, index, pos, dtype):
        b, c = memory.shape[:2]
        x = memory.flatten(2).transpose(1, 2) + level_weight[index].view(1, 1, -1)        # (B, L, C) f32
        key = x + pos
        ctx.index, ctx.mem_shape, ctx.mem_dtype = index, memory.shape, memory.dtype
        ctx.level_weight = level_weight
        return x.to(dtype), key.to(dtype)

    @staticmethod
    def backward(ctx, g_in, g_key):
        w, i = ctx.level_weight, ctx.index
        b, c = ctx.mem_shape[:2]
        if g_in is None and g_key is None:
            return None, None, None, None, None
        if g_in is None or g_key is None:
            g = (g_in if g_key is None else g_key).float()
        else:
            g = torch.add(g_in.float(), g_key)                         # (B, L, C) f32
        g_mem = g.transpose(1, 2).reshape(ctx.mem_shape).to(ctx.mem_dtype) if ctx.needs_input_grad[0] else None
        g_w = None
        if ctx.needs_input_grad[1]:
            g2 = g.reshape(-1, c)
            if getattr(w, '_mbv_arena', False) and w.grad is not None and w.grad.dtype == torch.float32 and g2.is_cuda:
                colsum_accum(g2, w.grad[i], persistent=True)
                _fire_grad_hooks(w)
            else:
                g_w = torch.zeros_like(w)
                g_w[i] = g2.sum(0).to(w.dtype)
        return g_mem, g_w, None, None, None


class _LevelPositions(torch.autograd.Function):
    """Query positions of the pixel decoder's encoder: ``cat_i(pos_i + level_encoding[i])`` over the levels' tokens
    (mmdet MSDeformAttnPixelDecoder: ``level_positional_encoding = level_encoding.weight[i] + pos_i``).  One node: the
    backward is three row-range column sums that join the pass's grouped accumulate (arena) — autograd's version was a
    ``sum`` per level, ``select_backward``'s zeros + copy per level, two adds of the (3, C) pieces and an ``add_``."""

    @staticmethod
    def forward(ctx, weight, lengths, *pos):
        ctx.lengths = lengths
        ctx.weight = weight
        out = torch.cat([p + weight[i].view(1, 1, -1) for i, p in enumerate(pos)], 1)
        return out

    @staticmethod
    def backward(ctx, g):
        w, lengths = ctx.weight, ctx.lengths
        c = g.shape[-1]
        g2 = g.reshape(-1, c) if g.shape[0] == 1 else g.sum(0)
        g2 = g2.contiguous()
        direct = (getattr(w, '_mbv_arena', False) and w.grad is not None and w.grad.dtype == torch.float32
                  and w.grad.is_contiguous() and g2.is_cuda and g2.dtype in _ACT_DTYPES)
        gw = None if direct else torch.zeros_like(w)
        start = 0
        for i, n in enumerate(lengths):
            if direct:
                if not _defer_colsum(g2, w.grad[i], n, c, c, offset=start * c):
                    _colsum_now(g2, w.grad[i], n, c, c, offset=start * c)
            else:
                gw[i] = g2[start:start + n].sum(0).to(w.dtype)
            start += n
        if direct:
            _fire_grad_hooks(w)
        return (gw, None) + (None,) * len(lengths)


def level_positions(weight: torch.Tensor, pos) -> torch.Tensor:
    """(1, sum_i N_i, C): ``pos[i] (1, N_i, C) + weight[i]`` concatenated over the levels."""
    return _LevelPositions.apply(weight, tuple(int(p.shape[1]) for p in pos), *pos)


def level_inputs(memory: torch.Tensor, level_weight: torch.Tensor, index: int, pos: torch.Tensor, dtype: torch.dtype):
    """(value tokens, key tokens) of memory level ``index`` in ``dtype``: memory (B, C, H, W), level_weight (levels, C),
    pos (1 or B, H*W, C)."""
    return _LevelInputs.apply(memory, level_weight, index, pos, dtype)


def shared_kv_project(key_in: torch.Tensor, val_in: torch.Tensor, packed_params) -> tuple:
    """``packed_params``: [(in_proj_weight (3E, E), in_proj_bias (3E,)), ...] of the layers that attend to this memory.
    Returns (holder, token) for :func:`attention_shared_kv`."""
    holder = SharedKV()
    flat = [t for wb in packed_params for t in wb]
    token = _SharedKVProject.apply(holder, key_in, val_in, *flat)
    return holder, token


def shared_kv_supported(num_queries: int, device) -> bool:
    """Strided key / value gradients are plain row stores: one 128-query super-block (K6)."""
    return device.type == 'cuda' and num_queries <= 128


class _AttentionSharedKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, token, blocked, num_heads, holder, slot):
        lib = _lib.load()
        k_cat, v_cat = holder.k_cat, holder.v_cat
        _need_gpu(q, k_cat, v_cat, blocked)
        dt = k_cat.dtype
        if dt not in _ACT_DTYPES:
            raise MaskBevHipError(f'attention supports f32, bf16 and fp16, got {dt}')
        ctx.q_dtype = q.dtype
        q = q.to(dt).contiguous()
        b, nq, e = q.shape
        nl = k_cat.shape[1]
        d = e // num_heads
        ld = holder.n * e
        off = slot * e * k_cat.element_size()
        mask = None
        if blocked is not None:
            mask = blocked.reshape(b, nq, nl).contiguous()
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        out = torch.empty_like(q)
        lse = torch.empty((b, num_heads, nq), dtype=torch.float32, device=q.device)
        ws = _workspace(lib.mbv_attn_workspace_bytes(b, nq, nl, num_heads, d), q.device)
        kp, vp = ctypes.c_void_p(k_cat.data_ptr() + off), ctypes.c_void_p(v_cat.data_ptr() + off)
        if _k6_split(dt, num_heads, d, ld, q, out) and (k_cat.data_ptr() + off) % 16 == 0 and (v_cat.data_ptr() + off) % 16 == 0:
            check(lib.mbv_attn_split_fwd_ld(_ptr(q), kp, vp, ld, _ptr(mask), b, nq, nl, num_heads, d, _ptr(out), _ptr(lse),
                                            _ptr(ws), ws.numel(), _stream()), 'mbv_attn_split_fwd_ld')
        else:
            rc = lib.mbv_attn_fwd_ld(_ptr(q), kp, vp, ld, _ptr(mask), _dt_flag(dt), b, nq, nl, num_heads, d, _ptr(out),
                                     _ptr(lse), _ptr(ws), ws.numel(), _stream())
            check(rc, 'mbv_attn_fwd_ld')
        ctx.save_for_backward(q, mask, out, lse)
        ctx.holder, ctx.slot, ctx.num_heads = holder, slot, num_heads
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        q, mask, out, lse = ctx.saved_tensors
        holder, slot, h = ctx.holder, ctx.slot, ctx.num_heads
        k_cat, v_cat = holder.k_cat, holder.v_cat
        b, nq, e = q.shape
        nl = k_cat.shape[1]
        ld = holder.n * e
        if holder.dk_cat is None:                         # first of the n layers to run backward allocates
            holder.dk_cat = torch.empty_like(k_cat)
            holder.dv_cat = torch.empty_like(v_cat)
        off = slot * e * k_cat.element_size()
        grad_out = grad_out.to(q.dtype).contiguous()
        g_q = torch.empty((b, nq, e), dtype=torch.float32, device=q.device)
        bf = _dt_flag(q.dtype)
        kp, vp = ctypes.c_void_p(k_cat.data_ptr() + off), ctypes.c_void_p(v_cat.data_ptr() + off)
        dkp, dvp = ctypes.c_void_p(holder.dk_cat.data_ptr() + off), ctypes.c_void_p(holder.dv_cat.data_ptr() + off)
        if (_k6_split(q.dtype, h, e // h, ld, q, out, grad_out) and (k_cat.data_ptr() + off) % 16 == 0
                and (v_cat.data_ptr() + off) % 16 == 0 and holder.dk_cat.dtype == torch.float32):
            check(lib.mbv_attn_split_bwd_ld(_ptr(q), kp, vp, ld, _ptr(mask), _ptr(out), _ptr(grad_out), _ptr(lse), b, nq, nl, h,
                                            e // h, _ptr(g_q), dkp, dvp, ld, _stream()), 'mbv_attn_split_bwd_ld')
        else:
            rc = lib.mbv_attn_bwd_ld(_ptr(q), kp, vp, ld, _ptr(mask), _ptr(out), _ptr(grad_out), _ptr(lse), bf, b, nq, nl, h,
                                     e // h, _ptr(g_q), dkp, dvp, ld, bf, _stream())
            check(rc, 'mbv_attn_bwd_ld')
        holder.written.add(slot)
        return g_q.to(ctx.q_dtype), None, None, None, None, None


def attention_shared_kv(q: torch.Tensor, token: torch.Tensor, blocked: Optional[torch.Tensor], num_heads: int,
                        holder: SharedKV, slot: int) -> torch.Tensor:
    """:func:`attention` against slot ``slot`` of a :class:`SharedKV` (keys / values already projected)."""
    return _AttentionSharedKV.apply(q, token, blocked, num_heads, holder, slot)


# --------------------------------------------------------------------------------------
# K7 per-query mask logits + attention mask of the next decoder layer
# --------------------------------------------------------------------------------------
class OutSlot:
    """A caller-provided output buffer handed to an op as a plain Python object (autograd does not see it as an
    input): ``mask_logits`` writes decoder output i straight into slice i of the stacked (D, B, Q, H, W) tensor the
    loss consumes, so neither the per-output f32 cast nor the stack copy exists."""

    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor


class _MaskLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask_embed, mask_feature, out_slot):
        lib = _lib.load()
        _need_gpu(mask_embed, mask_feature)
        dt = mask_feature.dtype
        if dt not in _ACT_DTYPES:
            raise MaskBevHipError(f'mask_logits supports f32, bf16 and fp16, got {dt}')
        e = mask_embed.to(dt).contiguous()
        f = mask_feature.contiguous()
        b, q, c = e.shape
        h, w = f.shape[-2:]
        if out_slot is not None:
            out = out_slot.tensor
            if out.shape != (b, q, h, w) or out.dtype != torch.float32 or not out.is_contiguous():
                raise MaskBevHipError('mask_logits: the output slot must be a contiguous f32 (B, Q, H, W) tensor')
        else:
            out = torch.empty((b, q, h, w), dtype=dt, device=f.device)
        hw = h * w
        if (dt in _GEMM16_DT and gemm16_enabled() and c % 8 == 0 and hw % 8 == 0 and e.data_ptr() % 16 == 0
                and f.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and c * hw * 2 < 0x7fff0000):
            # the same contraction as a batched NN product of K17: E (Q, C) . F (C, HW) with F's pixels contiguous is
            # exactly its K-strided operand form (LDS-DMA + ds_read_b64_tr_b16: no 2-byte transposing LDS stores) —
            # 15.7 against 26.9 us per launch at the bench shape, same sums (scratch/bench_k7.py)
            check(lib.mbv_gemm16_nn(_ptr(e), _ptr(f), _ptr(out), None, None, q, c, hw, c, hw, hw, 0, _GEMM16_DT[dt],
                                    1 if out.dtype == torch.float32 else 0, 0, b, q * c, c * hw, q * hw, None, 0, _stream()),
                  'mbv_gemm16_nn')
        elif dt == torch.float32 and switches.get('k7_f32_library'):
            # f32: the library's batched product (0.76 of the f32 MFMA peak on this shape); K7's own exact-f32 form streams
            # E through L2 per 128-pixel slab and sits at 0.07 of HBM — 179 against ≈ 35 us per launch in the fp32 step
            # (through .data: like the raw-pointer launches around it, the store must not count as an in-place update of the
            # stacked buffer this slot is a view of)
            torch.bmm(e, f.view(b, c, hw), out=out.data.view(b, q, hw))
        else:
            rc = lib.mbv_mask_logits_fwd(_ptr(e), _ptr(f), _dt_flag(dt), b, q, c, hw, _ptr(out),
                                         1 if out.dtype == torch.float32 else 0, _stream())
            check(rc, 'mbv_mask_logits_fwd')
        ctx.save_for_backward(e, f)
        ctx.embed_dtype = mask_embed.dtype
        return out

    @staticmethod
    def backward(ctx, grad_logits):
        # dE = dL . F^T, dF = E^T . dL: K17 for 16-bit operands (ops.mask_logits_backward)
        e, f = ctx.saved_tensors
        b, q, c = e.shape
        h, w = f.shape[-2:]
        dl = grad_logits.to(e.dtype).reshape(b, q, h * w)
        g_e, g_f = mask_logits_backward(dl, e, f.reshape(b, c, h * w))
        return g_e.to(ctx.embed_dtype), g_f.reshape(b, c, h, w), None


class _StackSlices(torch.autograd.Function):
    """``torch.stack(parts)`` when the parts already ARE the consecutive slices of ``buffer``: returns the buffer (no
    copy); the gradient of part i is the view grad[i]."""

    @staticmethod
    def forward(ctx, buffer, *parts):
        ctx.n = len(parts)
        return buffer.view_as(buffer)

    @staticmethod
    def backward(ctx, grad):
        return (None,) + tuple(grad[i] for i in range(ctx.n))


def stack_slices(buffer: torch.Tensor, parts) -> Optional[torch.Tensor]:
    """The stacked tensor of ``parts`` without copying, if every part i is exactly ``buffer[i]``; else None."""
    if buffer is None or len(parts) != buffer.shape[0]:
        return None
    step = buffer.stride(0) * buffer.element_size()
    for i, p in enumerate(parts):
        if (p.dtype != buffer.dtype or p.shape != buffer.shape[1:] or not p.is_contiguous()
                or p.data_ptr() != buffer.data_ptr() + i * step):
            return None
    return _StackSlices.apply(buffer, *parts)


def mask_logits(mask_embed: torch.Tensor, mask_feature: torch.Tensor, target_size, out: Optional[torch.Tensor] = None):
    """mask_embed (B, Q, C) · mask_feature (B, C, H, W) → logits (B, Q, H, W) (MFMA contraction, K7) and the
    boolean cross-attention mask of the next layer, (B, 1, Q, h*w), True = blocked:
    bilinear resize (align_corners=False) → sigmoid < 0.5, rows that would block every key unblocked
    (mask2former_head.py:459-470 and :538-539).  Kept once per query and broadcast over heads."""
    lib = _lib.load()
    logits = _MaskLogits.apply(mask_embed, mask_feature, None if out is None else OutSlot(out))
    b, q, h, w = logits.shape
    th, tw = int(target_size[0]), int(target_size[1])
    blocked = torch.empty((b, 1, q, th * tw), dtype=torch.bool, device=logits.device)
    src = logits.detach()
    rc = lib.mbv_attn_mask_from_logits(_ptr(src), _dt_flag(src.dtype), b * q, h, w, th, tw,
                                       _ptr(blocked), _stream())
    check(rc, 'mbv_attn_mask_from_logits')
    return logits, blocked


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
