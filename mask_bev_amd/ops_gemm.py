"""GEMM-shaped work: K17 (16-bit MFMA, csrc/gemm.hip), K20 (f32 products from IEEE-half pairs, csrc/gemm_f32s.hip: Linears,
the 4 x 4 patch projection, the 3 x 3 convolution), the library-GEMM Linear with its direct / deferred / grouped parameter
gradients (the per-backward-pass queues `_PENDING`, flushed by `flush_deferred_grads`), and the fused FFN pairs."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, switches
from ._lib import MaskBevHipError, check
from .ops_core import *          # noqa: F401,F403
from .ops_records import *          # noqa: F401,F403


# --------------------------------------------------------------------------------------
# K17 16-bit MFMA GEMMs (csrc/gemm.hip)
# --------------------------------------------------------------------------------------
_GEMM16_DT = {torch.bfloat16: 0, torch.float16: 1}
_ACT = {None: 0, 'none': 0, 'relu': 1, 'gelu': 2}


def gemm16_enabled() -> bool:
    """A/B switch: `switches.gemm16 = '0'` sends every Linear back to the library GEMM."""
    return switches.get('gemm16') != '0'


def gemm16_policy() -> str:
    """Which Linear work runs on K17 (csrc/gemm.hip) instead of the library GEMM.  `switches.gemm16` =
    ``auto`` (default): the fused forms — FFN input layer + activation, FFN output layer's data gradient + activation
    backward + bias gradient — and the arena-accumulating weight gradient, for token counts where K17 measured at or
    above the library (scratch/bench_gemm.py, profiles/r02); ``all``: every eligible Linear, forward and backward;
    ``0``: none (the round-1 path)."""
    import os
    v = switches.get('gemm16')
    return {'1': 'auto', '0': 'none'}.get(v, v)


# below these token counts the 128 x 128 tiles under-fill the chip and the library's split / stream-K kernels win
# (scratch/bench_gemm.py on the bench shapes, profiles/r02/c_gemm_shapes.txt).  The fused FFN forms pay down to 4096
# tokens (Swin stage 3): the K17 GEMM alone is slower there than the library's, but it replaces GEMM + GELU forward and
# GEMM + activation-backward/column-sum pass backward — step A/B 8192 / 4096 / 1024: 29.19 / 28.92 / 30.51 ms
def _k17_min_tokens(kind: str) -> Optional[int]:
    return {'fused': switches.get('k17_fused_min'), 'wgrad': 4096}.get(kind)


def _k17_wants(kind: str, tokens: int) -> bool:
    pol = gemm16_policy()
    if pol == 'none':
        return False
    if pol == 'all':
        return True
    floor = _k17_min_tokens(kind)
    return floor is not None and tokens >= floor


def _gemm16_ok(*ts: torch.Tensor) -> bool:
    dt = ts[0].dtype
    return (dt in _GEMM16_DT and all(t.is_cuda and t.dtype == dt and t.dim() == 2 and t.stride(1) == 1
                                     and t.stride(0) % 8 == 0 and t.shape[1] % 8 == 0 and t.data_ptr() % 16 == 0
                                     and t.shape[0] * t.stride(0) * 2 < 0x7fff0000 for t in ts))


def gemm16_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
              out_dtype: Optional[torch.dtype] = None, want_pre: bool = False):
    """``act(x (M, K) @ w (N, K)^T + bias)`` on K17 (bf16 / fp16 inputs, f32 accumulation).  Returns ``out`` or
    ``(out, pre_activation)`` with ``want_pre``.  ``bias`` f32 (N,).  Raises MaskBevHipError for shapes K17 does not take
    (check with :func:`gemm16_nt_ok`)."""
    lib = _lib.load()
    if not _gemm16_ok(x, w) or x.shape[1] != w.shape[1]:
        raise MaskBevHipError('gemm16_nt: unsupported operands')
    m, k = x.shape
    n = w.shape[0]
    od = out_dtype or x.dtype
    if od not in (x.dtype, torch.float32):
        raise MaskBevHipError('gemm16_nt: out dtype must be the input dtype or f32')
    out = torch.empty((m, n), dtype=od, device=x.device)
    pre = torch.empty((m, n), dtype=od, device=x.device) if (want_pre and _ACT[act]) else None
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.data_ptr() % 16):
        raise MaskBevHipError('gemm16_nt: bias must be contiguous f32, 16-byte aligned')
    check(lib.mbv_gemm16_nt(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(pre), m, n, k, x.stride(0), w.stride(0), n,
                            _GEMM16_DT[x.dtype], int(od == torch.float32), _ACT[act], 1, 0, 0, 0, _stream()),
          'mbv_gemm16_nt')
    return (out, pre) if want_pre else out


def gemm16_nn(g: torch.Tensor, w: torch.Tensor, act: Optional[str] = None, aux: Optional[torch.Tensor] = None,
              colsum: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``act'(aux) * (g (M, N) @ w (N, K))`` on K17: the data gradient of a Linear, optionally multiplied by the
    derivative of the activation in front of it (``aux``: ReLU output / GELU pre-activation, (M, K)) with the column
    sums of the result added to ``colsum`` (K,) f32."""
    lib = _lib.load()
    if not _gemm16_ok(g, w) or g.shape[1] != w.shape[0]:
        raise MaskBevHipError('gemm16_nn: unsupported operands')
    m, n = g.shape
    k = w.shape[1]
    a = _ACT[act]
    if a and (aux is None or not _gemm16_ok(aux) or aux.dtype != g.dtype or tuple(aux.shape) != (m, k)):
        raise MaskBevHipError('gemm16_nn: aux must be a (M, K) tensor of the input dtype')
    od = out_dtype or g.dtype
    out = torch.empty((m, k), dtype=od, device=g.device)
    if colsum is not None and (colsum.dtype != torch.float32 or not colsum.is_contiguous()):
        raise MaskBevHipError('gemm16_nn: colsum must be contiguous f32')
    if colsum is not None and switches.get('nn_colsum_defer') and _defer_ok():
        # inside a backward pass the per-wave-row partial sums join the pass's grouped column-sum launch (one small
        # reduction launch per fused data gradient less: 16 per step); the rows live in a tensor of their own until then
        rows = int(lib.mbv_gemm16_nn_part_rows(m, k, 1))
        parts = torch.empty(int(lib.mbv_gemm16_nn_workspace_bytes(m, k, 1)) // 4, dtype=torch.float32, device=g.device)
        check(lib.mbv_gemm16_nn_parts(_ptr(g), _ptr(w), _ptr(out), _ptr(aux if a else None), _ptr(parts), parts.numel() * 4,
                                      m, n, k, g.stride(0), w.stride(0), k, aux.stride(0) if a else 0,
                                      _GEMM16_DT[g.dtype], int(od == torch.float32), a, 1, 0, 0, 0, _stream()),
              'mbv_gemm16_nn_parts')
        if not _defer_colsum(parts, colsum, rows, k, k):
            _colsum_now(parts, colsum, rows, k, k)
        return out
    ws = _workspace(lib.mbv_gemm16_nn_workspace_bytes(m, k, 1), g.device) if colsum is not None else None
    check(lib.mbv_gemm16_nn(_ptr(g), _ptr(w), _ptr(out), _ptr(aux if a else None), _ptr(colsum), m, n, k, g.stride(0),
                            w.stride(0), k, aux.stride(0) if a else 0, _GEMM16_DT[g.dtype],
                            int(od == torch.float32), a, 1, 0, 0, 0, _ptr(ws), 0 if ws is None else ws.numel(),
                            _stream()), 'mbv_gemm16_nn')
    return out


def gemm16_tn_acc(acc: torch.Tensor, g: torch.Tensor, x: torch.Tensor, splits: int = 0) -> None:
    """``acc (N, K) f32 += g (M, N)^T @ x (M, K)`` on K17 (split over M, f32 atomic adds): the weight gradient of a
    Linear accumulated straight into the arena."""
    lib = _lib.load()
    if (not _gemm16_ok(g, x) or g.shape[0] != x.shape[0] or acc.dtype != torch.float32 or acc.stride(1) != 1
            or tuple(acc.shape) != (g.shape[1], x.shape[1]) or acc.data_ptr() % 16):
        raise MaskBevHipError('gemm16_tn_acc: unsupported operands')
    m, n = g.shape
    k = x.shape[1]
    ws = None
    if acc.is_contiguous():               # partial results + owner-adds instead of atomics
        ws = _workspace(lib.mbv_gemm16_tn_workspace_bytes(m, n, k), g.device)
    check(lib.mbv_gemm16_tn(_ptr(g), _ptr(x), _ptr(acc), m, n, k, g.stride(0), x.stride(0), acc.stride(0),
                            _GEMM16_DT[g.dtype], 1, 1, int(splits), 1, 0, 0, 0, _ptr(ws),
                            0 if ws is None else ws.numel(), _stream()), 'mbv_gemm16_tn')


def gemm16_nt_acc(x: torch.Tensor, w: torch.Tensor, splits: int = 0) -> torch.Tensor:
    """``x (B, M, K) @ w (B, N, K)^T`` → (B, M, N) f32 on K17 with the contraction split over workgroups (f32 atomic
    adds into a zeroed result): few rows, long K."""
    lib = _lib.load()
    if x.dim() != 3 or w.dim() != 3 or x.shape[0] != w.shape[0] or x.shape[2] != w.shape[2]:
        raise MaskBevHipError('gemm16_nt_acc: (B, M, K) and (B, N, K) operands')
    x, w = x.contiguous(), w.contiguous()
    if not _gemm16_ok(x[0], w[0]):
        raise MaskBevHipError('gemm16_nt_acc: unsupported operands')
    b, m, k = x.shape
    n = w.shape[1]
    out = torch.zeros((b, m, n), dtype=torch.float32, device=x.device)
    check(lib.mbv_gemm16_nt_acc(_ptr(x), _ptr(w), _ptr(out), m, n, k, k, k, n, _GEMM16_DT[x.dtype], int(splits), b,
                                m * k, n * k, m * n, _stream()), 'mbv_gemm16_nt_acc')
    return out


def mask_logits_backward(dl: torch.Tensor, embed: torch.Tensor, feature: torch.Tensor):
    """Backward of ``einsum('bqc,bcp->bqp', embed, feature)`` (/root/reference: mask_bev/models/networks/
    mask2former_head/mask2former_head.py:459) for dl (B, R, P), embed (B, R, C), feature (B, C, P):
    ``d_embed = dl . feature^T`` (B, R, C) f32 and ``d_feature = embed^T . dl`` (B, C, P) in the operands' dtype.  16-bit
    operands run on K17 (split-K NT with f32 atomics; batched TN stored once); anything else on the library GEMM."""
    if (dl.is_cuda and dl.dtype in _GEMM16_DT and embed.dtype == dl.dtype and feature.dtype == dl.dtype
            and gemm16_policy() != 'none' and dl.shape[2] % 8 == 0 and embed.shape[2] % 8 == 0):
        dl, embed, feature = dl.contiguous(), embed.contiguous(), feature.contiguous()
        if _gemm16_ok(dl[0], embed[0], feature[0]):
            return gemm16_nt_acc(dl, feature), gemm16_tn(embed, dl)
    return torch.bmm(dl, feature.transpose(1, 2)), torch.bmm(embed.transpose(1, 2), dl)


def gemm16_tn(g: torch.Tensor, x: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``g (B, M, N)^T @ x (B, M, K)`` → (B, N, K), stored once per tile (no split over M)."""
    lib = _lib.load()
    if g.dim() != 3 or x.dim() != 3 or g.shape[:2] != x.shape[:2] or not g.is_contiguous() or not x.is_contiguous():
        raise MaskBevHipError('gemm16_tn: (B, M, N) and (B, M, K) contiguous operands')
    if not _gemm16_ok(g[0], x[0]):
        raise MaskBevHipError('gemm16_tn: unsupported operands')
    b, m, n = g.shape
    k = x.shape[2]
    od = out_dtype or g.dtype
    out = torch.empty((b, n, k), dtype=od, device=g.device)
    check(lib.mbv_gemm16_tn(_ptr(g), _ptr(x), _ptr(out), m, n, k, n, k, k, _GEMM16_DT[g.dtype], 0,
                            int(od == torch.float32), 1, b, m * n, m * k, n * k, None, 0, _stream()), 'mbv_gemm16_tn')
    return out


def _gemm32s_ok(*ts: torch.Tensor) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
               and t.shape[1] % 8 == 0 and t.data_ptr() % 16 == 0 and t.shape[0] * t.stride(0) * 4 < 0x7fff0000 for t in ts)


def _amax_ptr(amax, i: int):
    """Pointer to record i of ``amax``: an (n, 64) tensor of records, or a tuple of one-record tensors."""
    if amax is None:
        return ctypes.c_void_p(0)
    if isinstance(amax, (tuple, list)):
        return ctypes.c_void_p(0 if amax[i] is None else amax[i].data_ptr())
    return ctypes.c_void_p(amax.data_ptr() + 4 * AMAX_SLOTS * i)


def gemm32s_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
               amax: Optional[torch.Tensor] = None, want_pre: bool = False, hint_out: bool = False):
    """``act(x (M, K) @ w (N, K)^T + bias)`` in f32 on K20.  ``amax`` = ``f32_absmax([x, w])`` (computed here when None).
    ``hint_out``: the epilogue max-combines |out| into an absmax record left as a hint for the next K20 product."""
    lib = _lib.load()
    if not _gemm32s_ok(x, w) or x.shape[1] != w.shape[1] or w.shape[0] % 8:
        raise MaskBevHipError('gemm32s_nt: unsupported operands')
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.data_ptr() % 16):
        raise MaskBevHipError('gemm32s_nt: bias must be contiguous f32, 16-byte aligned')
    if amax is None:
        amax = tuple(operand_amax([x, w], (True, False)))
    m, k = x.shape
    n = w.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    pre = torch.empty((m, n), dtype=torch.float32, device=x.device) if (want_pre and _ACT[act]) else None
    rec = amax_record(x.device) if hint_out else None
    AMAX_VERIFY.check(x, amax[0], 'gemm32s_nt x')
    AMAX_VERIFY.check(w, amax[1], 'gemm32s_nt w')
    check(lib.mbv_gemm32s_nt(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(pre), m, n, k, x.stride(0), w.stride(0), n,
                             _amax_ptr(amax, 0), _amax_ptr(amax, 1), _ptr(rec), _ACT[act], 1, 0, 0, 0, _stream()),
          'mbv_gemm32s_nt')
    amax_hint_set(out, rec)
    return (out, pre) if want_pre else out


def gemm32s_nn(g: torch.Tensor, w: torch.Tensor, amax_g: Optional[torch.Tensor] = None,
               amax_w: Optional[torch.Tensor] = None, hint_out: bool = False) -> torch.Tensor:
    """``g (M, N) @ w (N, K)`` in f32 on K20 (the data gradient of a Linear); amax_* = one-word tensors."""
    lib = _lib.load()
    if not _gemm32s_ok(g, w) or g.shape[1] != w.shape[0]:
        raise MaskBevHipError('gemm32s_nn: unsupported operands')
    if amax_g is None or amax_w is None:
        both = operand_amax([g, w], (True, False))
        amax_g = both[0] if amax_g is None else amax_g
        amax_w = both[1] if amax_w is None else amax_w
    m, n = g.shape
    k = w.shape[1]
    out = torch.empty((m, k), dtype=torch.float32, device=g.device)
    rec = amax_record(g.device) if hint_out else None
    AMAX_VERIFY.check(g, amax_g, 'gemm32s_nn g')
    AMAX_VERIFY.check(w, amax_w, 'gemm32s_nn w')
    check(lib.mbv_gemm32s_nn(_ptr(g), _ptr(w), _ptr(out), m, n, k, g.stride(0), w.stride(0), k, _amax_ptr(amax_g, 0),
                             _amax_ptr(amax_w, 0), _ptr(rec), 1, 0, 0, 0, _stream()), 'mbv_gemm32s_nn')
    amax_hint_set(out, rec)
    return out


def gemm32s_tn_acc(acc: torch.Tensor, g: torch.Tensor, x: torch.Tensor, amax_g: Optional[torch.Tensor] = None,
                   amax_x: Optional[torch.Tensor] = None) -> None:
    """``acc (N, K) f32 += g (M, N)^T @ x (M, K)`` on K20 (the weight gradient; token sum in parts, owner adds)."""
    lib = _lib.load()
    if (not _gemm32s_ok(g, x) or g.shape[0] != x.shape[0] or acc.dtype != torch.float32 or not acc.is_contiguous()
            or tuple(acc.shape) != (g.shape[1], x.shape[1]) or acc.data_ptr() % 16):
        raise MaskBevHipError('gemm32s_tn_acc: unsupported operands')
    if amax_g is None or amax_x is None:
        both = operand_amax([g, x])
        amax_g = both[0] if amax_g is None else amax_g
        amax_x = both[1] if amax_x is None else amax_x
    m, n = g.shape
    k = x.shape[1]
    nbytes = lib.mbv_gemm32s_tn_workspace_bytes(m, n, k)
    ws = _workspace(nbytes, g.device) if nbytes else None
    AMAX_VERIFY.check(g, amax_g, 'gemm32s_tn_acc g')
    AMAX_VERIFY.check(x, amax_x, 'gemm32s_tn_acc x')
    check(lib.mbv_gemm32s_tn_acc(_ptr(g), _ptr(x), _ptr(acc), m, n, k, g.stride(0), x.stride(0), _amax_ptr(amax_g, 0),
                                 _amax_ptr(amax_x, 0), _ptr(ws), int(nbytes), _stream()), 'mbv_gemm32s_tn_acc')


class _PatchEmbed32(torch.autograd.Function):
    """The backbone's 4 x 4 patch projection on the f32 NCHW pseudo-image as K20 products that gather / scatter the image
    directly (csrc/gemm_f32s.hip, GATHER modes): (B, C, H, W) -> tokens (B, H/4, W/4, E)."""

    @staticmethod
    def forward(ctx, image, weight, bias):
        lib = _lib.load()
        b, c, h, w = image.shape
        e = weight.shape[0]
        image = image.contiguous()
        w2 = weight.reshape(e, -1)
        amax = tuple(operand_amax([image.view(b * c * h, w), w2], (True, False)))      # (K3 leaves the image's record)
        out = torch.empty((b, h // 4, w // 4, e), dtype=torch.float32, device=image.device)
        AMAX_VERIFY.check(image, amax[0], 'patch_embed32 image')
        AMAX_VERIFY.check(w2, amax[1], 'patch_embed32 weight')
        check(lib.mbv_patch_embed32_fwd(_ptr(image), _ptr(w2), _ptr(bias), _ptr(out), b, c, h, w, e, _amax_ptr(amax, 0),
                                        _amax_ptr(amax, 1), _stream()), 'mbv_patch_embed32_fwd')
        ctx.save_for_backward(image, weight)
        ctx.amax, ctx.bias = amax, bias
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        image, weight = ctx.saved_tensors
        bias = ctx.bias
        b, c, h, w = image.shape
        e = weight.shape[0]
        g = g.contiguous()
        g2 = g.view(-1, e)
        amax_g = f32_absmax([g2])
        w2 = weight.reshape(e, -1)
        gi = gw = gb = None
        AMAX_VERIFY.check(g2, amax_g, 'patch_embed32_bwd g')
        AMAX_VERIFY.check(image, ctx.amax[0], 'patch_embed32_bwd image')
        AMAX_VERIFY.check(w2, ctx.amax[1], 'patch_embed32_bwd weight')
        if ctx.needs_input_grad[0]:
            gi = torch.empty_like(image)
            check(lib.mbv_patch_embed32_bwd_image(_ptr(g2), _ptr(w2), _ptr(gi), b, c, h, w, e, _amax_ptr(amax_g, 0),
                                                  _amax_ptr(ctx.amax, 1), _stream()), 'mbv_patch_embed32_bwd_image')
        if ctx.needs_input_grad[1]:
            direct = (getattr(weight, '_mbv_arena', False) and weight.grad is not None
                      and weight.grad.dtype == torch.float32 and weight.grad.is_contiguous())
            acc = weight.grad if direct else torch.zeros_like(weight)
            nbytes = lib.mbv_patch_embed32_bwd_weight_workspace_bytes(b, c, h, w, e)
            ws = _workspace(nbytes, g.device) if nbytes else None
            check(lib.mbv_patch_embed32_bwd_weight(_ptr(g2), _ptr(image), _ptr(acc), b, c, h, w, e, _amax_ptr(amax_g, 0),
                                                   _amax_ptr(ctx.amax, 0), _ptr(ws), int(nbytes), _stream()),
                  'mbv_patch_embed32_bwd_weight')
            if direct:
                _fire_grad_hooks(weight)
            else:
                gw = acc
        if bias is not None and ctx.needs_input_grad[2]:
            if (getattr(bias, '_mbv_arena', False) and bias.grad is not None and bias.grad.dtype == torch.float32):
                colsum_accum(g2, bias.grad, persistent=True)
                _fire_grad_hooks(bias)
            else:
                gb = g2.sum(0)
        return gi, gw, gb


def patch_embed32_ok(image: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    """fp32 compute, a 4 x 4 stride-4 projection, shapes K20's gather modes take (include/maskbev_hip.h)."""
    if not (switches.get('gemm32s') and image.is_cuda and image.dtype == torch.float32 and weight.dtype == torch.float32
            and image.dim() == 4 and weight.dim() == 4 and tuple(weight.shape[2:]) == (4, 4)
            and weight.shape[1] == image.shape[1] and not torch.is_autocast_enabled('cuda')):
        return False
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.data_ptr() % 16):
        return False
    b, c, h, w = image.shape
    return bool(weight.is_contiguous() and weight.data_ptr() % 16 == 0
                and _lib.load().mbv_patch_embed32_supported(b, c, h, w, weight.shape[0]))


def patch_embed32(image: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return _PatchEmbed32.apply(image, weight, bias)


def gemm32s_tn_group(items) -> None:
    """``acc (N, K) f32 += g (M, N)^T @ x (M, K)`` for every ``(g, x, acc[, amax_g, amax_x])`` of ``items`` (f32, pairwise
    disjoint ``acc``) in one K20 launch (+ one parts-add launch) per 48; the operands that come without an absmax record get
    theirs from one absmax launch per 64 of them."""
    if not items:
        return
    lib = _lib.load()
    n = len(items)
    items = [tuple(it) + (None, None) if len(it) == 3 else tuple(it) for it in items]
    for g, x, acc, _, _ in items:
        if (not _gemm32s_ok(g, x) or g.shape[0] != x.shape[0] or acc.dtype != torch.float32
                or not acc.is_contiguous() or tuple(acc.shape) != (g.shape[1], x.shape[1]) or acc.data_ptr() % 16):
            raise MaskBevHipError('gemm32s_tn_group: unsupported operands')
    need = [(i, j) for i, it in enumerate(items) for j in (0, 1) if it[3 + j] is None]
    recs = {}
    if switches.get('amax_hints'):                       # an earlier product of the pass read the same tensor
        for key in list(need):
            r = amax_hint_get(items[key[0]][key[1]])
            if r is not None:
                recs[key] = r
                need.remove(key)
    for c in range(0, len(need), 64):
        chunk = need[c:c + 64]
        r = f32_absmax([items[i][j] for i, j in chunk])
        for q, key in enumerate(chunk):
            recs[key] = r[q:q + 1]
    amax = [[it[3 + j] if it[3 + j] is not None else recs[(i, j)] for j in (0, 1)] for i, it in enumerate(items)]
    if switches.get('amax_verify'):
        for i, it in enumerate(items):
            AMAX_VERIFY.check(it[0], amax[i][0], 'gemm32s_tn_group g')
            AMAX_VERIFY.check(it[1], amax[i][1], 'gemm32s_tn_group x')
    PA, LA = ctypes.c_void_p * n, ctypes.c_int64 * n
    m, nn, k = LA(*[it[0].shape[0] for it in items]), LA(*[it[0].shape[1] for it in items]), LA(*[it[1].shape[1] for it in items])
    nbytes = lib.mbv_gemm32s_tn_group_workspace_bytes(m, nn, k, n)
    ws = _workspace(nbytes, items[0][0].device) if nbytes else None
    check(lib.mbv_gemm32s_tn_group(PA(*[it[0].data_ptr() for it in items]), PA(*[it[1].data_ptr() for it in items]),
                                   PA(*[it[2].data_ptr() for it in items]), m, nn, k,
                                   LA(*[it[0].stride(0) for it in items]), LA(*[it[1].stride(0) for it in items]),
                                   PA(*[_amax_ptr(a[0], 0).value for a in amax]), PA(*[_amax_ptr(a[1], 0).value for a in amax]),
                                   n, _ptr(ws), int(nbytes), _stream()), 'mbv_gemm32s_tn_group')


class _Conv3x3K20(torch.autograd.Function):
    """``conv2d(x, weight, padding=1)`` for a 3 x 3 kernel on an f32 (B, C, H, W) map as K20 products on a zero-bordered
    channels-last ROWS copy of the map (csrc/conv_pad.hip, mbv_conv3x3_gemm32s): forward and data gradient are one product
    over k = (tap, channel) each, the weight gradient nine entries of the grouped TN launch — no im2col, no MIOpen."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = _lib.load()
        b, c, h, w = x.shape
        cout = weight.shape[0]
        x = x.contiguous()
        rows = int(lib.mbv_conv_rows(b, h, w))
        xp = torch.zeros((rows, c), dtype=torch.float32, device=x.device)
        check(lib.mbv_conv_pad_rows(_ptr(x), _ptr(xp), b, c, h, w, 4, _stream()), 'mbv_conv_pad_rows')
        wm = weight.detach().permute(0, 2, 3, 1).reshape(cout, 9 * c).contiguous()
        rec = f32_absmax([xp, wm])
        outp = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        AMAX_VERIFY.check(xp, rec[0:1], 'conv3x3_gemm32s x')
        AMAX_VERIFY.check(wm, rec[1:2], 'conv3x3_gemm32s w')
        check(lib.mbv_conv3x3_gemm32s(_ptr(xp), _ptr(wm), _ptr(outp), b, h, w, c, cout, _amax_ptr(rec, 0), _amax_ptr(rec, 1),
                                      None, _stream()), 'mbv_conv3x3_gemm32s')
        y = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device)
        check(lib.mbv_conv_unpad_rows(_ptr(outp), _ptr(y), b, cout, h, w, 4, _stream()), 'mbv_conv_unpad_rows')
        ctx.save_for_backward(xp, weight)
        ctx.rec_x, ctx.dims = rec[0:1], (b, c, h, w, cout)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xp, weight = ctx.saved_tensors
        b, c, h, w, cout = ctx.dims
        rows = xp.shape[0]
        guard, mp = w + 3, b * (h + 2) * (w + 2)
        gy = gy.contiguous()
        gyp = torch.zeros((rows, cout), dtype=torch.float32, device=gy.device)
        check(lib.mbv_conv_pad_rows(_ptr(gy), _ptr(gyp), b, cout, h, w, 4, _stream()), 'mbv_conv_pad_rows')
        gx = gw = None
        wd = weight.detach()
        if ctx.needs_input_grad[0]:
            wflip = wd.flip(2, 3).permute(1, 2, 3, 0).reshape(c, 9 * cout).contiguous()
            rec = f32_absmax([gyp, wflip])
            rec_g = rec[0:1]
            gxp = torch.empty((rows, c), dtype=torch.float32, device=gy.device)
            check(lib.mbv_conv3x3_gemm32s(_ptr(gyp), _ptr(wflip), _ptr(gxp), b, h, w, cout, c, _amax_ptr(rec, 0),
                                          _amax_ptr(rec, 1), None, _stream()), 'mbv_conv3x3_gemm32s')
            gx = torch.empty((b, c, h, w), dtype=torch.float32, device=gy.device)
            check(lib.mbv_conv_unpad_rows(_ptr(gxp), _ptr(gx), b, c, h, w, 4, _stream()), 'mbv_conv_unpad_rows')
        else:
            rec_g = f32_absmax([gyp])
        if ctx.needs_input_grad[1]:
            # d weight[co][ci][dy][dx] = sum_m gyp[m][co] xp[m + shift_t][ci]: nine token-major products of the grouped launch
            dwm = torch.zeros((9, cout, c), dtype=torch.float32, device=gy.device)
            g2 = gyp[guard:guard + mp]
            items = []
            for t in range(9):
                sh = guard + (t // 3 - 1) * (w + 2) + (t % 3 - 1)
                items.append((g2, xp[sh:sh + mp], dwm[t], rec_g, ctx.rec_x))
            gemm32s_tn_group(items)
            gw = dwm.permute(1, 2, 0).reshape(cout, c, 3, 3)
            if (getattr(weight, '_mbv_arena', False) and weight.grad is not None and weight.grad.dtype == torch.float32):
                weight.grad.add_(gw)
                _fire_grad_hooks(weight)
                gw = None
        return gx, gw


class _Conv3x3K17(torch.autograd.Function):
    """The same convolution for the 16-bit compute modes: 16-bit rows, K17 products (mbv_conv3x3_gemm16; the weight gradient
    nine entries of mbv_gemm16_tn_group, f32).  ``x`` f32 or 16-bit (cast to ``dt``), the result and d x in ``dt``."""

    @staticmethod
    def forward(ctx, x, weight, dt):
        lib = _lib.load()
        b, c, h, w = x.shape
        cout = weight.shape[0]
        ctx.x_dtype = x.dtype
        x = x.to(dt).contiguous()
        rows = int(lib.mbv_conv_rows(b, h, w))
        xp = torch.zeros((rows, c), dtype=dt, device=x.device)
        check(lib.mbv_conv_pad_rows(_ptr(x), _ptr(xp), b, c, h, w, 2, _stream()), 'mbv_conv_pad_rows')
        wm = _compute_copy(weight, dt).detach().permute(0, 2, 3, 1).reshape(cout, 9 * c).contiguous()
        outp = torch.empty((rows, cout), dtype=dt, device=x.device)
        check(lib.mbv_conv3x3_gemm16(_ptr(xp), _ptr(wm), _ptr(outp), b, h, w, c, cout, _GEMM16_DT[dt], 0, _stream()),
              'mbv_conv3x3_gemm16')
        y = torch.empty((b, cout, h, w), dtype=dt, device=x.device)
        check(lib.mbv_conv_unpad_rows(_ptr(outp), _ptr(y), b, cout, h, w, 2, _stream()), 'mbv_conv_unpad_rows')
        ctx.save_for_backward(xp, weight)
        ctx.dims, ctx.dt = (b, c, h, w, cout), dt
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xp, weight = ctx.saved_tensors
        b, c, h, w, cout = ctx.dims
        dt = ctx.dt
        rows = xp.shape[0]
        guard, mp = w + 3, b * (h + 2) * (w + 2)
        gy = gy.to(dt).contiguous()
        gyp = torch.zeros((rows, cout), dtype=dt, device=gy.device)
        check(lib.mbv_conv_pad_rows(_ptr(gy), _ptr(gyp), b, cout, h, w, 2, _stream()), 'mbv_conv_pad_rows')
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wflip = _compute_copy(weight, dt).detach().flip(2, 3).permute(1, 2, 3, 0).reshape(c, 9 * cout).contiguous()
            gxp = torch.empty((rows, c), dtype=dt, device=gy.device)
            check(lib.mbv_conv3x3_gemm16(_ptr(gyp), _ptr(wflip), _ptr(gxp), b, h, w, cout, c, _GEMM16_DT[dt], 0, _stream()),
                  'mbv_conv3x3_gemm16')
            gx = torch.empty((b, c, h, w), dtype=dt, device=gy.device)
            check(lib.mbv_conv_unpad_rows(_ptr(gxp), _ptr(gx), b, c, h, w, 2, _stream()), 'mbv_conv_unpad_rows')
            gx = gx.to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            dwm = torch.zeros((9, cout, c), dtype=torch.float32, device=gy.device)
            g2 = gyp[guard:guard + mp]
            items = []
            for t in range(9):
                sh = guard + (t // 3 - 1) * (w + 2) + (t % 3 - 1)
                items.append((g2, xp[sh:sh + mp], dwm[t]))
            gemm16_tn_group(items)
            gw = dwm.permute(1, 2, 0).reshape(cout, c, 3, 3)
            if (getattr(weight, '_mbv_arena', False) and weight.grad is not None and weight.grad.dtype == torch.float32):
                weight.grad.add_(gw)
                _fire_grad_hooks(weight)
                gw = None
            else:
                gw = gw.to(weight.dtype)
        return gx, gw, None


def conv3x3_16_ok(x: torch.Tensor, conv) -> bool:
    """A 16-bit compute mode (autocast to bf16 / fp16, or 16-bit tensors), a 3 x 3 stride-1 padding-1 convolution without bias
    whose channel counts K17 takes."""
    if not (switches.get('conv3x3_k17') and gemm16_enabled() and x.is_cuda and x.dim() == 4):
        return False
    dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
    return bool(dt in _GEMM16_DT and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros'
                and x.shape[1] % 32 == 0 and conv.weight.shape[0] % 32 == 0 and x.shape[0] * x.shape[2] * x.shape[3] >= 1024)


def conv3x3_16(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
    with torch.autocast('cuda', enabled=False):
        return _Conv3x3K17.apply(x, weight, dt)


def conv3x3_32_ok(x: torch.Tensor, conv) -> bool:
    """fp32 compute, a 3 x 3 stride-1 padding-1 convolution without bias whose channel counts K20 takes."""
    return bool(switches.get('conv3x3_k20') and switches.get('gemm32s') and x.is_cuda and x.dtype == torch.float32
                and x.dim() == 4 and conv.weight.dtype == torch.float32 and not torch.is_autocast_enabled('cuda')
                and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros'
                and x.shape[1] % 32 == 0 and conv.weight.shape[0] % 32 == 0
                and gemm32s_wants(x.shape[0] * x.shape[2] * x.shape[3]))


def conv3x3_32(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    return _Conv3x3K20.apply(x, weight)


def mm32_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x (M, K) @ w (N, K)^T (+ bias)`` for f32 operands: K20 when the product is large enough and its operands fit
    (``switches.gemm32s``), else the library's f32 GEMM — the fp32 compute mode's stand-in for ``torch.mm`` / ``addmm``."""
    if (x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_cuda and x.dim() == 2 and gemm32s_wants(x.shape[0])
            and _gemm32s_ok(x, w) and w.shape[0] % 8 == 0
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))):
        return gemm32s_nt(x, w, bias)
    return torch.mm(x, w.t()) if bias is None else torch.addmm(bias, x, w.t())


def mm32_nn(g: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``g (M, N) @ w (N, K)`` for f32 operands: K20 or the library (see :func:`mm32_nt`)."""
    if (g.dtype == torch.float32 and w.dtype == torch.float32 and g.is_cuda and g.dim() == 2 and gemm32s_wants(g.shape[0])
            and _gemm32s_ok(g, w)):
        return gemm32s_nn(g, w)
    return torch.mm(g, w)


# --------------------------------------------------------------------------------------
# Linear layers: library GEMMs, with a split-K weight gradient for token-major activations
# --------------------------------------------------------------------------------------
def _wgrad_splits(tokens: int) -> int:
    """The weight gradient dW = dY^T X has tiny M x N (channels) and K = tokens (up to 65 536): one library GEMM
    under-fills the chip (measured 290 us vs 47 us at T = 65 536, 192 -> 576, MI355X).  Split K into chunks
    solved as one batched GEMM and reduce the partials in f32."""
    for s, t in ((128, 131072), (32, 32768), (8, 8192)):
        if tokens >= t:
            return s
    return 1


def _wgrad(g2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """dW (out, in) = g2^T x2 for token-major g2 (T, out), x2 (T, in); f32 result, split-K for large T."""
    t = g2.shape[0]
    s = _wgrad_splits(t)
    if s == 1:
        return g2.t().mm(x2).float()
    c = t // s                          # rows per chunk; the ragged tail (< s rows) is one more small GEMM
    gw = torch.bmm(g2[:s * c].view(s, c, -1).transpose(1, 2), x2[:s * c].view(s, c, -1)).sum(0, dtype=torch.float32)
    if s * c < t:                       # (the tail's product accumulates through the GEMM's beta = 1: no add launch)
        if g2.dtype == torch.float32:
            gw = torch.addmm(gw, g2[s * c:].t(), x2[s * c:])
        else:
            gw = gw + g2[s * c:].t().mm(x2[s * c:]).float()
    return gw


def _compute_copy(p: Optional[torch.Tensor], dt: torch.dtype) -> Optional[torch.Tensor]:
    """The parameter in the compute dtype: the arena's bf16 shadow when there is one (arena.py), else a cast."""
    if p is None or p.dtype == dt:
        return p
    sh = getattr(p, '_mbv_shadow', None)
    if sh is not None and sh.dtype == dt:
        return sh
    return p.to(dt)


def _fire_grad_hooks(p: torch.Tensor):
    """Gradients accumulated outside autograd still announce themselves to post-accumulate hooks (ddp.py)."""
    hooks = getattr(p, '_post_accumulate_grad_hooks', None)
    if hooks:
        for h in list(hooks.values()):
            h(p)


def colsum_accum(g2: torch.Tensor, out: torch.Tensor, persistent: bool = False):
    """out (N,) f32 += column sums of g2 (T, N) (bf16 or f32) — the bias gradient, in one launch.
    ``persistent``: ``out`` is an arena gradient — inside a backward pass the sum joins the grouped launch at its end."""
    lib = _lib.load()
    _need_gpu(g2, out)
    if g2.dtype not in _ACT_DTYPES or out.dtype != torch.float32 or not out.is_contiguous():
        raise MaskBevHipError('colsum_accum: g2 must be f32, bf16 or fp16 and out contiguous f32')
    g2 = g2.contiguous()
    if persistent and _defer_colsum(g2, out, g2.shape[0], g2.shape[1], g2.shape[1]):
        return
    check(lib.mbv_colsum_accum(_ptr(g2), _dt_flag(g2.dtype), g2.shape[0], g2.shape[1], _ptr(out),
                               _stream()), 'mbv_colsum_accum')


# Parameter gradients are nobody's input.  During a backward pass the small ones — exact-f32 weight gradients of the
# decoder's few-row Linears, bias gradients (column sums), the per-block partial rows of K12's LayerNorm-parameter
# gradients — are collected and issued as a few grouped launches (mbv_wgrad_small_f32_group, mbv_colsum_accum_group)
# from an autograd-engine callback at the end of that pass: ≈ 140 launches of 5-12 us with the chip mostly idle become
# four that fill it.  Only accumulations into ARENA gradients are deferred (nothing reads those before the pass ends).
# `switches.wgrad_group = False` keeps the per-layer launches (A/B).
_PENDING: dict = {}          # autograd graph-task id -> ([small weight gradients], [column sums]) of that backward pass
_PENDING_MAX = 32            # entries kept at most: nesting depth of re-entrant passes + leftovers of passes that raised


def _pending_lists():
    """The pending lists of the running backward pass (creating them and arming the end-of-pass callback on first use),
    or None outside a pass / with the switch off.  Keyed by the engine's graph-task id: a re-entrant pass (the deferred
    heads re-evaluate a sub-graph inside the outer backward) flushes its own work, and what a pass that raised left
    behind is never mistaken for the next pass's work."""
    if not switches.get('wgrad_group'):
        return None
    tid = torch._C._current_graph_task_id()
    if tid < 0:
        return None
    lists = _PENDING.get(tid)
    if lists is None:
        try:        # the callback runs when this pass has executed every node
            torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_deferred_grads(tid))
        except RuntimeError:
            return None
        # Leftovers of passes that raised before their callback ran hold (g, x) activations alive.  A live pass cannot be
        # told from a dead one by its id (an outer pass stays live while any number of inner passes come and go, each
        # with a higher id), but every pass that ENDS removes its entry, so the entries that exist are the nesting
        # depth plus the leaked ones: only when far more exist than passes can nest are the oldest dropped.
        if len(_PENDING) >= _PENDING_MAX:
            for old in sorted(_PENDING)[:len(_PENDING) - _PENDING_MAX + 1]:
                del _PENDING[old]
        lists = _PENDING[tid] = ([], [], [], [])
    return lists


def _defer_ok() -> bool:
    return _pending_lists() is not None


def _defer_small_wgrad(g2, x2, acc, bias_acc) -> bool:
    lists = _pending_lists()
    if lists is None:
        return False
    lists[0].append((g2, x2, acc, bias_acc, torch.cuda.current_stream()))
    return True


def _tn_group_mode() -> str:
    """`switches.tn_group`: ``1`` (default) — the K17 weight gradients of a backward pass are collected and issued as grouped
    launches at its end (mbv_gemm16_tn_group); ``all`` — every 16-bit arena weight gradient with at least 512 tokens joins
    the group, also those the per-layer policy leaves to the library (few tokens, wide inputs); ``0`` — per-layer launches."""
    return switches.get('tn_group')


def _defer_tn_wgrad(g2: torch.Tensor, x2: torch.Tensor, acc: torch.Tensor) -> bool:
    if _tn_group_mode() == '0' or not acc.is_contiguous():
        return False
    lists = _pending_lists()
    if lists is None:
        return False
    lists[2].append((g2, x2, acc, torch.cuda.current_stream()))
    return True


def _defer_tn32_wgrad(g2: torch.Tensor, x2: torch.Tensor, acc: torch.Tensor, amax) -> bool:
    """fp32 compute: a token-major K20 weight gradient joins the pass's grouped launch (``switches.tn32_group``)."""
    if not switches.get('tn32_group'):
        return False
    lists = _pending_lists()
    if lists is None:
        return False
    ag, ax = (None, None) if amax is None else (amax[0], amax[1])
    if switches.get('amax_hints'):      # resolved NOW: a hint lives as long as the tensor object it was left on, not until the flush
        ag = amax_hint_get(g2) if ag is None else ag
        ax = amax_hint_get(x2) if ax is None else ax
    lists[3].append((g2, x2, acc, ag, ax, torch.cuda.current_stream()))
    return True


_TN_SINK: Optional[list] = None


def set_tn_sink(sink: Optional[list]) -> None:
    """While a list is installed, the end-of-pass flush appends the pass's ``(g, x, acc)`` weight-gradient products to it
    instead of launching them (``None`` restores the launch)."""
    global _TN_SINK
    _TN_SINK = sink


def launch_tn_group(items) -> None:
    """The grouped launch(es) for a pass's products: deepest token sums first (their work items are the longest of a
    launch), one call per 16-bit dtype."""
    items = sorted(items, key=lambda it: -it[0].shape[0])
    for dt in {it[0].dtype for it in items}:
        for wave in _distinct_destination_waves([it for it in items if it[0].dtype == dt]):
            gemm16_tn_group(wave)


def _distinct_destination_waves(items):
    """Split ``(g, x, acc)`` products into successive launches whose ``acc`` ranges are pairwise disjoint.  Inside one
    grouped launch a destination is read-modified-written without atomics (single-range entries add their tile in
    place, multi-range entries are folded in by ``k_add_parts_group``), so a weight used twice in one backward pass —
    tied weights, one Linear applied twice — must not meet itself in a launch: its second product goes to the next
    one, which the stream orders behind the first."""
    waves = []                       # [(items, [(lo, hi) byte ranges])]
    for it in items:
        lo = it[2].data_ptr()
        hi = lo + it[2].numel() * it[2].element_size()
        for w_items, w_ranges in waves:
            if all(hi <= a or lo >= b for a, b in w_ranges):
                w_items.append(it)
                w_ranges.append((lo, hi))
                break
        else:
            waves.append(([it], [(lo, hi)]))
    return [w for w, _ in waves]


def gemm16_tn_group(items) -> None:
    """``acc (N, K) f32 += g (M, N)^T @ x (M, K)`` for every ``(g, x, acc)`` of ``items`` in one K17 launch per 48 (all of
    one 16-bit dtype, contiguous ``acc``)."""
    if not items:
        return
    lib = _lib.load()
    n = len(items)
    dt = items[0][0].dtype
    for g, x, acc in items:
        if (g.dtype != dt or not _gemm16_ok(g, x) or g.shape[0] != x.shape[0] or acc.dtype != torch.float32
                or not acc.is_contiguous() or tuple(acc.shape) != (g.shape[1], x.shape[1]) or acc.data_ptr() % 16):
            raise MaskBevHipError('gemm16_tn_group: unsupported operands')
    PA, LA = ctypes.c_void_p * n, ctypes.c_int64 * n
    m, nn, k = LA(*[g.shape[0] for g, _, _ in items]), LA(*[g.shape[1] for g, _, _ in items]), \
        LA(*[x.shape[1] for _, x, _ in items])
    nbytes = lib.mbv_gemm16_tn_group_workspace_bytes(m, nn, k, n)
    ws = _workspace(nbytes, items[0][0].device) if nbytes else None
    check(lib.mbv_gemm16_tn_group(PA(*[g.data_ptr() for g, _, _ in items]), PA(*[x.data_ptr() for _, x, _ in items]),
                                  PA(*[a.data_ptr() for _, _, a in items]), m, nn, k,
                                  LA(*[g.stride(0) for g, _, _ in items]), LA(*[x.stride(0) for _, x, _ in items]),
                                  n, _GEMM16_DT[dt], _ptr(ws), int(nbytes), _stream()), 'mbv_gemm16_tn_group')


def _defer_colsum(g2: torch.Tensor, out: torch.Tensor, rows: int, n: int, ld: int, offset: int = 0) -> bool:
    """out (n,) f32 += column sums of the (rows, n) block of ``g2`` that starts ``offset`` elements in, row stride ld."""
    if not g2.is_cuda or g2.dtype not in _ACT_DTYPES or out.dtype != torch.float32 or not out.is_contiguous():
        return False
    lists = _pending_lists()
    if lists is None:
        return False
    lists[1].append((g2, out, int(rows), int(n), int(ld), int(offset), torch.cuda.current_stream()))
    return True


def _colsum_now(g2: torch.Tensor, out: torch.Tensor, rows: int, n: int, ld: int, offset: int = 0) -> None:
    """The immediate form of :func:`_defer_colsum` (the kernel that produced ``g2`` was told its reduction comes later,
    so when the queue refuses it the reduction has to happen here — dropping it would lose the gradient silently)."""
    if g2.dtype not in _ACT_DTYPES or out.dtype != torch.float32:
        raise MaskBevHipError('column-sum accumulate: g2 must be f32, bf16 or fp16 and out f32')
    if not out.is_contiguous():
        tmp = torch.zeros(n, dtype=torch.float32, device=out.device)
        _colsum_now(g2, tmp, rows, n, ld, offset)
        out.add_(tmp)
        return
    lib = _lib.load()
    PA, IA, LA = ctypes.c_void_p * 1, ctypes.c_int32 * 1, ctypes.c_int64 * 1
    check(lib.mbv_colsum_accum_group(PA(g2.data_ptr() + offset * g2.element_size()), IA(_dt_flag(g2.dtype)),
                                     LA(int(rows)), IA(int(n)), LA(int(ld)), PA(out.data_ptr()), 1, _stream()),
          'mbv_colsum_accum_group')


def flush_deferred_grads(task_id: Optional[int] = None) -> None:
    """Issue the parameter-gradient work collected by backward pass ``task_id`` (default: by every pass that has some
    pending — callable directly; a no-op when nothing is pending)."""
    tids = [task_id] if task_id is not None else list(_PENDING)
    if task_id is not None:          # passes nested INSIDE this one have ended: what they left (they raised) is dropped
        for t in [t for t in _PENDING if t > task_id]:
            del _PENDING[t]
    wg, cs, tn, tn32 = [], [], [], []
    for t in tids:
        lists = _PENDING.pop(t, None)
        if lists is not None:
            wg += lists[0]
            cs += lists[1]
            tn += lists[2]
            tn32 += lists[3]
    if not wg and not cs and not tn and not tn32:
        return
    lib = _lib.load()
    cur = torch.cuda.current_stream()
    wg_all = list(wg)
    for st in {it[-1] for it in wg + cs + tn + tn32}:
        if st != cur:
            cur.wait_stream(st)
    if tn and _TN_SINK is not None:
        # the caller (graph.py, while it captures a backward pass) takes the pass's weight-gradient products over and
        # issues them itself — after the replay, on a side stream, underneath the eager encoder backward
        _TN_SINK.extend(it[:3] for it in tn)
        tn = []
    if tn:
        launch_tn_group([it[:3] for it in tn])
    if wg and switches.get('gemm32s') and switches.get('tn32_group'):
        # fp32 compute: the few-row products K20 takes (n, k multiples of 8, aligned rows) leave the exact-f32 MFMA group
        # for ONE grouped K20 launch (+ one absmax launch per 32 products); their bias column sums join the column-sum group
        k20 = [it for it in wg if (it[0].dtype == torch.float32 and it[1].dtype == torch.float32 and it[0].shape[0] <= 8192
                                   and _gemm32s_ok(it[0], it[1]) and it[2].dtype == torch.float32 and it[2].is_contiguous()
                                   and it[2].data_ptr() % 16 == 0
                                   and (it[3] is None or (it[3].dtype == torch.float32 and it[3].is_contiguous())))]
        if k20:
            ids = {id(it) for it in k20}
            wg = [it for it in wg if id(it) not in ids]
            tn32 = tn32 + [(it[0], it[1], it[2], None, None, it[-1]) for it in k20]
            for it in k20:
                if it[3] is not None:
                    cs.append((it[0], it[3], it[0].shape[0], it[0].shape[1], it[0].stride(0), 0, it[-1]))
    if tn32:
        # deepest token sums first (their work items are the longest of a launch); a weight used twice meets itself in the next launch
        for wave in _distinct_destination_waves(sorted(tn32, key=lambda it: -it[0].shape[0])):
            gemm32s_tn_group([it[:5] for it in wave])
    if wg:
        n = len(wg)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
        check(lib.mbv_wgrad_small_f32_group(
            PA(*[it[0].data_ptr() for it in wg]), PA(*[it[1].data_ptr() for it in wg]),
            PA(*[it[2].data_ptr() for it in wg]), PA(*[(it[3].data_ptr() if it[3] is not None else 0) for it in wg]),
            IA(*[it[0].shape[0] for it in wg]), IA(*[it[0].shape[1] for it in wg]), IA(*[it[1].shape[1] for it in wg]),
            n, _stream()), 'mbv_wgrad_small_f32_group')
    if cs:
        n = len(cs)
        PA, IA, LA = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n
        check(lib.mbv_colsum_accum_group(
            PA(*[it[0].data_ptr() + it[5] * it[0].element_size() for it in cs]), IA(*[_dt_flag(it[0].dtype) for it in cs]),
            LA(*[it[2] for it in cs]), IA(*[it[3] for it in cs]), LA(*[it[4] for it in cs]),
            PA(*[it[1].data_ptr() for it in cs]), n, _stream()), 'mbv_colsum_accum_group')
    for it in wg_all + tn + tn32:     # the producers' memory may be reused by later work on their own streams
        if it[-1] != cur:
            it[0].record_stream(cur)
            it[1].record_stream(cur)
    for it in cs:
        if it[-1] != cur:
            it[0].record_stream(cur)


flush_small_wgrads = flush_deferred_grads


def _wgrad_into(acc: torch.Tensor, g2: torch.Tensor, x2: torch.Tensor, bias_acc: Optional[torch.Tensor] = None,
                persistent: bool = False, amax=None) -> bool:
    """acc (out, in) f32 += g2^T x2, f32 accumulation inside the GEMM (no bf16 round trip, no separate add).
    Returns True when ``bias_acc`` (out,) f32 += column sums of g2 was done by the same launch.
    ``persistent``: ``acc`` / ``bias_acc`` are arena gradients nobody reads before the backward pass ends — the
    small-token form may then be deferred to the grouped launch at the end of the pass."""
    t = g2.shape[0]
    if ((amax is not None or (g2.dtype == torch.float32 and x2.dtype == torch.float32 and g2.is_cuda and gemm32s_wants(t)))
            and acc.dtype == torch.float32 and acc.is_contiguous() and acc.data_ptr() % 16 == 0 and _gemm32s_ok(g2, x2)):
        # fp32 compute: K20, token sum in parts, owner adds (the absmax words come from the layer's forward when it has them);
        # an arena gradient joins the pass's grouped launch
        if persistent and _defer_tn32_wgrad(g2, x2, acc, amax):
            return False
        gemm32s_tn_acc(acc, g2, x2, None if amax is None else amax[0], None if amax is None else amax[1])
        return False
    if (g2.dtype in _GEMM16_DT and x2.dtype == g2.dtype and acc.stride(-1) == 1 and acc.data_ptr() % 16 == 0
            and gemm16_policy() != 'none' and _gemm16_ok(g2, x2)):
        per_layer = _k17_wants('wgrad', t) and (x2.shape[1] <= switches.get('tn_max_in') or gemm16_policy() == 'all')   # 2048-wide patch rows: the library wins (77 vs 95 us)
        # few-token 16-bit products (the decoder's 400-row output projections: a 256 x 256 result over 400 rows) are a
        # handful of work items of the grouped launch; alone, the library ran them as ONE 256 x 256 tile — 30 us each
        few = t <= 512 and gemm16_policy() == 'auto'       # (Swin stage 4's 1024-token layers stay with the library: measured)
        if (persistent and (per_layer or few or (_tn_group_mode() == 'all' and t >= 512))
                and _defer_tn_wgrad(g2, x2, acc)):
            return False                         # K17, grouped with the pass's other weight gradients at its end
        if per_layer:
            gemm16_tn_acc(acc, g2, x2)           # K17: split over the tokens, parts added into the arena
            return False
    if (g2.dtype == torch.float32 and x2.dtype == torch.float32 and t <= _SMALL_F32_ROWS and g2.is_cuda
            and acc.is_contiguous()):
        lib = _lib.load()
        g2, x2 = g2.contiguous(), x2.contiguous()
        fuse = bias_acc is not None and bias_acc.is_contiguous() and bias_acc.dtype == torch.float32
        if persistent and _defer_small_wgrad(g2, x2, acc, bias_acc if fuse else None):
            return fuse
        check(lib.mbv_wgrad_small_f32(_ptr(g2), _ptr(x2), t, g2.shape[1], x2.shape[1], _ptr(acc),
                                      _ptr(bias_acc) if fuse else ctypes.c_void_p(0), _stream()),
              'mbv_wgrad_small_f32')
        return fuse
    s = _wgrad_splits(t)
    od = {} if g2.dtype == torch.float32 else dict(out_dtype=torch.float32)
    if s == 1:
        torch.addmm(acc, g2.t(), x2, out=acc, **od)
        return False
    c = t // s
    part = torch.bmm(g2[:s * c].view(s, c, -1).transpose(1, 2), x2[:s * c].view(s, c, -1), **od)
    if s * c < t:
        torch.addmm(acc, g2[s * c:].t(), x2[s * c:], out=acc, **od)
    if acc.is_contiguous() and part.is_cuda:
        colsum_accum(part.view(s, -1), acc.view(-1))         # Σ over the K-chunks, added in the same launch
    else:
        acc.add_(part.sum(0))
    return False


# Under autocast, f32 activations with at most this many rows (the decoder's B*Q query tokens) are multiplied in
# f32: the GEMM is microseconds either way, and the five cast kernels per layer and direction are not.
_SMALL_F32_ROWS = 2048
_SMALL_F32_MACS = 1 << 30          # … and only while the f32 GEMM itself stays in the microseconds


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rows, f32_out=False, skip_bias_grad=False):
        if torch.is_autocast_enabled('cuda') and not (
                x.dtype == torch.float32 and weight.dtype == torch.float32
                and x.numel() <= _SMALL_F32_ROWS * x.shape[-1] and x.numel() * weight.shape[0] <= _SMALL_F32_MACS):
            dt = torch.get_autocast_dtype('cuda')
            ctx.gx_f32 = x.dtype == torch.float32 and dt in _LO_DTYPES and x.is_cuda    # the caller's tensor is f32
            x, w, b = x.to(dt), _compute_copy(weight, dt), _compute_copy(bias, dt)
        else:
            ctx.gx_f32 = False
            w, b = weight, bias
        if rows is not None:
            w = w[rows[0]:rows[1]]
            b = None if b is None else b[rows[0]:rows[1]]
        x2k = x.reshape(-1, x.shape[-1]) if x.is_cuda and x.dtype in _GEMM16_DT else None
        ctx.amax = None
        x32 = None
        if x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() >= 2:
            x32 = x.reshape(-1, x.shape[-1])
            if not (gemm32s_wants(x32.shape[0]) and _gemm32s_ok(x32, w) and w.shape[0] % 8 == 0
                    and (b is None or (b.dtype == torch.float32 and b.is_contiguous() and b.data_ptr() % 16 == 0))):
                x32 = None
        with torch.autocast('cuda', enabled=False):
            if x32 is not None:
                # fp32 compute: K20 — f32 products from IEEE-half pairs on the 16-bit matrix pipe (csrc/gemm_f32s.hip).  The
                # operand scales: x's absmax record from its producer when it left one (K12, K20), else one pass over x;
                # the weight's once per parameter update
                hints = bool(switches.get('amax_hints'))
                hx = amax_hint_get(x32) if hints else None
                if hx is not None:
                    ctx.amax = (hx, weight_amax(w))
                else:
                    both = f32_absmax([x32, w])
                    ctx.amax = (both[0:1], both[1:2])
                y2 = gemm32s_nt(x32, w, b, amax=ctx.amax, hint_out=hints)
                y = y2.view(x.shape[:-1] + (w.shape[0],))
                amax_hint_set(y, amax_hint_get(y2))
            elif (x2k is not None and gemm16_policy() == 'all' and _gemm16_ok(x2k, w)
                    and (bias is None or bias.dtype == torch.float32)):
                bf = None if bias is None else (bias if rows is None else bias[rows[0]:rows[1]])
                y = gemm16_nt(x2k, w, bf, out_dtype=torch.float32 if f32_out else None)
                y = y.view(x.shape[:-1] + (w.shape[0],))
            elif f32_out and x.dtype in _LO_DTYPES and x.is_cuda:
                # 16-bit GEMM with the f32 accumulators stored as f32 (the consumer wants f32: no cast pass)
                x2 = x.reshape(-1, x.shape[-1])
                if bias is not None:
                    bf = bias if rows is None else bias[rows[0]:rows[1]]
                    y = torch.addmm(bf.float(), x2, w.t(), out_dtype=torch.float32)
                else:
                    y = torch.mm(x2, w.t(), out_dtype=torch.float32)
                y = y.view(x.shape[:-1] + (w.shape[0],))
            else:
                y = torch.nn.functional.linear(x, w, b)
        ctx.save_for_backward(x, w)
        ctx.weight, ctx.bias, ctx.rows = weight, bias, rows
        ctx.skip_bias_grad = skip_bias_grad
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        weight, bias, rows = ctx.weight, ctx.bias, ctx.rows
        gy = gy.to(x.dtype)
        g2 = gy.reshape(-1, gy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        gx = gw = gb = None
        amax_g = None
        if ctx.amax is not None:
            if not g2.is_contiguous():
                g2 = g2.contiguous()
            if _gemm32s_ok(g2, w) and _gemm32s_ok(x2):
                amax_g = amax_hint_get(g2) if switches.get('amax_hints') else None
                if amax_g is None:
                    amax_g = f32_absmax([g2])
        if ctx.needs_input_grad[0]:
            if amax_g is not None:
                gx = gemm32s_nn(g2, w, amax_g, ctx.amax[1], hint_out=bool(switches.get('amax_hints')))
                gx = _hinted_view(gx, x.shape)
            elif gemm16_policy() == 'all' and g2.is_cuda and _gemm16_ok(g2, w):
                gx = gemm16_nn(g2, w).view_as(x)
            elif ctx.gx_f32:         # an f32 input was cast for the GEMM: its gradient leaves the GEMM as f32 (no cast pass)
                gx = torch.mm(g2, w, out_dtype=torch.float32).view_as(x)
            else:
                gx = g2.mm(w).view_as(x)
        bias_direct = (bias is not None and ctx.needs_input_grad[2] and getattr(bias, '_mbv_arena', False)
                       and bias.grad is not None and bias.grad.dtype == torch.float32)
        bias_done = ctx.skip_bias_grad        # the consumer of this layer's output accumulates db (K12 / activation op)
        if bias_done:
            bias_direct = False
        if ctx.needs_input_grad[1]:
            if getattr(weight, '_mbv_arena', False) and weight.grad is not None and weight.grad.dtype == torch.float32:
                acc = weight.grad if rows is None else weight.grad[rows[0]:rows[1]]
                bacc = None
                if bias_direct:
                    bacc = bias.grad if rows is None else bias.grad[rows[0]:rows[1]]
                bias_done = _wgrad_into(acc, g2, x2, bacc, persistent=True,                  # straight into the arena
                                        amax=None if amax_g is None else (amax_g, ctx.amax[0])) or bias_done
                _fire_grad_hooks(weight)
                if bias_done:
                    _fire_grad_hooks(bias)
            elif amax_g is not None and weight.dtype == torch.float32 and weight.is_contiguous():
                gw = torch.zeros_like(weight)
                gemm32s_tn_acc(gw if rows is None else gw[rows[0]:rows[1]], g2, x2, amax_g, ctx.amax[0])
            elif rows is None:
                gw = _wgrad(g2, x2).to(weight.dtype)
            else:
                gw = torch.zeros_like(weight)
                gw[rows[0]:rows[1]] = _wgrad(g2, x2)
        if bias is not None and ctx.needs_input_grad[2] and not bias_done:
            if bias_direct:
                colsum_accum(g2, bias.grad if rows is None else bias.grad[rows[0]:rows[1]], persistent=True)
                _fire_grad_hooks(bias)
            elif rows is None:
                gb = g2.sum(0, dtype=torch.float32).to(bias.dtype)
            else:
                gb = torch.zeros_like(bias)
                gb[rows[0]:rows[1]] = g2.sum(0, dtype=torch.float32)
        return gx, gw, gb, None, None, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           rows: Optional[tuple] = None, f32_out: bool = False, skip_bias_grad: bool = False) -> torch.Tensor:
    """y = x W^T + b on the library GEMM (hipBLASLt) honouring autocast, with the split-K weight gradient.
    ``rows=(r0, r1)`` uses rows r0:r1 of the parameters (the q / k / v blocks of a packed ``in_proj_weight``)
    without materialising slices or zero-padded slice gradients.  Parameters that live in a
    :class:`~mask_bev_amd.arena.ParameterArena` are read through their bf16 shadow and receive their gradient by
    direct f32 accumulation (the autograd gradient returned for them is ``None``)."""
    _LAST_HINT[1] = None             # see amax_hint_refresh: only a hint THIS forward sets may be re-attached to y
    y = _Linear.apply(x, weight, bias, rows, f32_out, skip_bias_grad)
    amax_hint_refresh(y)
    return y


class _FFN(torch.autograd.Function):
    """``fc2(act(fc1(x)))`` of an mmcv FFN (/root/reference: mask_bev/models/networks/swin/swin.py:347-377) with the
    element-wise work folded into K17's epilogues: forward, fc1 + bias + activation in one launch (stores the
    pre-activation for GELU); backward, the data gradient of fc2 times the activation derivative with the column sums
    of the result (= d bias of fc1) in one launch, the two weight gradients accumulated straight into the arena, and no
    separate activation / bias kernels.  Parameters must live in a parameter arena (bf16 shadow, f32 gradients)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, kind, defer_out_bias):
        dt = torch.get_autocast_dtype('cuda')
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != dt:
            x2 = x2.to(dt)
        w1c, w2c = _compute_copy(w1, dt), _compute_copy(w2, dt)
        if kind == 'gelu':
            a, h = gemm16_nt(x2, w1c, b1, act='gelu', want_pre=True)
        else:
            a, h = gemm16_nt(x2, w1c, b1, act='relu'), None
        if gemm16_policy() == 'all':
            out = gemm16_nt(a, w2c, b2)
        else:
            out = torch.nn.functional.linear(a, w2c, _compute_copy(b2, dt))
        ctx.save_for_backward(x2, a if h is None else h, a, w1c, w2c)
        ctx.params = (w1, b1, w2, b2)
        ctx.kind, ctx.defer_out_bias, ctx.xshape = kind, defer_out_bias, x.shape
        ctx.x_f32 = x.dtype == torch.float32
        return out.view(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, gout):
        x2, aux, a, w1c, w2c = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        g2 = gout.reshape(-1, gout.shape[-1])
        if g2.dtype != x2.dtype:
            g2 = g2.to(x2.dtype)
        g2 = g2.contiguous()
        t = g2.shape[0]
        # d hidden = (g . W2) * act'(.), column sums -> d b1
        dh = gemm16_nn(g2, w2c, act=ctx.kind, aux=aux, colsum=b1.grad)
        _fire_grad_hooks(b1)
        _wgrad_into(w2.grad, g2, a, persistent=True)
        _fire_grad_hooks(w2)
        if not ctx.defer_out_bias:
            colsum_accum(g2, b2.grad)
            _fire_grad_hooks(b2)
        _wgrad_into(w1.grad, dh, x2, persistent=True)
        _fire_grad_hooks(w1)
        gx = None
        if ctx.needs_input_grad[0]:
            if gemm16_policy() == 'all':
                gx = gemm16_nn(dh, w1c)
            elif ctx.x_f32:        # an f32 input (post-LN residual stream) takes its gradient in f32: no 16-bit round trip + cast
                gx = torch.mm(dh, w1c, out_dtype=torch.float32)
            else:
                gx = dh.mm(w1c)
            gx = gx.view(ctx.xshape)
        return gx, None, None, None, None, None, None


class _FFN32(torch.autograd.Function):
    """``fc2(act(fc1(x)))`` of an mmcv FFN in fp32 compute on K20 (csrc/gemm_f32s.hip): forward, fc1 + bias + activation in one
    launch (stores the activation and the pre-activation, leaves the activation's absmax record for fc2); backward, the data
    gradient of fc2 times the activation's derivative with the partial column sums of the result (= d bias of fc1) in one
    launch — no activation kernels, no pass over the hidden gradient — then the two weight gradients and fc1's data gradient.
    /root/reference: mask_bev/models/networks/swin/swin.py:347-355."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, kind, defer_out_bias):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        hx = amax_hint_get(x2) if switches.get('amax_hints') else None
        if hx is None:
            both = f32_absmax([x2, w1])
            ax, aw1 = both[0:1], both[1:2]
        else:
            ax, aw1 = hx, weight_amax(w1)
        a, h = gemm32s_nt(x2, w1, b1, act=kind, amax=(ax, aw1), want_pre=True, hint_out=True)
        aa = amax_hint_get(a)
        aw2 = weight_amax(w2)
        out = gemm32s_nt(a, w2, b2, amax=(aa, aw2), hint_out=bool(switches.get('amax_hints')))
        ctx.save_for_backward(x2, h, a)
        ctx.params = (w1, b1, w2, b2)
        ctx.amax = (ax, aw1, aa, aw2)
        ctx.kind, ctx.defer_out_bias, ctx.xshape = kind, defer_out_bias, x.shape
        y = out.view(x.shape[:-1] + (w2.shape[0],))
        amax_hint_set(y, amax_hint_get(out))
        return y

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        x2, h, a = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        ax, aw1, aa, aw2 = ctx.amax
        g2 = gout.reshape(-1, gout.shape[-1])
        if g2.dtype != torch.float32:
            g2 = g2.float()
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        ag = amax_hint_get(g2) if switches.get('amax_hints') else None
        if ag is None:
            ag = f32_absmax([g2])
        t, f = h.shape
        # d hidden = (g . W2) * act'(pre), its partial column sums -> d b1, its absmax record for the products below
        dh = torch.empty_like(h)
        rows = lib.mbv_gemm32s_nn_part_rows(t, 1)
        parts = torch.empty((rows, f), dtype=torch.float32, device=h.device)
        adh = amax_record(h.device)
        AMAX_VERIFY.check(g2, ag, 'gemm32s_nn_act g')
        AMAX_VERIFY.check(w2, aw2, 'gemm32s_nn_act w2')
        check(lib.mbv_gemm32s_nn_act(_ptr(g2), _ptr(w2), _ptr(dh), _ptr(h), _ptr(parts), parts.numel() * 4, t, w2.shape[0], f,
                                     g2.stride(0), w2.stride(0), f, f, _ptr(ag), _ptr(aw2), _ptr(adh), _ACT[ctx.kind],
                                     _stream()), 'mbv_gemm32s_nn_act')
        if not _defer_colsum(parts, b1.grad, rows, f, f):
            _colsum_now(parts, b1.grad, rows, f, f)
        _fire_grad_hooks(b1)
        _wgrad_into(w2.grad, g2, a, persistent=True, amax=(ag, aa))
        _fire_grad_hooks(w2)
        if not ctx.defer_out_bias:
            colsum_accum(g2, b2.grad, persistent=True)
            _fire_grad_hooks(b2)
        _wgrad_into(w1.grad, dh, x2, persistent=True, amax=(adh, ax))
        _fire_grad_hooks(w1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx2 = gemm32s_nn(dh, w1, adh, aw1, hint_out=bool(switches.get('amax_hints')))
            gx = _hinted_view(gx2, ctx.xshape)
        return gx, None, None, None, None, None, None


def ffn32_ok(x: torch.Tensor, fc1_w, fc1_b, fc2_w, fc2_b) -> bool:
    """The K20 FFN applies: fp32 compute (no autocast) on the device, arena-resident f32 parameters with f32 gradients, a token
    count K20 takes, shapes in 8-element chunks."""
    if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and torch.is_grad_enabled()
            and switches.get('gemm32s') and switches.get('ffn32')):
        return False
    rows = x.numel() // max(1, x.shape[-1])
    if not gemm32s_wants(rows) or x.shape[-1] % 8:
        return False
    for p in (fc1_w, fc1_b, fc2_w, fc2_b):
        if (p is None or p.dtype != torch.float32 or not getattr(p, '_mbv_arena', False) or p.grad is None
                or p.grad.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 16
                or not p.grad.is_contiguous()):
            return False
    return fc1_w.shape[0] % 8 == 0 and fc1_w.shape[1] % 8 == 0 and fc2_w.shape[0] % 8 == 0


def ffn32(x: torch.Tensor, fc1_w, fc1_b, fc2_w, fc2_b, kind: str, defer_out_bias: bool = False) -> torch.Tensor:
    _LAST_HINT[1] = None
    y = _FFN32.apply(x, fc1_w, fc1_b, fc2_w, fc2_b, kind, defer_out_bias)
    amax_hint_refresh(y)
    return y


def ffn_fused_ok(x: torch.Tensor, fc1_w, fc1_b, fc2_w, fc2_b) -> bool:
    """The fused FFN (K17 epilogues) applies: 16-bit autocast on a ROCm device, arena-resident parameters with f32
    gradients, token count in K17's range, 16-byte-chunk shapes."""
    if not (x.is_cuda and torch.is_autocast_enabled('cuda') and torch.is_grad_enabled()):
        return False
    dt = torch.get_autocast_dtype('cuda')
    rows = x.numel() // max(1, x.shape[-1])
    if dt not in _GEMM16_DT or not _k17_wants('fused', rows):
        return False
    for p in (fc1_w, fc1_b, fc2_w, fc2_b):
        if p is None or not getattr(p, '_mbv_arena', False) or p.grad is None or p.grad.dtype != torch.float32:
            return False
        sh = getattr(p, '_mbv_shadow', None)
        if p.dim() == 2 and (sh is None or sh.dtype != dt):
            return False
    return fc1_w.shape[0] % 8 == 0 and fc1_w.shape[1] % 8 == 0 and fc2_w.shape[0] % 8 == 0


def ffn(x: torch.Tensor, fc1_w, fc1_b, fc2_w, fc2_b, kind: str, defer_out_bias: bool = False) -> torch.Tensor:
    """``fc2(act(fc1(x)))`` through :class:`_FFN` (check :func:`ffn_fused_ok` first).  ``defer_out_bias``: the
    consumer of the result (K12 with ``branch_bias``) accumulates d b2."""
    return _FFN.apply(x, fc1_w, fc1_b, fc2_w, fc2_b, kind, defer_out_bias)


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
