"""K12 residual-add + LayerNorm, bias + activation, position tokens, 1 x 1 convolutions on token rows, K18 GroupNorm,
patch-merging LayerNorm (swin.py:357-377,611-616; the pixel decoder's ConvModules)."""
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
# K12 fused residual-add + LayerNorm
# --------------------------------------------------------------------------------------
def add_layernorm_supported(channels: int) -> bool:
    return channels % 4 == 0 and 0 < channels <= 2048


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight, bias, eps, out_dtype, branch_bias=None, fanout=False, branch_dtype=None):
        lib = _lib.load()
        _need_gpu(a, b, weight, bias)
        ctx.branch_bias = branch_bias
        c = a.shape[-1]
        ok = _ACT_DTYPES
        if a.dtype not in ok or (b is not None and b.dtype not in ok) or out_dtype not in ok:
            raise MaskBevHipError('add_layernorm supports f32, bf16 and fp16 activations')
        if weight.dtype != torch.float32 or bias.dtype != torch.float32:
            raise MaskBevHipError('add_layernorm: f32 affine parameters')
        a2 = a.contiguous()
        b_rows = 0
        if b is not None and b.shape != a.shape:
            raise MaskBevHipError('add_layernorm: a and b must have the same shape')
        if b is not None and b.dim() >= 2 and b.shape[0] > 1 and b.stride(0) == 0 and b[0].is_contiguous():
            # one per-sample map expanded over the batch (ops.pos_tokens): read with its row index modulo, never materialised;
            # the gradient it gets back is the full-batch dx — the expanding op reduces it
            b2 = b[0]
            b_rows = b2.numel() // c
        else:
            b2 = None if b is None else b.contiguous()
        rows = a2.numel() // c
        need_sum = b2 is not None or a2.dtype != torch.float32
        s = torch.empty(a2.shape, dtype=torch.float32, device=a.device) if need_sum else None
        y = torch.empty(a2.shape, dtype=out_dtype, device=a.device)
        mean = torch.empty(rows, dtype=torch.float32, device=a.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
        w, bb = weight.contiguous(), bias.contiguous()
        # fanout with a branch dtype: the branch consumer's copy of y is written in ITS storage type by the same launch
        y_branch = None
        if fanout and branch_dtype is not None and branch_dtype != out_dtype:
            if branch_dtype not in ok:
                raise MaskBevHipError('add_layernorm: branch dtype must be f32, bf16 or fp16')
            y_branch = torch.empty(a2.shape, dtype=branch_dtype, device=a.device)
        check(lib.mbv_add_layernorm_fwd2(_ptr(a2), _dt_flag(a2.dtype), _ptr(b2),
                                         (_dt_flag(b2.dtype) if b2 is not None else 0), b_rows, _ptr(w), _ptr(bb), rows, c,
                                         float(eps), _ptr(s), _ptr(y), _dt_flag(out_dtype), _ptr(y_branch),
                                         _dt_flag(branch_dtype) if y_branch is not None else 0, _ptr(mean),
                                         _ptr(rstd), _stream()), 'mbv_add_layernorm_fwd2')
        ctx.save_for_backward(a2 if s is None else s, mean, rstd, w)
        ctx.weight, ctx.bias = weight, bias
        ctx.dtypes = (a.dtype, None if b is None else b.dtype)
        ctx.set_materialize_grads(False)
        # fanout: y leaves as two tensors over one buffer — one per consumer (the next residual add, the next branch) — so
        # that their gradients come back separately and K12's backward adds them on load (no autograd add launch)
        if y_branch is not None:
            return y, s, y_branch
        return y, s, (y.view_as(y) if fanout else None)   # s is None for a lone f32 input (it IS the input)

    @staticmethod
    def backward(ctx, gy, gs, gy2=None):
        lib = _lib.load()
        s, mean, rstd, w = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        da, db = ctx.dtypes
        if gy is None and gy2 is not None:
            gy, gy2 = gy2, None
        if gy is None:                                    # only the residual path carries gradient
            bb = ctx.branch_bias
            if gs is not None and bb is not None:         # the deferred bias gradient of the branch Linear: colsum(gs)
                g2 = gs.reshape(-1, gs.shape[-1])
                colsum_accum(g2 if g2.dtype in _ACT_DTYPES else g2.float(), bb.grad)
                _fire_grad_hooks(bb)
            ga = None if gs is None else gs.to(da)
            gb = None if (gs is None or db is None) else gs.to(db)
            return ga, gb, None, None, None, None, None, None, None
        c = s.shape[-1]
        rows = s.numel() // c
        gy = gy.contiguous()
        if gy.dtype not in _ACT_DTYPES:
            gy = gy.float()
        if gy2 is not None:
            gy2 = gy2.contiguous()
            if gy2.dtype not in _ACT_DTYPES:
                gy2 = gy2.float()
        if gs is not None:
            gs = gs.contiguous()
            if gs.dtype not in _ACT_DTYPES:
                gs = gs.float()
        dx = torch.empty(s.shape, dtype=torch.float32, device=s.device)
        lo = da if da in _LO_DTYPES else (db if db in _LO_DTYPES else None)      # a and b share their 16-bit type
        dx_lo = torch.empty(s.shape, dtype=lo, device=s.device) if lo is not None else None
        direct = (getattr(weight, '_mbv_arena', False) and getattr(bias, '_mbv_arena', False)
                  and weight.grad is not None and bias.grad is not None
                  and weight.grad.dtype == torch.float32 and bias.grad.dtype == torch.float32)
        if direct:
            dgamma, dbeta = weight.grad, bias.grad
        else:
            dgamma = torch.empty(c, dtype=torch.float32, device=s.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=s.device)
        nblk = lib.mbv_add_layernorm_bwd_blocks(rows, c)
        bb = ctx.branch_bias             # arena gradient of the Linear bias that produced b: += colsum(dx)
        ws = torch.empty(max(1, nblk * 3 * c), dtype=torch.float32, device=s.device)
        # arena gradients: the per-block partial rows of the large LayerNorms join the grouped column-sum launch at the
        # end of the backward pass instead of one reduction launch per layer
        np_ = 3 if bb is not None else 2
        defer = bool(direct and not lib.mbv_add_layernorm_bwd_direct(rows, c) and _defer_ok())
        # fp32 compute: dx is the output gradient of a Linear backward on K20 — its absmax record from this launch
        rec = (amax_record(s.device) if (lo is None and switches.get('amax_hints') and switches.get('ln_bound_hints')
                                         and amax_hint_wanted(rows)) else None)
        check(lib.mbv_add_layernorm_bwd3(_ptr(gy), _dt_flag(gy.dtype), _ptr(gy2),
                                         (_dt_flag(gy2.dtype) if gy2 is not None else 0), _ptr(gs),
                                         (_dt_flag(gs.dtype) if gs is not None else 0), _ptr(s), _ptr(mean),
                                         _ptr(rstd), _ptr(w), rows, c, _ptr(dx), _ptr(dx_lo),
                                         _dt_flag(lo) if lo is not None else 0, _ptr(dgamma), _ptr(dbeta),
                                         1 if direct else 0, _ptr(None if bb is None else bb.grad), _ptr(ws),
                                         1 if defer else 0, _ptr(rec), _stream()),
              'mbv_add_layernorm_bwd3')
        amax_hint_set(dx, rec)
        if defer:
            for j, dst in enumerate((dgamma, dbeta, None if bb is None else bb.grad)[:np_]):
                if not _defer_colsum(ws, dst, nblk, c, np_ * c, offset=j * c):
                    _colsum_now(ws, dst, nblk, c, np_ * c, offset=j * c)

        # (the branch Linear's own backward, which runs after this one, announces its bias gradient to the hooks)
        if direct:
            _fire_grad_hooks(weight)
            _fire_grad_hooks(bias)
            dgamma = dbeta = None
        else:
            dgamma, dbeta = dgamma.to(weight.dtype), dbeta.to(bias.dtype)
        ga = dx_lo if da in _LO_DTYPES else dx
        gb = None if db is None else (dx_lo if db in _LO_DTYPES else dx)
        return ga, gb, dgamma, dbeta, None, None, None, None, None


class _BiasAct(torch.autograd.Function):
    """act(z) (ReLU / erf-GELU) whose backward also accumulates the bias gradient of the Linear that produced z
    (K11 ``mbv_act_bwd_colsum``): one pass computes dz and its column sums."""

    @staticmethod
    def forward(ctx, z, bias, kind):
        ctx.bias, ctx.kind = bias, kind
        ctx.save_for_backward(z)
        out = torch.nn.functional.gelu(z) if kind == 1 else torch.relu(z)
        amax_hint_set(out, amax_hint_get(z))             # |gelu(z)|, |relu(z)| <= |z|: z's absmax record bounds the output
        return out

    @staticmethod
    def backward(ctx, ga):
        lib = _lib.load()
        (z,) = ctx.saved_tensors
        n = z.shape[-1]
        zc = z.contiguous()
        ga = ga.to(zc.dtype).contiguous()
        gz = torch.empty_like(zc)
        bias = ctx.bias
        check(lib.mbv_act_bwd_colsum(_ptr(ga), _ptr(zc), _dt_flag(zc.dtype), ctx.kind, zc.numel() // n, n,
                                     _ptr(gz), _ptr(None if bias is None else bias.grad), _stream()),
              'mbv_act_bwd_colsum')
        hg = amax_hint_get(ga) if gz.dtype == torch.float32 else None
        if hg is not None:
            # |act'| <= 1.13 (GELU) / 1 (ReLU): twice ga's bound bounds gz — one more binade in the record (64 words, one tiny launch)
            amax_hint_set(gz, hg + (1 << 23))
        if bias is not None:
            _fire_grad_hooks(bias)
        return gz, None, None


def bias_act(z: torch.Tensor, bias: Optional[torch.Tensor], kind: str) -> torch.Tensor:
    """``relu`` / ``gelu`` of a Linear output ``z``.  When ``bias`` (that Linear's arena-resident bias, the layer
    having been run with ``skip_bias_grad=True``) is given, the backward accumulates its gradient while it computes
    dz.  Falls back to the torch activation for shapes / dtypes the kernel does not take."""
    k = 1 if kind == 'gelu' else 0
    ok = (z.is_cuda and z.dtype in _ACT_DTYPES and z.shape[-1] % 4 == 0 and z.requires_grad)
    if not ok:
        if bias is not None and z.requires_grad:
            z = accumulate_bias_grad(z, bias)          # the deferred bias gradient must not be lost: dz reaches it here
        out = torch.nn.functional.gelu(z) if k == 1 else torch.relu(z)
        amax_hint_set(out, amax_hint_get(z))
        return out
    _LAST_HINT[1] = None
    out = _BiasAct.apply(z, bias, k)
    amax_hint_refresh(out)
    return out


class _AccumulateBiasGrad(torch.autograd.Function):
    """Identity whose backward adds the column sums of the gradient to ``bias.grad`` (the safety net for a bias
    gradient that was deferred to a K12 call which then took the non-fused path)."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.bias = bias
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, g.shape[-1])
        if g2.dtype not in _ACT_DTYPES:
            g2 = g2.float()
        colsum_accum(g2, ctx.bias.grad)
        _fire_grad_hooks(ctx.bias)
        return g, None


def accumulate_bias_grad(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return _AccumulateBiasGrad.apply(x, bias)


def bias_grad_deferrable(bias: Optional[torch.Tensor], channels: int) -> bool:
    """True when a Linear may leave its bias gradient to the K12 op that consumes its output as the residual branch:
    the bias lives in a parameter arena (so K12 can accumulate into its gradient) and K12 supports the width."""
    return (bias is not None and getattr(bias, '_mbv_arena', False) and bias.grad is not None
            and bias.grad.dtype == torch.float32 and bias.grad.is_contiguous() and bias.is_cuda
            and add_layernorm_supported(channels) and torch.is_grad_enabled())


def add_layernorm(a: torch.Tensor, b: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor,
                  eps: float = 1e-5, out_dtype: Optional[torch.dtype] = None, return_sum: bool = False,
                  branch_bias: Optional[torch.Tensor] = None, fanout: bool = False,
                  branch_dtype: Optional[torch.dtype] = None):
    """``y = LayerNorm_C(a + b)`` over the last axis in one pass (K12); ``b=None`` is a plain LayerNorm.
    ``out_dtype`` (default: the autocast dtype when autocast is on and the consumer is a GEMM — pass it explicitly —
    else f32) is the storage type of y; statistics and the sum are f32.  With ``return_sum`` the f32 sum ``a + b``
    (the new residual stream of a pre-LN block) is returned as well: ``(y, s)``.  ``fanout`` (post-LN layers, instead
    of ``return_sum``): returns ``(y, y')`` — the same values as two tensors, one for each of y's two consumers, whose
    gradients the backward kernel then adds on load instead of autograd adding them with a launch of its own; with
    ``branch_dtype`` y' is stored in that type (the 16-bit input of the branch GEMM) by the same launch."""
    if out_dtype is None:
        out_dtype = torch.float32
    y, s, y2 = _AddLayerNorm.apply(a, b, weight, bias, eps, out_dtype, branch_bias, fanout, branch_dtype)
    if (out_dtype == torch.float32 and y.is_cuda and switches.get('amax_hints') and switches.get('ln_bound_hints')
            and not torch.is_autocast_enabled('cuda') and amax_hint_wanted(y.numel() // y.shape[-1])):
        # fp32 compute: the consuming K20 product takes its scale from the LayerNorm's parameters, not from a pass over y
        rec = ln_bound(weight, bias)
        amax_hint_set(y, rec)
        if fanout and y2 is not None and y2.dtype == torch.float32:
            amax_hint_set(y2, rec)
    if fanout:
        return y, y2
    return (y, a if s is None else s) if return_sum else y


class _PosTokens(torch.autograd.Function):
    """The (1, C, H, W) absolute position embedding as (B, H, W, C) tokens: ONE transposed (1, H, W, C) copy seen through a
    stride-0 batch axis (K12 adds it to the patch tokens inside the first block's LayerNorm launch without materialising
    it), whose backward takes the full-batch gradient and accumulates its batch sum, transposed back, into the parameter's
    gradient in one pass — instead of a broadcast add forward and a batch reduction + a transposed accumulate backward.
    /root/reference: mask_bev/models/networks/swin/swin.py:579-586 (parameter), :750-760 (the add)."""

    @staticmethod
    def forward(ctx, ape, batch, h, w):
        c = int(ape.shape[1])           # the reference flattens the (rows, cols) map row-major into h * w tokens, whatever they are
        if int(ape.shape[2]) * int(ape.shape[3]) != h * w:
            raise MaskBevHipError('pos_tokens: the embedding has another number of positions')
        ctx.ape = ape
        ctx.dims = (int(batch), c, h, w)
        t = ape.detach().flatten(2).transpose(1, 2).reshape(1, h, w, c).contiguous()
        return t.expand(int(batch), h, w, c)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        ape = ctx.ape
        b, c, h, w = ctx.dims
        g = g.float().contiguous()
        direct = (getattr(ape, '_mbv_arena', False) and ape.grad is not None and ape.grad.dtype == torch.float32
                  and ape.grad.is_contiguous())
        acc = ape.grad if direct else torch.zeros((1, c, h * w), dtype=torch.float32, device=g.device)
        check(lib.mbv_transposed_batch_sum_accum(_ptr(g), b, h * w, c, _ptr(acc), _stream()),
              'mbv_transposed_batch_sum_accum')
        if direct:
            _fire_grad_hooks(ape)
            return None, None, None, None
        return acc.view(ape.shape).to(ape.dtype), None, None, None


def pos_tokens(ape: torch.Tensor, batch: int, h: int, w: int) -> torch.Tensor:
    """ape (1, C, rows, cols) → (batch, h, w, C) tokens (rows * cols == h * w) over a stride-0 batch axis (:class:`_PosTokens`)."""
    _need_gpu(ape)
    return _PosTokens.apply(ape, int(batch), int(h), int(w))


_ONES: dict = {}


def _ones_block(n: int, dtype, device) -> torch.Tensor:
    """Cached (n, 8) block of ones (a row-sum as a GEMM: eight identical columns keep the product off the GEMV paths)."""
    key = (int(n), dtype, str(device))
    t = _ONES.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise MaskBevHipError('conv1x1_tokens: first use inside a stream capture (run one eager step first)')
        t = _ONES[key] = torch.ones((n, 8), dtype=dtype, device=device)
        torch.cuda.current_stream(device).synchronize()      # (read by every stream from here on)
    return t


class _Conv1x1Tokens(torch.autograd.Function):
    """A 1 x 1 convolution of a CHANNELS-LAST map handed over as tokens: ``y (B, Cout, HW) = W (Cout, Cin) · x[b]^T + bias``
    for ``x (B, HW, Cin)`` — the backbone's stage outputs are token-major and the pixel decoder's ConvModules want NCHW,
    and the GEMM does that turn for free (the token matrix is the transposed operand), forward and backward:
    ``dx (B, HW, Cin) = dy[b]^T · W`` arrives token-major again.  No (B, C, H, W) copy of the stage outputs either way
    (four permute copies forward, four backward: 0.28 ms of a 29 ms step).  Library GEMMs (torch.bmm); the operands are
    cast to the autocast dtype, the gradient of x returns in x's dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
        with torch.autocast('cuda', enabled=False):
            xc = x.to(dt)
            wc = _compute_copy(weight, dt)
            b = x.shape[0]
            w3 = wc.unsqueeze(0).expand(b, -1, -1)
            if bias is None:
                y = torch.bmm(w3, xc.transpose(1, 2))
            else:
                y = torch.baddbmm(_compute_copy(bias, dt).view(1, -1, 1), w3, xc.transpose(1, 2))
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        x_dtype, w_dtype, b_dtype = ctx.meta
        gy = gy.to(xc.dtype).contiguous()
        b = xc.shape[0]
        od = {} if xc.dtype == torch.float32 else dict(out_dtype=torch.float32)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.bmm(gy.transpose(1, 2), wc.unsqueeze(0).expand(b, -1, -1), **od).to(x_dtype)
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(gy, xc, **od).sum(0).to(w_dtype)
        if b_dtype is not None and ctx.needs_input_grad[2]:
            # bias gradient = sum over (batch, pixels) per channel — as a product with a ones block, NOT `gy.sum((0, 2))`:
            # ATen reduces 65 536 elements per output in several workgroups that meet through a semaphore which it clears
            # with a memset, and inside a captured HIP graph that pair replays with STALE results on this stack
            # (scratch/dbg_graph_reduce.py).  The batch sum behind it is 4 elements per output: one block.
            ones = _ones_block(gy.shape[2], gy.dtype, gy.device)
            gb = torch.bmm(gy, ones.unsqueeze(0).expand(b, -1, -1), **od)[:, :, 0].sum(0).to(b_dtype)
        return gx, gw, gb


def conv1x1_tokens(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x (B, HW, Cin) tokens, weight (Cout, Cin) → (B, Cout, HW)."""
    return _Conv1x1Tokens.apply(x, weight, bias)


class _Conv1x1Rows(torch.autograd.Function):
    """The same 1 x 1 convolution for an NCHW-contiguous map, ``y (B, Cout, HW) = W · x[b] + bias`` with x (B, Cin, HW), as ONE
    node (round 6).  As plain ``torch.baddbmm`` the broadcast bias's gradient was autograd's ``sum`` of the (B, Cout, HW)
    gradient down to (1, Cout, 1) — 65 536 elements per output at the mask-feature projection: a multi-workgroup ATen reduction,
    which replays with stale results inside a captured HIP graph on this stack (see `_Conv1x1Tokens.backward`)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else x.dtype
        with torch.autocast('cuda', enabled=False):
            xc = x.to(dt)
            wc = _compute_copy(weight, dt)
            b = x.shape[0]
            w3 = wc.unsqueeze(0).expand(b, -1, -1)
            y = torch.bmm(w3, xc) if bias is None else torch.baddbmm(_compute_copy(bias, dt).view(1, -1, 1), w3, xc)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        x_dtype, w_dtype, b_dtype = ctx.meta
        gy = gy.to(xc.dtype).contiguous()
        b = xc.shape[0]
        od = {} if xc.dtype == torch.float32 else dict(out_dtype=torch.float32)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.bmm(wc.t().unsqueeze(0).expand(b, -1, -1), gy).to(x_dtype)
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(gy, xc.transpose(1, 2), **od).sum(0).to(w_dtype)
        if b_dtype is not None and ctx.needs_input_grad[2]:
            ones = _ones_block(gy.shape[2], gy.dtype, gy.device)
            gb = torch.bmm(gy, ones.unsqueeze(0).expand(b, -1, -1), **od)[:, :, 0].sum(0).to(b_dtype)
        return gx, gw, gb


def conv1x1_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x (B, Cin, HW) (an NCHW map, flattened), weight (Cout, Cin) → (B, Cout, HW)."""
    return _Conv1x1Rows.apply(x, weight, bias)


class _GroupNorm(torch.autograd.Function):
    """K18: ``y = GroupNorm(x) [+ bilinear-upsampled add] [ReLU]`` on an NCHW map, stored in ``out_dtype``."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, add, out_dtype):
        lib = _lib.load()
        _need_gpu(x, weight, bias)
        b, c, h, w = x.shape
        if (x.dtype not in _ACT_DTYPES or out_dtype not in _ACT_DTYPES or weight.dtype != torch.float32
                or bias.dtype != torch.float32 or not lib.mbv_groupnorm_supported(c, groups, h, w)):
            raise MaskBevHipError('group_norm: (B, C, H, W) f32 / bf16 / fp16 map with H*W % 4 == 0, f32 parameters')
        x2 = x.contiguous()
        add2 = None
        if add is not None:
            if add.dim() != 4 or add.shape[:2] != x.shape[:2] or add.dtype not in _ACT_DTYPES or w % 4:
                raise MaskBevHipError('group_norm: the added map must be (B, C, h, w) and W % 4 == 0')
            add2 = add.contiguous()
        y = torch.empty((b, c, h, w), dtype=out_dtype, device=x.device)
        mean = torch.empty(b * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty(b * groups, dtype=torch.float32, device=x.device)
        wc, bc = weight.contiguous(), bias.contiguous()
        nbytes = lib.mbv_groupnorm_workspace_bytes(b, c, groups, h, w)
        ws = _workspace(nbytes, x.device)
        check(lib.mbv_groupnorm_fwd(_ptr(x2), _dt_flag(x2.dtype), b, c, h, w, groups, _ptr(wc), _ptr(bc), float(eps),
                                    _ptr(add2), _dt_flag(add2.dtype) if add2 is not None else 0,
                                    add2.shape[2] if add2 is not None else 0, add2.shape[3] if add2 is not None else 0,
                                    1 if relu else 0, _ptr(y), _dt_flag(out_dtype), _ptr(mean), _ptr(rstd), _ptr(ws),
                                    int(nbytes), _stream()), 'mbv_groupnorm_fwd')
        ctx.save_for_backward(x2, mean, rstd, wc, bc)
        ctx.weight, ctx.bias = weight, bias
        ctx.meta = (groups, bool(relu), None if add is None else (tuple(add.shape), add.dtype), x.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, mean, rstd, w, bvec = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        groups, relu, add_meta, x_dtype = ctx.meta
        b, c, h, wd = x.shape
        gy = gy.contiguous()
        if gy.dtype not in _ACT_DTYPES:
            gy = gy.float()
        dx = torch.empty_like(x)
        direct = (getattr(weight, '_mbv_arena', False) and getattr(bias, '_mbv_arena', False)
                  and weight.grad is not None and bias.grad is not None
                  and weight.grad.dtype == torch.float32 and bias.grad.dtype == torch.float32)
        if direct:
            dgamma, dbeta = weight.grad, bias.grad
        else:
            dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        sums = torch.empty(b * c * 2, dtype=torch.float32, device=x.device)
        check(lib.mbv_groupnorm_bwd(_ptr(gy), _dt_flag(gy.dtype), _ptr(x), _dt_flag(x.dtype), _ptr(mean), _ptr(rstd),
                                    _ptr(w), _ptr(bvec), b, c, h, wd, groups, 1 if relu else 0, _ptr(dx),
                                    _dt_flag(dx.dtype), _ptr(dgamma), _ptr(dbeta), 1 if direct else 0, _ptr(sums),
                                    _stream()), 'mbv_groupnorm_bwd')
        if direct:
            _fire_grad_hooks(weight)
            _fire_grad_hooks(bias)
            dgamma = dbeta = None
        else:
            dgamma, dbeta = dgamma.to(weight.dtype), dbeta.to(bias.dtype)
        gadd = None
        if add_meta is not None and ctx.needs_input_grad[6]:
            shape, adt = add_meta                 # the added map entered through F.interpolate(bilinear, align_corners=False)
            gadd = torch.empty(shape, dtype=adt, device=gy.device)
            check(lib.mbv_upsample_bilinear_bwd(_ptr(gy), _dt_flag(gy.dtype), int(shape[0]) * int(shape[1]), h, wd,
                                                int(shape[2]), int(shape[3]), _ptr(gadd), _dt_flag(adt), _stream()),
                  'mbv_upsample_bilinear_bwd')
        return dx, dgamma, dbeta, None, None, None, gadd, None


def group_norm_supported(x: torch.Tensor, groups: int) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.dtype in _ACT_DTYPES and x.shape[1] % groups == 0
            and (x.shape[2] * x.shape[3]) % 4 == 0 and switches.get('groupnorm'))


def group_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, groups: int, eps: float = 1e-5,
               relu: bool = False, add_upsampled: Optional[torch.Tensor] = None,
               out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``relu?(GroupNorm(x) + interpolate(add_upsampled, size=x.shape[-2:], mode='bilinear', align_corners=False))`` for
    an NCHW map in two passes over ``x`` (K18); ``out_dtype`` (default f32) is the storage type of the result."""
    return _GroupNorm.apply(x, weight, bias, int(groups), float(eps), bool(relu), add_upsampled,
                            out_dtype or torch.float32)


class _MergeLayerNorm(torch.autograd.Function):
    """LayerNorm_{4C}(unfold_{2x2, stride 2}(x)) for a channels-last f32 (B, H, W, C) map, gathered / scattered by K12's
    addressing (mbv_merge_layernorm_*): the unfolded copy never exists, forward or backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        lib = _lib.load()
        _need_gpu(x, weight, bias)
        b, h, w, c = x.shape
        if x.dtype != torch.float32 or weight.dtype != torch.float32 or bias.dtype != torch.float32 \
                or out_dtype not in _ACT_DTYPES or not lib.mbv_merge_layernorm_supported(h, w, c):
            raise MaskBevHipError('merge_layernorm: f32 (B, H, W, C) map with even H, W and 4C <= 2048, f32 parameters')
        x2 = x.contiguous()
        rows = b * (h // 2) * (w // 2)
        y = torch.empty((b, h // 2, w // 2, 4 * c), dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        wc, bc = weight.contiguous(), bias.contiguous()
        check(lib.mbv_merge_layernorm_fwd(_ptr(x2), b, h, w, c, _ptr(wc), _ptr(bc), float(eps), _ptr(y),
                                          _dt_flag(out_dtype), _ptr(mean), _ptr(rstd), _stream()),
              'mbv_merge_layernorm_fwd')
        ctx.save_for_backward(x2, mean, rstd, wc)
        ctx.weight, ctx.bias = weight, bias
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, mean, rstd, w = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        b, h, wd, c = x.shape
        c4 = 4 * c
        rows = mean.numel()
        gy = gy.contiguous()
        if gy.dtype not in _ACT_DTYPES:
            gy = gy.float()
        dx = torch.empty_like(x)
        direct = (getattr(weight, '_mbv_arena', False) and getattr(bias, '_mbv_arena', False)
                  and weight.grad is not None and bias.grad is not None
                  and weight.grad.dtype == torch.float32 and bias.grad.dtype == torch.float32)
        if direct:
            dgamma, dbeta = weight.grad, bias.grad
        else:
            dgamma = torch.empty(c4, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(c4, dtype=torch.float32, device=x.device)
        nblk = lib.mbv_add_layernorm_bwd_blocks(rows, c4)
        ws = torch.empty(max(1, nblk * 2 * c4), dtype=torch.float32, device=x.device)
        defer = bool(direct and not lib.mbv_add_layernorm_bwd_direct(rows, c4) and _defer_ok())
        check(lib.mbv_merge_layernorm_bwd(_ptr(gy), _dt_flag(gy.dtype), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w), b, h,
                                          wd, c, _ptr(dx), _ptr(dgamma), _ptr(dbeta), 1 if direct else 0, _ptr(ws),
                                          1 if defer else 0, _stream()), 'mbv_merge_layernorm_bwd')
        if defer:
            for j, dst in enumerate((dgamma, dbeta)):
                if not _defer_colsum(ws, dst, nblk, c4, 2 * c4, offset=j * c4):
                    _colsum_now(ws, dst, nblk, c4, 2 * c4, offset=j * c4)
        if direct:
            _fire_grad_hooks(weight)
            _fire_grad_hooks(bias)
            dgamma = dbeta = None
        else:
            dgamma, dbeta = dgamma.to(weight.dtype), dbeta.to(bias.dtype)
        return dx, dgamma, dbeta, None, None


def merge_layernorm_supported(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
            and bool(_lib.load().mbv_merge_layernorm_supported(int(x.shape[1]), int(x.shape[2]), int(x.shape[3])))
            and switches.get('merge_ln'))


def merge_layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
                    out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(B, H, W, C) f32 → (B, H/2, W/2, 4C): the 2 x 2 neighbourhood concat of patch merging (channel order
    ``c*4 + kh*2 + kw``) and its LayerNorm in one pass (K12 with gather addressing)."""
    y = _MergeLayerNorm.apply(x, weight, bias, eps, out_dtype or torch.float32)
    if (y.dtype == torch.float32 and y.is_cuda and switches.get('amax_hints') and switches.get('ln_bound_hints')
            and not torch.is_autocast_enabled('cuda') and amax_hint_wanted(y.numel() // y.shape[-1])):
        amax_hint_set(y, ln_bound(weight, bias))
    return y


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
