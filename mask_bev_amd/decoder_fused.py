"""Fused query side of the Mask2Former transformer decoder (K19 row chains, csrc/rowchain.hip).

One decoder layer of /root/reference: mask_bev/models/networks/mask2former_head/mask2former_head.py:535-560 (mmdet
``Mask2FormerTransformerDecoderLayer``: masked cross-attention -> LN -> self-attention -> LN -> FFN -> LN, post-norm) plus
the prediction heads of ``_forward_head`` (:428-472) is, on the query side, a handful of few-row Linear / LayerNorm / ReLU
ops on B*Q = 400 tokens between attention calls.  Here each stretch between two attention kernels is ONE launch:

    _DecA:  O1 (cross-attention output) -> out-proj + residual + LN1 -> self-attention q / k / v projections
            -> [K6 self-attention]                                             returns (x1, O2)
    _DecB:  O2 -> out-proj + residual + LN2 -> FFN (ReLU) + residual + LN3 -> post-norm, class / mask-embed heads
            -> [K7 mask logits + next layer's attention mask] -> next layer's query projection
            -> [K6 masked cross-attention of the next layer]                    returns (x3, O1')

Both are autograd Functions whose backward is one row-chain launch (+ the attention kernel's backward); the weight
gradients leave as the (dY, X) pairs of the pass's grouped launches (ops._wgrad_into), bias and LayerNorm-parameter
gradients as per-block partial rows reduced by the pass's grouped column sums.  Data gradients ``dX = dY W`` are NT
products against transposed copies of the weights refreshed once per step by one grouped transpose launch.

GEMM operand types: f32 compute -> the f32 master weights on exact-f32 MFMA; bf16 / fp16 compute -> the 16-bit weight
copies, activations rounded to that type per GEMM, f32 accumulation (what autocast gives the reference's Linears).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib, ops, switches
from ._lib import MaskBevHipError, check

OP_LOAD, OP_STORE, OP_GEMM, OP_LN, OP_LN_BWD, OP_ADD, OP_COLSUM, OP_FFN, OP_FFN_IO, OP_SUM = range(10)
F_ACCUM, F_RELU, F_MASK, F_SAVE_SUM, F_FRAG = 4, 8, 16, 32, 128
F_SLICE = F_SPLIT = 64
ROWS = 16


class RowStage(ctypes.Structure):
    """``MbvRowStage`` of include/maskbev_hip.h."""
    _fields_ = [('op', ctypes.c_int32), ('dst', ctypes.c_int16), ('src', ctypes.c_int16), ('src2', ctypes.c_int16),
                ('reserved', ctypes.c_int16), ('n', ctypes.c_int32), ('k', ctypes.c_int32), ('flags', ctypes.c_int32),
                ('ld', ctypes.c_int32), ('ld2', ctypes.c_int32), ('p0', ctypes.c_void_p), ('p1', ctypes.c_void_p),
                ('p2', ctypes.c_void_p)]


def _dt(t: torch.Tensor) -> int:
    return ops._dt_flag(t.dtype)


def _addr(t: Optional[torch.Tensor], offset_elems: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + offset_elems * t.element_size()


class WRef:
    """A GEMM weight operand: a row-major matrix (f32 programs, tests) or a fragment-major 16-bit copy
    (``mbv_fragment_group``) of a logical (rows, cols) matrix."""

    def __init__(self, tensor: torch.Tensor, rows: int, cols: int, frag: bool, kmajor: bool = False):
        self.t, self.rows, self.cols, self.frag, self.kmajor = tensor, rows, cols, frag, kmajor

    @property
    def dtype(self):
        return self.t.dtype


def fragment_copy(w: torch.Tensor, transposed: bool = False, kmajor: bool = False) -> WRef:
    """Fragment-major copy of the 16-bit matrix ``w`` (or of its transpose): one launch; for tests / one-off use —
    the decoder refreshes all of its copies with one grouped launch per step (:class:`WeightCopies`)."""
    lib = _lib.load()
    if w.dtype not in (torch.bfloat16, torch.float16) or w.dim() != 2 or w.stride(1) != 1:
        raise MaskBevHipError('fragment_copy: a 16-bit row-major matrix')
    rows, cols = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    dst = torch.empty(((rows + 15) // 16 * 16) * cols, dtype=w.dtype, device=w.device)
    P1, I1 = ctypes.c_void_p * 1, ctypes.c_int32 * 1
    check(lib.mbv_fragment_group(P1(w.data_ptr()), P1(dst.data_ptr()), I1(rows), I1(cols), I1(w.stride(0)),
                                 I1((1 if transposed else 0) | (2 if kmajor else 0)), 1, ops._stream()), 'mbv_fragment_group')
    return WRef(dst, rows, cols, True, kmajor)


class Program:
    """A stage list for ``mbv_rowchain_run``.  Tensors named in stages are kept referenced until :meth:`run` returns
    (the launch is stream-ordered after that, like any other op on torch's current stream)."""

    TIMING = None       # a list -> run() brackets every launch with events and appends (label, start, end) (scratch/time_k19.py)

    def __init__(self, rows: int, q_mod: int, eps: float, wdtype: torch.dtype, label: str = '', split: int = 1):
        """``split`` > 1: that many workgroups per 16-row block (``mbv_rowchain_run_split``); stages added inside
        ``with P.only(j):`` run in workgroup j alone, all others in every workgroup of the block."""
        self.rows, self.q_mod, self.eps = int(rows), int(q_mod), float(eps)
        self.wdtype = wdtype
        self.label = label
        self.split = int(split)
        self.owner: Optional[int] = None
        self.stages: List[RowStage] = []
        self.keep: list = []

    def only(self, j: int):
        prog = self

        class _Only:
            def __enter__(self_inner):
                self_inner.prev = prog.owner
                prog.owner = (j % prog.split) if prog.split > 1 else None

            def __exit__(self_inner, *exc):
                prog.owner = self_inner.prev
                return False
        return _Only()

    @property
    def blocks(self) -> int:
        return (self.rows + ROWS - 1) // ROWS

    def _add(self, op, dst=0, src=0, src2=-1, n=0, k=0, flags=0, ld=0, ld2=0, p0=None, p1=None, p2=None):
        if self.owner is not None and op not in (OP_FFN, OP_FFN_IO):
            flags |= (self.owner + 1) << 8
        self.stages.append(RowStage(op, dst, src, src2, 0, n, k, flags, ld, ld2, p0, p1, p2))

    def load(self, dst: int, t: torch.Tensor, n: int, col0: int = 0, add: Optional[torch.Tensor] = None):
        """slot dst <- t[:, col0:col0 + n] (+ ``add`` rows at r % q_mod)."""
        self.keep += [t, add]
        self._add(OP_LOAD, dst=dst, n=n, flags=_dt(t), ld=t.stride(0), p0=_addr(t, col0),
                  ld2=0 if add is None else add.stride(0), p1=_addr(add))

    def load_slot_plus(self, dst: int, src: int, add: torch.Tensor, n: int):
        self.keep.append(add)
        self._add(OP_LOAD, dst=dst, src=src, n=n, ld2=add.stride(0), p1=_addr(add))

    def store(self, src: int, t: torch.Tensor, n: int, col0: int = 0, accum: bool = False):
        if accum and self.split > 1 and self.owner is None:
            # every one of the `split` workgroups of a row block would do the same non-atomic read-modify-write
            raise MaskBevHipError('rowchain store: an accumulating store of a split launch needs an owner (P.only(j))')
        self.keep.append(t)
        self._add(OP_STORE, src=src, n=n, flags=_dt(t) | (F_ACCUM if accum else 0), ld=t.stride(0), p0=_addr(t, col0))

    def store_part(self, src: int, parts: torch.Tensor, n: int):
        """parts[j] (split, rows, n) f32 <- slot src of workgroup j of every row block (a split launch's partial results)."""
        if parts.dim() != 3 or parts.shape[0] != self.split or parts.dtype != torch.float32 or not parts.is_contiguous():
            raise MaskBevHipError('rowchain store_part: (split, rows, n) contiguous f32')
        self.keep.append(parts)
        self._add(OP_STORE, src=src, n=n, flags=_dt(parts) | F_SPLIT, ld=parts.stride(1), ld2=parts.stride(0), p0=_addr(parts))

    def sum_parts(self, dst: int, parts: torch.Tensor, n: int):
        """slot dst <- sum over j of parts[j] (in order)."""
        if parts.dim() != 3 or parts.dtype != torch.float32 or not parts.is_contiguous():
            raise MaskBevHipError('rowchain sum_parts: (parts, rows, n) contiguous f32')
        self.keep.append(parts)
        self._add(OP_SUM, dst=dst, n=n, k=parts.shape[0], ld=parts.stride(1), ld2=parts.stride(0), p0=_addr(parts))

    def gemm(self, dst: int, src: int, w: torch.Tensor, n: int, k: int, bias: Optional[torch.Tensor] = None,
             row0: int = 0, col0: int = 0, relu: bool = False, accum: bool = False, mask: int = -1, bias0: int = 0,
             out: Optional[torch.Tensor] = None, out_col0: int = 0):
        """slot dst = act([dst +] src (16, k) @ w[row0:row0 + n, col0:col0 + k]^T + bias[bias0:bias0 + n]);
        ``out``: the result rows also go to out[:, out_col0:out_col0 + n] (saves a STORE stage)."""
        if isinstance(w, WRef) and w.frag:
            if w.dtype != self.wdtype or row0 % 16 or col0 % 32:
                raise MaskBevHipError('rowchain: fragment weight dtype / offsets')
            kbn = w.cols // 32
            self.keep += [w.t, bias]
            flags = (F_RELU if relu else 0) | (F_ACCUM if accum else 0) | (F_MASK if mask >= 0 else 0) | F_FRAG
            self._add(OP_GEMM, dst=dst, src=src, src2=mask, n=n, k=k, flags=flags, ld=kbn,
                      p0=_addr(w.t, ((row0 // 16) * kbn + col0 // 32) * 512), p1=_addr(bias, bias0))
        else:
            if isinstance(w, WRef):
                w = w.t
            if w.dtype != self.wdtype or w.stride(1) != 1:
                raise MaskBevHipError('rowchain: weight dtype / layout mismatch')
            self.keep += [w, bias]
            flags = (F_RELU if relu else 0) | (F_ACCUM if accum else 0) | (F_MASK if mask >= 0 else 0)
            self._add(OP_GEMM, dst=dst, src=src, src2=mask, n=n, k=k, flags=flags, ld=w.stride(0),
                      p0=_addr(w, row0 * w.stride(0) + col0), p1=_addr(bias, bias0))
        if out is not None:          # (an epilogue store from the MFMA accumulators was measured slower than a STORE stage)
            self.store(dst, out, n, col0=out_col0)

    def ln(self, dst: int, a: int, b: int, gamma: torch.Tensor, beta: torch.Tensor, n: int,
           stats: Optional[torch.Tensor] = None, save_sum: bool = False):
        self.keep += [gamma, beta, stats]
        self._add(OP_LN, dst=dst, src=a, src2=b, n=n, flags=F_SAVE_SUM if save_sum else 0, p0=_addr(gamma), p1=_addr(beta),
                  p2=_addr(stats))

    def ln_bwd(self, dst: int, g: int, s: int, gamma: torch.Tensor, stats: torch.Tensor, n: int,
               partial: Optional[torch.Tensor] = None):
        self.keep += [gamma, stats, partial]
        self._add(OP_LN_BWD, dst=dst, src=g, src2=s, n=n, p0=_addr(gamma), p1=_addr(partial), p2=_addr(stats))

    def add(self, dst: int, a: int, b: int, n: int):
        self._add(OP_ADD, dst=dst, src=a, src2=b, n=n)

    def colsum(self, src: int, partial: torch.Tensor, n: int, col0: int = 0):
        """partial[block, col0:col0 + n] = column sums of the block's rows of slot src."""
        self.keep.append(partial)
        self._add(OP_COLSUM, src=src, n=n, ld=partial.stride(0), p0=_addr(partial, col0))

    def ffn(self, dst: int, src: int, scratch: int, w_a: torch.Tensor, w_b: torch.Tensor, e: int, f: int,
            hid: torch.Tensor, bias_a: Optional[torch.Tensor] = None, bias_out: Optional[torch.Tensor] = None,
            backward: bool = False, d_hid: Optional[torch.Tensor] = None, partial: Optional[torch.Tensor] = None,
            partial_col0: int = 0, sliced: bool = False):
        """The MLP pair as ONE stage with the hidden chunks spread over the waves (16-bit weights).
        forward: slot dst = relu(src @ w_a^T + bias_a) @ w_b^T (+ bias_out), w_a = W1 (f, e), w_b = W2 (e, f); the hidden
        activations go to ``hid`` (rows, f) f32.  backward: slot dst = ((src @ w_a^T) * (hid > 0)) @ w_b^T with
        w_a = W2^T (f, e), w_b = W1^T (e, f); d(hidden) goes to ``d_hid``, its per-block column sums to
        ``partial[:, partial_col0:partial_col0 + f]``.  ``scratch``: first of 5 free consecutive slots.
        ``sliced`` (split launches, f == 256 * split): workgroup j computes hidden units [256 j, 256 j + 256) and slot dst
        holds its PARTIAL result (j == 0: plus ``bias_out``) — store it with :meth:`store_part`."""
        if (not isinstance(w_a, WRef) or not isinstance(w_b, WRef) or not w_a.frag or not w_b.frag
                or w_a.dtype != self.wdtype or w_b.dtype != self.wdtype or self.wdtype == torch.float32
                or (w_a.rows, w_a.cols) != (f, e) or (w_b.rows, w_b.cols) != (e, f)):
            raise MaskBevHipError('rowchain ffn: fragment-major 16-bit weights (f, e) and (e, f) in the program dtype')
        self.keep += [w_a.t, w_b.t, hid, bias_a, bias_out, d_hid, partial]
        if sliced and (f != 256 * self.split or not w_b.kmajor or w_a.kmajor or e % 16):
            raise MaskBevHipError('rowchain ffn: a sliced stage needs f == 256 * split and a k-major second weight')
        if not sliced and (w_a.kmajor or w_b.kmajor):
            raise MaskBevHipError('rowchain ffn: k-major weight copies belong to sliced stages')
        self._add(OP_FFN, dst=dst, src=src, src2=scratch, n=e, k=f, flags=(F_MASK if backward else 0) | (F_SLICE if sliced else 0), ld=e // 32,
                  ld2=f // 32, p0=_addr(w_a.t), p1=_addr(bias_a), p2=_addr(w_b.t))
        self._add(OP_FFN_IO, ld=hid.stride(0), ld2=0 if partial is None else partial.stride(0), p0=_addr(hid),
                  p1=_addr(d_hid if backward else bias_out), p2=_addr(partial, partial_col0))

    def run(self):
        lib = _lib.load()
        n = len(self.stages)
        if n == 0:
            return
        arr = (RowStage * n)(*self.stages)
        ev = None
        if Program.TIMING is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.split > 1:
            check(lib.mbv_rowchain_run_split(arr, n, self.rows, self.q_mod, self.eps, ops._dt_flag(self.wdtype), self.split,
                                             ops._stream()), 'mbv_rowchain_run_split')
        else:
            check(lib.mbv_rowchain_run(arr, n, self.rows, self.q_mod, self.eps, ops._dt_flag(self.wdtype), ops._stream()),
                  'mbv_rowchain_run')
        if ev is not None:
            ev[1].record()
            Program.TIMING.append((self.label, n, ev[0], ev[1]))
        self.keep = []


# ---------------------------------------------------------------------------------------------------------------------
# transposed weight copies (data gradients), refreshed once per step
# ---------------------------------------------------------------------------------------------------------------------
class WeightCopies:
    """The GEMM operands of the decoder's chains, refreshed once per step.

    16-bit compute: every weight — as it is for the forward products, transposed for the data gradients ``dX = dY W`` —
    as a fragment-major copy, all written by ONE grouped launch (``mbv_fragment_group``) from the 16-bit weight copies.
    f32 compute: the forward operands are the master weights themselves (row-major); the transposes are f32 copies
    (``mbv_transpose_group``).  ``get(param, rows, transposed)`` -> :class:`WRef`.  Buffers are allocated once."""

    def __init__(self):
        self.buf = {}
        self.ref = {}

    def refresh(self, entries, dt: torch.dtype):
        lib = _lib.load()
        # Entries are replaced, not wiped: a forward under no_grad (a sanity validation, an extra eager step) between a
        # training forward and its backward lists no transposed operands, and that backward still looks its own up.
        # They stay valid — the weights only change in the optimizer step, after the backward.
        if getattr(self, '_dt', None) != dt:
            self.ref = {}
            self._dt = dt
        keep = []
        if dt == torch.float32:
            src_p, dst_p, rows_l, cols_l = [], [], [], []
            for ent in entries:
                p, rows, tr = ent[:3]
                w = p.detach()
                if rows is not None:
                    w = w[rows[0]:rows[1]]
                key = (id(p), rows, tr)
                if not tr:
                    self.ref[key] = WRef(w, w.shape[0], w.shape[1], False)
                    continue
                t = self.buf.get(key)
                if t is None or t.dtype != dt or t.shape != (w.shape[1], w.shape[0]) or t.device != w.device:
                    t = self.buf[key] = torch.empty((w.shape[1], w.shape[0]), dtype=dt, device=w.device)
                keep.append(w)
                src_p.append(w.data_ptr()); dst_p.append(t.data_ptr()); rows_l.append(w.shape[0]); cols_l.append(w.shape[1])
                self.ref[key] = WRef(t, w.shape[1], w.shape[0], False)
            n = len(src_p)
            if n:
                PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
                check(lib.mbv_transpose_group(PA(*src_p), PA(*dst_p), IA(*rows_l), IA(*cols_l), n, 4, ops._stream()),
                      'mbv_transpose_group')
            return
        src_p, dst_p, rows_l, cols_l, ld_l, tr_l = [], [], [], [], [], []
        for ent in entries:
            p, rows, tr = ent[:3]
            km = len(ent) > 3 and bool(ent[3])          # k-major block order (second weight of a sliced MLP stage)
            w = ops._compute_copy(p, dt)
            if rows is not None:
                w = w[rows[0]:rows[1]]
            if w.stride(1) != 1:
                w = w.contiguous()
            lr, lc = (w.shape[1], w.shape[0]) if tr else (w.shape[0], w.shape[1])
            key = (id(p), rows, tr)
            numel = ((lr + 15) // 16 * 16) * lc
            t = self.buf.get(key)
            if t is None or t.dtype != dt or t.numel() != numel or t.device != w.device:
                t = self.buf[key] = torch.empty(numel, dtype=dt, device=w.device)
            keep.append(w)
            src_p.append(w.data_ptr()); dst_p.append(t.data_ptr()); rows_l.append(lr); cols_l.append(lc)
            ld_l.append(w.stride(0)); tr_l.append((1 if tr else 0) | (2 if km else 0))
            self.ref[key] = WRef(t, lr, lc, True, km)
        n = len(src_p)
        if n:
            PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
            check(lib.mbv_fragment_group(PA(*src_p), PA(*dst_p), IA(*rows_l), IA(*cols_l), IA(*ld_l), IA(*tr_l), n,
                                         ops._stream()), 'mbv_fragment_group')

    def get(self, p, rows=None, transposed: bool = False) -> WRef:
        return self.ref[(id(p), rows, transposed)]


# ---------------------------------------------------------------------------------------------------------------------
# parameter-gradient helpers: arena parameters accumulate in place (deferred grouped launches), plain ones get tensors
# ---------------------------------------------------------------------------------------------------------------------
def _is_arena(p) -> bool:
    return getattr(p, '_mbv_arena', False) and p.grad is not None and p.grad.dtype == torch.float32


def _weight_grad(p, rows, g2: torch.Tensor, x2: torch.Tensor, want: bool):
    """d(p[rows]) = g2^T x2.  Arena: accumulated into p.grad (returns None).  Otherwise a full-size gradient tensor."""
    if not want:
        return None
    if _is_arena(p):
        acc = p.grad if rows is None else p.grad[rows[0]:rows[1]]
        if g2.dtype != x2.dtype:
            x2 = x2.to(g2.dtype)
        ops._wgrad_into(acc, g2, x2, None, persistent=True)
        ops._fire_grad_hooks(p)
        return None
    gw = g2.float().t().mm(x2.float())
    if rows is None:
        return gw.to(p.dtype)
    full = torch.zeros_like(p)
    full[rows[0]:rows[1]] = gw
    return full


def _partial_grad(p, rows, partial: torch.Tensor, col0: int, n: int, want: bool):
    """d(p[rows]) (n,) = column sums of partial[:, col0:col0 + n] (per-block partial rows written by a row chain)."""
    if not want:
        return None
    nblk, ld = partial.shape
    if _is_arena(p):
        dst = p.grad if rows is None else p.grad[rows[0]:rows[1]]
        if not ops._defer_colsum(partial, dst, nblk, n, ld, offset=col0):
            ops._colsum_now(partial, dst, nblk, n, ld, offset=col0)
        ops._fire_grad_hooks(p)
        return None
    g = partial[:, col0:col0 + n].sum(0)
    if rows is None:
        return g.to(p.dtype)
    full = torch.zeros_like(p)
    full[rows[0]:rows[1]] = g
    return full


def _sum_grads(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a + b


class QueryPositions(torch.autograd.Function):
    """query_embed (Q, E) as used by the fused layers.  The layers' backward chains ACCUMULATE d(positions) row by row
    into ``acc`` (B*Q, E) (row-local read-modify-write, no atomics); this node, which autograd runs after all of its
    consumers, folds the batch and hands the sum to the embedding."""

    @staticmethod
    def forward(ctx, weight, acc_holder):
        ctx.acc_holder = acc_holder
        return weight.view_as(weight)

    @staticmethod
    def backward(ctx, g):
        acc = ctx.acc_holder.pop('acc', None)
        if acc is not None:
            q, e = g.shape
            g = g + acc.view(-1, q, e).sum(0)
        return g, None


# ---------------------------------------------------------------------------------------------------------------------
# _DecA: cross-attention output -> LN1 -> self-attention
# ---------------------------------------------------------------------------------------------------------------------
class LayerCtx:
    """Per-call constants shared by the two Functions of a layer (plain Python object, not seen by autograd)."""

    def __init__(self, batch, queries, embed, heads, ffn, eps, dt, tw: WeightCopies, dpos_holder, qpos):
        self.b, self.q, self.e, self.h, self.f, self.eps, self.dt = batch, queries, embed, heads, ffn, eps, dt
        self.m = batch * queries
        self.tw, self.dpos_holder, self.qpos = tw, dpos_holder, qpos
        self.wdt = torch.float32 if dt == torch.float32 else dt

    def w(self, p, rows=None):
        """The forward GEMM operand of parameter p[rows]."""
        return self.tw.get(p, rows, False)

    def wt(self, p, rows=None):
        """The data-gradient operand (p[rows] transposed)."""
        return self.tw.get(p, rows, True)

    def dpos(self, like: torch.Tensor) -> torch.Tensor:
        acc = self.dpos_holder.get('acc')
        if acc is None:
            acc = self.dpos_holder['acc'] = torch.zeros((self.m, self.e), dtype=torch.float32, device=like.device)
        return acc


def _self_attention_fwd(q, k, v, b, nq, heads):
    lib = _lib.load()
    e = q.shape[-1]
    out = torch.empty_like(q)
    lse = torch.empty((b, heads, nq), dtype=torch.float32, device=q.device)
    ws = ops._workspace(lib.mbv_attn_workspace_bytes(b, nq, nq, heads, e // heads), q.device)
    check(lib.mbv_attn_fwd(ops._ptr(q), ops._ptr(k), ops._ptr(v), ops._ptr(None), _dt(q), b, nq, nq, heads, e // heads,
                           ops._ptr(out), ops._ptr(lse), ops._ptr(ws), ws.numel(), ops._stream()), 'mbv_attn_fwd')
    return out, lse


class _DecA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lc: LayerCtx, x0, o1, wo, bo, g1, b1, w_in, b_in):
        m, e = lc.m, lc.e
        dev = x0.device
        f32 = dict(dtype=torch.float32, device=dev)
        ctx.in_shapes = (x0.shape, o1.shape)
        x0 = x0.reshape(m, e).contiguous()
        o1 = o1.reshape(m, e).contiguous()
        sum1 = torch.empty((m, e), **f32)
        stats1 = torch.empty((m, 2), **f32)
        x1 = torch.empty((m, e), **f32)
        t1 = torch.empty((m, e), **f32)
        # q / k / v in the compute dtype: the self-attention (K6) then runs its 16-bit MFMA form — 10 us instead of
        # 26 backward, 14 -> 8 forward, for 100 x 100 scores per head — like the per-op path's
        qkv = torch.empty((3, m, e), dtype=lc.dt, device=dev)
        w_o, w_i = lc.w(wo), lc.w(w_in)
        # three workgroups per row block: each one of the q / k / v projections and a share of the stores
        P = Program(m, lc.q, lc.eps, lc.wdt, 'A.fwd', split=spread(3))
        P.load(0, o1, e)
        o1f = o1
        if o1.dtype != torch.float32:         # an f32 copy for the weight gradient of the output projection (one STORE
            o1f = torch.empty((m, e), **f32)  # stage here instead of a conversion launch per layer in the backward pass)
            with P.only(0):
                P.store(0, o1f, e)
        P.gemm(1, 0, w_o, e, e, bias=bo)
        P.load(2, x0, e)
        P.ln(3, 2, 1, g1, b1, e, stats=stats1, save_sum=True)
        with P.only(2):
            P.store(2, sum1, e)
            P.store(3, x1, e)
        P.load_slot_plus(0, 3, lc.qpos, e)                      # x1 + positions: the q / k input
        with P.only(1):
            P.store(0, t1, e)
        with P.only(0):
            P.gemm(1, 0, w_i, e, e, bias=b_in, row0=0, bias0=0, out=qkv[0])
        with P.only(1):
            P.gemm(2, 0, w_i, e, e, bias=b_in, row0=e, bias0=e, out=qkv[1])
        with P.only(2):
            P.gemm(4, 3, w_i, e, e, bias=b_in, row0=2 * e, bias0=2 * e, out=qkv[2])      # (row offsets: multiples of 16)
        P.run()
        o2, lse = _self_attention_fwd(qkv[0], qkv[1], qkv[2], lc.b, lc.q, lc.h)
        ctx.lc = lc
        ctx.set_materialize_grads(False)
        ctx.params = (wo, bo, g1, b1, w_in, b_in)
        ctx.o1_dtype = o1.dtype
        ctx.save_for_backward(o1f, sum1, stats1, x1, t1, qkv, o2, lse)
        return x1, o2

    @staticmethod
    def backward(ctx, g_x1, g_o2):
        lib = _lib.load()
        lc = ctx.lc
        m, e = lc.m, lc.e
        o1, sum1, stats1, x1, t1, qkv, o2, lse = ctx.saved_tensors
        wo, bo, g1, b1, w_in, b_in = ctx.params
        dev = o1.device
        f32 = dict(dtype=torch.float32, device=dev)
        if g_x1 is None:
            g_x1 = torch.zeros((m, e), **f32)
        if g_o2 is None:
            g_o2 = torch.zeros((m, e), dtype=o2.dtype, device=dev)
        # K6 self-attention backward
        g_o2 = g_o2.reshape(m, e).to(o2.dtype).contiguous()
        g_qkv = torch.empty((3, m, e), **f32)
        check(lib.mbv_attn_bwd(ops._ptr(qkv[0]), ops._ptr(qkv[1]), ops._ptr(qkv[2]), ops._ptr(None), ops._ptr(o2),
                               ops._ptr(g_o2), ops._ptr(lse), _dt(o2), lc.b, lc.q, lc.q, lc.h, e // lc.h, ops._ptr(g_qkv[0]),
                               ops._ptr(g_qkv[1]), ops._ptr(g_qkv[2]), ops._stream()), 'mbv_attn_bwd')
        g_x1 = g_x1.reshape(m, e).to(torch.float32).contiguous()
        nblk = (m + ROWS - 1) // ROWS
        part_b = torch.empty((nblk, 4 * e), **f32)             # [bq | bk | bv | bo]
        part_ln = torch.empty((nblk, 2 * e), **f32)
        ds1 = torch.empty((m, e), **f32)
        g_o1 = torch.empty((m, e), dtype=ctx.o1_dtype, device=dev)
        tw = lc.tw
        # two workgroups per row block: one takes the last product, the other the column sums and the stores
        P = Program(m, lc.q, lc.eps, lc.wdt, 'A.bwd', split=spread(2))
        P.load(0, g_qkv[0], e)
        with P.only(1):
            P.colsum(0, part_b, e, 0)
        P.gemm(1, 0, lc.wt(w_in, (0, e)), e, e)
        P.load(2, g_qkv[1], e)
        with P.only(1):
            P.colsum(2, part_b, e, e)
        P.gemm(1, 2, lc.wt(w_in, (e, 2 * e)), e, e, accum=True)
        with P.only(1):
            P.store(1, lc.dpos(g_x1), e, accum=True)            # d(positions) of this layer's self-attention (ONE workgroup)
        P.load(3, g_qkv[2], e)
        with P.only(1):
            P.colsum(3, part_b, e, 2 * e)
        P.gemm(1, 3, lc.wt(w_in, (2 * e, 3 * e)), e, e, accum=True)
        P.load(4, g_x1, e)
        P.add(1, 1, 4, e)
        P.load(5, sum1, e)
        P.ln_bwd(6, 1, 5, g1, stats1, e, partial=part_ln)
        with P.only(1):
            P.store(6, ds1, e)
            P.colsum(6, part_b, e, 3 * e)
        with P.only(0):
            P.gemm(0, 6, lc.wt(wo), e, e, out=g_o1)
        P.run()
        ni = ctx.needs_input_grad
        gw_in = _sum_grads(_sum_grads(_weight_grad(w_in, (0, e), g_qkv[0], t1, ni[7]),
                                      _weight_grad(w_in, (e, 2 * e), g_qkv[1], t1, ni[7])),
                           _weight_grad(w_in, (2 * e, 3 * e), g_qkv[2], x1, ni[7]))
        gb_in = _sum_grads(_sum_grads(_partial_grad(b_in, (0, e), part_b, 0, e, ni[8]),
                                      _partial_grad(b_in, (e, 2 * e), part_b, e, e, ni[8])),
                           _partial_grad(b_in, (2 * e, 3 * e), part_b, 2 * e, e, ni[8]))
        gwo = _weight_grad(wo, None, ds1, o1, ni[3])
        gbo = _partial_grad(bo, None, part_b, 3 * e, e, ni[4])
        gg1 = _partial_grad(g1, None, part_ln, 0, e, ni[5])
        gb1 = _partial_grad(b1, None, part_ln, e, e, ni[6])
        return None, ds1.view(ctx.in_shapes[0]), g_o1.view(ctx.in_shapes[1]), gwo, gbo, gg1, gb1, gw_in, gb_in


# ---------------------------------------------------------------------------------------------------------------------
# _DecB: self-attention output -> LN2 -> FFN -> LN3 -> heads -> next layer's masked cross-attention
# ---------------------------------------------------------------------------------------------------------------------
class NextCross:
    """What _DecB needs to run the next layer's cross-attention: the packed in_proj parameters, the SharedKV slot,
    the mask features / stacked-logit slot / target size for K7."""

    def __init__(self, w_in, b_in, holder, slot, target_size):
        self.w_in, self.b_in, self.holder, self.slot, self.target_size = w_in, b_in, holder, slot, target_size


class HeadSpec:
    def __init__(self, post_g, post_b, cls_w, cls_b, mlp, mask_feature, out_slot, index: int = -1):
        self.post_g, self.post_b, self.cls_w, self.cls_b, self.mlp = post_g, post_b, cls_w, cls_b, mlp
        self.mask_feature, self.out_slot = mask_feature, out_slot
        self.index = index           # which decoder output this layer produces (the batched heads' backward keys on it)


class _DecB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lc: LayerCtx, head: HeadSpec, nxt: Optional[NextCross], x1, o2, token, wo, bo, g2, b2, w1, bb1, w2, bb2,
                g3, b3, nw_in, nb_in):
        lib = _lib.load()
        m, e, f = lc.m, lc.e, lc.f
        dev = x1.device
        f32 = dict(dtype=torch.float32, device=dev)
        ctx.in_shapes = (x1.shape, o2.shape)
        x1 = x1.reshape(m, e).contiguous()
        o2 = o2.reshape(m, e).contiguous()
        sum2, sum3 = torch.empty((m, e), **f32), torch.empty((m, e), **f32)
        stats2, stats3 = torch.empty((m, 2), **f32), torch.empty((m, 2), **f32)
        x2, x3 = torch.empty((m, e), **f32), torch.empty((m, e), **f32)
        hid = torch.empty((m, f), **f32)
        ncls = head.cls_w.shape[0]
        cls = torch.empty((m, ncls), **f32)
        oc = head.mlp[2][0].shape[0]
        me = torch.empty((m, oc), dtype=head.mask_feature.dtype, device=dev)
        w_o, w_1, w_2 = lc.w(wo), lc.w(w1), lc.w(w2)
        fused_ffn = lc.wdt != torch.float32 and f % 256 == 0 and e % 32 == 0 and switches.get('rc_ffn')
        # The MLP's 2 x 1 MB of weights behind ONE workgroup per 16 rows is 30 us of dependent loads on 25 CUs.  Split form:
        # f / 256 workgroups per row block each take 256 hidden units (launch 1: the cheap stages before the MLP run in all
        # of them), a second launch adds the parts and finishes the layer with its three independent branches (next
        # layer's query projection | class head | mask-embedding MLP) in three workgroups.
        S = ffn_split(f, e, lc.wdt)
        split = S > 1
        P = Program(m, lc.q, lc.eps, lc.wdt, 'B1.fwd' if split else 'B.fwd', split=S)
        P.load(2, o2, e)
        o2f = o2
        if o2.dtype != torch.float32:         # (f32 copy for d(wo), as in _DecA)
            o2f = torch.empty((m, e), **f32)
            with P.only(0):
                P.store(2, o2f, e)
        P.gemm(3, 2, w_o, e, e, bias=bo)
        P.load(4, x1, e)
        P.ln(0, 4, 3, g2, b2, e, stats=stats2, save_sum=True)          # x2 -> slot 0
        with P.only(1):
            P.store(4, sum2, e)
        with P.only(2):
            P.store(0, x2, e)
        if split:
            parts = torch.empty((S, m, e), **f32)
            P.ffn(1, 0, 2, w_1, w_2, e, f, hid, bias_a=bb1, bias_out=bb2, sliced=True)
            P.store_part(1, parts, e)
            P.run()
            P = Program(m, lc.q, lc.eps, lc.wdt, 'B2.fwd', split=3)
            P.sum_parts(1, parts, e)                                       # y -> slot 1
            P.load(0, x2, e)
        elif fused_ffn:
            P.ffn(1, 0, 2, w_1, w_2, e, f, hid, bias_a=bb1, bias_out=bb2)    # y -> slot 1; scratch slots 2..6
        else:
            ch = 256
            for c in range(0, f, ch):
                n = min(ch, f - c)
                P.gemm(2, 0, w_1, n, e, bias=bb1, row0=c, bias0=c, relu=True, out=hid, out_col0=c)
                P.gemm(1, 2, w_2, e, n, bias=bb2 if c == 0 else None, col0=c, accum=c > 0)
        P.ln(2, 0, 1, g3, b3, e, stats=stats3, save_sum=True)          # x3 -> slot 2, the sum -> slot 0
        qc = t3 = None
        with P.only(0):
            P.store(0, sum3, e)
            P.store(2, x3, e)
            if nxt is not None:
                t3 = torch.empty((m, e), **f32)
                qc = torch.empty((m, e), dtype=nxt.holder.k_cat.dtype, device=dev)
                P.load_slot_plus(0, 2, lc.qpos, e)
                P.store(0, t3, e)
                P.gemm(1, 0, lc.w(nxt.w_in), e, e, bias=nxt.b_in, row0=0, bias0=0, out=qc)
        # prediction heads (no gradient through here: ops._DeferredHeads re-evaluates them in one batched backward)
        P.ln(3, 2, -1, head.post_g, head.post_b, e)
        with P.only(1):
            P.gemm(4, 3, lc.w(head.cls_w), ncls, e, bias=head.cls_b, out=cls)
        (m1w, m1b), (m2w, m2b), (m3w, m3b) = head.mlp
        with P.only(2):
            P.gemm(5, 3, lc.w(m1w), m1w.shape[0], e, bias=m1b, relu=True)
            P.gemm(6, 5, lc.w(m2w), m2w.shape[0], m1w.shape[0], bias=m2b, relu=True)
            P.gemm(5, 6, lc.w(m3w), oc, m2w.shape[0], bias=m3b, out=me)
        P.run()
        b, q = lc.b, lc.q
        with torch.no_grad():
            size = nxt.target_size if nxt is not None else head.mask_feature.shape[-2:]
            mask_pred, blocked = ops.mask_logits(me.view(b, q, oc), head.mask_feature, size, head.out_slot)
        o1n = mask = lse = None
        if nxt is not None:
            holder = nxt.holder
            k_cat, v_cat = holder.k_cat, holder.v_cat
            nl = k_cat.shape[1]
            ldk = holder.n * e
            off = nxt.slot * e * k_cat.element_size()
            mask = blocked.reshape(b, q, nl).view(torch.uint8)
            o1n = torch.empty((m, e), dtype=k_cat.dtype, device=dev)
            lse = torch.empty((b, lc.h, q), **f32)
            ws = ops._workspace(lib.mbv_attn_workspace_bytes(b, q, nl, lc.h, e // lc.h), dev)
            check(lib.mbv_attn_fwd_ld(ops._ptr(qc), ctypes.c_void_p(k_cat.data_ptr() + off),
                                      ctypes.c_void_p(v_cat.data_ptr() + off), ldk, ops._ptr(mask), _dt(k_cat), b, q, nl,
                                      lc.h, e // lc.h, ops._ptr(o1n), ops._ptr(lse), ops._ptr(ws), ws.numel(),
                                      ops._stream()), 'mbv_attn_fwd_ld')
        ctx.lc, ctx.nxt = lc, nxt
        ctx.head_index = head.index
        # no zero tensors for the outputs that carry no gradient (class scores, mask logits, the attention mask): autograd
        # would otherwise MATERIALISE them for backward() — a 26 MB fill and a 1.6 MB bool fill per layer
        ctx.set_materialize_grads(False)
        ctx.params = (wo, bo, g2, b2, w1, bb1, w2, bb2, g3, b3, nw_in, nb_in)
        ctx.o2_dtype = o2.dtype
        ctx.save_for_backward(o2f, sum2, stats2, x2, hid, sum3, stats3, t3, qc, mask, o1n, lse)
        cls3 = cls.view(b, q, ncls)
        ctx.mark_non_differentiable(cls3, mask_pred)
        outs = (x3.view(b, q, e), cls3, mask_pred)
        if nxt is not None:
            ctx.mark_non_differentiable(blocked)
            return outs + (o1n.view(b, q, e), blocked)
        return outs

    @staticmethod
    def backward(ctx, g_x3, _g_cls, _g_mask, g_o1n=None, _g_blocked=None):
        lib = _lib.load()
        lc, nxt = ctx.lc, ctx.nxt
        m, e, f = lc.m, lc.e, lc.f
        b, q = lc.b, lc.q
        o2, sum2, stats2, x2, hid, sum3, stats3, t3, qc, mask, o1n, lse = ctx.saved_tensors
        wo, bo, g2, b2, w1, bb1, w2, bb2, g3, b3, nw_in, nb_in = ctx.params
        dev = o2.device
        f32 = dict(dtype=torch.float32, device=dev)
        tw = lc.tw
        nblk = (m + ROWS - 1) // ROWS
        g_qc = None
        if nxt is not None and g_o1n is not None:
            holder = nxt.holder
            k_cat, v_cat = holder.k_cat, holder.v_cat
            nl = k_cat.shape[1]
            ldk = holder.n * e
            if holder.dk_cat is None:
                holder.dk_cat = torch.empty_like(k_cat)
                holder.dv_cat = torch.empty_like(v_cat)
            off = nxt.slot * e * k_cat.element_size()
            g_o = g_o1n.reshape(m, e).to(qc.dtype).contiguous()
            g_qc = torch.empty((m, e), **f32)
            bf = _dt(qc)
            check(lib.mbv_attn_bwd_ld(ops._ptr(qc), ctypes.c_void_p(k_cat.data_ptr() + off),
                                      ctypes.c_void_p(v_cat.data_ptr() + off), ldk, ops._ptr(mask), ops._ptr(o1n),
                                      ops._ptr(g_o), ops._ptr(lse), bf, b, q, nl, lc.h, e // lc.h, ops._ptr(g_qc),
                                      ctypes.c_void_p(holder.dk_cat.data_ptr() + off),
                                      ctypes.c_void_p(holder.dv_cat.data_ptr() + off), ldk, bf, ops._stream()),
                  'mbv_attn_bwd_ld')
            holder.written.add(nxt.slot)
        # the batched heads' share of d(x3), left in the layer context instead of on the autograd edge (mask2former_head.
        # _DeferredHeads.backward): added by the program below while it loads the gradient
        g_heads = lc.dpos_holder.get('gq', {}).pop(ctx.head_index, None)
        if g_heads is not None:
            g_heads = g_heads.reshape(m, e).to(torch.float32).contiguous()
        if g_x3 is None:                  # (this layer's output fed nothing else that needed a gradient)
            g_x3, g_heads = (g_heads, None) if g_heads is not None else (torch.zeros((m, e), **f32), None)
        g_x3 = g_x3.reshape(m, e).to(torch.float32).contiguous()
        part_b = torch.empty((nblk, 3 * e + f), **f32)          # [bq' | b2 | bo | b1 (f)]
        part_ln3, part_ln2 = torch.empty((nblk, 2 * e), **f32), torch.empty((nblk, 2 * e), **f32)
        ds3, ds2 = torch.empty((m, e), **f32), torch.empty((m, e), **f32)
        dh = torch.empty((m, f), **f32)
        g_o2 = torch.empty((m, e), dtype=ctx.o2_dtype, device=dev)
        fused_ffn = lc.wdt != torch.float32 and f % 256 == 0 and e % 32 == 0 and switches.get('rc_ffn')
        S = ffn_split(f, e, lc.wdt)
        split = S > 1
        P = Program(m, lc.q, lc.eps, lc.wdt, 'B1.bwd' if split else 'B.bwd', split=S)
        if g_qc is not None:
            P.load(2, g_qc, e)
            with P.only(1):
                P.colsum(2, part_b, e, 0)
            P.gemm(3, 2, lc.wt(nxt.w_in, (0, e)), e, e)
            with P.only(2):
                P.store(3, lc.dpos(g_x3), e, accum=True)        # (an accumulating store: exactly one workgroup)
            P.load(4, g_x3, e)
            P.add(3, 3, 4, e)
        else:
            P.load(3, g_x3, e)
        if g_heads is not None:
            P.load(4, g_heads, e)
            P.add(3, 3, 4, e)
        P.load(4, sum3, e)
        P.ln_bwd(0, 3, 4, g3, stats3, e, partial=part_ln3)              # ds3 -> slot 0
        with P.only(3):
            P.store(0, ds3, e)
        with P.only(4):
            P.colsum(0, part_b, e, e)
        w2t, w1t = lc.wt(w2), lc.wt(w1)                         # (f, e) and (e, f)
        if split:
            parts = torch.empty((S, m, e), **f32)
            P.ffn(1, 0, 2, w2t, w1t, e, f, hid, backward=True, d_hid=dh, partial=part_b, partial_col0=3 * e, sliced=True)
            P.store_part(1, parts, e)
            P.run()
            P = Program(m, lc.q, lc.eps, lc.wdt, 'B2.bwd', split=spread(2))
            P.sum_parts(1, parts, e)
            P.load(0, ds3, e)
        elif fused_ffn:
            P.ffn(1, 0, 2, w2t, w1t, e, f, hid, backward=True, d_hid=dh, partial=part_b, partial_col0=3 * e)
        else:
            ch = 256
            for c in range(0, f, ch):
                n = min(ch, f - c)
                P.load(2, hid, n, col0=c)
                P.gemm(3, 0, w2t, n, e, row0=c, mask=2, out=dh, out_col0=c)   # d hidden chunk = (ds3 W2[:, chunk]) * (h > 0)
                P.colsum(3, part_b, n, 3 * e + c)
                P.gemm(1, 3, w1t, e, n, col0=c, accum=c > 0)        # d x2 += d hidden chunk . W1[chunk]
        P.add(1, 1, 0, e)
        P.load(2, sum2, e)
        P.ln_bwd(3, 1, 2, g2, stats2, e, partial=part_ln2)
        with P.only(1):
            P.store(3, ds2, e)
            P.colsum(3, part_b, e, 2 * e)
        with P.only(0):
            P.gemm(4, 3, lc.wt(wo), e, e, out=g_o2)
        P.run()
        ni = ctx.needs_input_grad         # lc, head, nxt, x1, o2, token, wo, bo, g2, b2, w1, bb1, w2, bb2, g3, b3, nw_in, nb_in
        gwo = _weight_grad(wo, None, ds2, o2, ni[6])
        gbo = _partial_grad(bo, None, part_b, 2 * e, e, ni[7])
        gg2 = _partial_grad(g2, None, part_ln2, 0, e, ni[8])
        gb2 = _partial_grad(b2, None, part_ln2, e, e, ni[9])
        gw1 = _weight_grad(w1, None, dh, x2, ni[10])
        gbb1 = _partial_grad(bb1, None, part_b, 3 * e, f, ni[11])
        gw2 = _weight_grad(w2, None, ds3, hid, ni[12])
        gbb2 = _partial_grad(bb2, None, part_b, e, e, ni[13])
        gg3 = _partial_grad(g3, None, part_ln3, 0, e, ni[14])
        gb3 = _partial_grad(b3, None, part_ln3, e, e, ni[15])
        gnw = gnb = None
        if g_qc is not None:
            gnw = _weight_grad(nw_in, (0, e), g_qc, t3, ni[16])
            gnb = _partial_grad(nb_in, (0, e), part_b, 0, e, ni[17])
        return (None, None, None, ds2.view(ctx.in_shapes[0]), g_o2.view(ctx.in_shapes[1]), None, gwo, gbo, gg2, gb2, gw1, gbb1, gw2, gbb2, gg3,
                gb3, gnw, gnb)


def spread(n: int) -> int:
    """Workgroups per row block of the programs WITHOUT a sliced stage (``n`` asked for): stores, column sums and
    independent products move to different workgroups of the block, every workgroup repeating the stages they depend
    on — a 128 KB weight block is ~2 us of one CU's L2 bandwidth, a store stage ~1.2 us, and 231 CUs are idle.
    MBV_RC_SPREAD=0: one workgroup per block (A/B)."""
    return n if switches.get('rc_spread') else 1


def ffn_split(f: int, e: int, dt: torch.dtype) -> int:
    """Workgroups per row block of the decoder MLP's split launches (1: the one-workgroup stage / the staged f32 form)."""
    fused = dt != torch.float32 and f % 256 == 0 and e % 32 == 0 and switches.get('rc_ffn')
    return f // 256 if (fused and switches.get('rc_split')) else 1


def enabled(dtype: torch.dtype = torch.bfloat16) -> bool:
    """Whether the row-chain decoder is wanted for compute dtype ``dtype`` (`switches.decoder_fused`).  In fp32 the chains'
    GEMM stages are exact-f32 MFMA products of 16-row blocks (25 workgroups, 32 MAC / clk / SIMD): 54 launches of ~ 110 us
    against the per-op path's few-row library GEMMs — 69.0 vs 72.1 scans/s for the fp32 step (round 4), so 'auto' keeps
    the chains to the 16-bit modes."""
    mode = str(switches.get('decoder_fused'))
    if mode in ('0', 'False'):
        return False
    if mode in ('1', 'True'):
        return True
    return dtype != torch.float32
