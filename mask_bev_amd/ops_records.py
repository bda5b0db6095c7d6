"""The absmax-record REGISTRIES of the fp32 compute mode (K20, csrc/gemm_f32s.hip) — one module that owns their invariants.

K20 forms f32 products from IEEE-half pairs and needs, per operand tensor, a power-of-two scale from max|x|.  That maximum
travels as a RECORD (64 device words whose maximum is the bits of max|x| or of a bound of it).  Four registries hand records out:

* pools (`amax_record`): zeroed records in blocks of 256, one pool per (device, thread, stream); a block never spans the start
  of a stream capture, and capture GENERATIONS (`amax_new_capture`) keep a second captured step from taking the first's;
* hints (`amax_hint_set / _get / _refresh`): the record a producer left for a tensor, found by the tensor's address, valid
  while that very tensor object is alive, unmodified (version) and in the capture state / generation it was made in.  A missed
  hint costs an absmax pass — never accuracy: callers clear `_LAST_HINT` before every `Function.apply` they refresh after;
* static records (`static_amax_register`): ONE persistent record for a long-lived buffer that crosses the eager / captured
  line (the graph step's input map; K3 clears and rewrites it every step);
* parameter records (`weight_amax`, `ln_bound`): keyed by the optimizer epoch (`note_parameters_changed`), refreshed in one
  grouped launch per 64.

`switches.amax_verify` (`AMAX_VERIFY`) checks every record a product consumes against a fresh maximum (debug; tests)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, switches
from ._lib import MaskBevHipError, check
from .ops_core import *          # noqa: F401,F403


# --------------------------------------------------------------------------------------
# K20 f32 GEMMs from IEEE-half pairs on the 16-bit matrix pipe (csrc/gemm_f32s.hip) — the fp32 compute mode's Linears
# --------------------------------------------------------------------------------------
AMAX_SLOTS = 64          # words per absmax record (csrc/gemm_f32s.hip kAmaxSlots)


# Capture generation: bumped once per captured training step (graph.py, before its first capture).  Records, hints and
# weight / LayerNorm-bound entries made inside a capture are tagged with it, so that a SECOND captured step of the same
# process (tests, a re-capture) never takes a record whose zero-fill / refresh launch was captured in the previous graph —
# the new graph's replays would not re-zero it and it would become a running maximum over all steps.
CAPTURE_ID = [0]


def amax_new_capture() -> int:
    CAPTURE_ID[0] += 1
    return CAPTURE_ID[0]


def _capture_tag() -> int:
    """0 outside a stream capture, the capture generation (>= 1) inside one."""
    return max(1, CAPTURE_ID[0]) if torch.cuda.is_current_stream_capturing() else 0


def static_amax_wanted() -> bool:
    """Whether K3 writes the absmax record of its f32 map (the condition `_ScatterLayerNorm.forward` tests): a static
    input buffer is registered (static_amax_register) only then — a registered record nobody writes would read as
    "max|x| = 0" and run the patch projection unscaled."""
    return bool(switches.get('amax_hints') and switches.get('ln_bound_hints') and switches.get('gemm32s'))


class _AmaxPool:
    """Absmax records for K20 (csrc/gemm_f32s.hip): 64 device words each, whose maximum is the BITS of max|x| (or of a bound
    of it), max-combined by `mbv_f32_absmax_group` or by a producer kernel — so a record must be zero before its tensor's
    launch.  Records are handed out one after the other from zero-filled blocks of 256 (one fill launch per block instead
    of one per tensor).  A block never spans the start of a stream capture: a block filled eagerly would not be zeroed
    again by the replay, and a record would then hold the maximum over ALL replays."""

    def __init__(self):
        self.block, self.used, self.capturing = None, 0, False

    def take(self, device, n: int = 1) -> torch.Tensor:
        cap = _capture_tag()
        if (self.block is None or self.used + n > self.block.shape[0] or cap != self.capturing
                or self.block.device != device):
            self.block = torch.zeros((256, AMAX_SLOTS), dtype=torch.int32, device=device)
            self.used, self.capturing = 0, cap
        out = self.block[self.used:self.used + n]
        self.used += n
        return out


_AMAX_POOLS: dict = {}


def amax_record(device, n: int = 1) -> torch.Tensor:
    """(n, 64) int32 zeroed absmax records on the current stream (see :class:`_AmaxPool`)."""
    # one pool per (device, thread, stream): a block is zero-filled on the stream that is current when it is made, and a
    # record handed to a launch on another stream could be read before that fill ran
    import threading
    pool = _AMAX_POOLS.setdefault((device, threading.get_ident(), torch.cuda.current_stream(device).cuda_stream), _AmaxPool())
    return pool.take(device, n)


def f32_absmax(tensors) -> torch.Tensor:
    """(len(tensors), 64) int32 absmax records of the f32 matrices (rows may be strided), one launch."""
    lib = _lib.load()
    n = len(tensors)
    dev = tensors[0].device
    for t in tensors:
        if (not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.shape[1] % 4
                or t.stride(0) % 4 or t.data_ptr() % 16):
            raise MaskBevHipError('f32_absmax: f32 matrices with contiguous, 16-byte aligned rows (cols % 4 == 0) only')
    out = amax_record(dev, n)
    PA, LA = ctypes.c_void_p * n, ctypes.c_int64 * n
    check(lib.mbv_f32_absmax_group(PA(*[t.data_ptr() for t in tensors]), LA(*[t.shape[0] for t in tensors]),
                                   LA(*[t.shape[1] for t in tensors]), LA(*[t.stride(0) for t in tensors]),
                                   PA(*[out.data_ptr() + 4 * AMAX_SLOTS * i for i in range(n)]), n, _stream()),
          'mbv_f32_absmax_group')
    return out


class _AmaxVerify:
    """``switches.amax_verify`` (debug, VERDICT r05 #6a): every absmax record a K20 product is about to consume — a producer's
    hint, a derived bound, a static / weight / LayerNorm-bound record — is compared with a FRESH max|operand|, taken by a
    torch reduction right in front of the product on the same stream.  Nothing is read on the host there (the pair
    (true maximum, record's value) goes into a small device tensor), so the check also runs inside a stream capture and
    again on every replay of the captured graph; :meth:`report` synchronises and returns the pairs.
    Invariant under test: record >= max|x| (a smaller one overflows IEEE half once x is scaled by 2^13 / record) and not
    absurdly larger (every binade of slack is a bit of the 22-bit product lost)."""

    CAPACITY = 8192

    def __init__(self):
        self.entries, self.fresh, self.cons = [], None, None

    def reset(self):
        self.entries = []

    def check(self, t: torch.Tensor, rec, what: str) -> None:
        if rec is None or not switches.get('amax_verify'):
            return
        if self.fresh is None or self.fresh.device != t.device:
            if torch.cuda.is_current_stream_capturing():
                raise MaskBevHipError('amax_verify: run one eager step first (the result rows are allocated outside the capture)')
            # persistent result rows, allocated outside any capture: a captured check rewrites ITS rows on every replay, and no
            # temporary of the check lives in a graph's private pool (the fresh maximum comes from the library's own absmax
            # kernel straight into its row — no torch reduction, no scratch)
            self.fresh = torch.zeros((self.CAPACITY, AMAX_SLOTS), dtype=torch.int32, device=t.device)
            self.cons = torch.zeros((self.CAPACITY, AMAX_SLOTS), dtype=torch.int32, device=t.device)
        i = len(self.entries)
        if i >= self.CAPACITY:
            raise MaskBevHipError('amax_verify: more checks than result rows')
        with torch.no_grad():
            t2 = t.detach()
            if t2.dim() != 2:
                t2 = t2.reshape(-1, t2.shape[-1])
            ok = (t2.dtype == torch.float32 and t2.stride(1) == 1 and t2.shape[1] % 4 == 0 and t2.stride(0) % 4 == 0
                  and t2.data_ptr() % 16 == 0)
            self.fresh[i].zero_()
            if ok:
                PA, LA = ctypes.c_void_p * 1, ctypes.c_int64 * 1
                check(_lib.load().mbv_f32_absmax_group(PA(t2.data_ptr()), LA(t2.shape[0]), LA(t2.shape[1]), LA(t2.stride(0)),
                                                       PA(self.fresh.data_ptr() + 4 * AMAX_SLOTS * i), 1, _stream()),
                      'mbv_f32_absmax_group')
            else:
                torch.add(t2.abs().max().float().view(1).view(torch.int32).expand(AMAX_SLOTS), 0, out=self.fresh[i])
            torch.add(rec.reshape(-1)[:AMAX_SLOTS], 0, out=self.cons[i])      # the record as the product is about to read it
        self.entries.append((what, tuple(t.shape), bool(torch.cuda.is_current_stream_capturing())))

    def report(self):
        """[(what, shape, captured, max|x|, record)] after a device synchronisation."""
        if not self.entries:
            return []
        torch.cuda.synchronize()
        n = len(self.entries)
        fresh = self.fresh[:n].max(1).values.view(torch.float32).cpu().tolist()
        cons = self.cons[:n].max(1).values.view(torch.float32).cpu().tolist()
        return [(w, s, c, f, r) for (w, s, c), f, r in zip(self.entries, fresh, cons)]


AMAX_VERIFY = _AmaxVerify()


def operand_amax(tensors, activations=None):
    """One-record tensors for the f32 matrices ``tensors``.  For the ACTIVATIONS among them (``activations[i]``; default: all)
    the record a producer — or an earlier product that read the same tensor — left as a hint is taken when there is one, and
    a record computed here is left as a hint in turn: the data gradient and the weight gradient of a layer read the same
    output gradient, a forward product and the weight gradient the same input (``switches.amax_hints``; a tensor rewritten
    through torch bumps its version and loses the hint).  Weights never take part: the optimizer rewrites them through raw
    pointers (their records are :func:`weight_amax`'s, keyed by the optimizer epoch).  One absmax launch for what is left."""
    hints = bool(switches.get('amax_hints'))
    act = [True] * len(tensors) if activations is None else list(activations)
    recs = [amax_hint_get(t) if (hints and a) else None for t, a in zip(tensors, act)]
    todo = [i for i, r in enumerate(recs) if r is None]
    if todo:
        new = f32_absmax([tensors[i] for i in todo])
        for j, i in enumerate(todo):
            recs[i] = new[j:j + 1]
            if hints and act[i]:
                amax_hint_set(tensors[i], recs[i])
    return recs


# Absmax HINTS: K20's epilogue can max-combine the values it stores into a record while they are in its registers, and the
# wrappers carry that record — or a bound derived from it: |gelu(z)| <= |z|, a window-attention output is a convex combination
# of v rows, |act'| <= 1.13 — to the next K20 product that reads the tensor (fc2's input behind fc1 + GELU, proj's input
# behind qkv + attention, fc1's output gradient behind fc2's data gradient) — found by the tensor's address, valid only while the very tensor object is alive and unmodified
# (weak reference + version).  A consumer without a valid hint runs the absmax pass: a missed hint costs time, never accuracy.
# A record made OUTSIDE a stream capture must not be baked into a captured launch (the replay would read the address of that one
# eager step's record for ever) and vice versa: a hint is valid only in the capture state it was made in.  What crosses that line
# — the eager encoder's map that a captured graph reads — has a REGISTERED persistent record instead (static_amax_register).
_AMAX_HINTS: dict = {}
_LAST_HINT = [0, None]
_STATIC_RECS: dict = {}


def static_amax_register(buf: torch.Tensor) -> torch.Tensor:
    """Give a long-lived buffer (the static input of a captured graph, graph.py) ONE persistent (1, 64) absmax record: the
    producer that refills the buffer every step (K3) clears and rewrites it, every K20 product that reads the buffer — under
    whatever tensor object, inside or outside a capture — finds it by the buffer's address while ``buf`` itself is alive."""
    import weakref
    for k in [k for k, (ref, _) in _STATIC_RECS.items() if ref() is None]:
        del _STATIC_RECS[k]
    rec = torch.zeros((1, AMAX_SLOTS), dtype=torch.int32, device=buf.device)
    _STATIC_RECS[(buf.data_ptr(), buf.numel())] = (weakref.ref(buf), rec)
    return rec


def static_amax_record(t: torch.Tensor) -> Optional[torch.Tensor]:
    e = _STATIC_RECS.get((t.data_ptr(), t.numel()))
    if e is None:
        return None
    src = e[0]()
    if src is None or src.data_ptr() != t.data_ptr() or src.numel() != t.numel() or e[1].device != t.device:
        return None
    return e[1]


def amax_hint_set(t: torch.Tensor, rec: Optional[torch.Tensor]) -> None:
    if rec is None or not torch.is_tensor(t) or not t.is_cuda:
        return
    import weakref
    if len(_AMAX_HINTS) > 512:
        for k in [k for k, e in _AMAX_HINTS.items() if e[0]() is None]:
            del _AMAX_HINTS[k]
        if len(_AMAX_HINTS) > 512:
            _AMAX_HINTS.clear()
    cap = _capture_tag()
    base = t._base
    if base is not None and base.data_ptr() == t.data_ptr() and base.numel() == t.numel() and base.dtype == t.dtype:
        t = base          # a reshaped view of the whole tensor: the hint lives with the tensor, not with the temporary view object
    _AMAX_HINTS[t.data_ptr()] = (weakref.ref(t), t._version, rec, cap)
    _LAST_HINT[0], _LAST_HINT[1] = t.data_ptr(), rec


def amax_hint_refresh(t) -> None:
    """After ``Function.apply``: the tensor object the caller holds may be a new wrapper of the one the forward hinted (or the
    same buffer with its version bumped by ``mark_dirty``).  Callers clear ``_LAST_HINT[1]`` BEFORE the apply: the match is
    by address, and a forward that sets no hint (library path) would otherwise re-attach the record of an earlier, already
    freed tensor whose address the caching allocator handed to this output (an f16 overflow, not "time")."""
    if torch.is_tensor(t) and t.is_cuda and _LAST_HINT[0] == t.data_ptr() and _LAST_HINT[1] is not None \
            and amax_hint_get(t) is None:
        amax_hint_set(t, _LAST_HINT[1])


def amax_hint_get(t: torch.Tensor) -> Optional[torch.Tensor]:
    if _STATIC_RECS:
        rec = static_amax_record(t)
        if rec is not None:
            return rec
    capturing = _capture_tag() if t.is_cuda else 0
    e = _AMAX_HINTS.get(t.data_ptr())
    if e is not None:
        ref, version, rec, cap = e
        src = ref()
        if (src is not None and src.data_ptr() == t.data_ptr() and src.numel() == t.numel() and src._version == version
                and rec.device == t.device and cap == capturing):
            return rec
    # a slice (column block, row range) of a hinted tensor: the whole tensor's record bounds it
    base = t._base
    if base is not None and base is not t and base.dtype == t.dtype:
        e = _AMAX_HINTS.get(base.data_ptr())
        if e is not None:
            ref, version, rec, cap = e
            src = ref()
            if (src is not None and src.data_ptr() == base.data_ptr() and src.numel() == base.numel()
                    and src._version == version and rec.device == t.device and cap == capturing):
                return rec
    return None


def _hinted_view(t: torch.Tensor, shape) -> torch.Tensor:
    """``t.view(shape)`` that keeps ``t``'s absmax hint (a view is another tensor object at the same address)."""
    v = t.view(shape)
    if v is not t:
        amax_hint_set(v, amax_hint_get(t))
    return v


def amax_hint_wanted(rows: int) -> bool:
    """Whether a producer of an f32 activation with this many rows should emit an absmax record (K20 will read it)."""
    return gemm32s_wants(rows)


# the absmax record of a WEIGHT is good until the parameters change: keyed by the optimizer epoch (FlatAdam / arena bump it),
# the tensor's version (torch optimizers and copy_ bump that) and whether a stream capture is running (a record computed
# eagerly would go stale inside a replayed graph: within a capture the first use computes it, as a captured launch)
PARAM_EPOCH = [0]
_WEIGHT_AMAX: dict = {}


def note_parameters_changed() -> None:
    PARAM_EPOCH[0] += 1
    _WEIGHT_AMAX.clear()


_WEIGHT_REG: dict = {}          # key -> weak reference: every weight K20 has asked a record for (the grouped refresh's list)
_LN_REG: dict = {}              # gamma's address -> [gamma ref, beta ref or None, record, tag]


def _amax_tag(dev):
    return (PARAM_EPOCH[0], _capture_tag(), torch.cuda.current_stream(dev).cuda_stream)


def weight_amax(w: torch.Tensor) -> torch.Tensor:
    """The absmax record of a weight, good until the parameters change.  A miss refreshes the records of EVERY weight seen so
    far in one launch per 64 (they all went stale together, with the optimizer step): ~ 30 single launches per step otherwise."""
    import weakref
    key = (w.data_ptr(), tuple(w.shape), w.stride(0))
    tag = _amax_tag(w.device)
    e = _WEIGHT_AMAX.get(key)
    if e is not None and e[0] == (tag, w._version):
        return e[1]
    # (the list is per stream: a weight used on a side stream is refreshed by that stream's first miss, not by every stream's)
    reg = _WEIGHT_REG.get(tag[2])
    if reg is None:
        # a stream seen for the first time (the capture stream of a graph): it starts from every weight any stream has used, so
        # that its first miss is one grouped refresh and not one launch per weight baked into the graph
        reg = _WEIGHT_REG[tag[2]] = {k2: r for other in list(_WEIGHT_REG.values()) for k2, r in other.items()}
    reg[key] = weakref.ref(w)
    todo = []
    for k2, ref in list(reg.items()):
        t = ref()
        if t is None or (t.data_ptr(), tuple(t.shape), t.stride(0)) != k2 or t.device != w.device:
            if t is None:
                del reg[k2]
            continue
        e2 = _WEIGHT_AMAX.get(k2)
        if e2 is None or e2[0] != (tag, t._version):
            todo.append((k2, t))
    if len(_WEIGHT_AMAX) > 4096:
        _WEIGHT_AMAX.clear()
    for c in range(0, len(todo), 64):
        chunk = todo[c:c + 64]
        recs = f32_absmax([t for _, t in chunk])
        for i, (k2, t) in enumerate(chunk):
            _WEIGHT_AMAX[k2] = ((tag, t._version), recs[i:i + 1])
    return _WEIGHT_AMAX[key][1]


def ln_bound(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The absmax-BOUND record of a LayerNorm's output, sqrt(C) max|weight| + max|bias| (mbv_ln_bound_group), good until the
    parameters change; a miss refreshes every LayerNorm seen so far in one launch."""
    import weakref
    if (not weight.is_cuda or weight.dtype != torch.float32 or not weight.is_contiguous()
            or (bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != weight.numel()))):
        return None
    tag = _amax_tag(weight.device)
    key = weight.data_ptr()

    e = _LN_REG.get(key)
    if e is not None and e[3] == (tag, weight._version, None if bias is None else bias._version) and e[0]() is weight:
        return e[2]
    _LN_REG[key] = [weakref.ref(weight), None if bias is None else weakref.ref(bias), amax_record(weight.device), None]
    todo = []
    for k2, e2 in list(_LN_REG.items()):
        g = e2[0]()
        b = e2[1]() if e2[1] is not None else None
        if g is None or g.data_ptr() != k2 or (e2[1] is not None and b is None) or g.device != weight.device:
            del _LN_REG[k2]
            continue
        t2 = (tag, g._version, None if b is None else b._version)
        if e2[3] != t2:
            # a record that a captured launch wrote must not be rewritten eagerly (and vice versa): a fresh one per refresh
            e2[2] = amax_record(g.device)
            todo.append((e2, g, b, t2))
    if todo:
        lib = _lib.load()
        n = len(todo)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
        check(lib.mbv_ln_bound_group(PA(*[g.data_ptr() for _, g, _, _ in todo]),
                                     PA(*[(0 if b is None else b.data_ptr()) for _, _, b, _ in todo]),
                                     IA(*[g.numel() for _, g, _, _ in todo]), PA(*[e2[2].data_ptr() for e2, _, _, _ in todo]),
                                     n, _stream()), 'mbv_ln_bound_group')
        for e2, _, _, t2 in todo:
            e2[3] = t2
    return _LN_REG[key][2]


def gemm32s_wants(tokens: int) -> bool:
    return bool(switches.get('gemm32s')) and tokens >= int(switches.get('gemm32s_min'))


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
