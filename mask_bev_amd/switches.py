"""Path selectors of the product, in ONE place and independent of the process environment.

Every alternative path the package can take (a per-op form next to a fused kernel, a grouped launch next to per-layer
launches) is selected by a named value here.  The defaults are the measured-best configuration DESIGN.md documents;
nothing in the product path reads ``os.environ`` for them, so two runs of the same build compute the same thing whatever
shell they were started from.  Tests that compare two forms of one operation use ``override(...)`` (a context manager),
A/B timing scripts under ``scratch/`` call ``load_from_env()`` explicitly.
"""
from __future__ import annotations

import contextlib
import os
from typing import Any, Dict

_DEFAULTS: Dict[str, Any] = {
    # --- decoder ---------------------------------------------------------------------------------------------------
    'decoder_fused': 'auto',      # K19 row chains for the decoder's query side: 'auto' = in the 16-bit modes (in fp32 the
                                  # per-op path measures faster: 72.1 vs 69.0 scans/s), '1' = wherever supported, '0' = never
    'rc_ffn': True,               # the MLP pair as one FFN stage of a row chain
    'rc_split': True,             # MLP stage sliced over f / 256 workgroups
    'rc_spread': True,            # independent stages of a program spread over 2-3 workgroups
    'gq_stash': True,             # the batched heads' query gradients join the row-chain backward programs (no autograd adds)
    'stack_grad_sink': True,      # K8 backward stores the stacked logit gradient as the deferred heads read it
    'deferred_heads': True,       # prediction heads' backward as one batched pass
    'shared_kv': True,            # one key / value projection per memory level
    'skv_direct': True,           # shared-K/V weight gradients straight into the arena rows
    # --- launch structure ------------------------------------------------------------------------------------------
    'tail_stream': True,          # FPN tail on a second stream
    'overlap_matcher': True,      # matcher branch beside the importance sampling
    'early_targets': True,        # the loss's batch-only preparation forks where the head's forward began
    'loss_node': True,            # dice / BCE algebra as one autograd node
    'match_fused': True,          # matcher products on MFMA from half pairs, terms never written (K13c)
    'loss_glue': True,            # matching-cost assembly and the class loss as single launches (K13)
    'k3_adam': True,              # one GPU, graph step: AdamW of the (C, ny, nx) LayerNorm affine inside K3's backward (their gradients never reach memory)
    'tn_overlap': False,          # early stages' grouped weight gradients beside the encoder backward (measured slower)
    'msda_bwd_overlap': False,    # K5 backward's two parts on two streams (measured slower)
    'wgrad_group': True,          # grouped small-token weight gradients / column sums at the end of a backward pass
    'tn_group': '1',              # '0' | '1' | 'all': which K17 weight gradients join the grouped launch
    'nn_colsum_defer': True,      # a fused data gradient's bias column sums join the pass's grouped column-sum launch
    # --- kernels selected over a library / ATen form ----------------------------------------------------------------
    'gemm16': 'auto',             # '0' | 'auto' | 'all': which Linear work runs on K17
    'tn_max_in': 1536,            # widest input of a Linear whose weight gradient K17 takes
    'k17_fused_min': 4096,        # fewest tokens of an FFN that takes the fused K17 pair
    'gemm32s': True,              # fp32 compute: token-major Linears on K20 (f32 products from IEEE-half pairs on the 16-bit MFMA)
    # K20's operand scales from absmax records / bounds a producing K20 product leaves (its epilogue max-combines what it stores:
    # one no-return atomic per workgroup), instead of a pass over the tensor: 92.6 / 92.9 -> 93.6 / 95.1 scans/s (fp32, one box)
    'amax_hints': True,
    'amax_verify': False,         # debug: every record a K20 product consumes is checked against a fresh absmax of its operand (ops.AMAX_VERIFY)
    'ffn32': True,                # fp32 compute: the FFN pair on K20 with the activation / its derivative in the GEMM epilogues
    'k4_split': True,             # fp32 compute: K4's products on the 16-bit matrix pipe from IEEE-half pairs (f32 tensors)
    'k6_split': True,             # fp32 compute: the decoder attention's products the same way (per-tile scales)
    'msda_packed_f32': True,      # fp32 compute: K5's value gradient in the packed fixed-point form of the 16-bit modes
    'ln_bound_hints': True,       # fp32 compute: a LayerNorm output's K20 scale from sqrt(C) max|gamma| + max|beta| instead of a pass over it
    'conv3x3_k20': True,          # fp32 compute: the pixel decoder's 3 x 3 convolution as K20 products on a zero-bordered channels-last copy (no MIOpen)
    'conv3x3_k17': True,          # 16-bit compute: the same convolution as K17 products
    'tn32_group': True,           # fp32 compute: the few-row weight gradients of a backward pass as one grouped K20 launch
    'gemm32s_min': 1024,          # fewest tokens of an f32 Linear that takes K20 (below: the library's f32 GEMM; measured: scratch/bench_gemm32s.py)
    'k7_f32_library': True,       # fp32 mask logits through the library's batched GEMM instead of K7's exact-f32 kernel
    'stage_out_lowp': True,       # backbone stage outputs stored in the autocast dtype by their LayerNorm launch
    'pos_fused': True,            # absolute position embedding added inside the first block's K12 launch
    'ln_branch_lowp': True,       # a post-LN output's branch copy written in 16 bits by K12 itself
    'ln_fanout': True,
    'conv1x1_tokens': True,
    'pos_share': True,            # one d(pos) product for the pixel decoder's six layers (ops.PosGradShare)
    'pfn_stream_stats': True,     # pillar term inside the Linear's launch + streaming BatchNorm statistics (one-call forward)
    'pfn_one_call': True,         # all PFN layers' forward behind one C-ABI call (host time of the eager section)
    'pfn_skinny': True,           # the PFN's f32 Linears on K2c (streaming exact-f32 MFMA GEMM)
    'msda_value_lowp': True,      # K5's value map in the compute dtype (16-bit modes)
    'msda_fused': True,
    'msda_packed': True,          # packed fixed-point value gradient in the 16-bit modes
    'k9_padded': True,
    'groupnorm': True,
    'merge_ln': True,
    'tuned_gemms': True,
}

_values: Dict[str, Any] = dict(_DEFAULTS)


def get(name: str) -> Any:
    return _values[name]


def defaults() -> Dict[str, Any]:
    return dict(_DEFAULTS)


def _coerce(name: str, value: Any) -> Any:
    d = _DEFAULTS[name]
    if isinstance(d, bool):
        if isinstance(value, str):
            return value not in ('0', '', 'false', 'False')
        return bool(value)
    if isinstance(d, int):
        return int(value)
    return str(value)


def set_value(name: str, value: Any) -> None:
    if name not in _DEFAULTS:
        raise KeyError(f'unknown switch {name!r}')
    _values[name] = _coerce(name, value)


@contextlib.contextmanager
def override(**kw: Any):
    """``with switches.override(decoder_fused='0'): ...`` — for tests that compare two forms of one operation."""
    old = {k: _values[k] for k in kw}
    try:
        for k, v in kw.items():
            set_value(k, v)
        yield
    finally:
        _values.update(old)


def load_from_env(prefix: str = 'MBV_') -> Dict[str, Any]:
    """A/B scripts only (``scratch/``): take ``MBV_<NAME>=value`` settings from the environment.  Never called by the
    package, bench.py or the launcher."""
    taken = {}
    for k in _DEFAULTS:
        e = os.environ.get(prefix + k.upper())
        if e is not None:
            set_value(k, e)
            taken[k] = _values[k]
    return taken


def patch(monkeypatch, **kw: Any) -> None:
    """pytest helper: set values through a ``monkeypatch`` fixture (restored at the end of the test)."""
    for k, v in kw.items():
        if k not in _DEFAULTS:
            raise KeyError(f'unknown switch {k!r}')
        monkeypatch.setitem(_values, k, _coerce(k, v))
