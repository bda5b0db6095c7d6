"""Algorithmic work of the C-ABI calls: for a call's argument tuple, the name of the kernel that
dominates it, the roofline that bounds that kernel and the algorithmic HBM bytes / MFMA flops of ONE call.
Used by bench.py to turn the HIP-event durations of an instrumented eager step into roofline fractions; the byte
counts are the operands each kernel must read and write once (no re-reads, no workspace traffic).

``MODELS[symbol](args) -> (kernel, bound, bytes, flops)``; argument positions follow include/maskbev_hip.h.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_TFLOPS = 2500.0        # dense bf16 / fp16 MFMA
MFMA_F32_TFLOPS = 157.3

Work = Tuple[str, str, float, float]


def _i(x) -> int:
    return int(getattr(x, 'value', x) or 0)


def _window_attn(a, bwd: bool) -> Work:
    if bwd:
        bf16, b, h, w, c, heads, ws = (_i(a[i]) for i in (6, 7, 8, 9, 10, 11, 12))
    else:
        bf16, b, h, w, c, heads, ws = (_i(a[i]) for i in (3, 4, 5, 6, 7, 8, 9))
    es = 2 if bf16 else 4
    t = b * h * w
    nwin = b * ((h + ws - 1) // ws) * ((w + ws - 1) // ws)
    n, d = ws * ws, c // max(1, heads)
    if bwd:      # reads qkv, out, grad_out; writes grad_qkv; five n x n x d products per (window, head)
        return ('k_window_attn_bwd', 'hbm', t * c * es * (3 + 1 + 1 + 3), 10.0 * nwin * heads * n * n * d)
    return ('k_window_attn_fwd', 'hbm', t * c * es * (3 + 1), 4.0 * nwin * heads * n * n * d)


def _window_attn_split(a, bwd: bool) -> Work:
    """K4 on f32 tensors in the split mode: the same bytes and ALGORITHMIC flops as the f32 form (the kernel issues three 16-bit
    matrix instructions per product term)."""
    if bwd:
        b, h, w, c, heads, ws = (_i(a[i]) for i in (6, 7, 8, 9, 10, 11))
    else:
        b, h, w, c, heads, ws = (_i(a[i]) for i in (3, 4, 5, 6, 7, 8))
    t = b * h * w
    nwin = b * ((h + ws - 1) // ws) * ((w + ws - 1) // ws)
    n, d = ws * ws, c // max(1, heads)
    if bwd:
        return ('k_window_attn_bwd', 'hbm', t * c * 4 * (3 + 1 + 1 + 3), 10.0 * nwin * heads * n * n * d)
    return ('k_window_attn_fwd', 'hbm', t * c * 4 * (3 + 1), 4.0 * nwin * heads * n * n * d)


def _msda(a, bwd: bool) -> Work:
    o = 1 if bwd else 0
    b, nv, heads, d, levels, nq, pts = (_i(a[i + o]) for i in (5, 6, 7, 8, 9, 10, 11))
    samples = b * nq * heads * levels * pts
    vmap, qmap = b * nv * heads * d * 4.0, b * nq * heads * d * 4.0
    if bwd:      # value, grad_out, loc, attn in; grad_value, grad_loc, grad_attn out
        part = _i(a[17])
        if part == 1:    # d(value): grad_out, loc, attn in; grad_value out
            return ('k_msda_bwd_value', 'hbm', vmap + qmap + samples * 12.0, 0.0)
        if part == 2:    # d(location), d(weight): value, grad_out, loc, attn in; grad_loc, grad_attn out
            return ('k_msda_bwd_locattn', 'hbm', vmap + qmap + 2 * samples * 12.0, 0.0)
        return ('k_msda_bwd', 'hbm', 2 * vmap + qmap + 2 * samples * 12.0, 0.0)
    return ('k_msda_fwd_v4', 'hbm', vmap + qmap + samples * 12.0, 0.0)


def _pfn_forward(a) -> Work:
    """mbv_pfn_forward: per layer the Linear(s) (rows read, y written), the statistics pass (y read), apply + max (y read, a
    written): one family, `k_pfn_forward`, for the 8-9 launches per layer it issues."""
    k, v, n = _i(a[5]), _i(a[6]), _i(a[14])
    c, by, fl = _i(a[1]), 0.0, 0.0
    for l in range(n):
        u = int(a[13][l])
        last = l == n - 1
        by += (k * (c + u) + k * u + k * u + (0 if last else k * u) + 6.0 * v * u) * 4.0
        fl += 2.0 * k * c * u + (4.0 * v * c * u if l else 0.0)
        c = u
    return ('k_pfn_forward', 'hbm', by, fl)


def _msda_typed(a, bwd: bool) -> Work:
    """mbv_ms_deform_attn_fwd_v / _bwd_locattn: as _msda, the value map in its own element size."""
    o = 1 if bwd else 0
    ves = 4.0 if _i(a[1 + o]) == 0 else 2.0
    b, nv, heads, d, levels, nq, pts = (_i(a[i + o]) for i in (6, 7, 8, 9, 10, 11, 12))
    samples = b * nq * heads * levels * pts
    vmap, qmap = b * nv * heads * d * ves, b * nq * heads * d * 4.0
    if bwd:
        return ('k_msda_bwd_locattn', 'hbm', vmap + qmap + 2 * samples * 12.0, 0.0)
    return ('k_msda_fwd_v4', 'hbm', vmap + qmap + samples * 12.0, 0.0)


def _msda_value_packed(a) -> Work:
    # grad_out, loc, attn in; grad_value out in its own dtype
    b, nv, heads, d, levels, nq, pts = (_i(a[i]) for i in (3, 4, 5, 6, 7, 8, 9))
    samples = b * nq * heads * levels * pts
    out_es = 4.0 if _i(a[12]) == 0 else 2.0
    # (the re-layout launch reads and writes grad_out / locations / weights once more: not algorithmic)
    return ('k_msda_bwd_value_fx', 'hbm', b * nv * heads * d * out_es + b * nq * heads * d * 4.0 + samples * 12.0, 0.0)


def _rowchain(a) -> Work:
    """K19: a stage program over `rows` token rows.  Algorithmic bytes: every LOAD / STORE row once, every GEMM's weight
    block once per launch (the 25 workgroups re-read it through L2), the LayerNorm statistics; flops: the GEMMs."""
    stages, n, rows, wdt = a[0], _i(a[1]), _i(a[2]), _i(a[5])
    wes = 4.0 if wdt == 0 else 2.0
    by = fl = 0.0
    for i in range(n):
        st = stages[i]
        es = 4.0 if (st.flags & 3) == 0 else 2.0
        if st.op == 0:                      # LOAD
            by += rows * st.n * (es if st.p0 else 0.0)
        elif st.op == 1:                    # STORE (accumulate: read + write)
            by += rows * st.n * es * (2.0 if st.flags & 4 else 1.0)
        elif st.op == 2:                    # GEMM
            by += st.n * st.k * wes
            fl += 2.0 * rows * st.n * st.k
        elif st.op in (3, 4):               # LayerNorm forward / backward: statistics, parameter partials
            by += rows * 8.0
        elif st.op == 7:                    # FFN pair: both weights once, the hidden activations (and their gradient)
            by += 2.0 * st.n * st.k * wes + rows * st.k * 4.0 * (2.0 if st.flags & 16 else 1.0)
            fl += 4.0 * rows * st.n * st.k
        elif st.op == 9:                    # SUM of a split launch's parts
            by += rows * st.n * 4.0 * st.k
    return ('k_rowchain', 'hbm', by, fl)


def _rowchain_split(a) -> Work:
    """The split form: the same stage list (a stage runs once whoever owns it; stages without an owner are REPEATED in
    every workgroup of a row block, which is not algorithmic work); MBV_RC_SPLIT stores write one part per workgroup."""
    name, bound, by, fl = _rowchain(a)
    stages, n, rows, split = a[0], _i(a[1]), _i(a[2]), _i(a[6])
    for i in range(n):
        st = stages[i]
        if st.op == 1 and (st.flags & 64):
            by += rows * st.n * 4.0 * (split - 1)
    return (name, bound, by, fl)


def _point_rows(rows: int, pts: int) -> float:
    """Coordinate reads + sample writes of a point-sampling call: the coordinate sets are shared by the rows that name
    them (one set per (decoder output, scan) for 100 queries), 8 B per point of a distinct set + 4 B per sample."""
    from . import _lib
    _, n_sets = _lib.WORK_HINT.get('point_sample', (rows, rows))
    return min(rows, n_sets) * pts * 8.0 + rows * pts * 4.0


def _point_sample(a, kernel: str, bytes_per_pixel: float) -> Work:
    """Every DISTINCT source map is read once (the ten decoder outputs' rows of a ground-truth mask share it: round 2
    charged one map per row and the model exceeded the measured traffic), plus the coordinate / sample traffic."""
    from . import _lib
    rows, pts, h, w = _i(a[4]), _i(a[5]), _i(a[6]), _i(a[7])
    n_src, _ = _lib.WORK_HINT.get('point_sample', (rows, rows))
    return (kernel, 'hbm', min(rows, n_src) * h * w * bytes_per_pixel + _point_rows(rows, pts), 0.0)


def _attn(a, bwd: bool, ld: bool) -> Work:
    if bwd:
        i0 = 8 if ld else 7
    else:
        i0 = 5 if ld else 4
    bf16, b, q, k, heads, d = (_i(a[i0 + j]) for j in range(6))
    es = 2 if bf16 else 4
    e = heads * d
    if bwd:
        return ('k_attn_bwd', 'hbm', b * (2 * q * e * es + 2 * k * e * es + q * k) + b * (q * e * 4 + 2 * k * e * 4),
                10.0 * b * heads * q * k * d)
    return ('k_attn_fwd_split', 'hbm', b * (q * e * es * 2 + 2 * k * e * es + q * k), 4.0 * b * heads * q * k * d)


def _attn_split(a, bwd: bool) -> Work:
    """K6 on f32 tensors in the split mode: the f32 form's bytes and ALGORITHMIC flops.  fwd: (q, k, v, ld, mask, b, Q, L, heads, d,
    ...); bwd: (q, k, v, ld, mask, out, dout, lse, b, Q, L, heads, d, ...)."""
    i0 = 8 if bwd else 5
    b, q, k, heads, d = (_i(a[i0 + j]) for j in range(5))
    e = heads * d
    if bwd:
        return ('k_attn_bwd', 'hbm', b * (2 * q * e * 4 + 2 * k * e * 4 + q * k) + b * (q * e * 4 + 2 * k * e * 4),
                10.0 * b * heads * q * k * d)
    return ('k_attn_fwd_split', 'hbm', b * (q * e * 4 * 2 + 2 * k * e * 4 + q * k), 4.0 * b * heads * q * k * d)


def _gemm16(a, layout: str) -> Work:
    if layout == 'nt':
        m, n, k, f32 = _i(a[5]), _i(a[6]), _i(a[7]), _i(a[12])
        by = (m * k + n * k) * 2.0 + m * n * (4 if f32 else 2) * (2 if a[4] is not None and _i(a[4]) else 1)
        return ('k_gemm16<NT>', 'mfma', by, 2.0 * m * n * k * max(1, _i(a[14])))
    if layout == 'nn':
        m, n, k, f32 = _i(a[5]), _i(a[6]), _i(a[7]), _i(a[13])
        by = (m * n + n * k) * 2.0 + m * k * (4 if f32 else 2) + (m * k * 2.0 if _i(a[14]) else 0.0)
        return ('k_gemm16<NN>', 'mfma', by, 2.0 * m * n * k * max(1, _i(a[15])))
    if layout == 'nn_parts':         # (g, w, out, aux, parts, parts_bytes, m, n, k, ldg, ldw, ldo, ldaux, dtype, out_f32, act, batch, ...)
        m, n, k, f32 = _i(a[6]), _i(a[7]), _i(a[8]), _i(a[14])
        by = (m * n + n * k) * 2.0 + m * k * (4 if f32 else 2) + (m * k * 2.0 if _i(a[15]) else 0.0)
        return ('k_gemm16<NN>', 'mfma', by, 2.0 * m * n * k * max(1, _i(a[16])))
    m, n, k = _i(a[3]), _i(a[4]), _i(a[5])
    return ('k_gemm16<TN>', 'mfma', (m * n + m * k) * 2.0 + n * k * 4.0 * 2, 2.0 * m * n * k * max(1, _i(a[13])))


def _add_ln(a, bwd: bool) -> Work:
    if bwd:
        rows, c = _i(a[8]), _i(a[9])
        by = rows * c * ((2 if _i(a[1]) else 4) + (0 if not _i(a[2]) else (2 if _i(a[3]) else 4)) + 4 + 4
                         + (2 if _i(a[11]) else 0))
        return ('k_add_ln_bwd', 'hbm', by, 0.0)
    rows, c = _i(a[6]), _i(a[7])
    by = rows * c * ((2 if _i(a[1]) else 4) + (0 if not _i(a[2]) else (2 if _i(a[3]) else 4))
                     + (4 if _i(a[9]) else 0) + (2 if _i(a[11]) else 4))
    return ('k_add_ln_fwd', 'hbm', by, 0.0)


def _add_ln_fwd2(a) -> Work:
    """mbv_add_layernorm_fwd2: as the forward, plus the second copy of y."""
    a = tuple(a[:4]) + tuple(a[5:])                     # without b_rows: the forward's argument order + (y2, y2_dtype)
    name, bound, by, fl = _add_ln(tuple(a[:12]) + tuple(a[14:]), False)
    rows, c = _i(a[6]), _i(a[7])
    return (name, bound, by + (rows * c * (2 if _i(a[13]) else 4) if _i(a[12]) else 0), fl)


def _add_ln_bwd2(a) -> Work:
    """mbv_add_layernorm_bwd2: dy (+ dy2) (+ ds) and the f32 sum read, dx (+ its 16-bit copy) written."""
    rows, c = _i(a[10]), _i(a[11])
    by = rows * c * ((2 if _i(a[1]) else 4) + (0 if not _i(a[2]) else (2 if _i(a[3]) else 4))
                     + (0 if not _i(a[4]) else (2 if _i(a[5]) else 4)) + 4 + 4 + (2 if _i(a[13]) else 0))
    return ('k_add_ln_bwd', 'hbm', by, 0.0)


def _wgrad_group(a) -> Work:
    n = _i(a[7])
    t, o, i = a[4], a[5], a[6]
    by = sum((t[j] * (o[j] + i[j]) + 2 * o[j] * i[j]) * 4.0 for j in range(n))
    fl = sum(2.0 * t[j] * o[j] * i[j] for j in range(n))
    return ('k_wgrad_small_group', 'mfma_f32', by, fl)


def _tn_group(a) -> Work:
    n = _i(a[8])
    m, o, k = a[3], a[4], a[5]
    by = sum(m[j] * (o[j] + k[j]) * 2.0 + 2 * o[j] * k[j] * 4.0 for j in range(n))
    fl = sum(2.0 * m[j] * o[j] * k[j] for j in range(n))
    return ('k_gemm16_tn_group', 'mfma', by, fl)        # bench.py prices the family against the roofline that bounds it


def _sz(flag) -> float:
    return 4.0 if _i(flag) == 0 else 2.0


def _groupnorm(a, bwd: bool) -> Work:
    if bwd:                                     # dy + x read, dx written (the per-plane sums pass re-reads dy and x)
        n = _i(a[8]) * _i(a[9]) * _i(a[10]) * _i(a[11])
        return ('k_gn_bwd', 'hbm', n * (_sz(a[1]) + _sz(a[3]) + _sz(a[15])), 0.0)
    n = _i(a[2]) * _i(a[3]) * _i(a[4]) * _i(a[5])
    add = _i(a[2]) * _i(a[3]) * _i(a[12]) * _i(a[13]) * _sz(a[11]) if _i(a[10]) else 0.0
    return ('k_gn_fwd', 'hbm', n * (_sz(a[1]) + _sz(a[16])) + add, 0.0)


def _merge_ln(a, bwd: bool) -> Work:
    if bwd:
        n = _i(a[6]) * _i(a[7]) * _i(a[8]) * _i(a[9])
        return ('k_add_ln_bwd', 'hbm', n * (_sz(a[1]) + 4.0 + 4.0), 0.0)
    n = _i(a[1]) * _i(a[2]) * _i(a[3]) * _i(a[4])
    return ('k_add_ln_fwd', 'hbm', n * (4.0 + _sz(a[9])), 0.0)


def _colsum_group(a) -> Work:
    n = _i(a[6])
    dt, rows, cols = a[1], a[2], a[3]
    by = sum(rows[j] * cols[j] * (4.0 if dt[j] == 0 else 2.0) + cols[j] * 8.0 for j in range(n))
    return ('k_colsum_group', 'hbm', by, 0.0)


def _pfn(a, kernel: str, rows_per_unit: float, pillars_per_unit: float, iv: int, iu: int) -> Work:
    """K2b per-pillar kernels: `rows_per_unit` (K, U) and `pillars_per_unit` (V, U) f32 tensors read or written once.
    The row count K is not an argument of these calls (rows live behind row_start): bench.py leaves it in WORK_HINT."""
    from . import _lib
    k = _lib.WORK_HINT.get('pfn_rows', 0)
    v, u = _i(a[iv]), _i(a[iu])
    return (kernel, 'hbm', (rows_per_unit * k + pillars_per_unit * v) * u * 4.0, 0.0)


def _decorate(a) -> Work:
    from . import _lib
    k, dim = _lib.WORK_HINT.get('pfn_rows', 0), _i(a[1])
    return ('k_pfn_decorate', 'hbm', k * (dim * 4.0 + (dim + 7) * 4.0 + 8.0 + 4.0), 0.0)     # point, row, row_pillar, index


def _voxelize(a) -> Work:
    """K1: every point read once (point_dim f32) and its (cell, index) key pair written / read by the sort passes'
    first and last step; pillar outputs (coors, counts, 32 point indices per pillar) and the dense cell map."""
    n, dim, batch, gx, gy = _i(a[2]), _i(a[1]), _i(a[4]), _i(a[14]), _i(a[15])
    pillars = _i(a[20])
    return ('k_voxelize (22 launches)', 'hbm', n * (dim * 4.0 + 16.0 + 8.0) + pillars * (16.0 + 4.0 + 128.0 + 4.0)
            + batch * gx * gy * 4.0, 0.0)


def _msda_prepare(a, bwd: bool) -> Work:
    if bwd:
        b, nq, heads, levels, pts, lo = (_i(a[i]) for i in (4, 5, 6, 7, 8, 9))
        per = b * nq * heads * levels * pts
        return ('k_msda_prepare_bwd', 'hbm', per * (12.0 + 4.0) + per * 3.0 * (2.0 if lo else 4.0), 0.0)
    lo, b, nq, heads, levels, pts = (_i(a[i]) for i in (2, 5, 6, 7, 8, 9))
    per = b * nq * heads * levels * pts
    return ('k_msda_prepare_fwd', 'hbm', per * 3.0 * (2.0 if lo else 4.0) + per * 12.0, 0.0)


def _copy_group(a, kernel: str, es_index: int) -> Work:
    rows, cols, n = a[2], a[3], _i(a[6] if kernel == 'k_fragment_group' else a[4])
    es = 2.0 if kernel == 'k_fragment_group' else float(_i(a[es_index]))
    return (kernel, 'hbm', sum(2.0 * rows[j] * cols[j] * es for j in range(n)), 0.0)


# ---- work of the library calls torch makes for the path (bench.py's dispatch-mode timer) -----------------------------
def gemm_work(m: int, n: int, k: int, batch: int, es_in: float, es_out: float, extra_bytes: float = 0.0) -> Tuple[float, float]:
    return batch * ((m * k + k * n) * es_in + m * n * es_out) + extra_bytes, 2.0 * batch * m * n * k


MODELS: Dict[str, Callable[[tuple], Work]] = {
    'mbv_voxelize': _voxelize,
    'mbv_pfn_decorate': lambda a: _decorate(a),
    # stats: y read (+ written when t is added); apply: y read, a written (not for the last layer), m written
    'mbv_pfn_stats': lambda a: _pfn(a, 'k_pfn_stats', 2.0 if a[1] else 1.0, 3.0 if a[1] else 1.0, 5, 6),
    'mbv_pfn_apply_max': lambda a: _pfn(a, 'k_pfn_apply_max', 2.0 if a[9] else 1.0, 3.0 if a[10] else 2.0, 6, 7),
    # route: y, dA (when present) read, dz written; dM, y_pad, dz_pad per pillar.  bn: y, dz read, dy written
    'mbv_pfn_bwd_route': lambda a: _pfn(a, 'k_pfn_bwd_route', 3.0 if _i(a[7]) else 2.0, 4.0, 12, 13),
    'mbv_pfn_bwd_bn': lambda a: _pfn(a, 'k_pfn_bwd_bn', 3.0, 4.0, 12, 13),
    'mbv_msda_prepare_fwd': lambda a: _msda_prepare(a, False),
    'mbv_msda_prepare_fwd_ld': lambda a: _msda_prepare((None, None) + tuple(a[6:7]) + (None, None) + tuple(a[9:14]), False),
    'mbv_msda_prepare_bwd': lambda a: _msda_prepare(a, True),
    'mbv_msda_prepare_bwd_ld': lambda a: _msda_prepare(a, True),
    'mbv_msda_query_inputs': lambda a: ('k_msda_query_inputs', 'hbm',
                                        _i(a[2]) * _i(a[4]) * (4.0 + 2.0 + 2.0) + _i(a[3]) * _i(a[4]) * 4.0, 0.0),
    'mbv_fragment_group': lambda a: _copy_group(a, 'k_fragment_group', 0),
    'mbv_transpose_group': lambda a: _copy_group(a, 'k_transpose_group', 5),
    'mbv_hungarian_padded': lambda a: ('k_hungarian', 'hbm', _i(a[1]) * _i(a[2]) * _i(a[3]) * 4.0 + _i(a[1]) * _i(a[2]) * 4.0, 0.0),
    'mbv_uniform_points': lambda a: ('k_uniform_points', 'hbm', _i(a[1]) * _i(a[2]) * 8.0, 0.0),
    'mbv_pack_binary_masks': lambda a: ('k_pack_binary', 'hbm', _i(a[1]) * _i(a[2]) * _i(a[3]) * (4.0 + 1.0 / 8.0), 0.0),
    # the bilinear resize reads four taps per OUTPUT pixel: a 16 x 16 mask of a 128 x 128 map touches 1 024 of its 16 384 logits
    'mbv_attn_mask_from_logits': lambda a: ('k_attn_mask', 'hbm',
                                            _i(a[2]) * (min(_i(a[3]) * _i(a[4]), 4 * _i(a[5]) * _i(a[6]))
                                                        * (2.0 if _i(a[1]) else 4.0) + _i(a[5]) * _i(a[6])), 0.0),
    'mbv_gemm16_nt_acc': lambda a: ('k_gemm16<NT,acc>', 'mfma',
                                    _i(a[11]) * ((_i(a[3]) + _i(a[4])) * _i(a[5]) * 2.0 + 2.0 * _i(a[3]) * _i(a[4]) * 4.0),
                                    2.0 * _i(a[11]) * _i(a[3]) * _i(a[4]) * _i(a[5])),
    'mbv_window_attn_fwd': lambda a: _window_attn(a, False),
    'mbv_window_attn_bwd': lambda a: _window_attn(a, True),
    'mbv_window_attn_split_fwd': lambda a: _window_attn_split(a, False),
    'mbv_window_attn_split_bwd': lambda a: _window_attn_split(a, True),
    'mbv_ms_deform_attn_fwd': lambda a: _msda(a, False),
    'mbv_ms_deform_attn_bwd': lambda a: _msda(a, True),
    'mbv_ms_deform_attn_fwd_v': lambda a: _msda_typed(a, False),
    'mbv_ms_deform_attn_bwd_locattn': lambda a: _msda_typed(a, True),
    'mbv_ms_deform_attn_bwd_value_packed': _msda_value_packed,
    'mbv_rowchain_run': _rowchain,
    'mbv_rowchain_run_split': _rowchain_split,
    'mbv_attn_fwd': lambda a: _attn(a, False, False),
    'mbv_attn_fwd_ld': lambda a: _attn(a, False, True),
    'mbv_attn_bwd': lambda a: _attn(a, True, False),
    'mbv_attn_bwd_ld': lambda a: _attn(a, True, True),
    'mbv_attn_split_fwd_ld': lambda a: _attn_split(a, False),
    'mbv_attn_split_bwd_ld': lambda a: _attn_split(a, True),
    'mbv_gemm16_nt': lambda a: _gemm16(a, 'nt'),
    'mbv_gemm16_nn': lambda a: _gemm16(a, 'nn'),
    'mbv_gemm16_nn_parts': lambda a: _gemm16(a, 'nn_parts'),
    'mbv_gemm16_tn': lambda a: _gemm16(a, 'tn'),
    'mbv_add_layernorm_fwd': lambda a: _add_ln(a, False),
    'mbv_add_layernorm_fwd2': lambda a: _add_ln_fwd2(a),
    'mbv_pfn_forward': lambda a: _pfn_forward(a),
    'mbv_skinny_gemm_f32_addrows': lambda a: ('k_skinny_f32', 'hbm',
                                              (_i(a[3]) * (_i(a[4]) + _i(a[5])) + _i(a[4]) * _i(a[5])) * 4.0,
                                              2.0 * _i(a[3]) * _i(a[4]) * _i(a[5])),
    'mbv_skinny_gemm_f32': lambda a: ('k_skinny_f32', 'hbm',
                                      (_i(a[3]) * (_i(a[4]) + _i(a[5])) + _i(a[4]) * _i(a[5])) * 4.0,
                                      2.0 * _i(a[3]) * _i(a[4]) * _i(a[5])),
    'mbv_copy_group': lambda a: ('k_copy_group', 'hbm', 2.0 * sum(int(a[2][j]) for j in range(_i(a[3]))), 0.0),
    'mbv_transposed_batch_sum_accum': lambda a: ('k_transposed_batch_sum', 'hbm',
                                                 (_i(a[1]) + 2.0) * _i(a[2]) * _i(a[3]) * 4.0, 0.0),
    'mbv_add_layernorm_bwd': lambda a: _add_ln(a, True),
    'mbv_add_layernorm_bwd2': lambda a: _add_ln_bwd2(a),
    'mbv_add_layernorm_bwd3': lambda a: _add_ln_bwd2(a),
    # importance sampling: every row's (H, W) f32 map is read once; the 3x over-sampled candidates never touch HBM
    'mbv_sample_select_uncertain': lambda a: ('k_sample_select', 'hbm',
                                              _i(a[4]) * (_i(a[7]) * _i(a[8]) * 4.0 + _i(a[6]) * 8.0), 0.0),
    'mbv_hungarian': lambda a: ('k_hungarian', 'hbm', _i(a[1]) * _i(a[2]) * _i(a[3]) * 4.0 + _i(a[1]) * _i(a[2]) * 4.0, 0.0),
    'mbv_mask_logits_fwd': lambda a: ('k_mask_logits', 'hbm',
                                      _i(a[3]) * (_i(a[4]) * _i(a[5]) + _i(a[5]) * _i(a[6])) * (2 if _i(a[2]) else 4)
                                      + _i(a[3]) * _i(a[4]) * _i(a[6]) * (4 if (_i(a[8]) or not _i(a[2])) else 2),
                                      2.0 * _i(a[3]) * _i(a[4]) * _i(a[5]) * _i(a[6])),
    'mbv_point_sample_fwd': lambda a: _point_sample(a, 'k_point_sample_fwd_lds', 4.0),
    'mbv_point_sample_bwd': lambda a: ('k_point_sample_bwd_lds', 'hbm',
                                       _i(a[8]) * _i(a[6]) * _i(a[7]) * 4.0 + _point_rows(_i(a[4]), _i(a[5])), 0.0),
    'mbv_point_sample_bwd_stack': lambda a: ('k_point_sample_bwd_lds', 'hbm',
                                             _i(a[4]) * _i(a[6]) * _i(a[7]) * (4.0 if _i(a[12]) == 0 else 2.0)
                                             + _point_rows(_i(a[4]), _i(a[5])), 0.0),
    'mbv_point_sample_packed_fwd': lambda a: _point_sample(a, 'k_point_sample_packed', 1.0 / 8.0),
    'mbv_mask_loss_rows_fwd': lambda a: ('k_mask_loss_rows_fwd', 'hbm', _i(a[2]) * _i(a[3]) * 8.0, 0.0),
    'mbv_mask_loss_rows_bwd': lambda a: ('k_mask_loss_rows_bwd', 'hbm', _i(a[3]) * _i(a[4]) * 12.0, 0.0),
    'mbv_mask_loss_rows_bwd_coef': lambda a: ('k_mask_loss_rows_bwd', 'hbm', _i(a[7]) * _i(a[9]) * 12.0, 0.0),
    'mbv_act_bwd_colsum': lambda a: ('k_act_bwd_colsum', 'hbm', _i(a[4]) * _i(a[5]) * 3.0 * (2 if _i(a[2]) else 4), 0.0),
    'mbv_wgrad_small_f32': lambda a: ('k_wgrad_small', 'mfma_f32',
                                      (_i(a[2]) * (_i(a[3]) + _i(a[4])) + 2 * _i(a[3]) * _i(a[4])) * 4.0,
                                      2.0 * _i(a[2]) * _i(a[3]) * _i(a[4])),
    'mbv_wgrad_small_f32_group': lambda a: _wgrad_group(a),
    'mbv_colsum_accum_group': lambda a: _colsum_group(a),
    'mbv_gemm16_tn_group': lambda a: _tn_group(a),
    # K20 (f32 products from IEEE-half pairs): the ALGORITHMIC flops of the f32 product, 2 m n k, against the 16-bit MFMA peak
    # (the kernel issues three matrix instructions per product term: a frac of 1/3 would be the pipe's own ceiling)
    'mbv_gemm32s_nt': lambda a: ('k_gemm32s<NT>', 'mfma', ((_i(a[5]) * _i(a[7]) + _i(a[6]) * _i(a[7])) * 4.0
                                                          + _i(a[5]) * _i(a[6]) * 4.0 * (2 if a[4] is not None and _i(a[4]) else 1))
                                 * max(1, _i(a[15])), 2.0 * _i(a[5]) * _i(a[6]) * _i(a[7]) * max(1, _i(a[15]))),
    'mbv_gemm32s_nn': lambda a: ('k_gemm32s<NN>', 'mfma', (_i(a[3]) * _i(a[4]) + _i(a[4]) * _i(a[5]) + _i(a[3]) * _i(a[5])) * 4.0
                                 * max(1, _i(a[12])), 2.0 * _i(a[3]) * _i(a[4]) * _i(a[5]) * max(1, _i(a[12]))),
    # the same product with the activation's derivative in the epilogue: one more read of the output-shaped pre-activation
    'mbv_gemm32s_nn_act': lambda a: ('k_gemm32s<NN,dact>', 'mfma',
                                     (_i(a[6]) * _i(a[7]) + _i(a[7]) * _i(a[8]) + 2 * _i(a[6]) * _i(a[8])) * 4.0,
                                     2.0 * _i(a[6]) * _i(a[7]) * _i(a[8])),
    'mbv_gemm32s_tn_acc': lambda a: ('k_gemm32s<TN>', 'mfma', (_i(a[3]) * _i(a[4]) + _i(a[3]) * _i(a[5])) * 4.0
                                     + _i(a[4]) * _i(a[5]) * 4.0 * 2, 2.0 * _i(a[3]) * _i(a[4]) * _i(a[5])),
    # the patch projection on K20's gather modes: (image, weight, bias, out, batch, C, h, w, E, ...): tokens x E x 16 C
    'mbv_patch_embed32_fwd': lambda a: ('k_gemm32s<NT,patch>', 'mfma',
                                        (_i(a[4]) * _i(a[5]) * _i(a[6]) * _i(a[7]) + _i(a[4]) * _i(a[6]) * _i(a[7]) // 16 * _i(a[8])) * 4.0,
                                        2.0 * _i(a[4]) * (_i(a[6]) // 4) * (_i(a[7]) // 4) * _i(a[8]) * 16 * _i(a[5])),
    'mbv_patch_embed32_bwd_image': lambda a: ('k_gemm32s<NN,patch>', 'mfma',
                                              (_i(a[3]) * _i(a[4]) * _i(a[5]) * _i(a[6]) + _i(a[3]) * _i(a[5]) * _i(a[6]) // 16 * _i(a[7])) * 4.0,
                                              2.0 * _i(a[3]) * (_i(a[5]) // 4) * (_i(a[6]) // 4) * _i(a[7]) * 16 * _i(a[4])),
    'mbv_patch_embed32_bwd_weight': lambda a: ('k_gemm32s<TN,patch>', 'mfma',
                                               (_i(a[3]) * _i(a[4]) * _i(a[5]) * _i(a[6]) + _i(a[3]) * _i(a[5]) * _i(a[6]) // 16 * _i(a[7])) * 4.0,
                                               2.0 * _i(a[3]) * (_i(a[5]) // 4) * (_i(a[6]) // 4) * _i(a[7]) * 16 * _i(a[4])),
    # the 3 x 3 convolution as a K20 product over (tap, channel): reads the rows buffer and the weights, writes the output rows;
    # 2 * positions * cout * 9 C flops.  (rows, wm, out_rows, batch, H, W, C, cout, ...)
    'mbv_conv3x3_gemm32s': lambda a: ('k_gemm32s<NT,conv3x3>', 'mfma',
                                      (_i(a[3]) * (_i(a[4]) + 2) * (_i(a[5]) + 2) * (_i(a[6]) + _i(a[7])) + 9 * _i(a[6]) * _i(a[7])) * 4.0,
                                      2.0 * _i(a[3]) * (_i(a[4]) + 2) * (_i(a[5]) + 2) * _i(a[7]) * 9 * _i(a[6])),
    'mbv_conv3x3_gemm16': lambda a: ('k_gemm16<NT,conv3x3>', 'mfma',
                                     (_i(a[3]) * (_i(a[4]) + 2) * (_i(a[5]) + 2) * (_i(a[6]) + _i(a[7])) + 9 * _i(a[6]) * _i(a[7])) * 2.0,
                                     2.0 * _i(a[3]) * (_i(a[4]) + 2) * (_i(a[5]) + 2) * _i(a[7]) * 9 * _i(a[6])),
    'mbv_conv_pad_rows': lambda a: ('k_conv_pad_rows', 'hbm', 2.0 * _i(a[2]) * _i(a[3]) * _i(a[4]) * _i(a[5]) * _i(a[6]), 0.0),
    'mbv_conv_unpad_rows': lambda a: ('k_conv_pad_rows', 'hbm', 2.0 * _i(a[2]) * _i(a[3]) * _i(a[4]) * _i(a[5]) * _i(a[6]), 0.0),
    'mbv_gemm32s_tn_group': lambda a: ('k_gemm32s_tn_group', 'mfma',
                                       sum((int(a[3][i]) * (int(a[4][i]) + int(a[5][i])) + 2 * int(a[4][i]) * int(a[5][i])) * 4.0
                                           for i in range(_i(a[10]))),
                                       sum(2.0 * int(a[3][i]) * int(a[4][i]) * int(a[5][i]) for i in range(_i(a[10])))),
    'mbv_f32_absmax_group': lambda a: ('k_absmax_group', 'hbm',
                                       4.0 * sum(int(a[1][i]) * int(a[2][i]) for i in range(_i(a[5]))), 0.0),
    'mbv_upsample_bilinear_bwd': lambda a: ('k_upsample_bilinear_bwd', 'hbm',
                                            _i(a[2]) * (_i(a[3]) * _i(a[4]) * _sz(a[1]) + _i(a[5]) * _i(a[6]) * _sz(a[8])), 0.0),
    'mbv_groupnorm_fwd': lambda a: _groupnorm(a, False),
    'mbv_groupnorm_bwd': lambda a: _groupnorm(a, True),
    'mbv_merge_layernorm_fwd': lambda a: _merge_ln(a, False),
    'mbv_merge_layernorm_bwd': lambda a: _merge_ln(a, True),
    'mbv_colsum_accum': lambda a: ('k_colsum', 'hbm', _i(a[2]) * _i(a[3]) * (2.0 if _i(a[1]) else 4.0), 0.0),
    'mbv_match_cost_terms': lambda a: ('k_match_cost_terms', 'hbm', _i(a[1]) * _i(a[2]) * _i(a[3]) * 16.0, 0.0),
    'mbv_match_cost': lambda a: ('k_match_cost', 'hbm',
                                 _i(a[4]) * ((3.0 * _i(a[5]) + 1) * _i(a[6]) + _i(a[5]) * _i(a[6])) * 4.0, 0.0),
    # K13c: reads the sampled logits and targets once, writes the sliced products; 2 (2Q + 1)(G + 1) P flops x 3 half products
    'mbv_match_products': lambda a: ('k_match_products', 'hbm',
                                     _i(a[2]) * ((_i(a[3]) + _i(a[4])) * _i(a[5]) * 4.0
                                                 + _i(a[6]) * ((2.0 * _i(a[3]) + 1) * (_i(a[4]) + 1) + _i(a[3])) * 4.0),
                                     _i(a[2]) * 6.0 * (2.0 * _i(a[3]) + 1) * (_i(a[4]) + 1) * _i(a[5])),
    'mbv_match_cost_split': lambda a: ('k_match_cost', 'hbm',
                                       _i(a[4]) * (_i(a[10]) * ((2.0 * _i(a[5]) + 1) * (_i(a[6]) + 1) + _i(a[5]))
                                                   + _i(a[5]) * _i(a[6])) * 4.0, 0.0),
    'mbv_cls_loss_fwd': lambda a: ('k_cls_loss', 'hbm', _i(a[4]) * _i(a[5]) * _i(a[6]) * (_i(a[8]) * 4.0 + 4.0), 0.0),
    'mbv_cls_loss_bwd': lambda a: ('k_cls_loss', 'hbm', _i(a[6]) * _i(a[7]) * _i(a[8]) * (_i(a[10]) * 8.0 + 4.0), 0.0),
}


def peak_of(bound: str) -> Tuple[float, str]:
    if bound == 'mfma':
        return MFMA_BF16_TFLOPS, 'TFLOP/s'
    if bound == 'mfma_f32':
        return MFMA_F32_TFLOPS, 'TFLOP/s'
    return HBM_PEAK_GBS, 'GB/s'


# ---------------------------------------------------------------------------------------------------------------------
# Library work torch issues for the path: hipBLASLt GEMMs, MIOpen convolutions, ATen element-wise / reduction kernels.
# bench.py times them with the same HIP-event brackets as the C-ABI calls, through a TorchDispatchMode that sees every
# ATen operator of the instrumented eager step (forward, the autograd engine's backward, the optimizer glue), so that
# `roofline_all` prices the WHOLE step and not only this repository's kernels.
# ---------------------------------------------------------------------------------------------------------------------
_NO_KERNEL = ('aten::empty', 'aten::new_empty', 'aten::as_strided', 'aten::detach', 'aten::alias', 'aten::_unsafe_view',
              'aten::view', 'aten::_reshape_alias', 'aten::resize_', 'aten::set_', 'aten::lift_fresh', 'aten::is_',
              'aten::sym_', 'aten::record_stream', 'aten::_local_scalar_dense', 'aten::stride', 'aten::size',
              'aten::_to_copy_view', 'prim::', 'aten::_has_', 'aten::is_same_size', 'aten::_nested')
_REDUCTIONS = ('sum', 'mean', 'amax', 'amin', 'max', 'min', 'softmax', 'norm', 'sort', 'topk', 'cumsum', 'argmax', 'argmin',
               'nll_loss', 'var', 'std', 'prod', 'all', 'any', 'logsumexp')


def _tensors(x, out):
    import torch
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for y in x:
            _tensors(y, out)


def _nbytes(ts) -> float:
    return float(sum(t.numel() * t.element_size() for t in ts))


def aten_work(func, args, kwargs, result, cuda_only: bool = True):
    """(family, bound, bytes, flops) of one ATen operator call, or None when it launches nothing (views, allocations).
    GEMMs: 2 M N K flops, operands + result bytes (hipBLASLt; family by operand dtype).  Convolutions: MIOpen.  Everything
    else: a streaming kernel that reads its tensor arguments and writes its results once."""
    import torch
    name = func._schema.name
    if getattr(func, 'is_view', False) or name.startswith(_NO_KERNEL):
        return None
    ins, outs = [], []
    _tensors(list(args) + list((kwargs or {}).values()), ins)
    _tensors(result, outs)
    if cuda_only:
        ins = [t for t in ins if t.is_cuda]
        outs = [t for t in outs if t.is_cuda]
    if not ins and not outs:
        return None
    short = name.split('::')[-1]
    short_out = str(getattr(func, '_overloadname', '')) == 'out' or (kwargs or {}).get('out') is not None
    if short in ('mm', 'addmm', 'bmm', 'baddbmm', '_scaled_mm'):
        # (an out= variant names its destination among the arguments: not an operand)
        dst = {t.data_ptr() for t in outs}
        mats = [t for t in ins if t.dim() >= 2 and not (short_out and t.data_ptr() in dst)][-2:]
        if len(mats) == 2:
            a, b = mats
            batch = a.shape[0] if a.dim() == 3 else 1
            m, k, n = a.shape[-2], a.shape[-1], b.shape[-1]
            f32 = a.dtype == torch.float32
            by = _nbytes([t for t in ins if not (short_out and t.data_ptr() in dst)]) + _nbytes(outs)
            return ('hipblaslt_f32' if f32 else 'hipblaslt_16bit', 'mfma_f32' if f32 else 'mfma', by, 2.0 * batch * m * n * k)
    if short in ('convolution', 'convolution_backward', 'miopen_convolution', 'cudnn_convolution', '_convolution'):
        bwd = short == 'convolution_backward'
        x = ins[1] if bwd else ins[0]
        w = ins[2] if bwd else ins[1]
        o = ins[0] if bwd else outs[0]
        cout, cin_g, kh, kw = w.shape[0], w.shape[1], w.shape[-2], w.shape[-1]
        fl = 2.0 * o.shape[0] * cout * o.shape[-2] * o.shape[-1] * cin_g * kh * kw
        f32 = x.dtype == torch.float32
        return ('miopen_conv', 'mfma_f32' if f32 else 'mfma', _nbytes(ins) + _nbytes(outs), fl * (2.0 if bwd else 1.0))
    # in-place / out= variants name their destination among the inputs: it is written once more
    family = 'aten_reduce' if any(r in short for r in _REDUCTIONS) else 'aten_elementwise'
    seen = {t.data_ptr() for t in ins}
    by = _nbytes(ins) + _nbytes([t for t in outs if t.data_ptr() not in seen]) \
        + _nbytes([t for t in outs if t.data_ptr() in seen and func._schema.is_mutable])
    return (family, 'hbm', by, 0.0)


def aten_timer(records: list, detail: list = None):
    """A TorchDispatchMode that brackets every kernel-launching ATen call with HIP events on the current stream and
    appends ``((family, bound, bytes, flops), start, end)`` to ``records`` (bench.py only)."""
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode

    class _Timer(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            name = func._schema.name
            if getattr(func, 'is_view', False) or name.startswith(_NO_KERNEL):
                return func(*args, **kwargs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = func(*args, **kwargs)
            b.record()
            work = aten_work(func, args, kwargs, out)
            if work is not None:
                records.append((work, a, b))
                if detail is not None:       # per-operator view for hunting the tail (bench.py --aten-detail)
                    ts = []
                    _tensors(list(args), ts)
                    import traceback
                    where = next((f'{fr.filename.split("/")[-1]}:{fr.lineno}' for fr in reversed(traceback.extract_stack(limit=14))
                                  if 'mask_bev_amd' in fr.filename and 'workmodel' not in fr.filename), '?')
                    detail.append((name, [tuple(t.shape) for t in ts[:3]], where, a, b))
            return out

    return _Timer()
