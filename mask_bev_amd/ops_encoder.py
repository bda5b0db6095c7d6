"""K1 voxelisation, K2 PillarFeatureNet, K3 scatter + (C, ny, nx) LayerNorm (mask_bev_encoders.py:63-123 upstream ops)."""
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
# K1 voxelisation
# --------------------------------------------------------------------------------------
@dataclass
class VoxelGeometry:
    """Grid description; mirrors the arguments of mmcv ``Voxelization`` built at
    mask_bev/models/encoders/mask_bev_encoders.py:67-69 (bounds are rounded to f32 by the kernel ABI)."""
    pc_range: Sequence[float]      # x_min, y_min, z_min, x_max, y_max, z_max
    voxel_size: Sequence[float]    # vx, vy, vz
    grid: Sequence[int]            # gx, gy, gz

    @staticmethod
    def from_ranges(pc_range: Sequence[float], voxel_size: Sequence[float]) -> 'VoxelGeometry':
        r = torch.tensor(list(pc_range), dtype=torch.float32)
        v = torch.tensor(list(voxel_size), dtype=torch.float32)
        grid = torch.round((r[3:] - r[:3]) / v).long().tolist()     # mmcv Voxelization.__init__ [upstream]
        return VoxelGeometry(list(pc_range), list(voxel_size), grid)

    @property
    def cells(self) -> int:
        return int(self.grid[0]) * int(self.grid[1]) * int(self.grid[2])


@dataclass
class Pillars:
    """Output of :func:`voxelize` (all device tensors except the python ints)."""
    points: torch.Tensor            # (N_total, D) f32, the concatenated scans
    scan_offsets: torch.Tensor      # (B+1,) i32
    coors: torch.Tensor             # (V, 4) i32 (b, z, y, x)
    num_points: torch.Tensor        # (V,) i32
    pillar_points: torch.Tensor     # (V, P) i32 index into points, -1 padded
    row_start: torch.Tensor         # (V+1,) i32
    cell_to_pillar: torch.Tensor    # (B, cells) i32
    pillar_batch_start: torch.Tensor  # (B+1,) i32
    pillars_per_scan: List[int]
    num_pillars: int
    num_rows: int
    max_points: int


def voxelize(point_clouds: Sequence[torch.Tensor], geom: VoxelGeometry, max_points: int, max_voxels: int,
             prefilter: bool = True) -> Pillars:
    """Range filter + hard voxelisation of a batch of scans (K1).  One host sync (reading V and K)."""
    lib = _lib.load()
    if len(point_clouds) == 0:
        raise ValueError('empty batch')
    _need_gpu(*point_clouds)
    dev = point_clouds[0].device
    dim = int(point_clouds[0].shape[1])
    lens = [int(p.shape[0]) for p in point_clouds]
    n = sum(lens)
    if (len(point_clouds) > 1 and n > 0 and all(p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 2
                                                 and p.shape[1] == dim and p.data_ptr() % 16 == 0
                                                 and (p.shape[0] * dim * 4) % 16 == 0 for p in point_clouds)):
        # the scans of a batch behind one another in ONE grouped-copy launch (ATen's batched cat: 70 us for 4 x 120 k points)
        points = torch.empty((n, dim), dtype=torch.float32, device=dev)
        src, dst, nb, off = [], [], [], 0
        for p, l in zip(point_clouds, lens):
            if l > 0:                                   # (an empty scan has nothing to copy — and no address to copy from)
                src.append(p.data_ptr())
                dst.append(points.data_ptr() + off)
                nb.append(l * dim * 4)
            off += l * dim * 4
        k = len(src)
        check(lib.mbv_copy_group((ctypes.c_void_p * k)(*src), (ctypes.c_void_p * k)(*dst), (ctypes.c_int64 * k)(*nb), k,
                                 _stream()), 'mbv_copy_group')
    else:
        points = torch.cat([p.reshape(-1, dim) for p in point_clouds], 0).to(torch.float32).contiguous()
    batch = len(point_clouds)
    if n == 0:                            # no points at all: empty pillars, like the reference (no kernel to launch)
        z = lambda *sh: torch.zeros(sh, dtype=torch.int32, device=dev)
        return Pillars(points=points, scan_offsets=z(batch + 1), coors=z(0, 4), num_points=z(0),
                       pillar_points=z(0, int(max_points)), row_start=z(1),
                       cell_to_pillar=torch.full((batch, geom.cells), -1, dtype=torch.int32, device=dev),
                       pillar_batch_start=z(batch + 1), pillars_per_scan=[0] * batch, num_pillars=0, num_rows=0,
                       max_points=int(max_points))
    offs = [0]
    for l in lens:
        offs.append(offs[-1] + l)
    scan_offsets = torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True)
    cap = min(n, batch * max_voxels) if max_voxels >= 0 else n
    coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nump = torch.empty((cap,), dtype=torch.int32, device=dev)
    ppts = torch.empty((cap, max_points), dtype=torch.int32, device=dev)
    row_start = torch.empty((cap + 1,), dtype=torch.int32, device=dev)
    c2p = torch.empty((batch, geom.cells), dtype=torch.int32, device=dev)
    counts = torch.empty((batch + 2,), dtype=torch.int32, device=dev)
    ws_bytes = lib.mbv_voxelize_workspace_bytes(n, batch, geom.cells)
    ws = _workspace(ws_bytes, dev)
    r, v, g = geom.pc_range, geom.voxel_size, geom.grid
    rc = lib.mbv_voxelize(_ptr(points), dim, n, _ptr(scan_offsets), batch,
                          r[0], r[1], r[2], r[3], r[4], r[5], v[0], v[1], v[2], int(g[0]), int(g[1]), int(g[2]),
                          1 if prefilter else 0, int(max_points), int(max_voxels), cap,
                          _ptr(coors), _ptr(nump), _ptr(ppts), _ptr(row_start), _ptr(c2p), _ptr(counts),
                          _ptr(ws), ws.numel(), _stream())
    check(rc, 'mbv_voxelize')
    counts_h = counts.cpu().tolist()                      # the one permitted sync
    per_scan, nv, nk = counts_h[:batch], counts_h[batch], counts_h[batch + 1]
    pbs = [0]
    for c in per_scan:
        pbs.append(pbs[-1] + c)
    return Pillars(points=points, scan_offsets=scan_offsets, coors=coors[:nv], num_points=nump[:nv],
                   pillar_points=ppts[:nv], row_start=row_start[:nv + 1], cell_to_pillar=c2p,
                   pillar_batch_start=torch.tensor(pbs, dtype=torch.int32).to(dev, non_blocking=True),
                   pillars_per_scan=per_scan, num_pillars=nv, num_rows=nk, max_points=int(max_points))


def gather_voxels(p: Pillars) -> torch.Tensor:
    """Dense zero-padded (V, P, D) voxel tensor — the first output of mmcv ``Voxelization``."""
    lib = _lib.load()
    dim = int(p.points.shape[1])
    out = torch.empty((p.num_pillars, p.max_points, dim), dtype=torch.float32, device=p.points.device)
    rc = lib.mbv_gather_voxels(_ptr(p.points), dim, _ptr(p.pillar_points), p.num_pillars, p.max_points, _ptr(out),
                               _stream())
    check(rc, 'mbv_gather_voxels')
    return out


def pfn_decorate(p: Pillars, voxel_size: Sequence[float], pc_range: Sequence[float]):
    """Compact decorated rows (K, D+7) of the real points + the pillar of each row (K2a)."""
    lib = _lib.load()
    dim = int(p.points.shape[1])
    dev = p.points.device
    _lib.WORK_HINT['pfn_rows'] = int(p.num_rows)
    rows = torch.empty((p.num_rows, dim + 7), dtype=torch.float32, device=dev)
    row_pillar = torch.empty((p.num_rows,), dtype=torch.int64, device=dev)
    vx, vy, vz = [float(v) for v in voxel_size]
    rc = lib.mbv_pfn_decorate(_ptr(p.points), dim, _ptr(p.pillar_points), _ptr(p.num_points), _ptr(p.row_start),
                              _ptr(p.coors), p.num_pillars, p.max_points, vx, vy, vz,
                              vx / 2 + pc_range[0], vy / 2 + pc_range[1], vz / 2 + pc_range[2],
                              _ptr(rows), _ptr(row_pillar), _stream())
    check(rc, 'mbv_pfn_decorate')
    return rows, row_pillar


# --------------------------------------------------------------------------------------
# K2b PillarFeatureNet layers (per-pillar kernels + f32 library GEMMs)
# --------------------------------------------------------------------------------------
class _PillarFeatureNet(torch.autograd.Function):
    """forward(rows, row_start, num_points, V, P, training, eps, momentum, W_0, gamma_0, beta_0, rmean_0, rvar_0, …)
    → (V, C_last).  Everything is f32 whatever the autocast state: the GEMMs are ~7 GFLOP and the BatchNorm
    statistics / pillar indices must not lose precision."""

    @staticmethod
    def forward(ctx, rows, row_start, num_points, v, p, training, eps, momentum, row_pillar, *params):
        lib = _lib.load()
        _need_gpu(rows, row_start, num_points)
        dev = rows.device
        n_layers = len(params) // 5
        k = int(rows.shape[0])
        _lib.WORK_HINT['pfn_rows'] = k          # (read by workmodel.py under bench.py's hook only)
        count = float(v * p)
        st = _stream()
        ctx.params = params
        ctx.meta = (row_start, num_points, v, p, training, count, [t.dtype for t in params])
        # One boundary crossing for all layers (mbv_pfn_forward: this is the eager section in front of the captured step, where
        # the host's time per launch is step time); every tensor of the pass is a piece of one workspace, cut into views only
        # when the backward asks for them.
        units = [int(params[5 * l].shape[0]) for l in range(n_layers)]
        if (switches.get('pfn_one_call') and switches.get('pfn_skinny') and rows.dtype == torch.float32 and rows.is_contiguous()
                and k > 0 and v > 0 and n_layers <= 8 and int(rows.shape[1]) <= 128
                and all(u % 32 == 0 and 32 <= u <= 128 for u in units)
                and all(t.dtype == torch.float32 and t.is_contiguous() for t in params)
                and all(tuple(params[5 * l].shape) == (units[l], int(rows.shape[1]) if l == 0 else 2 * units[l - 1])
                        for l in range(n_layers))):
            n = n_layers
            uarr = (ctypes.c_int32 * n)(*units)
            offs = (ctypes.c_int64 * (11 * n))()
            total = int(lib.mbv_pfn_forward_layout(k, v, uarr, n, offs))
            ws = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
            ptrs = [(ctypes.c_void_p * n)(*[params[5 * l + j].data_ptr() for l in range(n)]) for j in range(5)]
            rp = None
            if (row_pillar is not None and row_pillar.dtype == torch.int64 and row_pillar.is_contiguous()
                    and int(row_pillar.shape[0]) == k and switches.get('pfn_stream_stats')):
                rp = row_pillar          # the pillar term inside the Linear's launch, the statistics as a streaming pass
            check(lib.mbv_pfn_forward(_ptr(rows), int(rows.shape[1]), _ptr(row_start), _ptr(num_points), _ptr(rp), k, v, p, ptrs[0],
                                      ptrs[1], ptrs[2], ptrs[3], ptrs[4], uarr, n, float(eps), float(momentum),
                                      1 if training else 0, _ptr(ws), total, st), 'mbv_pfn_forward')
            ctx.saved = None
            ctx.packed = (rows, ws, list(offs), units)
            o = offs[11 * (n - 1) + 10]
            return ws[o:o + v * units[-1]].view(v, units[-1])
        ctx.packed = None
        with torch.autocast('cuda', enabled=False):
            a_prev, apad_prev, m_prev = rows.float().contiguous(), None, None
            saved = []
            for l in range(n_layers):
                w, g, b, rm, rv = params[5 * l:5 * l + 5]
                w = w.float()
                u = int(w.shape[0])
                if l == 0:
                    y = _pfn_mm(a_prev, w, True, st)
                    ypad = torch.zeros((v, u), dtype=torch.float32, device=dev)     # W . 0
                    t = None
                else:
                    ca = int(a_prev.shape[1])
                    y = _pfn_mm(a_prev, w[:, :ca], True, st)
                    ypad = _pfn_mm(apad_prev, w[:, :ca], True, st)
                    t = _pfn_mm(m_prev, w[:, ca:], True, st)
                sums = torch.empty(2 * u, dtype=torch.float64, device=dev)
                check(lib.mbv_pfn_stats(_ptr(y), _ptr(t), _ptr(ypad), _ptr(row_start), _ptr(num_points), v, u, p,
                                        _ptr(sums), st), 'mbv_pfn_stats')
                scale, shift, mean, rstd = (torch.empty(u, dtype=torch.float32, device=dev) for _ in range(4))
                check(lib.mbv_pfn_bn_finalize(_ptr(sums), count, _ptr(g.float()), _ptr(b.float()), float(eps),
                                              float(momentum), 1 if training else 0, _ptr(rm), _ptr(rv), u,
                                              _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), st),
                      'mbv_pfn_bn_finalize')
                last = l == n_layers - 1
                a = None if last else torch.empty((k, u), dtype=torch.float32, device=dev)
                apad = None if last else torch.empty((v, u), dtype=torch.float32, device=dev)
                m = torch.empty((v, u), dtype=torch.float32, device=dev)
                check(lib.mbv_pfn_apply_max(_ptr(y), _ptr(ypad), _ptr(scale), _ptr(shift), _ptr(row_start),
                                            _ptr(num_points), v, u, p, _ptr(a), _ptr(apad), _ptr(m), st),
                      'mbv_pfn_apply_max')
                saved.append((a_prev, apad_prev, m_prev, y, ypad, scale, shift, mean, rstd, w, g.float()))
                a_prev, apad_prev, m_prev = a, apad, m
        ctx.saved = saved
        return m_prev

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        row_start, num_points, v, p, training, count, dtypes = ctx.meta
        st = _stream()
        if ctx.packed is not None:                      # the one-call forward: cut its workspace into the tensors of each layer
            rows0, ws, offs, units = ctx.packed
            k = int(rows0.shape[0])

            def piece(l, j, r, u):
                o = offs[11 * l + j]
                return None if o < 0 else ws[o:o + r * u].view(r, u) if r else ws[o:o + u]
            saved = []
            a_prev, apad_prev, m_prev = rows0, None, None
            for l, u in enumerate(units):
                y, ypad = piece(l, 0, k, u), piece(l, 1, v, u)
                scale, shift, mean, rstd = (piece(l, j, 0, u) for j in (4, 5, 6, 7))
                saved.append((a_prev, apad_prev, m_prev, y, ypad, scale, shift, mean, rstd, ctx.params[5 * l],
                              ctx.params[5 * l + 1]))
                a_prev, apad_prev, m_prev = piece(l, 8, k, u), piece(l, 9, v, u), piece(l, 10, v, u)
            ctx.saved = saved
        n_layers = len(ctx.saved)
        grads = [None] * (5 * n_layers)
        with torch.autocast('cuda', enabled=False):
            dm = d_out.float().contiguous()
            da, sapad, d_rows = None, None, None
            for l in reversed(range(n_layers)):
                a_prev, apad_prev, m_prev, y, ypad, scale, shift, mean, rstd, w, g = ctx.saved[l]
                u = int(w.shape[0])
                dev = y.device
                dz = da if da is not None else torch.empty_like(y)
                dzpad = torch.empty_like(ypad)
                sums = torch.empty(2 * u, dtype=torch.float64, device=dev)
                check(lib.mbv_pfn_bwd_route(_ptr(y), _ptr(ypad), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
                                            _ptr(dz), 1 if da is not None else 0, _ptr(sapad), _ptr(dm),
                                            _ptr(row_start), _ptr(num_points), v, u, p, _ptr(dzpad), _ptr(sums), st),
                      'mbv_pfn_bwd_route')
                grads[5 * l + 1] = sums[u:].to(dtypes[5 * l + 1])          # d gamma = sum dz * xhat
                grads[5 * l + 2] = sums[:u].to(dtypes[5 * l + 2])          # d beta  = sum dz
                dt = torch.empty_like(ypad) if l > 0 else None
                check(lib.mbv_pfn_bwd_bn(_ptr(y), _ptr(ypad), _ptr(dz), _ptr(dzpad), _ptr(mean), _ptr(rstd), _ptr(g),
                                         _ptr(sums), count, 1 if training else 0, _ptr(row_start), _ptr(num_points),
                                         v, u, p, _ptr(dt), st), 'mbv_pfn_bwd_bn')
                dy, dypad = dz, dzpad
                if l == 0:
                    gw = _wgrad(dy, a_prev)
                    # d(rows): only a learnable per-point encoding in front of the PFN asks for it (A3, fourier)
                    d_rows = dy.mm(w) if ctx.needs_input_grad[0] else None
                    da = sapad = dm = None
                else:
                    ca = int(a_prev.shape[1])
                    wa, wb = w[:, :ca], w[:, ca:]
                    gw = torch.cat([_wgrad(dy, a_prev) + _wgrad(dypad, apad_prev), _wgrad(dt, m_prev)], dim=1)
                    da = _pfn_mm(dy, wa, False, st)
                    sapad = _pfn_mm(dypad, wa, False, st)
                    dm = _pfn_mm(dt, wb, False, st)
                grads[5 * l] = gw.to(dtypes[5 * l])
            # arena parameters: the 3 x 3 small gradients join the end-of-pass grouped accumulate (a (1, n) "column sum")
            # instead of one AccumulateGrad add_ launch each
            for i, g in enumerate(grads):
                if g is not None:
                    grads[i] = _param_grad_or_defer(ctx.params[i], g)
        ctx.saved = ctx.params = ctx.packed = None
        return (d_rows,) + (None,) * 8 + tuple(grads)


_PFN_SKINNY_MIN_ROWS = 8192


def _pfn_mm(x: torch.Tensor, w: torch.Tensor, weight_is_nk: bool, stream=None) -> torch.Tensor:
    """``x @ w.t()`` (weight_is_nk) or ``x @ w`` for the PFN's f32 Linears: K2c for the long row counts (w may be a column
    block of a wider weight — only its row stride has to be regular), the library otherwise.  (This runs in the eager
    section in front of the captured step, where host time is step time: the shape test is arithmetic here — the library
    repeats it — and the caller hands the stream over.)"""
    m, c = x.shape
    n = int(w.shape[0] if weight_is_nk else w.shape[1])
    if (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and m >= _PFN_SKINNY_MIN_ROWS
            and x.is_contiguous() and w.stride(1) == 1 and (w.shape[1] if weight_is_nk else w.shape[0]) == c
            and 1 <= c <= 128 and 32 <= n <= 128 and n % 32 == 0 and switches.get('pfn_skinny')):
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        check(_lib.load().mbv_skinny_gemm_f32(_ptr(x), _ptr(w), _ptr(y), m, c, n, int(w.stride(0)),
                                              1 if weight_is_nk else 0, stream if stream is not None else _stream()),
              'mbv_skinny_gemm_f32')
        return y
    return x.mm(w.t() if weight_is_nk else w)


def _param_grad_or_defer(p: torch.Tensor, g: torch.Tensor):
    """The gradient ``g`` of parameter ``p`` for autograd — or None when ``p`` lives in the arena and the add into its
    f32 gradient was queued with the pass's grouped accumulate launch (ops.flush_deferred_grads)."""
    if (getattr(p, '_mbv_arena', False) and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
            and g.is_cuda and g.numel() == p.grad.numel() and g.numel() < (1 << 31)):
        gf = g.float().contiguous()
        if _defer_colsum(gf.view(1, -1), p.grad.view(-1), 1, gf.numel(), gf.numel()):
            _fire_grad_hooks(p)
            return None
    return g.to(p.dtype)


def pfn_layers(rows: torch.Tensor, p: 'Pillars', layers, training: bool,
               row_pillar: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PFNLayer stack on the compact decorated rows (K2b).  ``layers``: sequence of (weight, bn_weight, bn_bias,
    running_mean, running_var, eps, momentum); running buffers are updated in place in training mode.  ``row_pillar`` (K,)
    i64, the pillar of every row (:func:`pfn_decorate` returns it): lets the one-call forward add a layer's pillar term inside
    its Linear and take the BatchNorm statistics as a streaming pass."""
    flat = []
    for (w, g, b, rm, rv, _eps, _mom) in layers:
        flat += [w, g, b, rm, rv]
    eps, mom = layers[0][5], layers[0][6]
    return _PillarFeatureNet.apply(rows, p.row_start, p.num_points, p.num_pillars, p.max_points, training, eps, mom,
                                   row_pillar, *flat)


# --------------------------------------------------------------------------------------
# K3 scatter + (C, H, W) LayerNorm
# --------------------------------------------------------------------------------------
class _ScatterLayerNorm(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda')          # inputs are cast by scatter_layernorm(); `out` keeps its dtype
    def forward(ctx, feats, weight, bias, cell_to_pillar, pillar_batch_start, batch, ny, nx, eps, patch=0, out=None,
                patch_dtype=torch.bfloat16):
        lib = _lib.load()
        _need_gpu(feats, weight, bias, cell_to_pillar, pillar_batch_start)
        feats = feats.contiguous()
        weight = weight.contiguous()
        bias = bias.contiguous()
        c = int(weight.shape[0])
        dev = feats.device
        if patch:
            if not lib.mbv_scatter_layernorm_patch_supported(c, ny, nx, patch):
                raise MaskBevHipError(f'scatter_layernorm: no patch-token layout for C={c}, {ny}x{nx}, patch {patch}')
            shape, dt = (batch, ny // patch, nx // patch, patch * patch * c), patch_dtype
        else:
            shape, dt = (batch, c, ny, nx), torch.float32
        owned = out is not None
        if out is None:
            out = torch.empty(shape, dtype=dt, device=dev)
        else:                      # caller-owned destination (the static input buffer of a captured graph)
            if tuple(out.shape) != shape or out.dtype != dt or not out.is_contiguous() or out.device != dev:
                raise MaskBevHipError(f'scatter_layernorm: out must be a contiguous {dt} tensor of shape {shape}')
            ctx.mark_dirty(out)
        stats = torch.empty((batch, 2), dtype=torch.float32, device=dev)
        ws = _workspace(lib.mbv_scatter_layernorm_workspace_bytes(batch), dev)
        # fp32 compute: the f32 map feeds the K20 patch projection — its absmax record from this launch (no pass over 0.5 GB)
        rec = None
        if not patch and static_amax_wanted():
            # (a registered caller-owned map — the static input of a captured graph — has ONE persistent record, cleared and
            # rewritten here every step: static_amax_register)
            rec = static_amax_record(out) if owned else None
            if rec is not None:
                rec.zero_()
            else:
                rec = amax_record(dev)
        rc = lib.mbv_scatter_layernorm_fwd2(_ptr(feats), _ptr(pillar_batch_start), _ptr(cell_to_pillar), _ptr(weight),
                                            _ptr(bias), batch, c, ny, nx, float(eps), int(patch),
                                            _dt_flag(dt) if patch else 0, _ptr(out), _ptr(stats), _ptr(ws), ws.numel(),
                                            _ptr(rec), _stream(), *TIMER.events('k_ln_apply')[2:])
        check(rc, 'mbv_scatter_layernorm_fwd2')
        amax_hint_set(out, rec)
        ctx.save_for_backward(feats, weight, stats, cell_to_pillar, pillar_batch_start)
        ctx.dims = (batch, c, ny, nx)
        ctx.params = (weight, bias)
        ctx.patch, ctx.patch_dtype = int(patch), dt
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_out):
        lib = _lib.load()
        feats, weight, stats, cell_to_pillar, pillar_batch_start = ctx.saved_tensors
        batch, c, ny, nx = ctx.dims
        grad_out = grad_out.to(ctx.patch_dtype if ctx.patch else torch.float32).contiguous()
        dev = feats.device
        g_feats = torch.empty_like(feats)
        wp, bp = ctx.params
        direct = all(getattr(t, '_mbv_arena', False) and t.grad is not None and t.grad.dtype == torch.float32
                     and t.grad.is_contiguous() for t in (wp, bp))
        if direct:          # the two 134 MB affine gradients accumulate straight into the arena (no temporaries)
            g_w, g_b = wp.grad, bp.grad
        else:
            g_w = torch.empty_like(weight)
            g_b = torch.empty_like(weight)
        ws = _workspace(lib.mbv_scatter_layernorm_workspace_bytes(batch), dev)
        fused = K3_ADAM[0].claim(wp, bp) if (direct and K3_ADAM[0] is not None) else None
        if fused is not None:
            # the step driver armed the optimizer for this pass (arena.FlatAdam.fuse_layernorm_affine): the AdamW update of
            # the two affine parameters happens inside the launch, their gradients never reach the arena
            rc = lib.mbv_scatter_layernorm_bwd_adamw(
                _ptr(grad_out), ctx.patch, _dt_flag(ctx.patch_dtype) if ctx.patch else 0, _ptr(feats),
                _ptr(pillar_batch_start), _ptr(cell_to_pillar), _ptr(wp.data), _ptr(bp.data), _ptr(stats), batch, c, ny, nx,
                int(feats.shape[0]), _ptr(g_feats), fused['m_w'], fused['v_w'], fused['m_b'], fused['v_b'], fused['sh_w'],
                fused['sh_b'], fused['shadow_flag'], fused['lr'], fused['beta1'], fused['beta2'], fused['eps'],
                fused['weight_decay'], fused['step'], fused['decoupled'], _ptr(ws), ws.numel(), _stream(),
                *TIMER.events('k_ln_bwd_dense')[2:])
            check(rc, 'mbv_scatter_layernorm_bwd_adamw')
            _fire_grad_hooks(wp)
            _fire_grad_hooks(bp)
            return (g_feats,) + (None,) * 11
        rc = lib.mbv_scatter_layernorm_bwd(_ptr(grad_out), ctx.patch, _dt_flag(ctx.patch_dtype) if ctx.patch else 0,
                                           _ptr(feats), _ptr(pillar_batch_start),
                                           _ptr(cell_to_pillar),
                                           _ptr(weight), _ptr(stats), batch, c, ny, nx, int(feats.shape[0]),
                                           _ptr(g_feats), _ptr(g_w), _ptr(g_b), 1 if direct else 0, _ptr(ws),
                                           ws.numel(), _stream(), *TIMER.events('k_ln_bwd_dense')[2:])
        check(rc, 'mbv_scatter_layernorm_bwd')
        if direct:
            _fire_grad_hooks(wp)
            _fire_grad_hooks(bp)
            return (g_feats,) + (None,) * 11
        return (g_feats, g_w, g_b) + (None,) * 9


class PatchTokens:
    """The BEV pseudo-image handed over as the input rows of a ``patch`` x ``patch`` non-overlapping projection:
    ``rows`` (B, ny/p, nx/p, p*p*C) bf16 / fp16 with element ``(y%p)*p*C + c*p + x%p`` (K3's patch-token layout)."""

    def __init__(self, rows: torch.Tensor, channels: int, patch: int):
        self.rows, self.channels, self.patch = rows, channels, patch

    def to_image(self) -> torch.Tensor:
        """(B, C, ny, nx) view of the same values (tests / staged callers)."""
        b, ty, tx, _ = self.rows.shape
        p, c = self.patch, self.channels
        return self.rows.view(b, ty, tx, p, c, p).permute(0, 4, 1, 3, 2, 5).reshape(b, c, ty * p, tx * p)


def patch_layout_supported(channels: int, ny: int, nx: int, patch: int) -> bool:
    return bool(_lib.load().mbv_scatter_layernorm_patch_supported(channels, ny, nx, patch))


# The optimizer that asked for the AdamW update of K3's two affine parameters to be fused into K3's backward (one entry:
# arena.FlatAdam.fuse_layernorm_affine arms it, FlatAdam.step() reads what was applied); None = the ordinary backward.
K3_ADAM = [None]


def scatter_layernorm(feats: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, p: Pillars, batch: int, ny: int,
                      nx: int, eps: float, patch: int = 0, out: Optional[torch.Tensor] = None):
    """``LayerNorm([C, ny, nx])(PointPillarsScatter(feats, coors))`` without building the canvas (K3).
    ``patch`` = 4 returns :class:`PatchTokens` (the 16-bit type of the autocast region, or of ``out``) instead of the
    (B, C, ny, nx) f32 map; ``out`` is an optional destination buffer (no grad) of the result's shape and dtype."""
    patch_dtype = out.dtype if (out is not None and patch) else lo_dtype()
    _LAST_HINT[1] = None             # (a forward that sets no hint must not hand `out` the record of an EARLIER tensor at its address)
    out = _ScatterLayerNorm.apply(feats.float(), weight.float(), bias.float(), p.cell_to_pillar,
                                  p.pillar_batch_start, batch, ny, nx, eps, patch, out, patch_dtype)
    amax_hint_refresh(out)           # (mark_dirty bumped a caller-owned buffer's version behind the forward's hint)
    return PatchTokens(out, int(weight.shape[0]), patch) if patch else out


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
