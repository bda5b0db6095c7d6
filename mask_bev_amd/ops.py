"""Torch-facing wrappers of the C-ABI kernels (``include/maskbev_hip.h``).

PyTorch is plumbing here: it owns device memory, the stream and the autograd graph; every function
below enqueues hand-written gfx950 kernels on ``torch.cuda.current_stream()`` through ctypes.
There is no CPU fallback — tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, switches
from ._lib import MaskBevHipError, check


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream() -> ctypes.c_void_p:
    """The current stream's handle for the C ABI.  Through torch's raw accessors when they exist: `torch.cuda.current_stream()`
    builds a Stream object behind three Python-level device look-ups — 9 us a call, and every launch of the eager sections of a
    step asks."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# storage types of activations and their flag in the C ABI (MBV_DT_F32 / MBV_DT_BF16 / MBV_DT_F16, maskbev_hip.h)
_ACT_DTYPES = (torch.float32, torch.bfloat16, torch.float16)
_LO_DTYPES = (torch.bfloat16, torch.float16)
_DT_FLAG = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _dt_flag(dtype: torch.dtype) -> int:
    try:
        return _DT_FLAG[dtype]
    except KeyError:
        raise MaskBevHipError(f'mask_bev_amd kernels take f32, bf16 or fp16 activations, got {dtype}') from None


def lo_dtype() -> torch.dtype:
    """The 16-bit type of the current autocast region (bf16 outside one)."""
    if torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        if dt in _LO_DTYPES:
            return dt
    return torch.bfloat16


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MaskBevHipError('mask_bev_amd kernels need ROCm device tensors (no CPU fallback); got a '
                                  f'{t.device} tensor')


class KernelTimer:
    """Optional HIP-event timing of the dominant kernel of a C-ABI call (used by bench.py's roofline leg).
    Events are recorded by the library itself on the launch stream, right around that one kernel."""

    def __init__(self):
        self.enabled = False
        self.records = {}          # name -> list of (start_event, stop_event)

    def events(self, name: str):
        if not self.enabled:
            return None, None, ctypes.c_void_p(0), ctypes.c_void_p(0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()                 # forces creation of the underlying hipEvent_t; re-recorded by the library
        b.record()
        self.records.setdefault(name, []).append((a, b))
        return a, b, ctypes.c_void_p(a.cuda_event), ctypes.c_void_p(b.cuda_event)

    def span(self, name: str):
        """Context manager: HIP events on torch's current stream around a C-ABI call that launches exactly one
        kernel on that stream (K11's optimizer step)."""
        timer = self

        class _Span:
            def __enter__(self_inner):
                self_inner.ev = None
                if timer.enabled:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    self_inner.ev = (a, b)
                return self_inner

            def __exit__(self_inner, *exc):
                if self_inner.ev is not None:
                    self_inner.ev[1].record()
                    timer.records.setdefault(name, []).append(self_inner.ev)
                return False

        return _Span()

    def summary_ms(self):
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in self.records.items()}

    def reset(self):
        self.records = {}


TIMER = KernelTimer()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


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
        if value.dtype in _LO_DTYPES:                    # the 16-bit value map of the 16-bit compute modes
            check(lib.mbv_ms_deform_attn_bwd_locattn(_ptr(g_out), _ptr(value), _dt_flag(value.dtype), _ptr(shapes_t),
                                                     _ptr(level_start), _ptr(loc), _ptr(attn), b, nv, nh, d, nl, nq, npnt,
                                                     _ptr(g_loc), _ptr(g_attn), _stream()), 'mbv_ms_deform_attn_bwd_locattn')
            return
        check(lib.mbv_ms_deform_attn_bwd(_ptr(g_out), _ptr(value), _ptr(shapes_t), _ptr(level_start), _ptr(loc),
                                         _ptr(attn), b, nv, nh, d, nl, nq, npnt, host, _ptr(None), _ptr(g_loc),
                                         _ptr(g_attn), 2, _stream()), 'mbv_ms_deform_attn_bwd')
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


def _gemm32s_ok(*ts: torch.Tensor) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
               and t.shape[1] % 8 == 0 and t.data_ptr() % 16 == 0 and t.shape[0] * t.stride(0) * 4 < 0x7fff0000 for t in ts)


def gemm32s_wants(tokens: int) -> bool:
    return bool(switches.get('gemm32s')) and tokens >= int(switches.get('gemm32s_min'))


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
            gb = gy.sum((0, 2), dtype=torch.float32).to(b_dtype)
        return gx, gw, gb


def conv1x1_tokens(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x (B, HW, Cin) tokens, weight (Cout, Cin) → (B, Cout, HW)."""
    return _Conv1x1Tokens.apply(x, weight, bias)


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
