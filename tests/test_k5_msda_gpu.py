"""K5 multi-scale deformable attention: HIP forward/backward vs the oracle's grid_sample form (f32).
Tolerance rtol 1e-4 / atol 1e-5 (f32 bilinear sums; grad_value uses f32 atomics)."""
import pytest
import torch

from oracle import maskbev_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,H,D,shapes,P', [
    (2, 8, 32, [(4, 4), (8, 8), (16, 16)], 4),
    (1, 4, 8, [(3, 5), (6, 10)], 2),
    (3, 8, 4, [(5, 5), (10, 10), (20, 20)], 4),
])
def test_msda_fwd_bwd(device, B, H, D, shapes, P):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(B * 100 + D)
    L = len(shapes)
    nv = sum(h * w for h, w in shapes)
    nq = nv
    value = torch.randn(B, nv, H, D, generator=g)
    # locations partly outside [0, 1] to exercise the zero padding and the border corners
    loc = torch.rand(B, nq, H, L, P, 2, generator=g) * 1.3 - 0.15
    attn = torch.rand(B, nq, H, L, P, generator=g).flatten(-2).softmax(-1).view(B, nq, H, L, P)
    go = torch.randn(B, nq, H * D, generator=g)
    v_r, l_r, a_r = value.clone().requires_grad_(), loc.clone().requires_grad_(), attn.clone().requires_grad_()
    out_ref = O.ms_deform_attn_core(v_r, shapes, l_r, a_r)
    out_ref.backward(go)
    shapes_t = torch.tensor(shapes, dtype=torch.int64, device=device)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    level_start = torch.tensor(starts, dtype=torch.int64, device=device)
    v_d, l_d, a_d = (t.clone().to(device).requires_grad_() for t in (value, loc, attn))
    out = ops.ms_deform_attn(v_d, shapes, shapes_t, level_start, l_d, a_d)
    out.backward(go.to(device))
    torch.testing.assert_close(out.detach().cpu(), out_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(v_d.grad.cpu(), v_r.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a_d.grad.cpu(), a_r.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(l_d.grad.cpu(), l_r.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shapes,points', [([(16, 16), (8, 8), (4, 4)], 4), ([(12, 20)], 3), ([(8, 8), (4, 6)], 8)])
def test_msda_prepare_matches_torch_composition(device, dt, shapes, points):
    """K16 vs the torch ops it replaces (softmax over L*P, offsets / (W, H), + reference points), including the
    dtype of the quotient under bf16 projections; forward and both gradients."""
    from mask_bev_amd import ops
    torch.manual_seed(7)
    B, H, L, P = 2, 4, len(shapes), points
    N = sum(h * w for h, w in shapes)
    assert ops.msda_prepare_supported(L, P)
    off = (torch.randn(B, N, H, L, P, 2, device=device) * 3).to(dt).requires_grad_()
    logit = torch.randn(B, N, H, L * P, device=device).to(dt).requires_grad_()
    ref = torch.rand(N, 2, device=device)
    g_loc = torch.randn(B, N, H, L, P, 2, device=device)
    g_aw = torch.randn(B, N, H, L, P, device=device)
    loc, aw = ops.msda_prepare(off, logit, ref, shapes)
    assert loc.dtype == torch.float32 and aw.dtype == torch.float32
    (loc * g_loc).sum().add((aw * g_aw).sum()).backward()
    off_r, logit_r = off.detach().clone().requires_grad_(), logit.detach().clone().requires_grad_()
    shapes_t = torch.tensor(shapes, device=device)
    normalizer = torch.stack([shapes_t[:, 1], shapes_t[:, 0]], -1).to(dt)
    aw_r = logit_r.float().softmax(-1).view(B, N, H, L, P)           # autocast evaluates softmax in f32
    loc_r = ref.view(1, N, 1, 1, 1, 2) + off_r / normalizer.view(1, 1, 1, L, 1, 2)
    (loc_r * g_loc).sum().add((aw_r * g_aw).sum()).backward()
    assert loc_r.dtype == torch.float32
    torch.testing.assert_close(aw, aw_r, rtol=1e-5, atol=1e-6)
    if dt == torch.float32:
        torch.testing.assert_close(loc, loc_r, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(off.grad, off_r.grad, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(logit.grad, logit_r.grad, rtol=1e-4, atol=1e-6)
    else:
        assert torch.equal(loc, loc_r)                               # same 16-bit quotient, same f32 add
        assert torch.equal(off.grad, off_r.grad)
        lo = 1.6e-2 if dt == torch.bfloat16 else 2e-3               # 8 / 11 significand bits
        torch.testing.assert_close(logit.grad.float(), logit_r.grad.float(), rtol=lo, atol=1e-3)


@pytest.mark.gpu
def test_msda_prepare_limits(device):
    from mask_bev_amd import ops
    assert ops.msda_prepare_supported(3, 4) and ops.msda_prepare_supported(4, 4)
    assert not ops.msda_prepare_supported(5, 4) and not ops.msda_prepare_supported(9, 1)


def _packed_case(device, B, H, shapes, ld_extra, dt, loc_mode, seed=0):
    import ctypes
    from mask_bev_amd import _lib, ops
    lib = _lib.load()
    D, P, L = 32, 4, len(shapes)
    nv = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(seed)
    if loc_mode == 'random':
        loc = torch.rand(B, nv, H, L, P, 2, generator=g) * 1.3 - 0.15
        attn = torch.rand(B, nv, H, L, P, generator=g).flatten(-2).softmax(-1).view(B, nv, H, L, P)
    else:          # worst case of the range analysis: every sample of a level sits on ONE pixel centre with all of the
        # (query, head)'s attention mass on that level -> |sum| = num_query * |grad_out| at that pixel
        loc = torch.zeros(B, nv, H, L, P, 2)
        for l, (h, w) in enumerate(shapes):
            loc[:, :, :, l, :, 0] = (1 + 0.5) / w
            loc[:, :, :, l, :, 1] = (2 + 0.5) / h
        attn = torch.zeros(B, nv, H, L, P)
        attn[:, :, :, :, :] = 1.0 / P
    go = torch.randn(B, nv, H * D, generator=g)
    if loc_mode != 'random':
        go = go.abs() + 1.0          # same sign: the sums really reach num_query * mean|g|
    host = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in shapes for v in hw])
    shapes_t = torch.tensor(shapes, dtype=torch.int64, device=device)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    level_start = torch.tensor(starts, dtype=torch.int64, device=device)
    value = torch.zeros(B, nv, H, D, device=device)
    go_d, loc_d, attn_d = go.to(device), loc.to(device), attn.to(device)
    assert lib.mbv_ms_deform_attn_bwd_value_packed_supported(D, L, P, nv, host) == 1
    # reference: the f64-accumulator kernel (itself tested against the oracle above) — its split form's value part, or, for
    # levels beyond that form's 4 096 pixels, the whole banded backward
    ref = torch.empty(B, nv, H, D, device=device)
    if lib.mbv_ms_deform_attn_bwd_split(D, L, host):
        ops.check(lib.mbv_ms_deform_attn_bwd(ops._ptr(go_d), ops._ptr(value), ops._ptr(shapes_t), ops._ptr(level_start),
                                             ops._ptr(loc_d), ops._ptr(attn_d), B, nv, H, D, L, nv, P, host, ops._ptr(ref),
                                             ops._ptr(None), ops._ptr(None), 1, ops._stream()), 'mbv_ms_deform_attn_bwd')
    else:
        g_loc, g_attn = torch.empty_like(loc_d), torch.empty_like(attn_d)
        ops.check(lib.mbv_ms_deform_attn_bwd(ops._ptr(go_d), ops._ptr(value), ops._ptr(shapes_t), ops._ptr(level_start),
                                             ops._ptr(loc_d), ops._ptr(attn_d), B, nv, H, D, L, nv, P, host, ops._ptr(ref),
                                             ops._ptr(g_loc), ops._ptr(g_attn), 3, ops._stream()), 'mbv_ms_deform_attn_bwd')
    ld = H * D + ld_extra
    outs = []
    for _ in range(2):
        out = torch.full((B * nv, ld), 7.0, dtype=dt, device=device)
        ws = torch.empty(lib.mbv_ms_deform_attn_bwd_value_packed_workspace_bytes(B, H, L, nv), dtype=torch.uint8, device=device)
        ops.check(lib.mbv_ms_deform_attn_bwd_value_packed(ops._ptr(go_d), ops._ptr(loc_d), ops._ptr(attn_d), B, nv, H, D, L,
                                                          nv, P, host, ops._ptr(out), ops._dt_flag(dt), ld, ops._ptr(ws),
                                                          ws.numel(), ops._stream()),
                  'mbv_ms_deform_attn_bwd_value_packed')
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])                       # integer sums: bit-reproducible
    if ld_extra:
        assert bool((outs[0][:, H * D:] == 7.0).all())         # nothing outside the d(value) columns is touched
    return outs[0][:, :H * D].float().view(B, nv, H, D), ref, float(go.abs().max())


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,shapes,ld_extra', [(4, 8, [(16, 16), (32, 32), (64, 64)], 288),
                                                 (2, 8, [(4, 4), (8, 8), (16, 16)], 0),
                                                 (1, 4, [(5, 7), (48, 80)], 24),
                                                 # round 6: levels of up to 128 x 128 pixels — one 128 KB plane per block (the
                                                 # 1024 x 1024 BEV configuration's levels; the reference is the banded f64 form)
                                                 (1, 8, [(32, 32), (64, 64), (128, 128)], 0),
                                                 (1, 2, [(72, 72), (36, 36)], 8)])
def test_msda_value_gradient_packed_fixed_point(device, dt, B, H, shapes, ld_extra):
    """mbv_ms_deform_attn_bwd_value_packed (two channels per ds_add_u64, all levels in one launch, output in the
    caller's dtype / row stride) against the f64-accumulator form on the same inputs: every addend is rounded to
    2^(e - 30) with 2^e the block's L1 bound (sum over queries of attention mass x largest |g|), a few 1e-6 here, and a
    coarse pixel collects a few hundred addends — bounded by 1e-4 * max|g| for f32 output (+ one rounding of the 16-bit
    output types)."""
    got, ref, gmax = _packed_case(device, B, H, shapes, ld_extra, dt, 'random', seed=B)
    err = (got - ref).abs()
    # the rounding unit follows the block's L1 bound, i.e. the number of queries: 21 504 of them (the 32 / 64 / 128 levels)
    # put it at 2^-15 where 5 376 leave 2^-17 — 3e-4 * max|g| there (still ~ 1e-4 of a typical d(value) element)
    nq = sum(h * w for h, w in shapes)
    lim = (1e-4 if nq <= 8192 else 3e-4) * gmax + {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dt] * ref.abs()
    assert bool((err <= lim).all()), float((err - lim).max())
    assert float(ref.abs().max()) > 0.1


def test_msda_value_gradient_packed_worst_case_range(device):
    """All 5 376 queries' samples on one pixel with full attention mass and same-sign gradients: the largest sum the
    32-bit halves must hold (num_query * max|g|); nothing wraps, neighbouring halves do not disturb each other."""
    got, ref, gmax = _packed_case(device, 1, 8, [(16, 16), (32, 32), (64, 64)], 0, torch.float32, 'one_pixel')
    assert float(ref.abs().max()) > 1000.0
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('side', [72, 136])
def test_msda_layer_with_a_level_beyond_the_split_backward(device, dt, side):
    """A 72 x 72 level (5 184 pixels) is beyond the f64 split backward (<= 4 096 pixels per level) but inside the packed
    value gradient's limit (round 6: one 128 KB plane per block, <= 16 384 pixels; its location / weight part is the gather
    kernel of mbv_ms_deform_attn_bwd_locattn, which has no limit); a 136 x 136 level is beyond both, the packed form must
    report itself unsupported and the layer's 16-bit backward must take the general path instead of raising.
    Gradients against the same layer in f32."""
    import ctypes
    from mask_bev_amd import _lib
    from mask_bev_amd.layers import MultiScaleDeformableAttention
    shapes = [(side, side), (36, 36)]
    host = (ctypes.c_int64 * 4)(side, side, 36, 36)
    lib = _lib.load()
    n = sum(h * w for h, w in shapes)
    assert lib.mbv_ms_deform_attn_bwd_split(32, 2, host) == 0
    assert lib.mbv_ms_deform_attn_bwd_value_packed_supported(32, 2, 4, n, host) == (1 if side * side <= 16384 else 0)
    ok = (ctypes.c_int64 * 4)(64, 64, 36, 36)
    assert lib.mbv_ms_deform_attn_bwd_value_packed_supported(32, 2, 4, 64 * 64 + 36 * 36, ok) == 1
    torch.manual_seed(0)
    m = MultiScaleDeformableAttention(256, 8, 2, 4).to(device)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.attention_weights.weight.normal_(0, 0.05)
    g = torch.Generator(device=device).manual_seed(1)
    x0 = torch.randn(1, n, 256, device=device, generator=g)
    pos = torch.randn(1, n, 256, device=device, generator=g)
    ref_pts = torch.rand(n, 2, device=device, generator=g)
    shapes_t = torch.tensor(shapes, dtype=torch.int64, device=device)
    level_start = torch.tensor([0, side * side], dtype=torch.int64, device=device)
    gy = torch.randn(1, n, 256, device=device, generator=g)
    res = {}
    for mode in (torch.float32, dt):
        x = x0.clone().requires_grad_()
        for p in m.parameters():
            p.grad = None
        with torch.autocast('cuda', dtype=dt, enabled=mode != torch.float32):
            y = m(x, pos, ref_pts, shapes, shapes_t, level_start)
        (y.float() * gy).sum().backward()
        res[mode] = (y.detach().float(), x.grad.clone(), m.value_proj.weight.grad.clone())
    tol = 4e-2 if dt == torch.bfloat16 else 1e-2
    for a, b in zip(res[dt], res[torch.float32]):
        assert torch.isfinite(a).all()
        assert float((a - b).norm() / b.norm()) < tol
