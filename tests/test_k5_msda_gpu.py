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
