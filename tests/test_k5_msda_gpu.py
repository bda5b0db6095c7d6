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
