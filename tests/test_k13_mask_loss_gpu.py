"""K13 fused dice/BCE row sums vs the torch expressions of the oracle's loss (forward and gradient)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('R,P', [(40, 12544), (7, 1001), (3, 4), (1, 1)])
def test_mask_loss_rows(device, R, P):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(R * 7 + P)
    x = (torch.randn(R, P, generator=g) * 6).to(device).requires_grad_()
    t = (torch.rand(R, P, generator=g) > 0.6).float().to(device)
    t[0] = torch.rand(P, generator=g).to(device)                     # non-binary targets are allowed
    xr = x.detach().double().requires_grad_()
    td = t.double()
    ps = xr.sigmoid()
    want = torch.stack([(ps * td).sum(1), ps.sum(1), td.sum(1),
                        F.binary_cross_entropy_with_logits(xr, td, reduction='none').sum(1)], 1)
    got = ops.mask_loss_rows(x, t)
    assert torch.allclose(got.double(), want, rtol=2e-5, atol=1e-4)
    w = torch.randn(R, 4, generator=g).to(device)
    (got * w).sum().backward()
    (want * w.double()).sum().backward()
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-4, atol=1e-6)


def test_match_cost_terms(device):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(5, 7, 1003, generator=g) * 5).to(device)
    terms, sums = ops.match_cost_terms(x)
    xd = x.double()
    want = torch.cat([F.softplus(-xd), F.softplus(xd), xd.sigmoid()], dim=1)
    assert terms.shape == (5, 21, 1003)
    assert torch.allclose(terms.double(), want, rtol=2e-6, atol=2e-6)
    assert torch.allclose(sums[..., 0].double(), F.softplus(xd).sum(-1), rtol=1e-5)
    assert torch.allclose(sums[..., 1].double(), xd.sigmoid().sum(-1), rtol=1e-5)


@pytest.mark.gpu
def test_dice_bce_node_equals_the_row_sums_algebra(device):
    """ops.mask_dice_bce (K13 + the dice / BCE algebra as one autograd node with an analytic gradient of the four sums)
    against ops.mask_loss_rows followed by the same algebra through autograd: values and the gradient of the logits."""
    import torch
    from mask_bev_amd import ops
    torch.manual_seed(5)
    d, g, p = 4, 13, 777
    x = (torch.randn(d * g, p, device=device) * 3).requires_grad_()
    t = (torch.rand(d * g, p, device=device) > 0.6).float()
    c_dice = torch.tensor(5.0 / 401.0, device=device)
    c_mask = 5.0 / (400.0 * p + 1.0)
    gd, gm = torch.randn(d, device=device), torch.randn(d, device=device)
    ld, lm = ops.mask_dice_bce(x, t, d, c_dice, c_mask)
    torch.autograd.backward([ld, lm], [gd, gm])
    got = x.grad.clone()
    x.grad = None
    sums = ops.mask_loss_rows(x, t)
    dice = (2 * sums[:, 0] + 1.0) / (sums[:, 1] + sums[:, 2] + 1.0)
    ld2 = (1 - dice).view(d, g).sum(1) * c_dice
    lm2 = sums[:, 3].view(d, g).sum(1) * c_mask
    torch.autograd.backward([ld2, lm2], [gd, gm])
    assert torch.allclose(ld, ld2, rtol=1e-6, atol=1e-7) and torch.allclose(lm, lm2, rtol=1e-6, atol=1e-7)
    assert float((got - x.grad).abs().max()) <= 1e-6 * float(x.grad.abs().max()) + 1e-12
