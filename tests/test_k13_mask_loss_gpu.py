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


@pytest.mark.gpu
@pytest.mark.parametrize('upstream', ['vectors', 'broadcast', 'dice_only'])
def test_fused_dice_bce_reduce_equals_the_tensor_form(device, upstream):
    """Round 6: with plain-float constants the algebra on the row sums is ONE launch (mbv_dice_bce_reduce) and the backward
    kernel assembles each row's gradient from its coefficients and the upstream gradients of the row's decoder output
    (mbv_mask_loss_rows_bwd_coef; a broadcast scalar arrives as a stride-0 view, a missing gradient as NULL) — against the
    tensor-constant form of the same node (ATen algebra + mbv_mask_loss_rows_bwd), values and d(logits)."""
    import torch
    from mask_bev_amd import ops
    torch.manual_seed(6)
    d, g, p = 10, 37, 1001
    x = (torch.randn(d * g, p, device=device) * 3).requires_grad_()
    t = (torch.rand(d * g, p, device=device) > 0.6).float()
    c_dice, c_mask = 5.0 / 401.0, 5.0 / (400.0 * p + 1.0)

    def run(cd, cm):
        x.grad = None
        ld, lm = ops.mask_dice_bce(x, t, d, cd, cm)
        if upstream == 'vectors':
            gd, gm = torch.linspace(-1, 2, d, device=device), torch.linspace(3, -1, d, device=device)
            torch.autograd.backward([ld, lm], [gd, gm])
        elif upstream == 'broadcast':
            torch.cat([ld, lm]).sum().backward()             # the head's `total`: expanded ones reach the node
        else:
            (ld * torch.linspace(1, 2, d, device=device)).sum().backward()
        return ld.detach(), lm.detach(), x.grad.clone()

    ld, lm, gx = run(c_dice, c_mask)
    ld2, lm2, gx2 = run(torch.tensor(c_dice, device=device), torch.tensor(c_mask, device=device))
    assert torch.allclose(ld, ld2, rtol=2e-6, atol=1e-7) and torch.allclose(lm, lm2, rtol=2e-6, atol=1e-7)
    assert float((gx - gx2).abs().max()) <= 2e-6 * float(gx2.abs().max()) + 1e-12


def test_match_cost_single_launch_equals_the_torch_form(device):
    """mbv_match_cost (+ the ones row of mbv_match_cost_terms) against the expression it replaces in
    Mask2FormerHead._match_cost: -2 softmax(cls)[label] + 5 BCE + 5 dice on the sampled points (mask2former_head.py:199-210)."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(3)
    d, b, q, ng, p, k1 = 3, 2, 10, 7, 96, 3
    cls = torch.randn(d, b, q, k1, generator=g).to(device)
    labels = torch.randint(0, k1 - 1, (b, ng), generator=g).to(device)
    mp = (torch.randn(d * b, q, p, generator=g) * 3).to(device)
    gp = torch.rand(d * b, ng, p, generator=g).to(device)
    gpt = gp.transpose(1, 2)
    # the torch form
    prob = cls.softmax(-1)
    cls_cost = -torch.gather(prob, 3, labels.view(1, b, 1, ng).expand(d, b, q, ng)) * 2.0
    terms, sums = ops.match_cost_terms(mp)
    prod = torch.matmul(terms, gpt).view(d, b, 3, q, ng)
    s = sums.view(d, b, q, 2)
    bce = (prod[:, :, 0] + s[..., 0:1] - prod[:, :, 1]) / p
    dice = 1 - (2 * prod[:, :, 2] + 1.0) / (s[..., 1:2] + gp.view(d, b, ng, p).sum(-1)[..., None, :] + 1.0)
    ref = (cls_cost + 5.0 * bce + 5.0 * dice).flatten(0, 1)
    # one launch
    terms1, sums1 = ops.match_cost_terms(mp, ones_row=True)
    assert torch.equal(sums1, sums) and torch.equal(terms1[:, :3 * q], terms) and bool((terms1[:, 3 * q] == 1).all())
    got = ops.match_cost(cls, labels, torch.matmul(terms1, gpt), sums1, p)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), float((got - ref).abs().max())


@pytest.mark.parametrize('shape', [(3, 2, 10, 7, 96, None), (2, 2, 100, 100, 12544 // 4, None), (1, 3, 37, 5, 200, 3),
                                   (2, 1, 111, 127, 40, 1), (1, 2, 16, 31, 1000, 32), (1, 1, 1, 1, 8, None)])
def test_match_products_on_half_pairs_equal_float64_products(device, shape):
    """K13c (mbv_match_products + mbv_match_cost_split): the matcher's products on MFMA from IEEE-half pairs, sliced over the
    points, against the float64 evaluation of mask2former_head.py:199-210 — every row / column tile count, a trailing
    partial chunk (P % 32 != 0), one slice and as many slices as chunks, fractional targets, large logits."""
    from mask_bev_amd import ops
    d, b, q, ng, p, splits = shape
    g = torch.Generator().manual_seed(11 + q)
    k1 = 3
    assert ops.match_products_supported(q, ng, p)
    cls = torch.randn(d, b, q, k1, generator=g).to(device)
    labels = torch.randint(0, k1 - 1, (b, ng), generator=g).to(device)
    mp = (torch.randn(d * b, q, p, generator=g) * 6).to(device)
    mp[0, 0, :4] = torch.tensor([90.0, -90.0, 1e-6, 7e4], device=device)[:min(4, p)]
    gp = torch.rand(d * b, ng, p, generator=g).to(device)
    gp[:, ::2] = (gp[:, ::2] > 0.5).float()                      # half the columns binary, as inside a mask
    prod, neg = ops.match_products(mp, gp, splits=splits)
    x64, t64 = mp.double().clamp(-6e4, 6e4), gp.double()
    assert torch.allclose(prod.sum(1)[:, :q, :ng].double(), x64 @ t64.transpose(1, 2), rtol=2e-6, atol=2e-4)
    assert torch.allclose(prod.sum(1)[:, q:2 * q, :ng].double(), x64.sigmoid() @ t64.transpose(1, 2), rtol=2e-6, atol=1e-5)
    assert torch.allclose(prod.sum(1)[:, 2 * q, :ng].double(), t64.sum(-1), rtol=2e-6, atol=1e-5)
    assert torch.allclose(prod.sum(1)[:, q:2 * q, ng].double(), x64.sigmoid().sum(-1), rtol=2e-6, atol=1e-5)
    assert torch.allclose(neg.sum(1).double(), F.softplus(x64).sum(-1), rtol=2e-6, atol=1e-5)
    got = ops.match_cost_split(cls, labels, prod, neg, p)
    prob = cls.double().softmax(-1)
    cls_cost = -torch.gather(prob, 3, labels.view(1, b, 1, ng).expand(d, b, q, ng)) * 2.0
    xt = (x64 @ t64.transpose(1, 2)).view(d, b, q, ng)
    st = (x64.sigmoid() @ t64.transpose(1, 2)).view(d, b, q, ng)
    bce = (F.softplus(x64).sum(-1).view(d, b, q, 1) - xt) / p
    dice = 1 - (2 * st + 1.0) / (x64.sigmoid().sum(-1).view(d, b, q, 1) + t64.sum(-1).view(d, b, 1, ng) + 1.0)
    ref = (cls_cost + 5.0 * bce + 5.0 * dice).flatten(0, 1)
    got, ref = got.double().flatten(0, 1)[1:], ref.flatten(0, 1)[1:]     # row 0 holds the 7e4 logit: its f32 sums cancel
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), float((got - ref).abs().max())
    # same inputs, other slicing: the products are the same sums in another order
    prod2, neg2 = ops.match_products(mp, gp, splits=1)
    assert torch.allclose(prod2.sum(1), prod.sum(1), rtol=1e-5, atol=5e-3)      # f32 sums of up to 3 136 O(10) terms
    # and the launch is reproducible bit for bit
    prod3, neg3 = ops.match_products(mp, gp, splits=splits)
    assert torch.equal(prod3, prod) and torch.equal(neg3, neg)


def test_match_products_unsupported_shapes_are_refused(device):
    from mask_bev_amd import ops
    assert not ops.match_products_supported(112, 10, 64) and not ops.match_products_supported(10, 128, 64)
    assert not ops.match_products_supported(10, 10, 100)
    with pytest.raises(ops.MaskBevHipError):
        ops.match_products(torch.zeros(1, 112, 64, device=device), torch.zeros(1, 10, 64, device=device))


def test_cls_loss_single_launch_equals_cross_entropy(device):
    """mbv_cls_loss_fwd / _bwd against F.cross_entropy(weight=class_weight, reduction='none') summed per decoder output and
    divided by the summed class weights of the targets (mask2former_head.py:393-404), unmatched queries -> 'no object'."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(5)
    d, b, q, ng, k1 = 4, 3, 9, 6, 3
    cls = torch.randn(d, b, q, k1, generator=g).to(device).requires_grad_()
    labels_gt = torch.randint(0, k1 - 1, (b, ng), generator=g).to(device)
    assigned = torch.randint(-1, ng, (d, b, q), generator=g).to(torch.int32).to(device)
    cw = torch.tensor([1.0, 0.7, 0.1], device=device)
    gout = torch.randn(d, generator=g).to(device)
    eps = float(torch.finfo(torch.float32).eps)
    loss = ops.cls_loss(cls, assigned, labels_gt, cw, 2.0, eps)
    (loss * gout).sum().backward()
    got_grad = cls.grad.clone()
    cls.grad = None
    matched = assigned >= 0
    lab = torch.where(matched, torch.gather(labels_gt.view(1, b, ng).expand(d, b, ng), 2, assigned.clamp(min=0).long()),
                      torch.full_like(assigned, k1 - 1, dtype=torch.int64))
    ce = F.cross_entropy(cls.flatten(0, 2), lab.flatten(), weight=cw, reduction='none').view(d, -1)
    ref = 2.0 * ce.sum(1) / (cw[lab].view(d, -1).sum(1) + eps)
    (ref * gout).sum().backward()
    assert torch.allclose(loss, ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got_grad, cls.grad, rtol=1e-4, atol=1e-6)
