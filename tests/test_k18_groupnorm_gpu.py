"""K18 (csrc/groupnorm.hip): GroupNorm of NCHW maps fused with the FPN up-sampled add / ReLU / output cast, against
torch's own ops in f64 (F.group_norm, F.interpolate(bilinear, align_corners=False), relu) on the same (already rounded)
inputs — forward, dx, dgamma, dbeta and the gradient of the added coarser map.  Tolerances: f32 results 2e-5 of the
tensor's largest entry; 16-bit outputs one rounding of the type (bf16 2^-8, fp16 2^-11 relative) — as K12's tests."""
import pytest
import torch
import torch.nn.functional as F
from mask_bev_amd import switches

pytestmark = pytest.mark.gpu

LO = {torch.bfloat16: 1.6e-2, torch.float16: 2e-3}


def _close(got, want, t):
    return float((got.double() - want).abs().max()) <= t * (float(want.abs().max()) + 1e-12)


@pytest.mark.parametrize('shape,groups', [((2, 64, 16, 16), 32), ((4, 256, 32, 32), 32), ((1, 96, 10, 6), 8),
                                          ((3, 256, 64, 64), 32), ((2, 32, 128, 128), 4)])
@pytest.mark.parametrize('x_dt,out_dt', [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                         (torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
@pytest.mark.parametrize('mode', ['plain', 'relu', 'add'])
def test_group_norm(device, shape, groups, x_dt, out_dt, mode):
    from mask_bev_amd import ops
    b, c, h, w = shape
    if mode == 'add' and w % 4:
        pytest.skip('the fused up-sampled add needs W % 4 == 0')
    g = torch.Generator().manual_seed(sum(shape) + groups)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(device).to(x_dt).requires_grad_()
    wt = (torch.rand(c, generator=g) + 0.5).to(device).requires_grad_()
    bias = (torch.randn(c, generator=g) * 0.5).to(device).requires_grad_()
    add = None
    if mode == 'add':
        add = torch.randn((b, c, max(1, h // 2), max(1, w // 2)), generator=g).to(device).requires_grad_()
    xr, wr, br = (t.detach().double().requires_grad_() for t in (x, wt, bias))
    yr = F.group_norm(xr, groups, wr, br, 1e-5)
    if add is not None:
        ar = add.detach().double().requires_grad_()
        yr = yr + F.interpolate(ar, size=(h, w), mode='bilinear', align_corners=False)
    if mode == 'relu':
        yr = F.relu(yr)
    y = ops.group_norm(x, wt, bias, groups, 1e-5, relu=(mode == 'relu'), add_upsampled=add, out_dtype=out_dt)
    assert y.dtype == out_dt and y.shape == x.shape
    tol = LO.get(out_dt, 2e-5)
    assert torch.allclose(y.double(), yr, rtol=tol, atol=tol)
    gy = torch.randn(shape, generator=g).to(device).to(out_dt)        # the gradient arrives in the output's dtype
    (y.double() * gy.double()).sum().backward()
    (yr * gy.double()).sum().backward()
    # ReLU: gates of outputs within rounding of zero may differ from the f64 reference; measured against the tensor's scale
    gt = max(LO.get(x_dt, 0.0), 5e-5 if mode != 'relu' else 2e-3)
    assert _close(x.grad, xr.grad, gt)
    pt = 1e-4 if mode != 'relu' else 2e-3
    assert _close(wt.grad, wr.grad, pt)
    assert _close(bias.grad, br.grad, pt)
    if add is not None:
        assert _close(add.grad, ar.grad, 2e-5 if out_dt == torch.float32 else LO[out_dt])


def test_conv_gn_module_equals_torch_path(device, monkeypatch):
    """layers.ConvGN (1 x 1 conv as GEMM → K18) against the same module with MBV_GROUPNORM=0 (torch GroupNorm, interpolate,
    add, relu), f32 compute: output and every gradient, with arena-free parameters."""
    from mask_bev_amd.layers import ConvGN
    torch.manual_seed(2)
    for relu, with_add in ((False, True), (True, False)):
        m = ConvGN(48, 64, 1, bias=not relu, relu=relu).to(device)
        with torch.no_grad():
            m.gn.weight.uniform_(0.5, 1.5)
            m.gn.bias.normal_()
        res = {}
        for mode in ('1', '0'):
            switches.patch(monkeypatch, groupnorm=mode)
            gen = torch.Generator(device=device).manual_seed(5)
            x = torch.randn(2, 48, 24, 16, device=device, generator=gen).requires_grad_()
            add = torch.randn(2, 64, 12, 8, device=device, generator=gen).requires_grad_() if with_add else None
            for p in m.parameters():
                p.grad = None
            y = m(x, add_upsampled=add)
            y.square().sum().backward()
            res[mode] = [y.detach(), x.grad] + [p.grad.clone() for p in m.parameters()] + ([add.grad] if with_add else [])
        for a, b in zip(res['1'], res['0']):
            assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max())), (relu, with_add)


@pytest.mark.parametrize('fine,coarse', [((16, 16), (8, 8)), ((20, 12), (7, 5)), ((9, 30), (3, 10)), ((5, 7), (5, 7)),
                                         ((12, 8), (1, 1)), ((128, 128), (64, 64))])
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_upsample_bilinear_backward_gather_form(device, fine, coarse, dt):
    """mbv_upsample_bilinear_bwd (the adjoint K18's fused FPN step owes its added map, gather form) against autograd through
    F.interpolate(bilinear, align_corners=False) in f64 — exact 2x, non-integer ratios, identity and a 1 x 1 source."""
    from mask_bev_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(fine) + sum(coarse))
    planes = 6
    gy = torch.randn(planes, *fine, generator=g).to(device).to(dt)
    a = torch.zeros(1, planes, *coarse, dtype=torch.float64, device=device, requires_grad=True)
    (F.interpolate(a, size=fine, mode='bilinear', align_corners=False) * gy.double().unsqueeze(0)).sum().backward()
    out = torch.empty(planes, *coarse, dtype=torch.float32, device=device)
    ops.check(lib.mbv_upsample_bilinear_bwd(ops._ptr(gy), ops._dt_flag(dt), planes, fine[0], fine[1], coarse[0], coarse[1],
                                            ops._ptr(out), 0, ops._stream()), 'mbv_upsample_bilinear_bwd')
    assert torch.allclose(out.double(), a.grad[0], rtol=1e-5, atol=1e-5 * float(a.grad.abs().max()))
