"""K12 fused residual-add + LayerNorm vs torch (f64 reference): forward, every gradient, both storage types, the
pre-LN form (sum returned and used) and direct accumulation of the affine gradients into a parameter arena."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(a, b, w, bias, eps, use_sum):
    a, w, bias = a.double(), w.double(), bias.double()
    s = a if b is None else a + b.double()
    y = F.layer_norm(s, (a.shape[-1],), w, bias, eps)
    return y, s


@pytest.mark.parametrize('shape', [(4, 100, 256), (3000, 192), (2, 7, 9, 1536), (37, 48), (5, 2048), (130, 768)])
@pytest.mark.parametrize('a_dt,b_dt,out_dt', [(torch.float32, None, torch.float32),
                                              (torch.float32, torch.bfloat16, torch.bfloat16),
                                              (torch.float32, torch.float32, torch.float32),
                                              (torch.bfloat16, torch.bfloat16, torch.float32),
                                              (torch.float32, torch.float16, torch.float16),
                                              (torch.float16, torch.float16, torch.float32)])
@pytest.mark.parametrize('use_sum', [False, True])
def test_add_layernorm(device, shape, a_dt, b_dt, out_dt, use_sum):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    a = (torch.randn(shape, generator=g) * 2 + 0.5).to(device).to(a_dt).requires_grad_()
    b = None if b_dt is None else torch.randn(shape, generator=g).to(device).to(b_dt).requires_grad_()
    w = (torch.rand(c, generator=g) + 0.5).to(device).requires_grad_()
    bias = torch.randn(c, generator=g).to(device).requires_grad_()
    ar = a.detach().double().requires_grad_()
    br = None if b is None else b.detach().double().requires_grad_()
    wr, biasr = w.detach().double().requires_grad_(), bias.detach().double().requires_grad_()
    yr, sr = _ref(ar, br, wr, biasr, 1e-5, use_sum)
    out = ops.add_layernorm(a, b, w, bias, 1e-5, out_dt, return_sum=use_sum)
    y, s = out if use_sum else (out, None)
    assert y.dtype == out_dt
    # bf16 holds 8 significand bits (half an ulp = 2^-9 relative, x the |y| <= ~8 of these inputs), IEEE half 11
    LO = {torch.bfloat16: 1.6e-2, torch.float16: 2e-3}
    tol = LO.get(out_dt, 2e-5)
    assert torch.allclose(y.double(), yr, rtol=tol, atol=tol)
    gy = torch.randn(shape, generator=g).to(device)
    gs = torch.randn(shape, generator=g).to(device)
    loss = (y.double() * gy.double()).sum()
    lossr = (yr * gy.double()).sum()
    if use_sum:
        assert torch.allclose(s.double(), sr, rtol=1e-6, atol=1e-6)
        loss = loss + (s.double() * gs.double()).sum()
        lossr = lossr + (sr * gs.double()).sum()
    loss.backward()
    lossr.backward()

    def close(got, want, t):
        scale = float(want.abs().max()) + 1e-12
        return float((got.double() - want).abs().max()) <= t * scale

    lowp = out_dt in LO                             # dy reaches the kernel rounded to 16 bits
    lo_t = max([LO[d] for d in (a_dt, b_dt, out_dt) if d in LO], default=None)
    gt = 3e-5 if (a_dt == torch.float32 and not lowp) else lo_t
    assert close(a.grad, ar.grad, gt)
    if b is not None:
        assert close(b.grad, br.grad, 3e-5 if (b_dt == torch.float32 and not lowp) else lo_t)
    assert close(w.grad, wr.grad, (1e-2 if out_dt == torch.bfloat16 else 2e-3) if lowp else 1e-4)
    assert close(bias.grad, biasr.grad, (1e-2 if out_dt == torch.bfloat16 else 2e-3) if lowp else 1e-4)


def test_layernorm_module_accumulates_into_arena(device):
    from mask_bev_amd.arena import ParameterArena
    from mask_bev_amd.layers import LayerNorm
    torch.manual_seed(0)
    ref = torch.nn.LayerNorm(192).to(device)
    mine = LayerNorm(192).to(device)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.normal_()
    mine.load_state_dict(ref.state_dict())
    arena = ParameterArena([('all', mine)], shadow_dtype=None)
    x = torch.randn(6, 50, 192, device=device)
    r = torch.randn(6, 50, 192, device=device)
    for it in range(2):                                    # two backward passes accumulate
        ref(x + r).square().mean().backward()
        mine(x, r).square().mean().backward()
    assert mine.weight.grad.data_ptr() == arena.grad.data_ptr() + 4 * arena.layout[0][1]
    assert torch.allclose(mine.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(mine.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-6)
