"""K12 fused residual-add + LayerNorm vs torch (f64 reference): forward, every gradient, both storage types, the
pre-LN form (sum returned and used) and direct accumulation of the affine gradients into a parameter arena."""
import pytest
import torch
import torch.nn.functional as F
from mask_bev_amd import switches

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


def _unfold_ref(x):
    """nn.Unfold(2, stride 2) of a channels-last map: (B, H, W, C) → (B, H/2, W/2, 4C), channel order c*4 + kh*2 + kw
    (mmdet PatchMerging's sampler, swin.py:611-616)."""
    b, h, w, c = x.shape
    u = F.unfold(x.permute(0, 3, 1, 2), kernel_size=2, stride=2)                  # (B, C*4, H/2 * W/2)
    return u.transpose(1, 2).reshape(b, h // 2, w // 2, 4 * c)


@pytest.mark.parametrize('shape', [(2, 8, 12, 48), (4, 32, 32, 192), (1, 10, 6, 384), (3, 4, 4, 512), (2, 64, 64, 192), (4, 32, 32, 768)])
@pytest.mark.parametrize('out_dt', [torch.float32, torch.bfloat16, torch.float16])
def test_merge_layernorm(device, shape, out_dt):
    """K12 with patch-merging addressing against nn.Unfold + F.layer_norm in f64: forward, dx (scattered back into the
    map's layout), dgamma, dbeta.  Tolerances as test_add_layernorm."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    b, h, w, c = shape
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(device).requires_grad_()
    wt = (torch.rand(4 * c, generator=g) + 0.5).to(device).requires_grad_()
    bias = torch.randn(4 * c, generator=g).to(device).requires_grad_()
    xr, wr, br = (t.detach().double().requires_grad_() for t in (x, wt, bias))
    yr = F.layer_norm(_unfold_ref(xr), (4 * c,), wr, br, 1e-5)
    y = ops.merge_layernorm(x, wt, bias, 1e-5, out_dt)
    assert y.dtype == out_dt and tuple(y.shape) == (b, h // 2, w // 2, 4 * c)
    LO = {torch.bfloat16: 1.6e-2, torch.float16: 2e-3}
    tol = LO.get(out_dt, 2e-5)
    assert torch.allclose(y.double(), yr, rtol=tol, atol=tol)
    gy = torch.randn(y.shape, generator=g).to(device)
    (y.double() * gy.double()).sum().backward()
    (yr * gy.double()).sum().backward()

    def close(got, want, t):
        return float((got.double() - want).abs().max()) <= t * (float(want.abs().max()) + 1e-12)

    lowp = out_dt in LO
    assert close(x.grad, xr.grad, LO[out_dt] if lowp else 3e-5)
    assert close(wt.grad, wr.grad, (1e-2 if out_dt == torch.bfloat16 else 2e-3) if lowp else 1e-4)
    assert close(bias.grad, br.grad, (1e-2 if out_dt == torch.bfloat16 else 2e-3) if lowp else 1e-4)


def test_patch_merging_module_equals_unfused_path(device, monkeypatch):
    """layers.PatchMerging through the gather kernel and through the copy + K12 path: same output and gradients (f32
    compute; the two differ only in the order of the f32 partial sums of dgamma / dbeta); odd maps take the padded path."""
    from mask_bev_amd.layers import PatchMerging
    torch.manual_seed(1)
    m = PatchMerging(96, 192).to(device)
    with torch.no_grad():
        m.norm.weight.uniform_(0.5, 1.5)
        m.norm.bias.normal_()
    res = {}
    for mode in ('1', '0'):
        switches.patch(monkeypatch, merge_ln=mode)
        x = torch.randn(2, 16, 20, 96, device=device, generator=torch.Generator(device=device).manual_seed(3)).requires_grad_()
        for p in m.parameters():
            p.grad = None
        y = m(x)
        y.square().sum().backward()
        res[mode] = (y.detach(), x.grad, m.norm.weight.grad.clone(), m.norm.bias.grad.clone(), m.reduction.weight.grad.clone())
    for a, b in zip(res['1'], res['0']):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    switches.patch(monkeypatch, merge_ln='1')
    y = m(torch.randn(1, 7, 9, 96, device=device))
    assert tuple(y.shape) == (1, 4, 5, 192)


@pytest.mark.parametrize('shape,dt2', [((4, 100, 256), torch.float32), ((3000, 192), torch.bfloat16), ((37, 768), torch.float32)])
def test_add_layernorm_fanout_adds_the_two_gradients_on_load(device, shape, dt2):
    """``fanout``: y leaves as two tensors over one buffer; the backward kernel adds their gradients on load
    (mbv_add_layernorm_bwd2) — against F.layer_norm in f64 with dy = g1 + g2; one of the two gradients missing works too."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    a = (torch.randn(shape, generator=g) * 2 + 0.5).to(device).requires_grad_()
    b = torch.randn(shape, generator=g).to(device).requires_grad_()
    w = (torch.rand(c, generator=g) + 0.5).to(device).requires_grad_()
    bias = torch.randn(c, generator=g).to(device).requires_grad_()
    g1 = torch.randn(shape, generator=g).to(device)
    g2 = torch.randn(shape, generator=g).to(device).to(dt2)
    ar, br, wr, biasr = (t.detach().double().requires_grad_() for t in (a, b, w, bias))
    yr = F.layer_norm(ar + br, (c,), wr, biasr, 1e-5)
    (yr * (g1.double() + g2.double())).sum().backward()
    y, y2 = ops.add_layernorm(a, b, w, bias, 1e-5, torch.float32, fanout=True)
    assert y.data_ptr() == y2.data_ptr() and torch.equal(y, y2)
    assert torch.allclose(y.double(), yr, rtol=2e-5, atol=2e-5)
    torch.autograd.backward([y, y2], [g1, g2.to(y2.dtype) if dt2 == torch.float32 else g2.float()])

    def close(got, want, t):
        return float((got.double() - want).abs().max()) <= t * (float(want.abs().max()) + 1e-12)

    for got, want in ((a.grad, ar.grad), (b.grad, br.grad)):
        assert close(got, want, 3e-5)
    assert close(w.grad, wr.grad, 1e-4) and close(bias.grad, biasr.grad, 1e-4)
    # only the second consumer sends a gradient
    a2 = a.detach().clone().requires_grad_()
    y, y2 = ops.add_layernorm(a2, b.detach(), w.detach(), bias.detach(), 1e-5, torch.float32, fanout=True)
    (y2 * g1).sum().backward()
    ar2 = a.detach().double().requires_grad_()
    (F.layer_norm(ar2 + b.detach().double(), (c,), w.detach().double(), bias.detach().double(), 1e-5) * g1.double()).sum().backward()
    assert close(a2.grad, ar2.grad, 3e-5)


@pytest.mark.parametrize('shape,dt2', [((4, 5376, 256), torch.bfloat16), ((300, 192), torch.float16), ((37, 768), torch.bfloat16)])
def test_add_layernorm_fanout_branch_copy_in_16_bits(device, shape, dt2):
    """``fanout`` with ``branch_dtype`` (mbv_add_layernorm_fwd2): the branch consumer's copy of y is written in 16 bits by the
    same launch — the f32 y rounded to nearest, a tensor of its own — and its 16-bit gradient is added to the f32 one on
    load by the backward kernel; against F.layer_norm in f64."""
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + 1)
    c = shape[-1]
    a = (torch.randn(shape, generator=g) * 2 + 0.5).to(device).requires_grad_()
    b = torch.randn(shape, generator=g).to(device).requires_grad_()
    w = (torch.rand(c, generator=g) + 0.5).to(device).requires_grad_()
    bias = torch.randn(c, generator=g).to(device).requires_grad_()
    g1 = torch.randn(shape, generator=g).to(device)
    g2 = torch.randn(shape, generator=g).to(device).to(dt2)
    ar, br, wr, biasr = (t.detach().double().requires_grad_() for t in (a, b, w, bias))
    yr = F.layer_norm(ar + br, (c,), wr, biasr, 1e-5)
    (yr * (g1.double() + g2.double())).sum().backward()
    y, y2 = ops.add_layernorm(a, b, w, bias, 1e-5, torch.float32, fanout=True, branch_dtype=dt2)
    assert y.dtype == torch.float32 and y2.dtype == dt2 and y.data_ptr() != y2.data_ptr()
    assert torch.equal(y2, y.to(dt2))
    assert torch.allclose(y.double(), yr, rtol=2e-5, atol=2e-5)
    torch.autograd.backward([y, y2], [g1, g2])

    def close(got, want, t):
        return float((got.double() - want).abs().max()) <= t * (float(want.abs().max()) + 1e-12)

    for got, want in ((a.grad, ar.grad), (b.grad, br.grad)):
        assert close(got, want, 3e-5)
    assert close(w.grad, wr.grad, 1e-4) and close(bias.grad, biasr.grad, 1e-4)


@pytest.mark.parametrize('shape', [(4, 16, 24, 96), (2, 5, 7, 192)])
def test_position_tokens_ride_on_the_layernorm_launch(device, shape):
    """ops.pos_tokens + K12's repeating ``b`` (mbv_add_layernorm_fwd2 b_rows; mbv_transposed_batch_sum_accum): the (1, C, rows,
    cols) absolute position embedding added to (B, H, W, C) patch tokens inside the LayerNorm launch, its gradient the batch
    sum of dx transposed back — against the explicit broadcast add + F.layer_norm in f64 (swin.py:750-760)."""
    from mask_bev_amd import ops
    b, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(device).requires_grad_()
    ape = torch.randn(1, c, w, h, generator=g).to(device).requires_grad_()          # (w, h) on purpose, as the reference has it
    gam = (torch.rand(c, generator=g) + 0.5).to(device).requires_grad_()
    bet = torch.randn(c, generator=g).to(device).requires_grad_()
    gy = torch.randn(shape, generator=g).to(device)
    gs = torch.randn(shape, generator=g).to(device)
    pos = ops.pos_tokens(ape, b, h, w)
    assert pos.shape == x.shape and pos.stride(0) == 0
    y, s = ops.add_layernorm(x, pos, gam, bet, 1e-5, torch.float32, return_sum=True)
    torch.autograd.backward([y, s], [gy, gs])
    xr, ar, gr, br = (t.detach().double().requires_grad_() for t in (x, ape, gam, bet))
    sr = xr + ar.flatten(2).transpose(1, 2).reshape(1, h, w, c)
    yr = F.layer_norm(sr, (c,), gr, br, 1e-5)
    torch.autograd.backward([yr, sr], [gy.double(), gs.double()])
    assert torch.allclose(y.double(), yr, rtol=2e-5, atol=2e-5) and torch.allclose(s.double(), sr, rtol=1e-6, atol=1e-6)
    for got, want in ((x.grad, xr.grad), (ape.grad, ar.grad), (gam.grad, gr.grad), (bet.grad, br.grad)):
        assert float((got.double() - want).abs().max()) <= 1e-4 * (float(want.abs().max()) + 1e-12)
