"""K6 decoder attention (split-L forward + combine, flash-style backward) vs dense f32 attention.
f32 path: rtol 1e-4; bf16 path on bf16-rounded inputs: 2e-2 / 3e-2 (declared bf16 tolerance)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_attention(q, k, v, blocked, heads):
    b, nq, e = q.shape
    nl = k.shape[1]
    d = e // heads
    qh = q.view(b, nq, heads, d).transpose(1, 2)
    kh = k.view(b, nl, heads, d).transpose(1, 2)
    vh = v.view(b, nl, heads, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / d ** 0.5
    if blocked is not None:
        s = s.masked_fill(blocked.view(b, 1, nq, nl), float('-inf'))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(b, nq, e)


CASES = [(2, 100, 1024, 8, 32, True), (1, 100, 100, 8, 32, False), (2, 8, 25, 8, 16, True), (1, 130, 300, 4, 16, True),
         (2, 100, 4096, 8, 32, True), (1, 6, 7, 2, 64, False)]


@pytest.mark.parametrize('B,Q,L,heads,D,masked', CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_attention_fwd_bwd(device, B, Q, L, heads, D, masked, dtype):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(Q * 7 + L)
    E = heads * D
    q, k, v = (torch.randn(B, n, E, generator=g) for n in (Q, L, L))
    go = torch.randn(B, Q, E, generator=g)
    blocked = None
    if masked:
        blocked = torch.rand(B, Q, L, generator=g) < 0.7
        blocked[:, 0] = True
        blocked[:, 0, L // 2] = False                      # a row with a single attendable key
        blocked[:, 1, :min(L, 128)] = True                 # a whole first split blocked (if L > 128 other keys remain)
        blocked[:, 1, -1] = False
    if dtype == 'bf16':
        q, k, v, go = (t.bfloat16().float() for t in (q, k, v, go))
    qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
    out_ref = ref_attention(qr, kr, vr, blocked, heads)
    out_ref.backward(go)
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    qd, kd, vd = (t.to(device=device, dtype=tdt).requires_grad_() for t in (q, k, v))
    bd = None if blocked is None else blocked.to(device).unsqueeze(1)
    out = ops.attention(qd, kd, vd, bd, heads)
    out.backward(go.to(device=device, dtype=tdt))
    tol = dict(rtol=1e-4, atol=2e-5) if dtype == 'f32' else dict(rtol=2e-2, atol=2e-2)
    gtol = dict(rtol=2e-4, atol=1e-4) if dtype == 'f32' else dict(rtol=3e-2, atol=6e-2)
    torch.testing.assert_close(out.detach().float().cpu(), out_ref.detach(), **tol)
    torch.testing.assert_close(qd.grad.float().cpu(), qr.grad, **gtol)
    torch.testing.assert_close(kd.grad.float().cpu(), kr.grad, **gtol)
    torch.testing.assert_close(vd.grad.float().cpu(), vr.grad, **gtol)
