"""K4 fused shifted-window attention: HIP forward/backward vs a dense f32 restatement of
swin.py:80-118,179-253 built from the oracle's helpers.  f32 path (exact-f32 MFMA): rtol 1e-4;
bf16 path: inputs rounded to bf16 on both sides, tolerance 2e-2 (declared bf16 tolerance)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import maskbev_oracle as O

pytestmark = pytest.mark.gpu


def ref_window_attention(qkv, qkv_bias, table, heads, ws, shift):
    b, h, w, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    pad_b, pad_r = (ws - h % ws) % ws, (ws - w % ws) % ws
    hp, wp = h + pad_b, w + pad_r
    full = qkv_bias.view(1, 1, 1, c3).expand(b, hp, wp, c3).clone()
    full[:, :h, :w] = qkv
    mask = None
    if shift:
        full = torch.roll(full, shifts=(-shift, -shift), dims=(1, 2))
        img = torch.zeros((1, hp, wp, 1))
        sl = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
        cnt = 0
        for a in sl:
            for bb in sl:
                img[:, a, bb, :] = cnt
                cnt += 1
        mw = O._window_partition(img, ws).view(-1, ws * ws)
        mask = mw.unsqueeze(1) - mw.unsqueeze(2)
        mask = mask.masked_fill(mask != 0, -100.0).masked_fill(mask == 0, 0.0)
    win = O._window_partition(full, ws).view(-1, ws * ws, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = win[0] * d ** -0.5, win[1], win[2]
    attn = q @ k.transpose(-2, -1)
    bias = table[O.rel_position_index(ws).view(-1)].view(ws * ws, ws * ws, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(b, nw, heads, ws * ws, ws * ws) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, ws * ws, ws * ws)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, ws, ws, c)
    out = O._window_reverse(out, hp, wp, ws)
    if shift:
        out = torch.roll(out, shifts=(shift, shift), dims=(1, 2))
    return out[:, :h, :w].contiguous()


CASES = [
    # B, H, W, heads, D, ws, shift
    (2, 10, 10, 2, 16, 5, 0),
    (2, 10, 10, 2, 16, 5, 2),
    (1, 13, 9, 3, 16, 4, 2),       # padding on both axes + shift
    (2, 23, 30, 3, 64, 10, 5),     # the production shape class: 100-token windows, head dim 64, padded
    (1, 20, 20, 2, 32, 10, 0),
    (1, 7, 7, 1, 32, 7, 3),
]


@pytest.mark.parametrize('B,H,W,heads,D,ws,shift', CASES)
@pytest.mark.parametrize('dtype', ['f32', 'f32_exact', 'bf16', 'fp16'])
def test_window_attention_fwd_bwd(device, B, H, W, heads, D, ws, shift, dtype):
    """'f32': the split mode where the head dimension allows it (IEEE-half pairs on the 16-bit matrix pipe); 'f32_exact':
    v_mfma_f32_32x32x2_f32 (switches.k4_split off)."""
    from mask_bev_amd import ops, switches
    if dtype == 'f32_exact':
        with switches.override(k4_split=False):
            return test_window_attention_fwd_bwd(device, B, H, W, heads, D, ws, shift, 'f32')
    g = torch.Generator().manual_seed(H * 31 + W + shift)
    C = heads * D
    qkv = torch.randn(B, H, W, 3 * C, generator=g)
    bias = torch.randn(3 * C, generator=g) * 0.5
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    go = torch.randn(B, H, W, C, generator=g)
    tdt = dict(f32=torch.float32, bf16=torch.bfloat16, fp16=torch.float16)[dtype]
    qkv, go = qkv.to(tdt).float(), go.to(tdt).float()          # both sides see the inputs as the 16-bit type holds them
    q_r, b_r, t_r = qkv.clone().requires_grad_(), bias.clone().requires_grad_(), table.clone().requires_grad_()
    out_ref = ref_window_attention(q_r, b_r, t_r, heads, ws, shift)
    out_ref.backward(go)
    q_d = qkv.to(device=device, dtype=tdt).requires_grad_()
    b_d, t_d = bias.clone().to(device).requires_grad_(), table.clone().to(device).requires_grad_()
    out = ops.window_attention(q_d, b_d, t_d, heads, ws, shift)
    assert out.dtype == tdt and out.shape == (B, H, W, C)
    out.backward(go.to(device=device, dtype=tdt))
    if dtype == 'f32':
        tol = dict(rtol=1e-4, atol=2e-5)
        gtol = dict(rtol=2e-4, atol=1e-4)
    elif dtype == 'bf16':
        tol = dict(rtol=2e-2, atol=2e-2)
        gtol = dict(rtol=3e-2, atol=6e-2)
    else:                       # IEEE half: 11 significand bits against bf16's 8 -> a quarter of the bf16 tolerance
        tol = dict(rtol=5e-3, atol=5e-3)
        gtol = dict(rtol=8e-3, atol=1.5e-2)
    ptol = {'f32': 2e-4, 'bf16': 3e-2, 'fp16': 8e-3}[dtype]
    torch.testing.assert_close(out.detach().float().cpu(), out_ref.detach(), **tol)
    torch.testing.assert_close(q_d.grad.float().cpu(), q_r.grad, **gtol)
    scale = float(t_r.grad.abs().max().clamp(min=1.0))
    assert float((t_d.grad.cpu() - t_r.grad).abs().max()) / scale < ptol
    scale = float(b_r.grad.abs().max().clamp(min=1.0))
    assert float((b_d.grad.cpu() - b_r.grad).abs().max()) / scale < ptol


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,heads,D,ws,shift,gscale', [(2, 23, 17, 3, 32, 10, 5, 1e-6), (1, 20, 20, 2, 64, 10, 0, 1.0),
                                                        (2, 16, 16, 4, 16, 7, 3, 3e-4), (1, 30, 30, 6, 32, 10, 5, 1e3)])
def test_split_mode_against_float64(device, B, H, W, heads, D, ws, shift, gscale):
    """K4's split mode (f32 tensors, products from IEEE-half pairs) against the float64 attention: output and every gradient
    within 4e-6 of the result's maximum — the exact-f32 MFMA form is held to the same bar beside it — for output gradients of
    1e-6 ... 1e+3 (the tensors' power-of-two scales come from absmax records), and its absmax record of d(qkv) is exact."""
    from mask_bev_amd import ops, switches
    g = torch.Generator().manual_seed(B * 1000 + H + D)
    C = heads * D
    qkv = torch.randn(B, H, W, 3 * C, generator=g) * 2.0
    bias = torch.randn(3 * C, generator=g) * 0.5
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    go = torch.randn(B, H, W, C, generator=g) * gscale
    q_r, b_r, t_r = (t.double().requires_grad_() for t in (qkv, bias, table))
    ref = ref_window_attention(q_r, b_r, t_r, heads, ws, shift)
    ref.backward(go.double())

    def run(split):
        with switches.override(k4_split=split, amax_hints=True):
            q_d = qkv.to(device).requires_grad_()
            b_d, t_d = bias.to(device).requires_grad_(), table.to(device).requires_grad_()
            mid, seen = q_d * 1.0, []
            mid.register_hook(lambda gr: seen.append(ops.amax_hint_get(gr)))      # what the qkv projection's backward finds
            out = ops.window_attention(mid, b_d, t_d, heads, ws, shift)
            out.backward(go.to(device))
            return out.detach(), q_d.grad, b_d.grad, t_d.grad, seen[0]

    def err(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp(min=1e-30))

    res = {split: run(split) for split in (True, False)}
    for split, (out, gq, gb, gt, rec) in res.items():
        for name, a, b in (('out', out, ref.detach()), ('dqkv', gq, q_r.grad), ('dbias', gb, b_r.grad), ('dtable', gt, t_r.grad)):
            assert err(a, b) <= 4e-6, (split, name, err(a, b))
    rec = res[True][4]
    assert rec is not None and res[False][4] is None
    assert int(rec.max()) == int(res[True][1].abs().max().view(torch.int32))
