"""K7: mask-logit contraction (MFMA) and attention-mask generation vs the dense restatement of
mask2former_head.py:459-470,538-539.  f32: rtol 1e-4; bf16 inputs: 2e-2 (declared bf16 tolerance)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref(embed, feat, size):
    logits = torch.einsum('bqc,bchw->bqhw', embed, feat)
    a = F.interpolate(logits, size, mode='bilinear', align_corners=False).flatten(2)
    blocked = a.sigmoid() < 0.5
    blocked[torch.where(blocked.sum(-1) == blocked.shape[-1])] = False
    return logits, blocked


@pytest.mark.parametrize('B,Q,C,H,W,size', [(2, 100, 256, 128, 128, (32, 32)), (1, 8, 32, 20, 20, (5, 5)),
                                            (2, 130, 64, 31, 25, (8, 7)), (1, 6, 128, 16, 24, (16, 24)),
                                            (2, 100, 256, 125, 125, (63, 63))])
@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'fp16'])
def test_mask_logits_and_attn_mask(device, B, Q, C, H, W, size, dtype):
    from mask_bev_amd import ops
    g = torch.Generator().manual_seed(Q + H)
    embed = torch.randn(B, Q, C, generator=g) / C ** 0.5
    feat = torch.randn(B, C, H, W, generator=g)
    embed[0, 0] = -embed[0, 0].abs() * 0 - 0.0        # a zero query
    feat_neg = feat.clone()
    go = torch.randn(B, Q, H, W, generator=g)
    tdt = dict(f32=torch.float32, bf16=torch.bfloat16, fp16=torch.float16)[dtype]
    embed, feat, go = embed.to(tdt).float(), feat.to(tdt).float(), go.to(tdt).float()
    e_r, f_r = embed.clone().requires_grad_(), feat.clone().requires_grad_()
    logits_ref, blocked_ref = ref(e_r, f_r, size)
    logits_ref.backward(go)
    e_d = embed.to(device=device, dtype=tdt).requires_grad_()
    f_d = feat.to(device=device, dtype=tdt).requires_grad_()
    logits, blocked = ops.mask_logits(e_d, f_d, size)
    logits.backward(go.to(device=device, dtype=tdt))
    assert blocked.shape == (B, 1, Q, size[0] * size[1]) and blocked.dtype == torch.bool
    # declared: f32 exact; bf16 8 significand bits; IEEE half 11 bits (a quarter of the bf16 figures)
    tol = {'f32': dict(rtol=1e-4, atol=1e-4), 'bf16': dict(rtol=2e-2, atol=5e-2), 'fp16': dict(rtol=5e-3, atol=1.2e-2)}[dtype]
    torch.testing.assert_close(logits.detach().float().cpu(), logits_ref.detach(), **tol)
    gtol = {'f32': dict(rtol=1e-3, atol=1e-3), 'bf16': dict(rtol=3e-2, atol=0.3), 'fp16': dict(rtol=8e-3, atol=8e-2)}[dtype]
    torch.testing.assert_close(e_d.grad.float().cpu(), e_r.grad, **gtol)
    torch.testing.assert_close(f_d.grad.float().cpu(), f_r.grad, **gtol)
    # the boolean mask may differ only where the resized logit is within rounding of 0
    small = F.interpolate(logits_ref.detach(), size, mode='bilinear', align_corners=False).flatten(2)
    diff = blocked[:, 0].cpu() != blocked_ref
    assert not (diff & (small.abs() > {'f32': 1e-4, 'bf16': 0.1, 'fp16': 0.025}[dtype])).any()
    assert not blocked.all(-1).any()


def test_all_blocked_rows_are_unblocked(device):
    from mask_bev_amd import ops
    embed = torch.ones(1, 4, 16, device=device)
    feat = -torch.ones(1, 16, 8, 8, device=device)
    feat[0, :, :4] = 1.0                                  # top half positive logits, bottom half negative
    embed[0, 1] = -1.0                                    # query 1: inverted
    embed[0, 2] = 0.0                                     # query 2: logits 0 → sigmoid = 0.5 → not blocked
    feat2 = -torch.ones(1, 16, 8, 8, device=device)
    _, blocked = ops.mask_logits(embed, feat2, (4, 4))    # every logit negative for queries 0, 3 → rows unblocked
    assert not blocked.any()
    _, blocked = ops.mask_logits(embed, feat, (4, 4))
    b = blocked[0, 0].view(4, 4, 4)
    assert not b[0, :2].any() and b[0, 2:].all()          # query 0: bottom half blocked
    assert b[1, :2].all() and not b[1, 2:].any()
