import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from mask_bev_amd import ops, switches
from test_k6_attention_gpu import ref_attention
B, Q, L, heads, D, gscale = 2, 100, 4096, 8, 32, 1e3
g = torch.Generator().manual_seed(Q + L + D)
E = heads * D
q = torch.randn(B, Q, E, generator=g)
k = torch.randn(B, L, E, generator=g) * torch.logspace(-2, 0, heads).repeat_interleave(D)
v = torch.randn(B, L, E, generator=g) * torch.logspace(1, -3, heads).repeat_interleave(D)
go = torch.randn(B, Q, E, generator=g) * gscale
blocked = torch.rand(B, Q, L, generator=g) < 0.5
blocked[:, 0] = True
blocked[:, 0, L // 2] = False
qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
ref = ref_attention(qr, kr, vr, blocked, heads)
ref.backward(go.double())
dev = torch.device('cuda', 0)
for split in (True, False):
    with switches.override(k6_split=split):
        qd, kd, vd = (t.to(dev).requires_grad_() for t in (q, k, v))
        out = ops.attention(qd, kd, vd, blocked.to(dev).unsqueeze(1), heads)
        out.backward(go.to(dev))
    for name, a, b in (('out', out.detach(), ref.detach()), ('dq', qd.grad, qr.grad), ('dk', kd.grad, kr.grad), ('dv', vd.grad, vr.grad)):
        e = (a.double().cpu() - b).abs()
        per_head = [float(e[..., h * D:(h + 1) * D].max() / b[..., h * D:(h + 1) * D].abs().max()) for h in range(heads)]
        print(split, name, 'global', float(e.max() / b.abs().max()), 'per head', ['%.1e' % x for x in per_head])
    if split:
        e = (kd.grad.double().cpu() - kr.grad).abs()
        idx = e.argmax(); print('worst dk element', idx.item(), 'key index', (idx // E % L).item(), 'col', (idx % E).item(), float(kr.grad.flatten()[idx]), float(kd.grad.flatten()[idx]))
