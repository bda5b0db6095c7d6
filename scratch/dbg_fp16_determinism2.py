import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd import ops
dev = torch.device('cuda', 0)
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    B, Q, L, E, H = 2, 8, 576, 256, 8
    q = torch.randn(B, Q, E, device=dev).to(dt)
    k = torch.randn(B, L, E, device=dev).to(dt)
    v = torch.randn(B, L, E, device=dev).to(dt)
    blocked = torch.rand(B, 1, Q, L, device=dev) > 0.5
    r = [ops.attention(q, k, v, blocked, H) for _ in range(4)]
    print(dt, 'K6 attention reproducible:', all(torch.equal(r[0], x) for x in r[1:]))
    e = torch.randn(B, Q, E, device=dev).to(dt)
    f = torch.randn(B, E, 24, 24, device=dev).to(dt)
    r = [ops.mask_logits(e, f, (12, 12)) for _ in range(4)]
    print(dt, 'K7 reproducible:', all(torch.equal(r[0][0], x[0]) and torch.equal(r[0][1], x[1]) for x in r[1:]))
    for (m, kk, n) in [(16, 256, 256), (16, 256, 2048), (16, 2048, 256), (16, 256, 768), (1152, 256, 256)]:
        x = torch.randn(m, kk, device=dev).to(dt)
        w = torch.randn(n, kk, device=dev).to(dt)
        b = torch.randn(n, device=dev).to(dt)
        r = [torch.nn.functional.linear(x, w, b) for _ in range(6)]
        r2 = [torch.addmm(b.float(), x, w.t(), out_dtype=torch.float32) for _ in range(6)]
        print(dt, (m, kk, n), 'linear reproducible:', all(torch.equal(r[0], y) for y in r[1:]),
              'addmm f32-out:', all(torch.equal(r2[0], y) for y in r2[1:]))
    mha = torch.nn.MultiheadAttention(E, H, batch_first=False).to(dev)
    x = torch.randn(Q, B, E, device=dev)
    with torch.autocast('cuda', dtype=dt), torch.no_grad():
        r = [mha(x, x, x)[0] for _ in range(4)]
    print(dt, 'torch MHA reproducible:', all(torch.equal(r[0], y) for y in r[1:]))
