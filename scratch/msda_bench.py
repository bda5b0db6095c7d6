import sys, torch, math
sys.path.insert(0, '.')
from mask_bev_amd import ops
dev = torch.device('cuda:0')
B, H, D, P = 4, 8, 32, 4
shapes = [(16, 16), (32, 32), (64, 64)]
L = len(shapes)
nv = sum(h * w for h, w in shapes); nq = nv
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, nv, H, D, device=dev, generator=g, requires_grad=True)
# reference points + small offsets (what the model produces early in training)
refs = []
for (h, w) in shapes:
    xs = (torch.arange(w, device=dev) + 0.5) / w; ys = (torch.arange(h, device=dev) + 0.5) / h
    refs.append(torch.stack([xs.repeat(h), ys.view(-1, 1).repeat(1, w).view(-1)], -1))
ref = torch.cat(refs, 0)
norm = torch.tensor([[w, h] for h, w in shapes], device=dev, dtype=torch.float32)
off = torch.randn(B, nq, H, L, P, 2, device=dev, generator=g) * 2.0
loc = (ref.view(1, nq, 1, 1, 1, 2) + off / norm.view(1, 1, 1, L, 1, 2)).requires_grad_()
attn = torch.rand(B, nq, H, L, P, device=dev, generator=g).flatten(-2).softmax(-1).view(B, nq, H, L, P).requires_grad_()
shapes_t = torch.tensor(shapes, dtype=torch.int64, device=dev)
starts = [0]
for h, w in shapes[:-1]: starts.append(starts[-1] + h * w)
ls = torch.tensor(starts, dtype=torch.int64, device=dev)
go = torch.randn(B, nq, H * D, device=dev, generator=g)
def run(sh):
    out = ops.ms_deform_attn(value, sh, shapes_t, ls, loc, attn)
    out.backward(go)
    gv = value.grad.clone(); value.grad = None; loc.grad = None; attn.grad = None
    return gv
def t(sh, n=10):
    for _ in range(3): run(sh)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): run(sh)
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
g1 = run(shapes); g2 = run(None)
print('max diff lds vs atomic', (g1 - g2).abs().max().item(), g2.abs().max().item())
print('fwd+bwd ms  lds-slab:', t(shapes), ' atomics:', t(None))
