import torch, time
from mask_bev_amd import ops
dev = torch.device('cuda')
B, N, H, L, P = 4, 5376, 8, 3, 4
shapes = [(64, 64), (32, 32), (16, 16)]
off = torch.randn(B, N, H, L, P, 2, device=dev).bfloat16().requires_grad_()
lg = torch.randn(B, N, H, L * P, device=dev).bfloat16().requires_grad_()
ref = torch.rand(N, 2, device=dev)
gl = torch.randn(B, N, H, L, P, 2, device=dev); ga = torch.randn(B, N, H, L, P, device=dev)
st = torch.tensor(shapes, device=dev)
def fused():
    loc, aw = ops.msda_prepare(off, lg, ref, shapes)
    torch.autograd.backward([loc, aw], [gl, ga])
def composed():
    aw = lg.float().softmax(-1).view(B, N, H, L, P)
    nrm = torch.stack([st[:, 1], st[:, 0]], -1).to(off.dtype)
    loc = ref.view(1, N, 1, 1, 1, 2) + off / nrm.view(1, 1, 1, L, 1, 2)
    torch.autograd.backward([loc, aw], [gl, ga])
for name, f in (('fused', fused), ('composed', composed)):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 50 * 1e6, 'us fwd+bwd')
