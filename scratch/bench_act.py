import torch, time
from mask_bev_amd import ops
dev = torch.device('cuda')
for rows, n in ((65536, 768), (16384, 1536), (4096, 3072), (21504, 1024)):
    z = torch.randn(rows, n, device=dev).bfloat16().requires_grad_()
    bias = torch.nn.Parameter(torch.zeros(n, device=dev)); bias.grad = torch.zeros(n, device=dev)
    go = torch.randn(rows, n, device=dev).bfloat16()
    kind = 'gelu' if n != 1024 else 'relu'
    y = ops.bias_act(z, bias, kind)
    for _ in range(3): y.backward(go, retain_graph=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): y.backward(go, retain_graph=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(rows, n, kind, f'{dt*1e6:.1f} us  {rows*n*6/dt/1e12:.2f} TB/s')
