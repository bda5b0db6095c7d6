"""The pixel decoder's 3 x 3 convolution (4, 256, 128, 128) f32: MIOpen against the K20 form, forward and forward + backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scratch._timeit import timeit
from mask_bev_amd import ops
dev = torch.device('cuda', 0)
conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
x = torch.randn(4, 256, 128, 128, device=dev).requires_grad_()
g = torch.randn(4, 256, 128, 128, device=dev) * 1e-3
def run(fn, bwd):
    def f():
        if bwd:
            x.grad = None; conv.weight.grad = None
            fn().backward(g)
        else:
            with torch.no_grad():
                fn()
    return timeit(f, iters=5)
for name, fn in (('MIOpen', lambda: conv(x)), ('K20', lambda: ops.conv3x3_32(x, conv.weight))):
    print(f'{name}: forward {run(fn, False):.0f} us, forward + backward {run(fn, True):.0f} us', flush=True)
