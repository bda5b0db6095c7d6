"""MIOpen's choice for the pixel decoder's 3 x 3 convolution (4, 256, 128, 128) with and without its find (benchmark) mode,
f32 and bf16: forward and backward times (graph-timed)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scratch._timeit import timeit
dev = torch.device('cuda', 0)
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for dt in (torch.float32, torch.bfloat16):
        conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev).to(dt)
        x = torch.randn(4, 256, 128, 128, device=dev, dtype=dt).requires_grad_()
        g = torch.randn(4, 256, 128, 128, device=dev, dtype=dt)
        t0 = time.time()
        y = conv(x); y.backward(g); torch.cuda.synchronize()
        first = time.time() - t0
        def fwd():
            with torch.no_grad():
                conv(x)
        def both():
            x.grad = None; conv.weight.grad = None
            conv(x).backward(g)
        tf, tb = timeit(fwd), timeit(both, iters=5)
        print(f'benchmark={bench} {dt}: first call {first:.2f} s, forward {tf:.0f} us, forward + backward {tb:.0f} us', flush=True)
