"""Prototype timing: few-row f32 Linear on exact-f32 MFMA from global memory vs torch / hipBLASLt (graph-timed)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'small_linear.so')
if not os.path.exists(so):
    subprocess.check_call(['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', os.path.join(here, 'small_linear.hip'), '-o', so])
lib = ctypes.CDLL(so)
dev = torch.device('cuda', 0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def gtime(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn(s)
        g.replay(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(5): g.replay()
        b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / 100
for (M, K, N, ks) in [(400, 256, 256, 1), (400, 256, 2048, 1), (400, 2048, 256, 1), (400, 2048, 256, 4), (400, 256, 768, 1), (400, 256, 512, 1), (400, 256, 2, 1), (4000, 256, 256, 1)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    def mine(s):
        if ks > 1: y.copy_(b.expand(M, N))
        assert lib.small_nt(P(x), P(w), None if ks > 1 else P(b), P(y), M, N, K, 0, ks, ctypes.c_void_p(s.cuda_stream)) == 0
    t_mine = gtime(mine)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    err = float((y.double() - ref).abs().max())
    t_lib = gtime(lambda s: torch.nn.functional.linear(x, w, b))
    print(f'M {M:5d} K {K:5d} N {N:5d} ksplit {ks}: mine {t_mine:6.1f} us  torch {t_lib:6.1f} us  max err {err:.2e}')
