"""Where K4's backward spends a workgroup's time: experiment builds of the K4 translation units alone
(scratch/_k4var/build.sh NAME -DK4X_...) timed on the four Swin stages of the bench workload (bf16, shifted).
python scratch/bench_k4_variants.py NAME..."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd import _lib

dev = torch.device('cuda', 0)
dt, flag = torch.bfloat16, 1
P = lambda t: ctypes.c_void_p(t.data_ptr())
here = os.path.dirname(os.path.abspath(__file__))
for name in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(here, '_k4var', f'k4_{name}.so'))
    for fn in ('mbv_window_attn_fwd', 'mbv_window_attn_bwd', 'mbv_window_attn_lse_elems'):
        res, args = _lib.SIGNATURES[fn]
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    line = []
    for (hw, c, heads, nblocks) in [(128, 192, 3, 2), (64, 384, 6, 2), (32, 768, 12, 6), (16, 1536, 24, 2)]:
        B, ws = 4, 10
        shift = ws // 2
        qkv = torch.randn(B, hw, hw, 3 * c, device=dev).to(dt)
        bias = torch.randn(3 * c, device=dev)
        table = torch.randn((2 * ws - 1) ** 2, heads, device=dev)
        go = torch.randn(B, hw, hw, c, device=dev).to(dt)
        out = torch.empty(B, hw, hw, c, device=dev, dtype=dt)
        lse = torch.empty(lib.mbv_window_attn_lse_elems(B, hw, hw, heads, ws), device=dev)
        g_qkv, g_table, g_bias = torch.empty_like(qkv), torch.empty_like(table), torch.empty_like(bias)
        s = torch.cuda.Stream()
        st = ctypes.c_void_p(s.cuda_stream)

        def fwd():
            assert lib.mbv_window_attn_fwd(P(qkv), P(bias), P(table), flag, B, hw, hw, c, heads, ws, shift, P(out), P(lse), st) == 0

        def bwd():
            assert lib.mbv_window_attn_bwd(P(qkv), P(bias), P(table), P(out), P(go), P(lse), flag, B, hw, hw, c, heads, ws,
                                           shift, P(g_qkv), P(g_table), P(g_bias), 1, 0, st) == 0
        with torch.cuda.stream(s):
            fwd(); bwd()
            torch.cuda.synchronize()
            ts = []
            for fn in (fwd, bwd):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(20):
                        fn()
                g.replay()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s)
                for _ in range(5):
                    g.replay()
                b.record(s)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / 100)
        line.append(f'{hw}^2: fwd {ts[0]:6.1f} bwd {ts[1]:6.1f}')
    print(f'{name:12s} ' + ' | '.join(line), flush=True)
