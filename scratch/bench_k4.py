"""K4 window attention at the four Swin stages of the bench workload (B 4, 512 x 512 BEV, window 10, D 64): launch
time of mbv_window_attn_fwd / _bwd (C ABI, preallocated buffers) from HIP-graph replays of 20 calls."""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd import _lib

lib = _lib.load()
dev = torch.device('cuda', 0)
name = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'f32': torch.float32}[name]
flag = {'f32': 0, 'bf16': 1, 'fp16': 2}[name]
P = lambda t: ctypes.c_void_p(t.data_ptr())
tot_f = tot_b = 0.0
for (hw, c, heads, nblocks) in [(128, 192, 3, 2), (64, 384, 6, 2), (32, 768, 12, 6), (16, 1536, 24, 2)]:
    B, ws = 4, 10
    qkv = torch.randn(B, hw, hw, 3 * c, device=dev).to(dt)
    bias = torch.randn(3 * c, device=dev)
    table = torch.randn((2 * ws - 1) ** 2, heads, device=dev)
    go = torch.randn(B, hw, hw, c, device=dev).to(dt)
    out = torch.empty(B, hw, hw, c, device=dev, dtype=dt)
    lse = torch.empty(lib.mbv_window_attn_lse_elems(B, hw, hw, heads, ws), device=dev)
    g_qkv, g_table, g_bias = torch.empty_like(qkv), torch.empty_like(table), torch.empty_like(bias)
    res = {}
    for shift in (0, ws // 2):
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
            graphs = []
            for fn in (fwd, bwd):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(20):
                        fn()
                graphs.append(g)
            torch.cuda.synchronize()
            ts = []
            for g in graphs:
                g.replay()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s)
                for _ in range(5):
                    g.replay()
                b.record(s)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / 100)
        res[shift] = ts
    tot_f += (res[0][0] + res[ws // 2][0]) / 2 * nblocks
    tot_b += (res[0][1] + res[ws // 2][1]) / 2 * nblocks
    print(f'{hw:4d}^2 C {c:5d} heads {heads:3d}: fwd {res[0][0]:6.1f} / {res[ws // 2][0]:6.1f} us   bwd (+2 fills) {res[0][1]:6.1f} / {res[ws // 2][1]:6.1f} us   (unshifted / shifted)')
print(f'{name} per step (12 blocks): fwd {tot_f:.0f} us, bwd {tot_b:.0f} us')
