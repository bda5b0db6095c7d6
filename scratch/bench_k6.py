"""K6 decoder attention at the bench shapes (B 4, Q 100, 8 heads x 32): masked cross-attention over L = 256 / 1024 / 4096
memory tokens (shared-KV layout: row stride 768) and the 100 x 100 self-attention; forward / backward launch time of
the C entry points from HIP-graph replays of 20 calls."""
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
B, Q, H, D = 4, 100, 8, 32
E = H * D
for L, ld, masked in [(256, 768, True), (1024, 768, True), (4096, 768, True), (100, 256, False)]:
    q = torch.randn(B, Q, E, device=dev).to(dt)
    kc = torch.randn(B, L, ld, device=dev).to(dt)
    vc = torch.randn(B, L, ld, device=dev).to(dt)
    mask = (torch.rand(B, Q, L, device=dev) < 0.6).to(torch.uint8) if masked else None
    if mask is not None:
        mask[:, :, 0] = 0
    out = torch.empty(B, Q, E, device=dev, dtype=dt)
    go = torch.randn(B, Q, E, device=dev).to(dt)
    lse = torch.empty(B, H, Q, device=dev)
    wsb = lib.mbv_attn_workspace_bytes(B, Q, L, H, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    gq = torch.empty(B, Q, E, device=dev)
    lo = dt != torch.float32
    gk = torch.empty(B, L, ld, device=dev, dtype=dt if lo else torch.float32)
    gv = torch.empty_like(gk)
    s = torch.cuda.Stream()
    st = ctypes.c_void_p(s.cuda_stream)
    mp = P(mask) if mask is not None else ctypes.c_void_p(0)

    def fwd():
        rc = lib.mbv_attn_fwd_ld(P(q), P(kc), P(vc), ld, mp, flag, B, Q, L, H, D, P(out), P(lse), P(ws), wsb, st)
        assert rc == 0, rc

    def bwd():
        rc = lib.mbv_attn_bwd_ld(P(q), P(kc), P(vc), ld, mp, P(out), P(go), P(lse), flag, B, Q, L, H, D, P(gq), P(gk), P(gv),
                                 ld, flag if lo else 0, st)
        assert rc == 0, rc
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
    print(f'{name} L {L:5d} ld {ld}: fwd (split + combine) {ts[0]:6.1f} us   bwd {ts[1]:6.1f} us')
