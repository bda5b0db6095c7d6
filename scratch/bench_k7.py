"""K7 forward: k_mask_logits_bf16 against K17's NN GEMM on the same product (B = 4, Q = 100, C = 256, 128 x 128 pixels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import ops, _lib
from scratch._timeit import timeit as graph_time_us
lib = _lib.load()
dev = torch.device('cuda:0')
b, q, c, hw = 4, 100, 256, 128 * 128
e = torch.randn(b, q, c, device=dev).bfloat16()
f = torch.randn(b, c, hw, device=dev).bfloat16()
out1 = torch.empty(b, q, hw, device=dev)
out2 = torch.empty(b, q, hw, device=dev)
def k7():
    ops.check(lib.mbv_mask_logits_fwd(ops._ptr(e), ops._ptr(f), 1, b, q, c, hw, ops._ptr(out1), 1, ops._stream()), 'k7')
def k17():
    ops.check(lib.mbv_gemm16_nn(ops._ptr(e), ops._ptr(f), ops._ptr(out2), None, None, q, c, hw, c, hw, hw, 0, 0, 1, 0, b,
                                q * c, c * hw, q * hw, None, 0, ops._stream()), 'k17')
k7(); k17(); torch.cuda.synchronize()
ref = torch.bmm(e.float(), f.float())
print('K7 err', float((out1 - ref).abs().max()), 'K17 err', float((out2 - ref).abs().max()), 'ref max', float(ref.abs().max()))
print('K7  us', graph_time_us(k7))
print('K17 us', graph_time_us(k17))
