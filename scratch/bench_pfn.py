"""K2: PillarFeatureNet forward + backward on the bench batch (4 x 120k points).
usage (GPU box): python scratch/bench_pfn.py   (kernel times: bash scratch/prof_pfn_cmd.sh)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import synthetic, switches, ops
from mask_bev_amd.mask_bev_module import MaskBevModule

dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4)
torch.manual_seed(0)
m = MaskBevModule(**kw).to(dev).train()
enc = m._encoder
scans, _ = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
pil = enc._voxel_layer.pillars(scans, prefilter=True)
print('rows', pil.num_rows, 'pillars', pil.num_pillars)
go = torch.randn(pil.num_pillars, 128, device=dev)
res = {}
for mode in (False, True):                          # library f32 GEMMs, then K2c
    switches.set_value('pfn_skinny', mode)
    for p in enc._voxel_encoder.parameters():
        p.grad = None
    out = enc._voxel_encoder(pil)
    out.backward(go)
    torch.cuda.synchronize()
    res[mode] = (out.detach().clone(), [p.grad.clone() for p in enc._voxel_encoder.parameters()])
    ts = []
    for _ in range(5):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = enc._voxel_encoder(pil)
        e1.record()
        out.backward(go)
        e2.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    print('run', int(mode), 'fwd ms', min(t[0] for t in ts), 'bwd ms', min(t[1] for t in ts))
a, b = res[True], res[False]
print('out rel', float((a[0] - b[0]).abs().max() / b[0].abs().max()))
for (n, _), x, y in zip(enc._voxel_encoder.named_parameters(), a[1], b[1]):
    print(n, float((x - y).norm() / y.norm().clamp(min=1e-12)))
