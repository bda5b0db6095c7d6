"""gpurun helper: where the eager encoder forward of the bench step spends its wall time — K1 (+ its host read), the PFN, K3:
host time to issue each part, the wall clock when the device finished it, device time between events."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import synthetic, ops
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
torch.manual_seed(0)
m = MaskBevModule(**kw).to(dev).train(); m.flatten_parameters()
enc = m._encoder
scans, _ = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
patch = m._patch_handoff()
with torch.no_grad(), m._autocast():
    x = enc(scans, patch=patch)
out = torch.zeros_like(x.rows if hasattr(x, 'rows') else x)
acc = [0.0] * 6
n = 0
for it in range(25):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter(); ev[0].record()
    pillars = enc._voxel_layer.pillars(scans, prefilter=True)
    t1 = time.perf_counter(); ev[1].record()
    feats = enc._voxel_encoder(pillars)
    t2 = time.perf_counter(); ev[2].record()
    ln = enc._layer_norm
    y = ops.scatter_layernorm(feats, ln.weight, ln.bias, pillars, 4, enc._num_voxel_y, enc._num_voxel_x, ln.eps, patch, out)
    t3 = time.perf_counter(); ev[3].record()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 5:
        n += 1
        for i, v in enumerate(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t0) * 1e3, ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]))):
            acc[i] += v
print('host ms: K1 + its host read %.3f | PFN issue %.3f | K3 issue %.3f | wall to completion %.3f || device ms: PFN span %.3f, K3 span %.3f'
      % tuple(a / n for a in acc))
