"""gpurun helper: wall time against device time of the eager encoder forward (K1 -> host read -> K2 -> K3) of the bench step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
torch.manual_seed(0)
m = MaskBevModule(**kw).to(dev).train(); m.flatten_parameters()
scans, _ = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
patch = m._patch_handoff()
with torch.no_grad(), m._autocast():
    x = m._encoder(scans, patch=patch)
out = torch.zeros_like(x.rows if hasattr(x, 'rows') else x)
for phase in ('with grad',):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with m._autocast():
            y = m._encoder(scans, patch=patch, out=out)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, e0.elapsed_time(e1)))
    ts = ts[5:]
    print(phase, 'host enqueue ms %.3f  wall to completion ms %.3f  device span ms %.3f' % tuple(sum(t[i] for t in ts) / len(ts) for i in range(3)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    with m._autocast():
        y = m._encoder(scans, patch=patch, out=out)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
