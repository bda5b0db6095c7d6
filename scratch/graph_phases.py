import sys, torch, time
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
from mask_bev_amd.graph import GraphedTrainStep
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
import os
if os.environ.get("ARENA","1")=="1": m.flatten_parameters()
opt = m.configure_optimizers()['optimizer']
pool = [synthetic.make_batch('semantic_kitti_512', 4, 0, s, dev) for s in range(2)]
g = GraphedTrainStep(m, opt, pool[0])
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
acc = {}; host = {}
for it in range(12):
    scans, (labels, masks) = pool[it % 2]
    torch.cuda.synchronize(); t = [ev()]; h = [time.perf_counter()]
    with m._autocast(): x = m._encoder(scans)
    t.append(ev()); h.append(time.perf_counter())
    g.x_static.data.copy_(x.detach()); g.labels.copy_(labels); g.masks.copy_(masks)
    t.append(ev()); h.append(time.perf_counter())
    g.graph.replay()
    t.append(ev()); h.append(time.perf_counter())
    x.backward(g.x_static.grad)
    t.append(ev()); h.append(time.perf_counter())
    opt.step()
    for p in m._encoder.parameters(): p.grad = None
    t.append(ev()); h.append(time.perf_counter())
    torch.cuda.synchronize()
    if it >= 4:
        for i, name in enumerate(['encoder_fwd', 'copies', 'graph_replay', 'encoder_bwd', 'optimizer']):
            acc[name] = acc.get(name, 0) + t[i].elapsed_time(t[i + 1]) / 8
            host[name] = host.get(name, 0) + (h[i + 1] - h[i]) * 1e3 / 8
print('GPU ms ', {k: round(v, 2) for k, v in acc.items()}, 'total', round(sum(acc.values()), 2))
print('host ms', {k: round(v, 2) for k, v in host.items()}, 'total', round(sum(host.values()), 2))
