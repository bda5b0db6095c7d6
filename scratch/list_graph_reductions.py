"""ATen reductions issued while the graph step is CAPTURED (bench workload), with elements reduced per output: the ones that
reduce many elements per output run as several workgroups meeting through a memset-cleared semaphore — which replays with stale
results inside a HIP graph on this stack (scratch/dbg_graph_reduce.py).   usage: python scratch/list_graph_reductions.py [bf16|fp32]"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from mask_bev_amd import synthetic, tuning
from mask_bev_amd.graph import GraphedTrainStep
from mask_bev_amd.mask_bev_module import MaskBevModule
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda', 0)
tuning.use_tuned_gemms(None)
torch.manual_seed(420)
m = MaskBevModule(**synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype=dtype)).to(dev).train()
m.log_scalars = False
m.flatten_parameters()
opt = m.configure_optimizers()['optimizer']
batch = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
RED = ('sum', 'amax', 'amin', 'max', 'min', 'mean', 'norm', 'prod', 'any', 'all', 'argmax', 'argmin', 'var', 'std', 'logsumexp')
seen = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split('.')[0]
        if name in RED and torch.cuda.is_current_stream_capturing():
            t = next((a for a in args if torch.is_tensor(a)), None)
            o = out[0] if isinstance(out, (tuple, list)) else out
            if t is not None and torch.is_tensor(o) and t.is_cuda and o.numel() > 0:
                per = t.numel() // max(1, o.numel())
                where = '?'
                for f in reversed(traceback.extract_stack(limit=14)):
                    if 'mask_bev_amd' in f.filename and 'torch' not in f.filename:
                        where = f'{os.path.basename(f.filename)}:{f.lineno}'
                        break
                seen[(str(func), tuple(t.shape), tuple(o.shape), per, where)] += 1
        return out


with Spy():
    g = GraphedTrainStep(m, opt, batch)
print(f'{dtype}: ATen reductions inside the captured graphs, by elements reduced per output')
for (f, si, so, per, where), n in sorted(seen.items(), key=lambda kv: -kv[0][3]):
    print(f'  {per:9d} per output  x{n}  {f:28s} {str(si):28s} -> {str(so):18s} @ {where}')
