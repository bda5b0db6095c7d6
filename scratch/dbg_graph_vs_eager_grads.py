"""Per-PARAMETER comparison of the graph step's gradients with the eager step's on the bench workload (same parameters, same batch;
the loss draws fresh sampling points, so a healthy parameter agrees to a few percent — a stale / garbage gradient does not).
Round 6: ATen multi-block reductions were found to replay wrongly inside captured graphs on this stack (scratch/dbg_graph_reduce.py).
usage: python scratch/dbg_graph_vs_eager_grads.py [bf16|fp32] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import synthetic, tuning, switches
from mask_bev_amd.graph import GraphedTrainStep
from mask_bev_amd.mask_bev_module import MaskBevModule
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for kv in sys.argv[3:]:
    k, _, v = kv.partition('=')
    switches.set_value(k, v)
dev = torch.device('cuda', 0)
tuning.use_tuned_gemms(None)
torch.manual_seed(420)
m = MaskBevModule(**synthetic.module_kwargs('semantic_kitti_512', B, compute_dtype=dtype)).to(dev).train()
m.log_scalars = False
arena = m.flatten_parameters()
data = [synthetic.make_batch('semantic_kitti_512', B, 0, s, dev) for s in range(3)]


class NoOpt:
    grad_scale = 1.0
    zero_grad_in_step = True                 # (the caller clears the arena gradient itself, after reading it)
    def step(self): pass


def eager(batch, n=2):
    acc = None
    for _ in range(n):                       # average over n evaluations (fresh sampling points each)
        arena.zero_grad()
        loss = m.training_step(batch, 0)
        loss.backward()
        acc = arena.grad.clone() if acc is None else acc + arena.grad
    arena.zero_grad()
    return acc / n


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ge = [eager(data[i]) for i in (1, 2)]
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = GraphedTrainStep(m, NoOpt(), data[0])
res = []
for i in (1, 2, 1, 2):
    arena.zero_grad()
    g.step(data[i])
    torch.cuda.synchronize()
    res.append(arena.grad.clone())
names = {id(p): n for n, p in m.named_parameters()}
rows = []
for p, o in arena.layout:
    n = p.numel()
    for k, (gi, ei) in enumerate(((0, 0), (1, 1), (2, 0), (3, 1))):
        a, b = res[gi][o:o + n].double(), ge[ei][o:o + n].double()
        den = (a.norm() * b.norm()).clamp(min=1e-30)
        cos = float((a * b).sum() / den)
        rel = float((a - b).norm() / b.norm().clamp(min=1e-30))
        rows.append((cos, rel, k, names[id(p)], n, float(b.norm())))
rows.sort()
print('worst per-parameter agreement (cos, rel l2, replay#, name, numel, |eager|):')
for r in rows[:25]:
    print(f'  cos {r[0]:7.4f} rel {r[1]:8.3f} replay {r[2]} {r[3][-70:]:70s} n={r[4]} |g|={r[5]:.3e}')
bad = [r for r in rows if r[0] < 0.9 and r[5] > 1e-12]
print('parameters x replays with cos < 0.9:', len(bad), 'of', len(rows))
