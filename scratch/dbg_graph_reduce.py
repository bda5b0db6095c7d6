"""Do ATen multi-block reductions replay correctly inside captured graphs on this stack?  (round 6: the first amax_verify used
`t.abs().max()` and read stale values inside the graph step)
usage: python scratch/dbg_graph_reduce.py MODE     MODE: shared (two graphs, one pool) | separate (two pools) | single (one graph)
                                                         | sum (torch.sum over rows instead of max) """
import sys
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else 'shared'
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N = 12
xs = [torch.randn(2048 * (i + 1), 768, device=dev) for i in range(N)]
out1 = torch.zeros(N, device=dev)
out2 = torch.zeros(N, device=dev)


def red(x):
    if mode == 'sum':
        return x.sum(0)[:1]                    # a column sum (many rows per output: a multi-block reduction)
    return x.abs().max().view(1)


def body1():
    for i in range(N):
        torch.mul(red(xs[i]), 1.0, out=out1[i:i + 1])


def body2():
    for i in range(N):
        torch.mul(red(xs[i] * 2), 1.0, out=out2[i:i + 1])


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body1(); body2()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    body1()
if mode != 'single':
    with torch.cuda.graph(g2, pool=g1.pool() if mode in ('shared', 'sum') else None):
        body2()
bad = 0
for rep in range(5):
    for i in range(N):
        xs[i].normal_()
        xs[i][rep, 0] = 100.0 + 10 * rep + i          # a known maximum / a dominant term of column 0
    g1.replay()
    if mode != 'single':
        g2.replay()
    torch.cuda.synchronize()
    if mode == 'sum':
        want = torch.stack([x.double().sum(0)[0] for x in xs]).float()
        e1 = int(((out1 - want).abs() > 1e-2).sum())
        e2 = int(((out2 - 2 * want).abs() > 2e-2).sum())
    else:
        want = torch.tensor([100.0 + 10 * rep + i for i in range(N)], device=dev)
        e1 = int((out1 != want).sum())
        e2 = int((out2 != 2 * want).sum()) if mode != 'single' else 0
    bad += e1 + e2
    if e1 or e2:
        print(rep, 'graph1 wrong', e1, 'graph2 wrong', e2, [round(v, 1) for v in out1.tolist()], [round(v, 1) for v in want.tolist()])
print(mode, 'BAD' if bad else 'ok')
