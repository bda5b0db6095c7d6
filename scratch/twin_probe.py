#!/usr/bin/env python3
"""Round 6, VERDICT r05 #1 — upper bound of the "two micro-batch branches" idea before building it.

Three graph steps in one process: one at B = 4 (the bench step) and two independent ones at B = 2 (own modules, own
arenas).  Only the two captured graphs of each are replayed (no encoder, no optimizer), N times:
  t4          B = 4, one stream
  t2          one B = 2 step, one stream
  t2+2 serial both B = 2 steps back to back on one stream
  t2|2        both B = 2 steps on two streams (what two branches of one graph could reach at best)
usage: python scratch/twin_probe.py [bf16|fp32] [replays]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import synthetic, tuning                       # noqa: E402
from mask_bev_amd.graph import GraphedTrainStep                  # noqa: E402
from mask_bev_amd.mask_bev_module import MaskBevModule           # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
WL = 'semantic_kitti_512'
dev = torch.device('cuda', 0)
tuning.use_tuned_gemms(None)


def build(batch, seed):
    torch.manual_seed(420)
    m = MaskBevModule(**synthetic.module_kwargs(WL, batch, compute_dtype=dtype)).to(dev).train()
    m.log_scalars = False
    m.flatten_parameters()
    opt = m.configure_optimizers()['optimizer']
    b = synthetic.make_batch(WL, batch, 0, seed, dev)
    g = GraphedTrainStep(m, opt, b)
    for _ in range(3):
        g.step(b)
    torch.cuda.synchronize()
    return m, g


def replay(g, stream=None):
    g.graph.replay()
    g.graph_late.replay()


def timed(fn, n=N):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


m4, g4 = build(4, 0)
ma, ga = build(2, 1)
mb, gb = build(2, 2)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def both_parallel():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        replay(ga)
    with torch.cuda.stream(s2):
        replay(gb)
    main.wait_stream(s1)
    main.wait_stream(s2)


def clear():
    for m in (m4, ma, mb):
        m._arena.zero_grad()


res = {}
for name, fn in (('t4', lambda: replay(g4)), ('t2', lambda: replay(ga)),
                 ('t2+2 serial', lambda: (replay(ga), replay(gb))), ('t2|2 streams', both_parallel)):
    clear()
    res[name] = timed(fn)
    print(f'{name:14s} {res[name]:8.3f} ms', flush=True)
clear()
print(f'{dtype}: graphs only — B=4 {res["t4"]:.2f} ms; two B=2 on two streams {res["t2|2 streams"]:.2f} ms '
      f'({100 * (1 - res["t2|2 streams"] / res["t4"]):+.1f} % saved); serial {res["t2+2 serial"]:.2f}')
