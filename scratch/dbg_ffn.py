import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import layers, ops
from mask_bev_amd.arena import ParameterArena
dev = torch.device('cuda', 0)
for act in ('gelu', 'relu'):
    torch.manual_seed(3)
    c, rows = 192, 9000
    x0 = torch.randn(2, rows // 2, c, device=dev)
    g0 = torch.randn(2, rows // 2, c, device=dev).to(torch.bfloat16)
    res = {}
    for pol in ('auto', '0', 'all'):
        os.environ['MBV_GEMM16'] = pol
        torch.manual_seed(5)
        m = layers.FFN(c, 4 * c, act=act).to(dev)
        arena = ParameterArena([('ffn', m)])
        x = x0.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = m(x, add_identity=False)
        y.backward(g0)
        res[pol] = (y.float(), x.grad.float(), arena.grad.clone(), arena)
    for pol in ('auto', 'all'):
        y, gx, ga, ar = res[pol]
        y0, gx0, ga0, _ = res['0']
        print(act, pol, 'y', ((y - y0).abs().max() / y0.abs().max()).item(), 'gx', ((gx - gx0).abs().max() / gx0.abs().max()).item(),
              'ga', ((ga - ga0).abs().max() / ga0.abs().max()).item())
        for p, o in ar.layout:
            n = p.numel()
            d = (ga[o:o+n] - ga0[o:o+n]).abs().max() / ga0[o:o+n].abs().max()
            print('   ', tuple(p.shape), d.item())
