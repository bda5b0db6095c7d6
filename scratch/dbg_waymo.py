import sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
# dirty the allocator's pools with NaN bit patterns first
junk = [torch.full((n,), float('nan'), device=dev) for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20)]
del junk
for workload in ('kitti_496x432', 'waymo_1024'):
    torch.manual_seed(0)
    kw = synthetic.module_kwargs(workload, 1, compute_dtype='bf16')
    m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
    arena = m.flatten_parameters()
    data = synthetic.make_batch(workload, 1, 0, 0, dev)
    loss = m.training_step(data, 0); loss.backward()
    print(workload, float(loss), bool(torch.isfinite(arena.grad).all()))
    for k, p in m.named_parameters():
        if p.grad is not None and not bool(torch.isfinite(p.grad).all()):
            print('   non-finite:', k, tuple(p.shape), int((~torch.isfinite(p.grad)).sum()))
    del m, arena
