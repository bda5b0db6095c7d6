"""Which framework ops launch the small kernels between two named autograd nodes of one eager training step (torch.profiler)."""
import sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = MaskBevModule(**synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')).to(dev).train()
m.log_scalars = False
m.flatten_parameters()
opt = m.configure_optimizers()['optimizer']
batch = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
def one(i):
    loss = m.training_step(batch, i); m.scale_loss(loss).backward(); opt.step()
one(0); one(1); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one(2); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
evs.sort(key=lambda e: e.time_range.start)
start_key, end_key = sys.argv[1], sys.argv[2]
on = False
for e in evs:
    if start_key in e.name: on = True
    if on and end_key in e.name: break
    if on and (e.name.startswith('aten::') or 'Backward' in e.name or e.name.startswith('autograd')) and e.cpu_parent is not None and ('Backward' in e.cpu_parent.name or 'evaluate_function' in e.cpu_parent.name or e.cpu_parent.cpu_parent is None):
        kern = [k.name[:50] for k in e.kernels][:3]
        print(f'{e.name[:60]:60s} {str(list(e.input_shapes)[:3])[:70] if e.input_shapes else "":70s} {kern}')
