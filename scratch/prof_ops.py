import sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
opt = m.configure_optimizers()['optimizer']
pool = [synthetic.make_batch('semantic_kitti_512', 4, 0, s, dev) for s in range(2)]
def step(i):
    loss = m.training_step(pool[i % 2], i); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3); torch.cuda.synchronize()
evs = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::copy_', 'aten::sum', 'aten::convolution_backward', 'aten::miopen_convolution', 'aten::add', 'aten::mul')]
evs.sort(key=lambda e: -e.self_device_time_total)
for e in evs[:40]:
    print(f'{e.key:28s} {e.self_device_time_total/1e3:8.2f} ms  n={e.count:4d}  {str(e.input_shapes)[:120]}')
