"""Which aten ops produce the step's kernels (eager step under torch.profiler, grouped by op + input shapes)."""
import torch, collections, sys
from torch.profiler import profile, ProfilerActivity
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda')
torch.manual_seed(420)
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
m.flatten_parameters()
opt = m.configure_optimizers()['optimizer']
batch = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
def step():
    loss = m.training_step(batch, 0); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and e.count > 0:
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total self device time ms', tot / 1e3)
filt = sys.argv[1].split(',') if len(sys.argv) > 1 else None
rows=[r for r in rows if r[2].startswith('aten::') and (filt is None or r[2][6:] in filt)]
rows.sort(key=(lambda r:-r[0]) if (len(sys.argv) > 3 and sys.argv[3] == 'time') else (lambda r:-r[1]))
print('aten ops by count')
for t, n, k, sh in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f'{t/1e3:7.2f} ms n={n:4d} {k[:34]:34s} {sh}')
