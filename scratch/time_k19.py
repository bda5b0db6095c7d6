"""Per-program durations of the fused decoder (K19) inside one eager training step of the bench workload."""
import sys, collections, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic, decoder_fused as DF
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
torch.manual_seed(0)
m = MaskBevModule(**synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype=dtype)).to(dev).train()
m.log_scalars = False
m.flatten_parameters()
opt = m.configure_optimizers()['optimizer']
batch = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
def one(i):
    loss = m.training_step(batch, i); m.scale_loss(loss).backward(); opt.step()
one(0); one(1); torch.cuda.synchronize()
DF.Program.TIMING = []
one(2); torch.cuda.synchronize()
agg = collections.defaultdict(list)
for label, n, a, b in DF.Program.TIMING:
    agg[(label, n)].append(a.elapsed_time(b) * 1e3)
for k, v in sorted(agg.items()):
    print(f'{k[0]:8s} stages={k[1]:3d} launches={len(v):3d} avg={sum(v)/len(v):8.1f} us  min={min(v):8.1f}')
