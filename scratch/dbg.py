import os, sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic, ops
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
scans, (labels, masks) = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
torch.cuda.synchronize(); print('built', flush=True)
cls, mk, _ = m(scans); torch.cuda.synchronize(); print('fwd ok', mk[0].dtype, mk[0].shape, flush=True)
head = m._panoptic_head._panoptic_head
from mask_bev_amd.mask2former_head import PointSource
pts = PointSource(dev, None)
a = head._assign(cls[0], mk[0], labels, masks, pts); torch.cuda.synchronize(); print('assign ok', a.shape, flush=True)
l = head._loss_single(cls[0], mk[0], labels, masks, pts); torch.cuda.synchronize(); print('loss ok', [float(x) for x in l], flush=True)
sum(l).backward(retain_graph=True); torch.cuda.synchronize(); print('bwd ok', flush=True)
