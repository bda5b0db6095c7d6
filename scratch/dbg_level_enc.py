import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import synthetic, switches
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda', 0)
for wl, B, dt in (('smoke_96', 2, 'bf16'), ('smoke_96', 2, 'fp32'), ('semantic_kitti_512', 2, 'bf16'), ('semantic_kitti_512', 2, 'fp32')):
    for arena in (False, True):
        torch.manual_seed(0)
        m = MaskBevModule(**synthetic.module_kwargs(wl, B, compute_dtype=dt)).to(dev).train()
        m.log_scalars = False
        if arena:
            m.flatten_parameters()
        batch = synthetic.make_batch(wl, B, 0, 0, dev)
        loss = m.training_step(batch, 0)
        loss.backward()
        pd = m._panoptic_head._panoptic_head.pixel_decoder
        h = m._panoptic_head._panoptic_head
        print(wl, dt, 'arena' if arena else 'plain', 'level_encoding', float(pd.level_encoding.weight.grad.abs().sum()) if pd.level_encoding.weight.grad is not None else None,
              'level_embed', float(h.level_embed.weight.grad.abs().sum()), 'query_embed', float(h.query_embed.weight.grad.abs().sum()))
