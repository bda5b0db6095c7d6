import time, torch, sys
sys.path.insert(0,'.')
from oracle import maskbev_oracle as O
from mask_bev_amd import synthetic
torch.set_num_threads(6)
kw = synthetic.module_kwargs('semantic_kitti_512', 1)
cfg = O.make_cfg(**kw)
sd = O.make_state_dict(cfg, 0)
scans, (labels, masks) = synthetic.make_batch('semantic_kitti_512', 1, 0, 0, torch.device('cpu'))
t=time.perf_counter()
with torch.no_grad():
    x, parts = O.encoder_forward(cfg, sd, scans, True, return_parts=True)
t1=time.perf_counter(); print('encoder fwd', t1-t, 'V', parts['coors'].shape[0], flush=True)
with torch.no_grad():
    f = O.swin_forward(cfg, sd, x)
t2=time.perf_counter(); print('swin fwd', t2-t1, flush=True)
with torch.no_grad():
    cls, mk, _ = O.head_forward(cfg, sd, f)
t3=time.perf_counter(); print('head fwd', t3-t2, flush=True)
with torch.no_grad():
    ld = O.loss_dict(cfg, cls, mk, labels, masks, O.PointSource(0))
t4=time.perf_counter(); print('loss fwd', t4-t3, flush=True)
