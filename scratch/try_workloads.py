import sys, torch, time
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda')
for wl, b in [('kitti_496x432', 2), ('waymo_1024', 1)]:
    try:
        torch.manual_seed(0)
        kw = synthetic.module_kwargs(wl, b, compute_dtype='bf16')
        m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
        m.flatten_parameters(); opt = m.configure_optimizers()['optimizer']
        batch = synthetic.make_batch(wl, b, 0, 0, dev)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss = m.training_step(batch, it); loss.backward(); opt.step(); opt.zero_grad()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(wl, 'B', b, 'loss', float(loss), 'step ms', round(dt * 1e3, 1), 'mem GB', round(torch.cuda.max_memory_allocated() / 2**30, 1))
        del m, opt, batch; torch.cuda.empty_cache()
    except Exception as e:
        import traceback; traceback.print_exc(limit=3); print(wl, 'FAILED', type(e).__name__, str(e)[:300])
