"""gpurun helper: K2a (decoration of the real points) on the bench batch.  python scratch/bench_decorate.py [lib.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from mask_bev_amd import synthetic, ops
from mask_bev_amd.mask_bev_module import MaskBevModule
from _timeit import timeit
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4)
torch.manual_seed(0)
m = MaskBevModule(**kw).to(dev).train()
enc = m._encoder
scans, _ = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
pil = enc._voxel_layer.pillars(scans, prefilter=True)
ve = enc._voxel_encoder
print('rows', pil.num_rows, 'pillars', pil.num_pillars)
rows, rp = ops.pfn_decorate(pil, ve.voxel_size, ve.point_cloud_range)
torch.cuda.synchronize()
print('checksum rows %.9e |rows| %.9e row_pillar %d' % (rows.double().sum().item(), rows.double().abs().sum().item(), int(rp.sum().item())))
print('decorate %.1f us' % timeit(lambda: ops.pfn_decorate(pil, ve.voxel_size, ve.point_cloud_range)))
