"""Golden vectors for the F2 readers: small synthetic SemanticKITTI files (`poses.txt`, `calib.txt`, one `.bin`, one
`.label`) parsed by the reference's OWN reader methods (mask_bev/datasets/semantic_kitti/semantic_kitti_dataset.py:
336-385, called unbound on the unmodified class — the module imports here: numpy + yaml only).  Writes the input files
under tests/golden/semantic_kitti_sample/ and the parsed arrays to tests/golden/readers.npz.  Runs only where
/root/reference exists; the test-suite reads the committed files."""
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
from mask_bev.datasets.semantic_kitti.semantic_kitti_dataset import SemanticKittiDataset  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
d = os.path.join(here, 'semantic_kitti_sample')
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(7)
# poses.txt: 5 lines of 12 numbers (3 x 4, row-major)
poses = rng.normal(size=(5, 12))
np.savetxt(os.path.join(d, 'poses.txt'), poses, fmt='%.9e')
# calib.txt in the KITTI odometry format
with open(os.path.join(d, 'calib.txt'), 'w') as f:
    for k in ('P0', 'P1', 'P2', 'P3', 'Tr'):
        f.write(k + ': ' + ' '.join(f'{v:.9e}' for v in rng.normal(size=12)) + '\n')
# one scan and its labels (semantic in the lower 16 bits, instance in the upper 16)
n = 257
scan = rng.normal(size=(n, 4)).astype(np.float32)
scan.tofile(os.path.join(d, '000000.bin'))
sem = rng.choice([0, 1, 10, 10, 10, 40, 252], size=n).astype(np.uint32)
inst = rng.integers(0, 6, size=n).astype(np.uint32)
(sem | (inst << 16)).astype(np.uint32).tofile(os.path.join(d, '000000.label'))

lut = np.zeros(260, dtype=np.uint32)          # a learning map: raw 10 / 252 -> 1 (car), everything else unlabeled
lut[10] = 1
lut[252] = 1


class _Stub:
    _learning_map_lut = lut


out = dict(
    poses=SemanticKittiDataset._load_poses(None, os.path.join(d, 'poses.txt')),
    scan=SemanticKittiDataset._load_scan(None, os.path.join(d, '000000.bin')),
    learning_map_lut=lut)
sem_l, inst_l = SemanticKittiDataset._load_label(_Stub(), os.path.join(d, '000000.label'))
out['sem'], out['inst'] = sem_l, inst_l
calib = SemanticKittiDataset._load_scan_calib(None, os.path.join(d, 'calib.txt'))
for k in ('p0', 'p1', 'p2', 'p3', 'velo_to_cam'):
    out['calib_' + k] = np.asarray(getattr(calib, k))
np.savez_compressed(os.path.join(here, 'readers.npz'), **out)
print({k: v.shape for k, v in out.items()})
