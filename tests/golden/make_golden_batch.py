#!/usr/bin/env python3
"""Golden vectors for the batch-producer row: instance-id map -> (labels, masks).

Runs only in the build container.  The reference's own
    mask_bev/datasets/semantic_kitti/semantic_kitti_transforms.py   (FilterSmallMasks :11-26,
                                                                     MaskToLabelInstanceMasks :66-81)
is imported UNMODIFIED from /root/reference; its import chain needs ``cv2`` (not installed, not used by these two
classes), for which an empty stand-in module is registered.  Inputs and outputs are committed as
tests/golden/instance_masks.npz.

    python tests/golden/make_golden_batch.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
sys.path.insert(0, '/root/reference')
from mask_bev.datasets.semantic_kitti.semantic_kitti_mask_dataset import SemanticKittiMaskScan  # noqa: E402
from mask_bev.datasets.semantic_kitti.semantic_kitti_transforms import (FilterSmallMasks,  # noqa: E402
                                                                         MaskToLabelInstanceMasks)


def make_iou_golden():
    """mask_bev/evaluation/average_precision.py:78-81 run unmodified (its module imports cv2: stand-in above)."""
    from mask_bev.evaluation.average_precision import batched_mask_iou
    g = torch.Generator().manual_seed(5)
    m1 = (torch.rand(6, 40, 36, generator=g) > 0.6).float()
    m2 = torch.rand(6, 40, 36, generator=g) > 0.5
    m1[2] = 0
    m2[3] = False
    m1[4] = 0
    m2[4] = False
    np.savez_compressed(os.path.join(HERE, 'mask_iou.npz'), masks1=m1.numpy().astype(np.uint8),
                        masks2=m2.numpy().astype(np.uint8), iou=batched_mask_iou(m1, m2).numpy())
    print('wrote mask_iou.npz')


def make_map(rng, nx, ny, ids, sizes):
    m = np.zeros((nx, ny), dtype=np.int64)
    for inst, (sx, sy) in zip(ids, sizes):
        x0 = rng.integers(0, nx - sx + 1)
        y0 = rng.integers(0, ny - sy + 1)
        m[x0:x0 + sx, y0:y0 + sy] = inst                      # later instances overwrite earlier ones
    return m


def main():
    rng = np.random.default_rng(7)
    out = {}
    cases = [  # (name, nx, ny, ids, sizes, num_queries, min_pixels)
        ('a', 48, 40, [3, 17, 65540, 9, 131077], [(6, 5), (3, 3), (10, 4), (2, 2), (7, 7)], 8, 5),
        ('b', 64, 64, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], [(5, 9)] * 12, 16, 1),
        ('c', 32, 48, [], [], 4, 3),                                       # empty scene
        ('d', 40, 40, [70000, 5], [(40, 40), (4, 4)], 3, 20),              # one instance fills the map, one too small
    ]
    for name, nx, ny, ids, sizes, q, minpix in cases:
        m = make_map(rng, nx, ny, ids, sizes)
        s = FilterSmallMasks(minpix)(SemanticKittiMaskScan(scan=None, mask=m.copy()))
        labels, masks = MaskToLabelInstanceMasks(q)(torch.from_numpy(s.mask))
        # the reference enumerates `set(mask.unique().numpy()) - {0}` in set order; record the order it used
        order = [int(m_i) for m_i in (set(torch.from_numpy(s.mask).T.unique().numpy()) - {0})]
        out[f'{name}_map'] = m
        out[f'{name}_cfg'] = np.array([q, minpix], dtype=np.int64)
        out[f'{name}_labels'] = labels.numpy()
        out[f'{name}_masks'] = masks.numpy().astype(np.uint8)
        out[f'{name}_order'] = np.array(order, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'instance_masks.npz'), **out)
    print('wrote', os.path.join(HERE, 'instance_masks.npz'), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
    make_iou_golden()
