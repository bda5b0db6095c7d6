"""K14 batch producer: instance-id map → (labels, masks) on the device vs the oracle (which is pinned against the
reference's own transforms), f32 and bit-packed forms, and the loss fed with the packed targets."""
import os

import numpy as np
import pytest
import torch

from oracle import batch_oracle as BO
from tests.util_cfg import random_scans, tiny_kwargs

pytestmark = pytest.mark.gpu


def _check(device, maps, q, minpix):
    from mask_bev_amd import batch, ops
    t = torch.from_numpy(np.stack(maps)).to(device)
    labels, masks = batch.instance_targets(t, q, minpix, check_overflow=True)
    lp, packed = batch.instance_targets(t, q, minpix, packed=True)
    assert torch.equal(labels, lp)
    want_words = ops.pack_binary_masks(masks.flatten(0, 1)).words
    assert torch.equal(packed.words, want_words)
    for b, m in enumerate(maps):
        ol, om, _ = BO.instance_targets(m, q, minpix)
        assert np.array_equal(labels[b].cpu().numpy(), ol)
        assert np.array_equal(masks[b].cpu().numpy(), om)


def test_golden_maps(device):
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'instance_masks.npz'))
    for name in ('a', 'b', 'c', 'd'):
        q, minpix = (int(v) for v in z[f'{name}_cfg'])
        _check(device, [z[f'{name}_map']], q, minpix)


def test_bev_sized_batch(device):
    rng = np.random.default_rng(3)
    maps = []
    for s in range(3):
        m = np.zeros((512, 512), dtype=np.int64)
        for k in range(37 + s):
            inst = int(rng.integers(1, 2 ** 20))
            x0, y0 = rng.integers(0, 480, 2)
            m[x0:x0 + rng.integers(2, 30), y0:y0 + rng.integers(2, 30)] = inst
        maps.append(m)
    _check(device, maps, 100, 12)


def test_more_instances_than_queries_raises(device):
    from mask_bev_amd import batch
    m = np.zeros((1, 16, 16), dtype=np.int64)
    for k in range(6):
        m[0, k * 2, :4] = k + 1
    with pytest.raises(IndexError):
        batch.instance_targets(torch.from_numpy(m).to(device), 4, 1, check_overflow=True)


def test_loss_with_packed_targets_equals_dense(device):
    from mask_bev_amd import batch
    from mask_bev_amd.mask_bev_module import MaskBevModule
    torch.manual_seed(0)
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    m = MaskBevModule(**kw).to(device).eval()
    head = m._panoptic_head._panoptic_head
    head.num_points = 500
    head.point_seed = 11
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    rng = np.random.default_rng(0)
    maps = np.zeros((2, 96, 96), dtype=np.int64)
    for b in range(2):
        for k in range(4):
            x0, y0 = rng.integers(0, 70, 2)
            maps[b, x0:x0 + 12, y0:y0 + 9] = 100 + k
    t = torch.from_numpy(maps).to(device)
    labels, dense = batch.instance_targets(t, 8, 3)
    _, packed = batch.instance_targets(t, 8, 3, packed=True)
    with torch.no_grad():
        cls, masks, h = m(scans)
        l0 = m.loss(m.compute_loss(cls, masks, labels, dense, h, None))
        l1 = m.loss(m.compute_loss(cls, masks, labels, packed, h, None))
    assert torch.allclose(l0, l1, rtol=1e-6, atol=1e-6)
