"""K15 matched mask IoU (metrics path) vs the oracle's F.interpolate → sigmoid > 0.5 → batched_mask_iou chain, and the
metric objects fed from the loss's own assignment."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO
from tests.util_cfg import random_gt, random_scans, tiny_kwargs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,Q,h,w,H,W', [(2, 7, 24, 20, 96, 80), (1, 5, 128, 128, 512, 512), (2, 4, 10, 10, 10, 10),
                                         (1, 3, 17, 13, 50, 41)])
def test_matched_mask_iou(device, B, Q, h, w, H, W):
    from mask_bev_amd import metrics, ops
    g = torch.Generator().manual_seed(B * 100 + h)
    logits = torch.randn(B, Q, h, w, generator=g) * 3
    gt = (torch.rand(B, Q, H, W, generator=g) > 0.7).float()
    gt[0, 1] = 0                                               # an empty (padded) ground-truth mask
    assign = torch.stack([torch.randperm(Q, generator=g) for _ in range(B)])
    assign[0, 0] = -1                                          # an unmatched prediction
    want = MO.matched_mask_iou(logits, assign, gt)
    packed = ops.pack_binary_masks(gt.flatten(0, 1).to(device))
    got = metrics.matched_mask_iou(logits.to(device), assign.to(device), packed).cpu()
    # a pixel whose interpolated logit is within float rounding of 0 may fall on either side: allow 2 px per pair
    px = H * W
    assert torch.allclose(got, want, atol=3.0 / max(1.0, 0.25 * px) + 1e-6), (got - want).abs().max()


def test_metrics_from_loss_assignment(device):
    from mask_bev_amd import metrics
    from mask_bev_amd.mask_bev_module import MaskBevModule
    torch.manual_seed(0)
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    m = MaskBevModule(**kw).to(device).eval()
    head = m._panoptic_head._panoptic_head
    head.num_points = 500
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    labels, gt = labels.to(device), gt.to(device)
    with torch.no_grad():
        cls, masks, hts = m(scans)
        m.compute_loss(cls, masks, labels, gt, hts, None)
    miou, clsm = metrics.MeanIoU(), metrics.BinaryClassifScores()
    for layer in (0, len(cls) - 1):
        metrics.update_metrics(head, layer, cls, masks, labels, gt, clsm, miou)
    v = float(miou.compute())
    assert 0.0 <= v <= 1.0
    assign = head.last_assignment[len(cls) - 1].cpu().long()
    want = MO.matched_mask_iou(masks[-1].float().cpu(), assign, gt.cpu())
    got = metrics.matched_mask_iou(masks[-1], head.last_assignment[len(cls) - 1], head.last_gt_packed).cpu()
    assert torch.allclose(got, want, atol=2e-3)
    ap = float(clsm.compute())
    assert 0.0 <= ap <= 1.0


def test_module_step_feeds_metrics(device):
    from mask_bev_amd.mask_bev_module import MaskBevModule
    torch.manual_seed(0)
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    m = MaskBevModule(**kw).to(device).train()
    m.log_scalars = False
    m._panoptic_head._panoptic_head.num_points = 500
    m.enable_metrics(layers=(0, 9))
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    loss = m.training_step((scans, (labels.to(device), gt.to(device))), 0)
    assert torch.isfinite(loss)
    cls_metric, _, miou = m._train_metric_per_layer[9]
    assert len(miou.ious) == 1 and miou.ious[0].numel() == 2 * 8
    assert 0.0 <= float(miou.compute()) <= 1.0
    m.on_train_epoch_end()
    assert miou.ious == []


def test_module_step_feeds_mask_map_metric(device):
    """The `map_metric` slot (mask_bev_panoptic_head.py:87-96): fed from a training step with device-side pairwise
    IoUs; the stored IoU matrix equals the oracle's dense pairwise IoU of the upsampled, thresholded masks, and
    compute() returns the twelve COCO numbers equal to the oracle's plain-loop COCOeval on the same state."""
    import torch.nn.functional as F
    from mask_bev_amd.mask_bev_module import MaskBevModule
    from oracle import metrics_oracle as MO
    torch.manual_seed(0)
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    m = MaskBevModule(**kw).to(device).train()
    m.log_scalars = False
    m._panoptic_head._panoptic_head.num_points = 500
    m.enable_metrics(layers=(9,), val=False, mask_map=True)
    scans = [x.to(device) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    with torch.no_grad():
        _, masks, _ = m(scans)
    loss = m.training_step((scans, (labels.to(device), gt.to(device))), 0)
    assert torch.isfinite(loss)
    _, map_metric, _ = m._train_metric_per_layer[9]
    assert len(map_metric.images) == 2
    for i, im in enumerate(map_metric.images):
        up = F.interpolate(masks[9][i].float().cpu().unsqueeze(1), gt.shape[-2:], mode='bilinear',
                           align_corners=False).squeeze(1)
        ref = MO.pairwise_mask_iou(torch.sigmoid(up) > 0.5, gt[i] > 0.5)
        # bf16 GEMM inputs can move a logit across 0 only where |logit| is at rounding level: compare loosely there
        assert np.abs(im['ious'].numpy() - ref).max() < 2e-2
    got = map_metric.compute()
    ref = MO.coco_mask_map([{k: v.numpy() for k, v in im.items()} for im in map_metric.images])
    assert set(got) == {'map', 'map_50', 'map_75', 'map_small', 'map_medium', 'map_large', 'mar_1', 'mar_10',
                        'mar_100', 'mar_small', 'mar_medium', 'mar_large'}
    for k in ref:
        assert got[k] == pytest.approx(ref[k], abs=1e-12), k
    m.on_train_epoch_end()
    assert map_metric.images == []
