"""ORACLE (test infrastructure, never imported by mask_bev_amd): CPU restatement of the mask-IoU part of the
reference's metrics path.

Follows /root/reference:
  mask_bev/models/head/mask_bev_panoptic_head.py:74-85   upsample (bilinear, align_corners=False) → sigmoid > 0.5 →
                                                          batched_mask_iou(mask_targets, pred)
  mask_bev/evaluation/average_precision.py:78-81         batched_mask_iou = Σ min / (Σ max + 1e-12)
PINNED for batched_mask_iou: tests/golden/mask_iou.npz holds inputs/outputs of the reference's own function
(tests/golden/make_golden_batch.py); the interpolation is torch's own F.interpolate, as in the reference.
"""
import torch
import torch.nn.functional as F

_ESP = 1e-12


def batched_mask_iou(masks1: torch.Tensor, masks2: torch.Tensor) -> torch.Tensor:
    union = torch.maximum(masks1, masks2)
    inter = torch.minimum(masks1, masks2)
    return inter.sum(-1).sum(-1) / (union.sum(-1).sum(-1) + _ESP)


def matched_mask_iou(pred_logits: torch.Tensor, assignment: torch.Tensor, masks_gt: torch.Tensor) -> torch.Tensor:
    """pred_logits (B, Q, h, w), assignment (B, Q) long (−1 unmatched), masks_gt (B, G, ny, nx) {0,1} → IoU (B, Q)."""
    b, q = assignment.shape
    out = torch.zeros(b, q)
    for i in range(b):
        up = F.interpolate(pred_logits[i].float().unsqueeze(1), masks_gt.shape[-2:], mode='bilinear',
                           align_corners=False).squeeze(1)
        pred = (torch.sigmoid(up) > 0.5).float()
        for j in range(q):
            a = int(assignment[i, j])
            if a >= 0:
                out[i, j] = batched_mask_iou(masks_gt[i, a].float()[None], pred[j][None])[0]
    return out
