"""Metrics path of the training / validation step, GPU-native.

Reference: ``MaskBevPanopticHead.update_mAP_metrics`` (mask_bev/models/head/mask_bev_panoptic_head.py:34-96), called
per decoder layer and step from ``MaskBevModule.training_step`` (mask_bev_module.py:273-275), with the metric objects
of mask_bev/evaluation/detection_metric.py.  There it re-runs the Hungarian matcher per sample, upsamples all Q masks
to ground-truth resolution, thresholds and hands dense tensors to torchmetrics.  Here:

* the assignment is the one the loss just computed (``Mask2FormerHead.last_assignment``, K9) — no second matching;
* the mask IoU of the matched pairs is computed by K15 straight from the low-resolution logits and the bit-packed
  ground truth (``matched_mask_iou``);
* ``MeanIoU`` / ``BinaryClassifScores`` mirror the reference's metric objects without torchmetrics (states are device
  tensors; ``compute`` is the only synchronisation).

torchmetrics' COCO-style ``MeanAveragePrecision`` over masks is not reproduced (torchmetrics is not available in this
image and is outside the forward/backward path).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib, ops
from ._lib import MaskBevHipError, check

_EPS = 1e-12          # average_precision.py:7


@torch.no_grad()
def matched_mask_iou(pred_logits: torch.Tensor, assignment: torch.Tensor, gt: 'ops.PackedMasks') -> torch.Tensor:
    """``pred_logits`` (B, Q, h, w) mask logits of one decoder output, ``assignment`` (B, Q) int32 (query → ground-
    truth slot of its image, −1 = unmatched), ``gt``: the B*G bit-packed ground-truth masks at (ny, nx).
    → IoU (B, Q) f32 at ground-truth resolution: bilinear upsampling (align_corners=False), sigmoid > 0.5,
    ``inter / (union + 1e-12)`` (mask_bev_panoptic_head.py:74-85, average_precision.py:78-81)."""
    lib = _lib.load()
    if not pred_logits.is_cuda:
        raise MaskBevHipError('matched_mask_iou needs ROCm device tensors (no CPU fallback)')
    b, q, h, w = pred_logits.shape
    logits = pred_logits.float().contiguous()
    g = gt.words.shape[0] // b
    dev = logits.device
    assignment = assignment.to(torch.int32)
    img = torch.arange(b, device=dev, dtype=torch.int32).view(b, 1)
    gt_row = torch.where(assignment >= 0, img * g + assignment, torch.full_like(assignment, -1)).contiguous()
    pred_row = torch.arange(b * q, device=dev, dtype=torch.int32)
    inter = torch.empty(b * q, dtype=torch.int32, device=dev)
    union = torch.empty(b * q, dtype=torch.int32, device=dev)
    check(lib.mbv_matched_mask_iou(logits.data_ptr(), pred_row.data_ptr(), gt.words.data_ptr(), gt_row.data_ptr(),
                                   b * q, h, w, gt.h, gt.w, inter.data_ptr(), union.data_ptr(),
                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
          'mbv_matched_mask_iou')
    return (inter.float() / (union.float() + _EPS)).view(b, q)


class MeanIoU:
    """detection_metric.py:77-92 — mean of all IoUs seen since ``reset``."""

    def __init__(self):
        self.ious: List[torch.Tensor] = []

    def update(self, ious: torch.Tensor):
        self.ious.append(ious.detach().flatten())

    def compute(self):
        if not self.ious:
            return 0.0
        return torch.cat(self.ious).mean()

    def reset(self):
        self.ious = []


class BinaryClassifScores:
    """State of detection_metric.py:10-31 (scores of the evaluated class and the matched targets); the 11-threshold
    average precision of ``compute`` is the binned binary AP of torchmetrics ``binary_average_precision``."""

    def __init__(self, thresholds: int = 11):
        self.y_score: List[torch.Tensor] = []
        self.y_true: List[torch.Tensor] = []
        self.thresholds = thresholds

    def update(self, y_score: torch.Tensor, y_true: torch.Tensor):
        self.y_score.append(y_score.detach().flatten())
        self.y_true.append(y_true.detach().flatten())

    def compute(self):
        if not self.y_score:
            return 0.0
        s, t = torch.cat(self.y_score).float(), torch.cat(self.y_true) > 0
        th = torch.linspace(0, 1, self.thresholds, device=s.device)
        pred = s.view(1, -1) >= th.view(-1, 1)                                   # (T, N)
        tp = (pred & t.view(1, -1)).sum(1).float()
        fp = (pred & ~t.view(1, -1)).sum(1).float()
        fn = (~pred & t.view(1, -1)).sum(1).float()
        precision = torch.where(tp + fp > 0, tp / (tp + fp), torch.ones_like(tp))
        recall = torch.where(tp + fn > 0, tp / (tp + fn), torch.zeros_like(tp))
        precision = torch.cat([precision, precision.new_ones(1)])
        recall = torch.cat([recall, recall.new_zeros(1)])
        return -torch.sum((recall[1:] - recall[:-1]) * precision[:-1])

    def reset(self):
        self.y_score, self.y_true = [], []


@torch.no_grad()
def update_metrics(head, layer_index: int, pred_cls, pred_masks, labels_gt: torch.Tensor, masks_gt,
                   cls_metric: Optional[BinaryClassifScores], miou_metric: Optional[MeanIoU]):
    """``MaskBevPanopticHead.update_mAP_metrics`` for the classification and mIoU metrics, batched over the images and
    reusing the assignment of the loss that was just evaluated on the same predictions (``head`` is the
    ``Mask2FormerHead``; call after ``compute_loss``).  ``pred_cls`` / ``pred_masks``: the per-layer output lists."""
    if getattr(head, 'last_assignment', None) is None:
        raise MaskBevHipError('update_metrics: evaluate the loss first (it provides the assignment)')
    assigned = head.last_assignment[layer_index]                                 # (B, Q)
    cls = pred_cls[layer_index]
    b, q = cls.shape[:2]
    safe = assigned.clamp(min=0).long()
    labels = torch.where(assigned >= 0, torch.gather(labels_gt, 1, safe), torch.full_like(safe, head.num_classes))
    if cls_metric is not None:
        cls_metric.update(cls.float().softmax(-1)[..., 0], labels)               # evaluated_class = 0 (:66-71)
    if miou_metric is not None:
        gt = head.last_gt_packed
        if gt is None:
            gt = masks_gt if isinstance(masks_gt, ops.PackedMasks) else ops.pack_binary_masks(
                masks_gt.float().flatten(0, 1))
        miou_metric.update(matched_mask_iou(pred_masks[layer_index], assigned, gt))
