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

* the ``map_metric`` slot (torchmetrics ``MeanAveragePrecision(iou_type='segm')`` in the reference, which hands the
  dense masks to pycocotools at ``compute``) is :class:`MaskMeanAveragePrecision`: the (Q, G) IoU matrix of every image
  is formed on the GPU at ``update`` (bilinear upsampling, threshold, two {0,1} GEMMs) and only that matrix, the
  scores, labels and areas are kept; ``compute`` runs the COCO protocol on them.
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


class MaskMeanAveragePrecision:
    """COCO-protocol mask mAP with torchmetrics' defaults (IoU thresholds 0.50:0.05:0.95, 101 recall thresholds, max
    detections 1 / 10 / 100, area ranges all / small / medium / large, classes = labels present) — the object the
    reference puts in the ``map_metric`` slot (mask_bev_module.py:85-94) and feeds at mask_bev_panoptic_head.py:87-96
    with ``scores = softmax[:, 0]``, ``labels = argmax``, all Q upsampled binary masks and ALL ground-truth slots (the
    zero-mask padding included: class-0 ground truths of area 0).  ``compute()`` returns the twelve COCO numbers under
    torchmetrics' names (map, map_50, map_75, map_small/medium/large, mar_1/10/100, mar_small/medium/large).
    The greedy matching of COCOeval.evaluateImg is vectorised over the IoU thresholds; checked against the plain-loop
    restatement in oracle/metrics_oracle.py (parity with torchmetrics itself is unpinned: it is not installed)."""

    IOU_THRS = torch.linspace(0.5, 0.95, 10, dtype=torch.float64)
    REC_THRS = torch.linspace(0.0, 1.0, 101, dtype=torch.float64)
    MAX_DETS = (1, 10, 100)
    AREAS = (('all', 0.0, 1e10), ('small', 0.0, 32.0 ** 2), ('medium', 32.0 ** 2, 96.0 ** 2), ('large', 96.0 ** 2, 1e10))

    def __init__(self):
        self.images: List[dict] = []

    def reset(self):
        self.images = []

    @torch.no_grad()
    def update(self, pred_logits: torch.Tensor, scores: torch.Tensor, pred_labels: torch.Tensor,
               gt_masks: torch.Tensor, gt_labels: torch.Tensor):
        """One batch: ``pred_logits`` (B, Q, h, w) mask logits, ``scores`` / ``pred_labels`` (B, Q), ``gt_masks``
        (B, G, ny, nx) {0, 1}, ``gt_labels`` (B, G).  Pairwise IoU at ground-truth resolution on the device."""
        b, q = scores.shape
        gt = gt_masks.to(torch.bfloat16).flatten(2)                                         # {0,1}: exact in bf16
        for i in range(b):
            up = torch.nn.functional.interpolate(pred_logits[i].float().unsqueeze(1), gt_masks.shape[-2:],
                                                 mode='bilinear', align_corners=False).squeeze(1)
            pm = (up > 0).to(torch.bfloat16).flatten(1)                                     # sigmoid(x) > 0.5
            inter = torch.mm(pm, gt[i].t(), out_dtype=torch.float32) if pm.is_cuda else (pm.float() @ gt[i].float().t())
            pa, ga = pm.float().sum(1), gt[i].float().sum(1)
            union = pa.view(-1, 1) + ga.view(1, -1) - inter
            iou = torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(inter))
            self.images.append(dict(ious=iou.double().cpu(), scores=scores[i].double().cpu(),
                                    pred_labels=pred_labels[i].long().cpu(), pred_areas=pa.double().cpu(),
                                    gt_labels=gt_labels[i].long().cpu(), gt_areas=ga.double().cpu()))

    @classmethod
    def _evaluate_image(cls, iou, sc, d_area, g_area, lo, hi, max_det):
        """COCOeval.evaluateImg for one (image, class, area range), all IoU thresholds at once."""
        g_ig = ~((g_area >= lo) & (g_area <= hi))
        gtind = torch.sort(g_ig.to(torch.int8), stable=True).indices
        dtind = torch.sort(-sc, stable=True).indices[:max_det]
        g_ig = g_ig[gtind]
        d, g, t = len(dtind), len(gtind), len(cls.IOU_THRS)
        iou = iou[dtind][:, gtind] if d and g else torch.zeros((d, g), dtype=torch.float64)
        taken = torch.zeros((t, g), dtype=torch.bool)
        dtm = torch.zeros((t, d), dtype=torch.bool)
        dt_ig = torch.zeros((t, d), dtype=torch.bool)
        thr = torch.clamp(cls.IOU_THRS, max=1 - 1e-10)
        rev = torch.arange(g - 1, -1, -1)
        for k in range(d):
            if g == 0:
                break
            ok = (~taken) & (iou[k].view(1, g) >= thr.view(t, 1))                          # (T, G) candidates
            best = torch.full((t,), -1, dtype=torch.long)
            for ig in (False, True):                      # a non-ignored ground truth wins over any ignored one
                cand = ok & (g_ig.view(1, g) == ig)
                val = torch.where(cand, iou[k].view(1, g).expand(t, g), torch.full((t, g), -1.0, dtype=torch.float64))
                # among equal IoUs the LAST ground truth in the sorted order wins (the loop's `>=` update)
                arg = rev[torch.argmax(val[:, rev], dim=1)]
                hit = (val.gather(1, arg.view(t, 1)).view(t) >= 0) & (best < 0)
                best = torch.where(hit, arg, best)
            m = best >= 0
            if m.any():
                rows = torch.nonzero(m).flatten()
                taken[rows, best[rows]] = True
                dtm[rows, k] = True
                dt_ig[rows, k] = g_ig[best[rows]]
        out_rng = ~((d_area[dtind] >= lo) & (d_area[dtind] <= hi))
        dt_ig = dt_ig | (~dtm & out_rng.view(1, d))
        return sc[dtind], dtm, dt_ig, g_ig

    def compute(self) -> dict:
        imgs = self.images
        classes = sorted({int(c) for im in imgs for c in im['pred_labels'].tolist() + im['gt_labels'].tolist()})
        t, r, k_n, a_n, m_n = len(self.IOU_THRS), len(self.REC_THRS), len(classes), len(self.AREAS), len(self.MAX_DETS)
        precision = -torch.ones((t, r, k_n, a_n, m_n), dtype=torch.float64)
        recall = -torch.ones((t, k_n, a_n, m_n), dtype=torch.float64)
        eps = torch.finfo(torch.float64).eps
        for k, c in enumerate(classes):
            for a, (_, lo, hi) in enumerate(self.AREAS):
                per_img = []
                for im in imgs:
                    di = torch.nonzero(im['pred_labels'] == c).flatten()
                    gi = torch.nonzero(im['gt_labels'] == c).flatten()
                    if len(di) == 0 and len(gi) == 0:
                        continue
                    iou = im['ious'][di][:, gi]
                    per_img.append(self._evaluate_image(iou, im['scores'][di], im['pred_areas'][di], im['gt_areas'][gi],
                                                        lo, hi, self.MAX_DETS[-1]))
                if not per_img:
                    continue
                for m, max_det in enumerate(self.MAX_DETS):
                    scores = torch.cat([e[0][:max_det] for e in per_img])
                    inds = torch.sort(-scores, stable=True).indices
                    dtm = torch.cat([e[1][:, :max_det] for e in per_img], 1)[:, inds]
                    dtig = torch.cat([e[2][:, :max_det] for e in per_img], 1)[:, inds]
                    npig = int(sum(int((~e[3]).sum()) for e in per_img))
                    if npig == 0:
                        continue
                    tp = torch.cumsum((dtm & ~dtig).double(), 1)
                    fp = torch.cumsum((~dtm & ~dtig).double(), 1)
                    nd = tp.shape[1]
                    rc = tp / npig
                    pr = tp / (fp + tp + eps)
                    recall[:, k, a, m] = rc[:, -1] if nd else 0.0
                    if nd:
                        pr = torch.flip(torch.cummax(torch.flip(pr, (1,)), 1).values, (1,))     # monotone envelope
                        idx = torch.searchsorted(rc.contiguous(), self.REC_THRS.view(1, r).expand(t, r).contiguous(),
                                                 right=False)
                        q = torch.where(idx < nd, pr.gather(1, idx.clamp(max=nd - 1)), torch.zeros((t, r), dtype=torch.float64))
                    else:
                        q = torch.zeros((t, r), dtype=torch.float64)
                    precision[:, :, k, a, m] = q

        def ap(thr=None, area=0):
            s = precision[:, :, :, area, m_n - 1]
            if thr is not None:
                s = s[torch.isclose(self.IOU_THRS, torch.tensor(thr, dtype=torch.float64))]
            s = s[s > -1]
            return float(s.mean()) if s.numel() else -1.0

        def ar(mi, area=0):
            s = recall[:, :, area, mi]
            s = s[s > -1]
            return float(s.mean()) if s.numel() else -1.0

        return dict(map=ap(), map_50=ap(0.5), map_75=ap(0.75), map_small=ap(area=1), map_medium=ap(area=2),
                    map_large=ap(area=3), mar_1=ar(0), mar_10=ar(1), mar_100=ar(2), mar_small=ar(2, 1),
                    mar_medium=ar(2, 2), mar_large=ar(2, 3))


@torch.no_grad()
def update_metrics(head, layer_index: int, pred_cls, pred_masks, labels_gt: torch.Tensor, masks_gt,
                   cls_metric: Optional[BinaryClassifScores], miou_metric: Optional[MeanIoU],
                   map_metric: Optional[MaskMeanAveragePrecision] = None):
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
    if map_metric is not None:                            # mask_bev_panoptic_head.py:87-96
        if isinstance(masks_gt, ops.PackedMasks):
            raise MaskBevHipError('the mask-mAP metric needs the dense (B, G, ny, nx) ground-truth masks')
        sm = cls.float().softmax(-1)
        map_metric.update(pred_masks[layer_index], sm[..., 0], cls.argmax(-1), masks_gt, labels_gt)
