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


# --------------------------------------------------------------------------------------------------------------
# COCO-protocol mask mAP (the `map_metric` slot of update_mAP_metrics, mask_bev_panoptic_head.py:87-96, which the
# reference fills with torchmetrics.detection.MeanAveragePrecision(iou_type='segm'), mask_bev_module.py:85-94).
# PARITY UNPINNED: torchmetrics / pycocotools are not vendored in /root/reference and not installed here.  This is a
# plain-loop restatement of the published COCOeval algorithm those packages implement (pycocotools cocoeval.py:
# evaluateImg + accumulate + summarize), with torchmetrics' defaults: IoU thresholds 0.50:0.05:0.95, 101 recall
# thresholds, max detections (1, 10, 100), area ranges all / small (< 32^2) / medium / large (>= 96^2), no crowd
# regions, classes = every label present in predictions or targets.
# --------------------------------------------------------------------------------------------------------------
import numpy as np

COCO_IOU_THRS = np.linspace(0.5, 0.95, 10)
COCO_REC_THRS = np.linspace(0.0, 1.0, 101)
COCO_MAX_DETS = (1, 10, 100)
COCO_AREAS = {'all': (0.0, 1e10), 'small': (0.0, 32.0 ** 2), 'medium': (32.0 ** 2, 96.0 ** 2), 'large': (96.0 ** 2, 1e10)}


def pairwise_mask_iou(pred: torch.Tensor, gt: torch.Tensor) -> np.ndarray:
    """pred (Q, H, W), gt (G, H, W) boolean masks → (Q, G) IoU, 0 where the union is empty."""
    p, g = pred.flatten(1).double(), gt.flatten(1).double()
    inter = p @ g.t()
    union = p.sum(1, keepdim=True) + g.sum(1).view(1, -1) - inter
    return torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(inter)).numpy()


def coco_evaluate_image(ious, dt_scores, dt_areas, gt_areas, area_rng, max_det):
    """pycocotools COCOeval.evaluateImg for one (image, category): → dt order, dtm (T, D) matched flags, dtIg (T, D),
    gtIg (G,)."""
    g_ig = np.array([not (area_rng[0] <= a <= area_rng[1]) for a in gt_areas], dtype=bool)
    gtind = np.argsort(g_ig, kind='mergesort')                  # non-ignored ground truths first
    dtind = np.argsort(-np.asarray(dt_scores, dtype=np.float64), kind='mergesort')[:max_det]
    g_ig = g_ig[gtind]
    iou = ious[dtind][:, gtind] if len(gtind) and len(dtind) else np.zeros((len(dtind), len(gtind)))
    T, D, G = len(COCO_IOU_THRS), len(dtind), len(gtind)
    gtm = -np.ones((T, G), dtype=np.int64)
    dtm = -np.ones((T, D), dtype=np.int64)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(COCO_IOU_THRS):
        for d in range(D):
            best = min(t, 1 - 1e-10)
            m = -1
            for g in range(G):
                if gtm[ti, g] >= 0:
                    continue
                if m > -1 and not g_ig[m] and g_ig[g]:
                    break
                if iou[d, g] < best:
                    continue
                best = iou[d, g]
                m = g
            if m == -1:
                continue
            dt_ig[ti, d] = g_ig[m]
            dtm[ti, d] = m
            gtm[ti, m] = d
    out_rng = np.array([not (area_rng[0] <= dt_areas[i] <= area_rng[1]) for i in dtind], dtype=bool)
    dt_ig = dt_ig | ((dtm < 0) & out_rng[None, :])
    return dtind, dtm >= 0, dt_ig, g_ig


def coco_mask_map(images):
    """images: list of dicts with 'ious' (Q, G), 'scores' (Q,), 'pred_labels' (Q,), 'pred_areas' (Q,),
    'gt_labels' (G,), 'gt_areas' (G,).  → dict of the twelve COCO summary numbers (map, map_50, map_75, map_small,
    map_medium, map_large, mar_1, mar_10, mar_100, mar_small, mar_medium, mar_large)."""
    classes = sorted({int(c) for im in images for c in list(im['pred_labels']) + list(im['gt_labels'])})
    T, R, K, A, M = len(COCO_IOU_THRS), len(COCO_REC_THRS), len(classes), len(COCO_AREAS), len(COCO_MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k, c in enumerate(classes):
        for a, rng in enumerate(COCO_AREAS.values()):
            per_img = []
            for im in images:
                di = np.nonzero(np.asarray(im['pred_labels']) == c)[0]
                gi = np.nonzero(np.asarray(im['gt_labels']) == c)[0]
                if len(di) == 0 and len(gi) == 0:
                    continue
                iou = np.asarray(im['ious'])[di][:, gi] if len(di) and len(gi) else np.zeros((len(di), len(gi)))
                sc = np.asarray(im['scores'], dtype=np.float64)[di]
                dtind, dtm, dtig, gig = coco_evaluate_image(iou, sc, np.asarray(im['pred_areas'])[di],
                                                            np.asarray(im['gt_areas'])[gi], rng, COCO_MAX_DETS[-1])
                per_img.append((sc[dtind], dtm, dtig, gig))
            if not per_img:
                continue
            for m, max_det in enumerate(COCO_MAX_DETS):
                scores = np.concatenate([e[0][:max_det] for e in per_img])
                inds = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e[1][:, :max_det] for e in per_img], axis=1)[:, inds]
                dtig = np.concatenate([e[2][:, :max_det] for e in per_img], axis=1)[:, inds]
                npig = int(sum((~e[3]).sum() for e in per_img))
                if npig == 0:
                    continue
                tps = np.cumsum(dtm & ~dtig, axis=1).astype(np.float64)
                fps = np.cumsum(~dtm & ~dtig, axis=1).astype(np.float64)
                for t in range(T):
                    tp, fp = tps[t], fps[t]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = np.zeros(R)
                    idx = np.searchsorted(rc, COCO_REC_THRS, side='left')
                    for ri, pi in enumerate(idx):
                        if pi < nd:
                            q[ri] = pr[pi]
                    precision[t, :, k, a, m] = q

    def ap(thr=None, area='all'):
        s = precision[:, :, :, list(COCO_AREAS).index(area), M - 1]
        if thr is not None:
            s = s[np.isclose(COCO_IOU_THRS, thr)]
        s = s[s > -1]
        return float(s.mean()) if s.size else -1.0

    def ar(max_det, area='all'):
        s = recall[:, :, list(COCO_AREAS).index(area), COCO_MAX_DETS.index(max_det)]
        s = s[s > -1]
        return float(s.mean()) if s.size else -1.0

    return dict(map=ap(), map_50=ap(0.5), map_75=ap(0.75), map_small=ap(area='small'), map_medium=ap(area='medium'),
                map_large=ap(area='large'), mar_1=ar(1), mar_10=ar(10), mar_100=ar(100), mar_small=ar(100, 'small'),
                mar_medium=ar(100, 'medium'), mar_large=ar(100, 'large'))
