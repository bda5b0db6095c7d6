"""ORACLE (test infrastructure, never imported by mask_bev_amd): CPU restatement of the reference's target
construction for one scan — the row SURVEY.md §8f-2.

Follows /root/reference:
  mask_bev/datasets/semantic_kitti/semantic_kitti_transforms.py:11-26   FilterSmallMasks
  mask_bev/datasets/semantic_kitti/semantic_kitti_transforms.py:66-81   MaskToLabelInstanceMasks
  mask_bev/datasets/semantic_kitti/semantic_kitti_dataset.py:175        SemanticKittiLearningLabel.CAR = 1
(the shipped data module composes exactly these two, semantic_kitti_mask_data_module.py:91-101; the
LabelMaskToMask2FormerLabel step is commented out there).

PINNED: tests/test_oracle_golden.py checks this file against tests/golden/instance_masks.npz, which
tests/golden/make_golden_batch.py produced by running the reference's own, unmodified classes.
The reference enumerates the instances in the iteration order of a Python ``set``; this restatement uses ascending
ids.  The set of (label, mask) pairs is identical, and the Hungarian matcher makes the loss independent of the order.
"""
import numpy as np

CAR = 1


def filter_small_masks(mask: np.ndarray, min_num_inst_pixels: int) -> np.ndarray:
    """semantic_kitti_transforms.py:19-26 — instances with fewer than ``min_num_inst_pixels`` pixels become 0."""
    mask = mask.copy()
    for inst in np.unique(mask):
        if inst == 0:
            continue
        if np.sum(mask == inst) < min_num_inst_pixels:
            mask[mask == inst] = 0
    return mask


def mask_to_label_instance_masks(mask: np.ndarray, num_pred: int):
    """semantic_kitti_transforms.py:70-81 — ``mask`` (nx, ny) int → labels (num_pred,) int64, masks
    (num_pred, ny, nx) f32; raises IndexError like the reference when there are more instances than ``num_pred``."""
    m = mask.T
    h, w = m.shape
    instances = sorted(set(np.unique(m).tolist()) - {0})
    labels = np.zeros((num_pred,), dtype=np.int64)
    masks = np.zeros((num_pred, h, w), dtype=np.float32)
    for i, inst in enumerate(instances):
        if i >= num_pred:
            raise IndexError('more instances than queries')
        labels[i] = CAR
        masks[i][m == inst] = 1.0
    return labels, masks, instances


def instance_targets(mask: np.ndarray, num_pred: int, min_num_inst_pixels: int):
    return mask_to_label_instance_masks(filter_small_masks(mask, min_num_inst_pixels), num_pred)
