"""Batch producer for the MaskBEV step: what the reference does on the host between the dataset
and ``training_step`` — reading a scan, and turning the cached instance-id map into padded (labels, masks) targets —
with the target construction moved to the GPU (K14, csrc/instance_masks.hip).

Reference: mask_bev/datasets/semantic_kitti/semantic_kitti_dataset.py (``.bin`` / ``.label`` readers),
semantic_kitti_mask_dataset.py:121-137 (``.npy`` mask cache), semantic_kitti_transforms.py:11-26,66-81,98-121
(FilterSmallMasks, MaskToLabelInstanceMasks, MaskListCollate[Height]).

What crosses PCIe per scan is the point cloud (1.9 MB) and the (nx, ny) int32 instance map (1 MB) instead of the
dense (Q, ny, nx) f32 masks (105 MB).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import MaskBevHipError, check

CAR = 1           # SemanticKittiLearningLabel.CAR (semantic_kitti_dataset.py:175): the label of every real instance


# ---------------------------------------------------------------------------------------------------------
# host readers (plain numpy; the formats of the SemanticKITTI distribution)
# ---------------------------------------------------------------------------------------------------------
def read_velodyne_bin(path) -> np.ndarray:
    """``sequences/SS/velodyne/NNNNNN.bin`` → (N, 4) f32 x, y, z, remission."""
    pc = np.fromfile(str(path), dtype=np.float32)
    if pc.size % 4:
        raise ValueError(f'{path}: not a multiple of 4 floats')
    return pc.reshape(-1, 4)


def read_semantic_kitti_label(path) -> Tuple[np.ndarray, np.ndarray]:
    """``sequences/SS/labels/NNNNNN.label`` → (semantic (N,) u32 = lower 16 bits, instance (N,) u32 = upper 16 bits)."""
    raw = np.fromfile(str(path), dtype=np.uint32)
    return raw & 0xFFFF, raw >> 16


def apply_learning_map(sem: np.ndarray, inst: np.ndarray, learning_map_lut: np.ndarray, unlabeled: int = 0):
    """Class remap of semantic_kitti_dataset.py:369-372: ``sem`` through the learning-map look-up table, instance ids of
    points that become UNLABELED cleared."""
    sem = learning_map_lut[sem]
    inst = inst.copy()
    inst[sem == unlabeled] = 0
    return sem, inst


def read_poses(path) -> np.ndarray:
    """``sequences/SS/poses.txt`` → (N, 4, 4) f64 origin-to-scan transforms: each line holds the upper 3 x 4 block,
    the omitted last row is (0, 0, 0, 1) (semantic_kitti_dataset.py:336-349)."""
    reduced = np.loadtxt(str(path), ndmin=2)
    n = reduced.shape[0]
    full = np.zeros((n, 4, 4))
    full[:, :3, :] = reduced.reshape(n, 3, 4)
    full[:, 3, 3] = 1
    return full


def read_calib(path) -> dict:
    """``sequences/SS/calib.txt`` → ``{'p0': (3, 4), ..., 'velo_to_cam': (4, 4)}`` (``Tr`` completed with the row
    (0, 0, 0, 1); the other keys lower-cased — the fields of SemanticKittiCalib, semantic_kitti_dataset.py:374-385)."""
    calib = {}
    with open(str(path), 'r') as f:
        for line in f:
            if ':' not in line:
                continue
            k, v = line.split(':')
            mat = np.array(v.split(), dtype=np.float64).reshape(3, 4)
            if k == 'Tr':
                calib['velo_to_cam'] = np.vstack((mat, [0, 0, 0, 1]))
            else:
                calib[k.lower()] = mat
    return calib


def read_mask_cache(path) -> np.ndarray:
    """The reference's per-scan mask cache (``np.save`` of the (nx, ny) instance map,
    semantic_kitti_mask_dataset.py:121-137)."""
    with open(str(path), 'rb') as f:
        return np.load(f)


# ---------------------------------------------------------------------------------------------------------
# K14: instance map → targets on the device
# ---------------------------------------------------------------------------------------------------------
@torch.no_grad()
def instance_targets(instance_maps: torch.Tensor, num_queries: int, min_num_inst_pixels: int = 0,
                     packed: bool = False, check_overflow: bool = False):
    """``instance_maps`` (B, nx, ny) integer device tensor (0 = background) → ``(labels (B, Q) int64, masks)`` with
    ``masks`` (B, Q, ny, nx) f32 {0, 1} — the reference's batch contract — or, ``packed=True``, an
    :class:`ops.PackedMasks` of the B*Q maps that ``MaskBevModule.compute_loss`` accepts in their place.
    Instances are enumerated in ascending id order; ``check_overflow`` synchronises and raises ``IndexError`` like
    the reference when a scan holds more instances than ``num_queries``."""
    lib = _lib.load()
    if not instance_maps.is_cuda:
        raise MaskBevHipError('instance_targets needs a ROCm device tensor (no CPU fallback)')
    if instance_maps.dim() != 3:
        raise ValueError('instance_maps must be (B, nx, ny)')
    m = instance_maps.to(torch.int32).contiguous()
    b, nx, ny = m.shape
    dev = m.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ids = torch.empty((b, num_queries), dtype=torch.int32, device=dev)
    counts = torch.empty((b,), dtype=torch.int32, device=dev)
    status = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib.mbv_instance_ids(m.data_ptr(), b, nx, ny, num_queries, int(min_num_inst_pixels), ids.data_ptr(),
                               counts.data_ptr(), status.data_ptr(), stream), 'mbv_instance_ids')
    if packed:
        words = torch.empty((b * num_queries, lib.mbv_packed_mask_words(ny, nx)), dtype=torch.int32, device=dev)
        check(lib.mbv_expand_instance_masks(m.data_ptr(), ids.data_ptr(), b, nx, ny, num_queries, None,
                                            words.data_ptr(), stream), 'mbv_expand_instance_masks')
        masks = ops.PackedMasks(words, ny, nx)
        masks.batch_shape = (b, num_queries)
    else:
        masks = torch.empty((b, num_queries, ny, nx), dtype=torch.float32, device=dev)
        check(lib.mbv_expand_instance_masks(m.data_ptr(), ids.data_ptr(), b, nx, ny, num_queries, masks.data_ptr(),
                                            None, stream), 'mbv_expand_instance_masks')
    labels = (torch.arange(num_queries, device=dev).view(1, -1) < counts.view(-1, 1)).to(torch.int64) * CAR
    if check_overflow:
        st = int(status.item())
        if st & 1:
            raise IndexError('a scan has more instances than num_queries '
                             '(semantic_kitti_transforms.py:78-80 indexes past num_pred)')
        if st & 2:
            raise MaskBevHipError('more than 4096 distinct instance ids in one scan')
    return labels, masks


class InstanceMapCollate:
    """Collate of ``(point_cloud (N, pc_dim) f32 array/tensor, instance_map (nx, ny) int array/tensor[, metadata])``
    samples into the batch ``MaskBevModule.training_step`` takes — the reference's ``MaskListCollate[Height]``
    (semantic_kitti_transforms.py:98-121) with the masks built on ``device`` by K14."""

    def __init__(self, num_queries: int, device, min_num_inst_pixels: int = 0, packed: bool = False):
        self.num_queries, self.device = num_queries, torch.device(device)
        self.min_num_inst_pixels, self.packed = min_num_inst_pixels, packed

    def __call__(self, batch: Sequence):
        pcs = [torch.as_tensor(s[0], dtype=torch.float32).to(self.device, non_blocking=True) for s in batch]
        maps = torch.stack([torch.as_tensor(np.asarray(s[1])).to(torch.int32) for s in batch]).to(self.device,
                                                                                                   non_blocking=True)
        labels, masks = instance_targets(maps, self.num_queries, self.min_num_inst_pixels, self.packed)
        if len(batch[0]) > 2:
            return pcs, (labels, masks), [s[2] for s in batch]
        return pcs, (labels, masks)
