"""Synthetic SemanticKITTI-shaped inputs for bench.py / smoke tests.  No dataset is read.

Seeded ``420 + 1000 * rank + step`` (420 = the ``seed`` of the shipped SemanticKITTI config,
/root/reference: configs/training/semantic_kitti/01_point_mask_data_aug_gentle.yml:2).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

# named workloads of BASELINE.json `configs` (grid sizes verified in SURVEY.md §0)
WORKLOADS: Dict[str, Dict] = {
    # S2 / S3: SemanticKITTI-shaped, 512x512 BEV, 100 queries, 120k points
    'semantic_kitti_512': dict(x_range=(-40, 40), y_range=(-40, 40), z_range=(-3, 1), voxel_size=0.15625,
                               num_queries=100, points=120000, pc_point_dim=4),
    # S4: KITTI-shaped, 0.16 m pillars, 496x432 BEV, 200 queries
    'kitti_496x432': dict(x_range=(0, 69.12), y_range=(-39.68, 39.68), z_range=(-3, 1), voxel_size=0.16,
                          num_queries=200, points=120000, pc_point_dim=4),
    # S5: Waymo-scale, 1024x1024 BEV, 300 queries, 180k xyz-only points
    'waymo_1024': dict(x_range=(-40, 40), y_range=(-40, 40), z_range=(-3, 1), voxel_size=0.078125,
                       num_queries=300, points=180000, pc_point_dim=3),
    # tiny plumbing case for smoke()
    'smoke_96': dict(x_range=(-12, 12), y_range=(-12, 12), z_range=(-3, 1), voxel_size=0.25,
                     num_queries=8, points=6000, pc_point_dim=4),
}


def module_kwargs(workload: str, batch_size: int, compute_dtype: str = 'fp32', **overrides) -> Dict:
    """Constructor kwargs of MaskBevModule for a named workload; the remaining hyper-parameters are those of
    the shipped SemanticKITTI YAML (configs/training/semantic_kitti/01_point_mask_data_aug_gentle.yml:6-29)."""
    w = WORKLOADS[workload]
    kw = dict(x_range=w['x_range'], y_range=w['y_range'], z_range=w['z_range'], voxel_size=w['voxel_size'],
              num_queries=w['num_queries'], max_num_points=32, encoder_feat_channels=[128, 128, 128],
              backbone_embed_dim=192, head_feat_channels=256, head_out_channels=256, optimiser_type='adam_w',
              lr=1e-4, weight_decay=1e-4, lr_schedulers_type='plateau', differential_lr=False,
              differential_lr_scaling=1.0, backbone_window_size=10, pc_point_dim=w['pc_point_dim'],
              batch_size=batch_size, seed=420, compute_dtype=compute_dtype)
    if workload == 'smoke_96':
        kw.update(encoder_feat_channels=[32, 32, 32], backbone_embed_dim=48, head_feat_channels=128,
                  head_out_channels=128, backbone_window_size=6, max_num_points=8)
    kw.update(overrides)
    return kw


def lidar_scan(n_points: int, pc_dim: int, gen: torch.Generator, device, max_range: float = 52.0) -> torch.Tensor:
    """LiDAR-shaped scan: 64 beams (elevation -24.8°..+2°) x azimuths, sensor at 1.73 m over a flat ground,
    random 'walls' for the beams that do not hit the ground, range noise sigma = 0.02 m, intensity U(0,1),
    randomly permuted (mirrors ShufflePointCloud, semantic_kitti_transforms.py:58-61)."""
    beams = 64
    az = n_points // beams
    n = beams * az
    elev = torch.linspace(math.radians(-24.8), math.radians(2.0), beams, device=device).view(beams, 1)
    theta = (torch.arange(az, device=device, dtype=torch.float32) / az * 2 * math.pi).view(1, az)
    # piecewise-constant wall distance per azimuth sector
    sectors = 90
    wall = 6.0 + torch.rand(sectors, generator=gen, device=device) * (max_range - 6.0)
    wall_r = wall[(theta / (2 * math.pi) * sectors).long().clamp(max=sectors - 1)].expand(beams, az)
    ground_r = torch.where(elev < -1e-3, 1.73 / torch.tan(-elev).clamp(min=1e-3), torch.full_like(elev, 1e9))
    r = torch.minimum(ground_r.expand(beams, az), wall_r)
    r = r + torch.randn(beams, az, generator=gen, device=device) * 0.02
    x = r * torch.cos(elev) * torch.cos(theta)
    y = r * torch.cos(elev) * torch.sin(theta)
    z = r * torch.sin(elev)                      # sensor frame: ground at z = -1.73
    cols = [x.reshape(-1), y.reshape(-1), z.reshape(-1)]
    for _ in range(pc_dim - 3):
        cols.append(torch.rand(n, generator=gen, device=device))
    pts = torch.stack(cols, 1)
    if n < n_points:                             # top up with uniform points to the requested count
        extra = torch.rand(n_points - n, pc_dim, generator=gen, device=device)
        extra[:, :2] = extra[:, :2] * 2 * max_range - max_range
        extra[:, 2] = extra[:, 2] * 4 - 3
        pts = torch.cat([pts, extra], 0)
    return pts[torch.randperm(pts.shape[0], generator=gen, device=device)].contiguous()


def gt_masks(batch: int, num_queries: int, ny: int, nx: int, gen: torch.Generator, device,
             k_range: Tuple[int, int] = (5, 40), cell: float = 0.15625) -> Tuple[torch.Tensor, torch.Tensor]:
    """K ~ U{5..40} car-sized boxes rasterised to {0,1} masks, padded to Q with zero masks / label 0
    (the dataset quirk of semantic_kitti_transforms.py:77-81: label 1 = object)."""
    labels = torch.zeros(batch, num_queries, dtype=torch.long, device=device)
    masks = torch.zeros(batch, num_queries, ny, nx, dtype=torch.float32, device=device)
    ys = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
    xs = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
    for b in range(batch):
        k = int(torch.randint(k_range[0], min(k_range[1], num_queries) + 1, (1,), generator=gen, device=device))
        cx = torch.rand(k, generator=gen, device=device) * nx
        cy = torch.rand(k, generator=gen, device=device) * ny
        ang = torch.rand(k, generator=gen, device=device) * math.pi
        hl = (4.5 / 2) / cell
        hw = (1.8 / 2) / cell
        dx = xs - cx.view(k, 1, 1)
        dy = ys - cy.view(k, 1, 1)
        u = dx * torch.cos(ang).view(k, 1, 1) + dy * torch.sin(ang).view(k, 1, 1)
        v = -dx * torch.sin(ang).view(k, 1, 1) + dy * torch.cos(ang).view(k, 1, 1)
        masks[b, :k] = ((u.abs() <= hl) & (v.abs() <= hw)).float()
        labels[b, :k] = 1
    return labels, masks


def uniform_scan(n_points: int, pc_dim: int, gen: torch.Generator, device, x_range, y_range) -> torch.Tensor:
    """x, y ~ U(range), z ~ U(-3, 1), remaining channels U(0, 1): the worst case for the pillar count
    (≈ 96 k non-empty pillars per 120 k-point scan on the 512 x 512 grid; SURVEY.md §8d distribution (ii))."""
    pts = torch.rand(n_points, pc_dim, generator=gen, device=device)
    pts[:, 0] = pts[:, 0] * (x_range[1] - x_range[0]) + x_range[0]
    pts[:, 1] = pts[:, 1] * (y_range[1] - y_range[0]) + y_range[0]
    pts[:, 2] = pts[:, 2] * 4.0 - 3.0
    return pts.contiguous()


def make_batch(workload: str, batch: int, rank: int, step: int, device, distribution: str = 'lidar'):
    """One training batch in the reference's batch contract:
    (list of (Ni, pc_dim) tensors, (labels (B, Q) int64, masks (B, Q, ny, nx) f32)).
    ``distribution``: 'lidar' (64-beam scan over a flat ground, the headline input) or 'uniform'."""
    w = WORKLOADS[workload]
    gen = torch.Generator(device=device).manual_seed(420 + 1000 * rank + step)
    nx = int((w['x_range'][1] - w['x_range'][0]) / w['voxel_size'])
    ny = int((w['y_range'][1] - w['y_range'][0]) / w['voxel_size'])
    scans = []
    for _ in range(batch):
        if distribution == 'uniform':
            s = uniform_scan(w['points'], w['pc_point_dim'], gen, device, w['x_range'], w['y_range'])
        elif distribution == 'lidar':
            s = lidar_scan(w['points'], w['pc_point_dim'], gen, device)
            if w['x_range'][0] >= 0:             # forward-facing range (KITTI): fold the scan into x >= 0
                s[:, 0] = s[:, 0].abs()
        else:
            raise ValueError(f'unknown point distribution {distribution!r}')
        scans.append(s)
    labels, masks = gt_masks(batch, w['num_queries'], ny, nx, gen, device, cell=w['voxel_size'])
    return scans, (labels, masks)
