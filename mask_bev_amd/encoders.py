"""Point-cloud → pseudo-image encoder of MaskBEV on gfx950.

Interface and checkpoint keys of ``MaskBevEncoder``
(/root/reference: mask_bev/models/encoders/mask_bev_encoders.py:21-123).  The hot path is
``forward``: K1 voxelise → K2 pillar feature net over REAL points only → K3 scatter fused with the
(C, ny, nx) LayerNorm; neither the zero-padded (V, 32, ·) tensors nor the dense canvas are built.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import Linear
from .ops import Pillars, VoxelGeometry


class EncodingType:
    Vanilla = 'vanilla'
    Fourier = 'fourier'
    Cosine = 'cosine'


class Voxelization(nn.Module):
    """Hard voxelisation layer with the constructor of ``mmcv.ops.Voxelization`` (mask_bev_encoders.py:69).
    ``forward(points)`` handles one scan and returns (voxels, coors(z,y,x), num_points) like mmcv."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.max_num_points = int(max_num_points)
        # mmcv: an int serves both modes, a (train, test) pair is picked by ``self.training``
        mv = tuple(max_voxels) if isinstance(max_voxels, (tuple, list)) else (max_voxels, max_voxels)
        self.max_voxels_train_test = (int(mv[0]), int(mv[1]))
        self.geometry = VoxelGeometry.from_ranges(self.point_cloud_range, self.voxel_size)

    @property
    def max_voxels(self) -> int:
        return self.max_voxels_train_test[0 if self.training else 1]

    def pillars(self, point_clouds: Sequence[torch.Tensor], prefilter: bool) -> Pillars:
        return ops.voxelize(point_clouds, self.geometry, self.max_num_points, self.max_voxels, prefilter)

    def forward(self, points: torch.Tensor):
        p = self.pillars([points], prefilter=False)
        return ops.gather_voxels(p), p.coors[:, 1:].contiguous(), p.num_points


class _PFNLayer(nn.Module):
    """Keys ``linear.weight`` / ``norm.*`` (mmdet3d ``PFNLayer``)."""

    def __init__(self, cin: int, cout: int, last: bool):
        super().__init__()
        self.last_vfe = last
        self.units = cout if last else cout // 2
        self.linear = Linear(cin, self.units, bias=False)
        self.norm = nn.BatchNorm1d(self.units, eps=1e-3, momentum=0.01)


class PillarFeatureNet(nn.Module):
    """PillarFeatureNet(with_distance=True, legacy=True) evaluated on real points only.

    mmdet3d runs Linear → BatchNorm1d → ReLU → max on the zero-padded (V, P, ·) tensor, so padded rows
    take part in the batch statistics and — being relu(β − γμ/σ) ≠ 0 after BN — in the max
    (SURVEY.md §7 "Padded-row algebra").  Here each pillar carries ONE representative padded row with
    multiplicity P − n: identical results, ≈ n/P of the work and memory.
    """

    def __init__(self, in_channels=4, feat_channels=(64,), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=None, mode='max', legacy=True):
        super().__init__()
        if not (with_distance and with_cluster_center and with_voxel_center and legacy and mode == 'max'):
            raise NotImplementedError('only the MaskBEV configuration of PillarFeatureNet '
                                      '(with_distance, cluster + voxel centre, legacy, max) is built')
        self.in_channels = in_channels
        chans = [in_channels + 7] + list(feat_channels)
        self.pfn_layers = nn.ModuleList([
            _PFNLayer(chans[i], chans[i + 1], last=(i == len(chans) - 2)) for i in range(len(chans) - 1)])
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def forward(self, p: Pillars) -> torch.Tensor:
        rows, row_pillar = ops.pfn_decorate(p, self.voxel_size, self.point_cloud_range)     # K2a: (K, D+7)
        return self.forward_rows(rows, p, row_pillar)

    def forward_rows(self, rows: torch.Tensor, p: Pillars, row_pillar: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The PFN layers on already decorated compact rows (K, in_channels + 7)."""
        layers = [(l.linear.weight, l.norm.weight, l.norm.bias, l.norm.running_mean, l.norm.running_var, l.norm.eps,
                   l.norm.momentum) for l in self.pfn_layers]
        out = ops.pfn_layers(rows, p, layers, self.training, row_pillar)                    # K2b
        if self.training:
            with torch.no_grad():
                torch._foreach_add_([l.norm.num_batches_tracked for l in self.pfn_layers], 1)      # one launch
        return out


class LearnableFourierPositionalEncoding(nn.Module):
    """Per-point learnable Fourier features — same constructor, parameter names (``Wr``, ``mlp.0``, ``mlp.2``) and
    arithmetic as /root/reference: mask_bev/models/positional_encoding/learnable_fourier_positional_encoding.py:6-59
    (x (N, G, M) → (N, D)); evaluated on the real points only (see :meth:`MaskBevEncoder._fourier_rows`)."""

    def __init__(self, G: int, M: int, F_dim: int, H_dim: int, D: int, gamma: float):
        super().__init__()
        self.G, self.M, self.F_dim, self.H_dim, self.D, self.gamma = G, M, F_dim, H_dim, D, gamma
        self.Wr = nn.Linear(M, F_dim // 2, bias=False)
        self.mlp = nn.Sequential(nn.Linear(F_dim, H_dim, bias=True), nn.GELU(), nn.Linear(H_dim, D // G))
        nn.init.normal_(self.Wr.weight.data, mean=0, std=gamma ** -2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n = x.shape[0]
        projected = self.Wr(x)
        f = torch.cat([torch.cos(projected), torch.sin(projected)], dim=-1) / (self.F_dim ** 0.5)
        return self.mlp(f).reshape(n, self.D)


class PointPillarsScatter(nn.Module):
    """Dense canvas scatter (API parity with mmdet3d ``PointPillarsScatter``, mask_bev_encoders.py:74).
    Not used by ``MaskBevEncoder.forward`` — K3 fuses it with the LayerNorm."""

    def __init__(self, in_channels: int, output_shape):
        super().__init__()
        self.in_channels = in_channels
        self.ny, self.nx = int(output_shape[0]), int(output_shape[1])

    def forward(self, voxel_features, coors, batch_size=None):
        if batch_size is None:
            batch_size = int(coors[:, 0].max().item()) + 1
        ops._need_gpu(voxel_features)
        canvas = voxel_features.new_zeros(batch_size, self.ny * self.nx, self.in_channels)
        idx = (coors[:, 2] * self.nx + coors[:, 3]).long()
        canvas[coors[:, 0].long(), idx] = voxel_features
        return canvas.transpose(1, 2).reshape(batch_size, self.in_channels, self.ny, self.nx)


class MaskBevEncoder(nn.Module):
    """Same constructor as the reference class (mask_bev_encoders.py:22-26)."""

    def __init__(self, feat_channels: List[int], x_range, y_range, z_range, voxel_size_x: float, voxel_size_y: float,
                 voxel_size_z: float, max_num_points: int, encoding_type: str, fourier_enc_group: int,
                 max_voxels: Union[tuple, int] = 500 * 500, deterministic: bool = True,
                 encoder_params: Optional[Dict] = None, pc_point_dim: int = 4):
        super().__init__()
        encoder_params = encoder_params or {}
        if encoding_type == EncodingType.Vanilla:
            self._pos_encoder = None
            pc_in_channels = pc_point_dim
        elif encoding_type == EncodingType.Fourier:                      # mask_bev_encoders.py:51-58
            pc_in_channels = 128
            self._pos_encoder_group = fourier_enc_group
            self._pos_encoder_M = 4 // fourier_enc_group
            self._pos_encoder = LearnableFourierPositionalEncoding(G=fourier_enc_group, M=self._pos_encoder_M,
                                                                   F_dim=32, H_dim=32, D=pc_in_channels, gamma=1.0)
        else:
            raise NotImplementedError(f'{encoding_type}')
        self._feat_channels = list(feat_channels)
        self._out_features = feat_channels[-1]
        self._x_range, self._y_range, self._z_range = x_range, y_range, z_range
        self._num_voxel_x = int((x_range[1] - x_range[0]) / voxel_size_x)
        self._num_voxel_y = int((y_range[1] - y_range[0]) / voxel_size_y)
        self._num_voxel_z = 1
        point_cloud_range = [x_range[0], y_range[0], z_range[0], x_range[1], y_range[1], z_range[1]]
        voxel_size = [voxel_size_x, voxel_size_y, voxel_size_z]
        self._voxel_layer = Voxelization(voxel_size, point_cloud_range, max_num_points, max_voxels, deterministic)
        self._voxel_encoder = PillarFeatureNet(in_channels=pc_in_channels, feat_channels=self._feat_channels,
                                               voxel_size=voxel_size, point_cloud_range=point_cloud_range,
                                               **encoder_params)
        out_shape = [self._num_voxel_y, self._num_voxel_x]
        self._middle_encoder = PointPillarsScatter(in_channels=self._out_features, output_shape=out_shape)
        self._layer_norm = nn.LayerNorm([self._out_features, *out_shape], eps=1e-3)

    def forward(self, point_clouds: Sequence[torch.Tensor], patch: int = 0, out: Optional[torch.Tensor] = None):
        """list of (Ni, pc_dim) device tensors → (B, C, ny, nx); with ``patch`` = 4 the same values as bf16
        ``ops.PatchTokens`` for a backbone whose first layer is a 4 x 4 patch projection.  ``out``: optional
        destination buffer for the result."""
        batch = len(point_clouds)
        pillars = self._voxel_layer.pillars(point_clouds, prefilter=True)
        if self._pos_encoder is not None:
            feats = self._voxel_encoder.forward_rows(self._fourier_rows(pillars), pillars)
        else:
            feats = self._voxel_encoder(pillars)
        ln = self._layer_norm
        return ops.scatter_layernorm(feats, ln.weight, ln.bias, pillars, batch, self._num_voxel_y, self._num_voxel_x,
                                     ln.eps, patch, out)

    def _fourier_rows(self, p: Pillars) -> torch.Tensor:
        """Decorated PFN input rows (K, 128 + 7) of the REAL points under the Fourier per-point encoding
        (mask_bev_encoders.py:85-89 + mmdet3d PillarFeatureNet, legacy).  The reference encodes every slot of the dense
        zero-padded (V, P, 4) tensor, so a padded slot carries the constant encoding e0 of the origin; the decoration
        then zeroes padded rows again, and the only trace e0 leaves is in ``points_mean``, which sums the first three
        channels over all P slots: mean = (sum over real points + (P - n) e0) / n.  That term is added here, and the
        PFN layers see exactly the rows of the dense evaluation.  Plain differentiable torch ops (the encoder's MLP is
        a few hundred parameters on K ~ 10^5 rows); the PFN layers themselves stay on K2b."""
        enc = self._pos_encoder
        dev = p.points.device
        idx = p.pillar_points.reshape(-1)
        idx = idx[idx >= 0].long()                                  # compact row order: pillar by pillar, slot by slot
        pts = p.points[idx][:, :4]
        e = enc(pts.reshape(-1, enc.G, enc.M))                      # (K, 128)
        e0 = enc(torch.zeros(1, enc.G, enc.M, device=dev, dtype=pts.dtype))[0, :3]
        n = p.num_points.to(e.dtype)                                # (V,)
        row_pillar = torch.repeat_interleave(torch.arange(p.num_pillars, device=dev), p.num_points.long())
        sums = torch.zeros(p.num_pillars, 3, device=dev, dtype=e.dtype).index_add_(0, row_pillar, e[:, :3])
        mean = (sums + (p.max_points - n).unsqueeze(1) * e0.unsqueeze(0)) / n.unsqueeze(1)
        vs, rng = self._voxel_encoder.voxel_size, self._voxel_encoder.point_cloud_range
        c = p.coors.to(e.dtype)
        center = torch.stack([c[:, 3] * vs[0] + (vs[0] / 2 + rng[0]), c[:, 2] * vs[1] + (vs[1] / 2 + rng[1]),
                              c[:, 1] * vs[2] + (vs[2] / 2 + rng[2])], 1)
        f_cluster = e[:, :3] - mean[row_pillar]
        f_center = e[:, :3] - center[row_pillar]                    # legacy: written over channels 0..2
        dist = f_center.norm(dim=1, keepdim=True)
        return torch.cat([f_center, e[:, 3:], f_cluster, f_center, dist], 1)

    def patch_layout(self, patch: int) -> bool:
        return ops.patch_layout_supported(self._out_features, self._num_voxel_y, self._num_voxel_x, patch)

    # --- staged API of the reference (mask_bev_encoders.py:95-123) -------------------------------
    def voxelize(self, point_clouds: Sequence[torch.Tensor]):
        """→ (voxels (V, P, D), num_points (V,), coors (V, 4) = (b, z, y, x))."""
        pillars = self._voxel_layer.pillars(point_clouds, prefilter=True)
        self._last_pillars = pillars
        return ops.gather_voxels(pillars), pillars.num_points, pillars.coors

    def _filter_in_range(self, point_cloud: torch.Tensor) -> torch.Tensor:
        m = (self._x_range[0] < point_cloud[:, 0]) & (point_cloud[:, 0] < self._x_range[1]) & \
            (self._y_range[0] < point_cloud[:, 1]) & (point_cloud[:, 1] < self._y_range[1]) & \
            (self._z_range[0] < point_cloud[:, 2]) & (point_cloud[:, 2] < self._z_range[1])
        return point_cloud[m]

    def encode(self, voxel, num_points, coords):
        """Staged call: must follow ``voxelize`` on the same batch (the compact pillar lists are reused)."""
        pillars = getattr(self, '_last_pillars', None)
        if pillars is None or pillars.coors.data_ptr() != coords.data_ptr():
            raise RuntimeError('encode() expects the tensors returned by the preceding voxelize() call')
        return self._voxel_encoder(pillars)

    def middle_encode(self, voxel_features, coors, batch_size=None):
        return self._middle_encoder(voxel_features, coors, batch_size)
