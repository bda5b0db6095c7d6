"""ORACLE — test infrastructure, NOT product code (see oracle/__init__.py).

Torch-fp32 CPU restatement, in reference-faithful *dense* form, of the MaskBEV
``scan -> BEV -> mask`` path.  Everything is a pure
function of a flat ``state_dict`` whose keys are the reference's checkpoint keys
(``_encoder.*``, ``_backbone._backbone.*``, ``_panoptic_head._panoptic_head.*``),
so that the product module's ``state_dict()`` can be fed to it unchanged.

File:line citations refer to /root/reference.  ``[upstream]`` marks behaviour of
the un-vendored mmcv 2.0.0 / mmdet 3.0.0 / mmdet3d 1.1.0 packages, restated from
their published algorithms (PARITY UNPINNED — SURVEY.md §8c, Appendix A).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

ENC = '_encoder.'
BB = '_backbone._backbone.'
HEAD = '_panoptic_head._panoptic_head.'


# --------------------------------------------------------------------------------------
# configuration  (mask_bev/mask_bev_module.py:35-80)
# --------------------------------------------------------------------------------------
def make_cfg(x_range, y_range, z_range, voxel_size, num_queries, max_num_points,
             encoder_feat_channels, backbone_embed_dim, head_feat_channels, head_out_channels,
             backbone_patch_size=4, backbone_window_size=10, backbone_strides=(4, 2, 2, 2),
             backbone_use_abs_emb=True, backbone_swap_dims=False, head_reverse_class_weights=False,
             head_num_classes=1, pc_point_dim=4, max_voxels=500 * 500,
             depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), mlp_ratio=4,
             pd_layers=6, pd_heads=8, pd_levels=3, pd_points=4, pd_ffn=1024,
             dec_layers=9, dec_heads=8, dec_ffn=2048, num_points=12544,
             oversample_ratio=3.0, importance_sample_ratio=0.75, encoder_encoding_type='vanilla',
             encoder_fourier_enc_group=1, **_ignored) -> SimpleNamespace:
    """Same keyword names as ``MaskBevModule.__init__`` (mask_bev_module.py:35-43).
    The trailing architecture keywords default to the values hard-coded at
    mask_bev_backbone.py:41-64 and mask_bev_panoptic_head.py:105-215 and exist only
    so that golden fixtures can use tiny networks."""
    c = SimpleNamespace()
    c.x_range, c.y_range, c.z_range = tuple(x_range), tuple(y_range), tuple(z_range)
    c.voxel_size = voxel_size
    c.voxel_size3 = [voxel_size, voxel_size, z_range[1] - z_range[0]]        # mask_bev_module.py:62
    c.pc_range = [x_range[0], y_range[0], z_range[0], x_range[1], y_range[1], z_range[1]]
    c.nx = int((x_range[1] - x_range[0]) / voxel_size)                       # mask_bev_module.py:68
    c.ny = int((y_range[1] - y_range[0]) / voxel_size)                       # mask_bev_module.py:69
    # mmcv Voxelization [upstream]: grid = round((max - min) / vs) in f32
    pcr = torch.tensor(c.pc_range, dtype=torch.float32)
    vs = torch.tensor(c.voxel_size3, dtype=torch.float32)
    c.grid3 = [int(v) for v in torch.round((pcr[3:] - pcr[:3]) / vs).long()]
    c.num_queries = num_queries
    c.max_num_points = max_num_points
    c.max_voxels = max_voxels                                               # mask_bev_encoders.py:25
    c.feat_channels = list(encoder_feat_channels)
    c.embed_dim = backbone_embed_dim
    c.head_feat = head_feat_channels
    c.head_out = head_out_channels
    c.patch_size = backbone_patch_size
    c.window_size = backbone_window_size
    c.strides = tuple(backbone_strides)
    c.use_abs_emb = backbone_use_abs_emb
    c.swap_dims = backbone_swap_dims
    c.reverse_class_weights = head_reverse_class_weights
    c.num_classes = head_num_classes
    c.pc_dim = pc_point_dim
    # mask_bev_encoders.py:51-58: 'fourier' feeds a 128-channel per-point encoding to the PillarFeatureNet
    c.encoding_type = encoder_encoding_type
    c.fourier_group = encoder_fourier_enc_group
    c.pfn_in = 128 if encoder_encoding_type == 'fourier' else pc_point_dim
    c.depths, c.num_heads, c.mlp_ratio = tuple(depths), tuple(num_heads), mlp_ratio
    c.pd_layers, c.pd_heads, c.pd_levels, c.pd_points, c.pd_ffn = pd_layers, pd_heads, pd_levels, pd_points, pd_ffn
    c.dec_layers, c.dec_heads, c.dec_ffn = dec_layers, dec_heads, dec_ffn
    c.num_points = num_points
    c.oversample_ratio = oversample_ratio
    c.importance_sample_ratio = importance_sample_ratio
    cw = [1.0] * c.num_classes + [0.1]                                       # mask_bev_panoptic_head.py:101-103
    c.class_weight = list(reversed(cw)) if c.reverse_class_weights else cw
    return c


# --------------------------------------------------------------------------------------
# A1/A2  range filter + hard voxelisation
# --------------------------------------------------------------------------------------
_ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/voxelize_ref.c with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_ORACLE_DIR, 'libmbv_oracle.so')
    src = os.path.join(_ORACLE_DIR, 'voxelize_ref.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _ORACLE_DIR, 'libmbv_oracle.so'])
    return so


def _clib():
    global _CLIB
    if _CLIB is None:
        lib = ctypes.CDLL(build_c_oracle())
        lib.mbv_oracle_hard_voxelize.restype = ctypes.c_int
        lib.mbv_oracle_hard_voxelize.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p]
        _CLIB = lib
    return _CLIB


def filter_in_range(cfg, pc: Tensor) -> Tensor:
    """mask_bev_encoders.py:113-117 — strict ``<`` on x, y, z; torch compares a f32 tensor with a
    python scalar in f32 (verified in the build container), i.e. bounds rounded to f32."""
    m = ((cfg.x_range[0] < pc[:, 0]) & (pc[:, 0] < cfg.x_range[1]) &
         (cfg.y_range[0] < pc[:, 1]) & (pc[:, 1] < cfg.y_range[1]) &
         (cfg.z_range[0] < pc[:, 2]) & (pc[:, 2] < cfg.z_range[1]))
    return pc[m]


def hard_voxelize_py(points: np.ndarray, pc_range, voxel_size3, grid3, max_points: int, max_voxels: int):
    """Pure-python/numpy hard voxelisation (small inputs only); same algorithm as
    oracle/voxelize_ref.c — mmcv 2.0.0 ``hard_voxelize_forward_cpu_kernel`` [upstream]."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, dim = pts.shape
    lo = np.asarray(pc_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size3, dtype=np.float32)
    lut: Dict[Tuple[int, int, int], int] = {}
    voxels: List[np.ndarray] = []
    coors: List[Tuple[int, int, int]] = []
    nump: List[int] = []
    for i in range(n):
        c = []
        ok = True
        for j in range(3):
            q = np.floor((pts[i, j] - lo[j]) / vs[j])          # all np.float32
            cj = int(q)
            if cj < 0 or cj >= grid3[j]:
                ok = False
                break
            c.append(cj)
        if not ok:
            continue
        key = (c[2], c[1], c[0])
        vid = lut.get(key, -1)
        if vid == -1:
            if max_voxels != -1 and len(coors) >= max_voxels:
                continue
            vid = len(coors)
            lut[key] = vid
            coors.append(key)
            voxels.append(np.zeros((max_points, dim), np.float32))
            nump.append(0)
        if nump[vid] < max_points:
            voxels[vid][nump[vid]] = pts[i]
            nump[vid] += 1
    v = len(coors)
    return (np.stack(voxels) if v else np.zeros((0, max_points, dim), np.float32),
            np.asarray(coors, np.int32).reshape(v, 3), np.asarray(nump, np.int32))


def hard_voxelize_c(points: np.ndarray, pc_range, voxel_size3, grid3, max_points: int, max_voxels: int,
                    prefilter: bool = False, return_point_map: bool = False):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, dim = pts.shape
    cap = int(min(n, max_voxels if max_voxels != -1 else n))
    voxels = np.zeros((max(cap, 1), max_points, dim), np.float32)
    coors = np.zeros((max(cap, 1), 3), np.int32)
    nump = np.zeros((max(cap, 1),), np.int32)
    pv = np.zeros((max(n, 1),), np.int32)
    ps = np.zeros((max(n, 1),), np.int32)
    r6 = np.asarray(pc_range, np.float32)
    v3 = np.asarray(voxel_size3, np.float32)
    g3 = np.asarray(grid3, np.int32)
    v = _clib().mbv_oracle_hard_voxelize(
        pts.ctypes.data, n, dim, r6.ctypes.data, v3.ctypes.data, g3.ctypes.data, max_points, max_voxels,
        1 if prefilter else 0, voxels.ctypes.data, coors.ctypes.data, nump.ctypes.data, pv.ctypes.data,
        ps.ctypes.data)
    if v < 0:
        raise MemoryError('oracle voxeliser LUT allocation failed')
    out = (voxels[:v], coors[:v], nump[:v])
    if return_point_map:
        out = out + (pv[:n], ps[:n])
    return out


def voxelize(cfg, point_clouds: Sequence[Tensor], use_c: bool = True):
    """MaskBevEncoder.voxelize, mask_bev_encoders.py:95-111 → (voxels, num_points, coors_batch(b,z,y,x))."""
    voxels, coors, nump = [], [], []
    for i, res in enumerate(point_clouds):
        res = filter_in_range(cfg, res)
        fn = hard_voxelize_c if use_c else hard_voxelize_py
        v, c, n = fn(res.detach().cpu().numpy(), cfg.pc_range, cfg.voxel_size3, cfg.grid3,
                     cfg.max_num_points, cfg.max_voxels)
        voxels.append(torch.from_numpy(np.ascontiguousarray(v)))
        nump.append(torch.from_numpy(np.ascontiguousarray(n)))
        coors.append(F.pad(torch.from_numpy(np.ascontiguousarray(c)), (1, 0), mode='constant', value=i))
    return torch.cat(voxels, 0), torch.cat(nump, 0), torch.cat(coors, 0)


# --------------------------------------------------------------------------------------
# A4  PillarFeatureNet (mmdet3d 1.1.0, legacy=True, with_distance=True) [upstream]
# --------------------------------------------------------------------------------------
def pfn_decorate(cfg, voxels: Tensor, num_points: Tensor, coors: Tensor) -> Tensor:
    """Dense (V, P, pc+7) decoration with the *legacy in-place aliasing*: the centre offsets are written
    through a view of ``features[:, :, :3]`` so channels 0-2 become centre offsets and the distance is the
    norm of the centre offset (SURVEY.md §7 'Legacy in-place decoration', Appendix A)."""
    feats = voxels.clone()
    vx, vy, vz = cfg.voxel_size3
    x_off, y_off, z_off = vx / 2 + cfg.pc_range[0], vy / 2 + cfg.pc_range[1], vz / 2 + cfg.pc_range[2]
    points_mean = feats[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(feats).view(-1, 1, 1)
    f_cluster = feats[:, :, :3] - points_mean                       # computed BEFORE the aliasing write
    f_center = feats[:, :, :3]                                      # a view (legacy=True)
    f_center[:, :, 0] = f_center[:, :, 0] - (coors[:, 3].type_as(feats).unsqueeze(1) * vx + x_off)
    f_center[:, :, 1] = f_center[:, :, 1] - (coors[:, 2].type_as(feats).unsqueeze(1) * vy + y_off)
    f_center[:, :, 2] = f_center[:, :, 2] - (coors[:, 1].type_as(feats).unsqueeze(1) * vz + z_off)
    dist = torch.norm(feats[:, :, :3], 2, 2, keepdim=True)          # AFTER the write → ‖centre offset‖
    out = torch.cat([feats, f_cluster, f_center, dist], dim=-1)
    p = out.shape[1]
    mask = (torch.arange(p).view(1, -1) < num_points.view(-1, 1)).unsqueeze(-1).type_as(out)
    return out * mask


def pfn_forward(cfg, sd: SD, voxels: Tensor, num_points: Tensor, coors: Tensor, training: bool = True,
                bn_buffers: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """3 x PFNLayer: Linear(no bias) → BatchNorm1d(eps 1e-3, momentum 0.01) over ALL V·P rows (padding
    included) → ReLU → max over P → concat [x, max] (non-last).  (V, P, pc) → (V, C_last)."""
    x = pfn_decorate(cfg, voxels, num_points, coors)
    n_layers = len(cfg.feat_channels)
    for i in range(n_layers):
        p = f'{ENC}_voxel_encoder.pfn_layers.{i}.'
        last = i == n_layers - 1
        y = F.linear(x, sd[p + 'linear.weight'])
        rm = sd[p + 'norm.running_mean'].clone() if bn_buffers is None else bn_buffers[p + 'norm.running_mean']
        rv = sd[p + 'norm.running_var'].clone() if bn_buffers is None else bn_buffers[p + 'norm.running_var']
        y = F.batch_norm(y.permute(0, 2, 1).contiguous(), rm, rv, sd[p + 'norm.weight'], sd[p + 'norm.bias'],
                         training, 0.01, 1e-3).permute(0, 2, 1).contiguous()
        y = F.relu(y)
        y_max = torch.max(y, dim=1, keepdim=True)[0]
        if last:
            x = y_max
        else:
            x = torch.cat([y, y_max.repeat(1, x.shape[1], 1)], dim=2)
    return x.squeeze(1)


def scatter_to_canvas(cfg, feats: Tensor, coors: Tensor, batch_size: int) -> Tensor:
    """PointPillarsScatter.forward_batch (mmdet3d 1.1.0) [upstream]; mask_bev_encoders.py:122-123."""
    c = feats.shape[1]
    out = []
    for b in range(batch_size):
        canvas = torch.zeros(c, cfg.nx * cfg.ny, dtype=feats.dtype)
        m = coors[:, 0] == b
        this = coors[m]
        idx = (this[:, 2] * cfg.nx + this[:, 3]).long()
        canvas[:, idx] = feats[m].t()
        out.append(canvas)
    return torch.stack(out, 0).view(batch_size, c, cfg.ny, cfg.nx)


def fourier_encode(sd: SD, x: Tensor, p: str = ENC + '_pos_encoder.') -> Tensor:
    """LearnableFourierPositionalEncoding.forward (mask_bev/models/positional_encoding/
    learnable_fourier_positional_encoding.py:41-59): x (N, G, M) → (N, D).  PINNED by tests/golden/fourier.npz
    (the reference class itself, run by tests/golden/make_golden_fourier.py)."""
    n = x.shape[0]
    projected = F.linear(x, sd[p + 'Wr.weight'])                                     # (N, G, F/2), no bias
    f_dim = 2 * projected.shape[-1]
    feats = torch.cat([torch.cos(projected), torch.sin(projected)], dim=-1) / math.sqrt(f_dim)
    y = F.linear(F.gelu(F.linear(feats, sd[p + 'mlp.0.weight'], sd[p + 'mlp.0.bias'])),
                 sd[p + 'mlp.2.weight'], sd[p + 'mlp.2.bias'])
    return y.reshape(n, -1)


def encoder_forward(cfg, sd: SD, point_clouds: Sequence[Tensor], training: bool = True, return_parts: bool = False):
    """MaskBevEncoder.forward, mask_bev_encoders.py:77-93."""
    voxels, nump, coors = voxelize(cfg, point_clouds)
    if getattr(cfg, 'encoding_type', 'vanilla') == 'fourier':
        # mask_bev_encoders.py:85-89: EVERY slot of the dense (V, P, 4) tensor is encoded, the zero padding included —
        # a padded slot becomes the constant encoding of the origin and takes part in the PFN's points_mean
        v, npts, _ = voxels.shape
        g = cfg.fourier_group
        x = voxels.reshape(v * npts, -1).reshape(-1, g, 4 // g)
        voxels = fourier_encode(sd, x).reshape(v, npts, -1)
    feats = pfn_forward(cfg, sd, voxels, nump, coors, training)
    canvas = scatter_to_canvas(cfg, feats, coors, len(point_clouds))
    w, b = sd[ENC + '_layer_norm.weight'], sd[ENC + '_layer_norm.bias']
    out = F.layer_norm(canvas, list(w.shape), w, b, 1e-3)                   # mask_bev_encoders.py:75,92
    if return_parts:
        return out, dict(voxels=voxels, num_points=nump, coors=coors, pillar_feats=feats, canvas=canvas)
    return out


# --------------------------------------------------------------------------------------
# A7-A9  Swin backbone (in-repo: mask_bev/models/networks/swin/swin.py)
# --------------------------------------------------------------------------------------
def _ln(sd: SD, p: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], eps)


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def _corner_pad(x: Tensor, k: int, s: int) -> Tensor:
    """mmdet AdaptivePadding('corner') [upstream]: pad bottom/right so the kernel tiles the input."""
    h, w = x.shape[-2:]
    pad_h = max((math.ceil(h / s) - 1) * s + (k - 1) + 1 - h, 0)
    pad_w = max((math.ceil(w / s) - 1) * s + (k - 1) + 1 - w, 0)
    if pad_h > 0 or pad_w > 0:
        x = F.pad(x, [0, pad_w, 0, pad_h])
    return x


def ffn(sd: SD, p: str, x: Tensor, identity: Optional[Tensor] = None, act: str = 'gelu') -> Tensor:
    """mmcv FFN [upstream]: Linear → act → Linear, + identity. Keys ``layers.0.0`` / ``layers.1``."""
    h = _lin(sd, p + '.layers.0.0', x)
    h = F.gelu(h) if act == 'gelu' else F.relu(h)
    h = _lin(sd, p + '.layers.1', h)
    return (x if identity is None else identity) + h


def rel_position_index(ws: int) -> Tensor:
    """swin.py:64-68,120-124."""
    seq1 = torch.arange(0, (2 * ws - 1) * ws, 2 * ws - 1)
    seq2 = torch.arange(0, ws, 1)
    coords = (seq1[:, None] + seq2[None, :]).reshape(1, -1)
    idx = coords + coords.T
    return idx.flip(1).contiguous()


def window_msa(sd: SD, p: str, x: Tensor, num_heads: int, ws: int, mask: Optional[Tensor]) -> Tensor:
    """WindowMSA.forward, swin.py:80-118."""
    b, n, c = x.shape
    qkv = _lin(sd, p + '.qkv', x).reshape(b, n, 3, num_heads, c // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((c // num_heads) ** -0.5)
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + '.relative_position_bias_table'][rel_position_index(ws).view(-1)].view(ws * ws, ws * ws, -1)
    attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.view(b // nw, nw, num_heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, n, n)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(b, n, c)
    return _lin(sd, p + '.proj', x)


def _window_partition(x: Tensor, ws: int) -> Tensor:
    b, h, w, c = x.shape                                                      # swin.py:271-284
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


def _window_reverse(win: Tensor, h: int, w: int, ws: int) -> Tensor:
    b = int(win.shape[0] / (h * w / ws / ws))                                 # swin.py:255-269
    x = win.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def shift_window_msa(sd: SD, p: str, query: Tensor, hw: Tuple[int, int], num_heads: int, ws: int, shift: int) -> Tensor:
    """ShiftWindowMSA.forward, swin.py:179-253."""
    b, l, c = query.shape
    h, w = hw
    query = query.view(b, h, w, c)
    pad_r = (ws - w % ws) % ws
    pad_b = (ws - h % ws) % ws
    query = F.pad(query, (0, 0, 0, pad_r, 0, pad_b))
    hp, wp = query.shape[1], query.shape[2]
    if shift > 0:
        shifted = torch.roll(query, shifts=(-shift, -shift), dims=(1, 2))
        img_mask = torch.zeros((1, hp, wp, 1))
        slices = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
        cnt = 0
        for hs in slices:
            for wsl in slices:
                img_mask[:, hs, wsl, :] = cnt
                cnt += 1
        mw = _window_partition(img_mask, ws).view(-1, ws * ws)
        attn_mask = mw.unsqueeze(1) - mw.unsqueeze(2)
        attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
    else:
        shifted, attn_mask = query, None
    win = _window_partition(shifted, ws).view(-1, ws * ws, c)
    out = window_msa(sd, p + '.w_msa', win, num_heads, ws, attn_mask).view(-1, ws, ws, c)
    x = _window_reverse(out, hp, wp, ws)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    if pad_r > 0 or pad_b:
        x = x[:, :h, :w, :].contiguous()
    return x.view(b, h * w, c)


def swin_block(sd: SD, p: str, x: Tensor, hw, num_heads: int, ws: int, shift: bool) -> Tensor:
    """SwinBlock.forward, swin.py:357-377."""
    identity = x
    x = _ln(sd, p + '.norm1', x)
    x = shift_window_msa(sd, p + '.attn', x, hw, num_heads, ws, ws // 2 if shift else 0)
    x = x + identity
    identity = x
    x = _ln(sd, p + '.norm2', x)
    return ffn(sd, p + '.ffn', x, identity=identity, act='gelu')


def patch_embed(sd: SD, p: str, x: Tensor, patch: int):
    """mmdet PatchEmbed [upstream]: corner pad → Conv2d(k=s=patch) → flatten → LN. swin.py:579-586."""
    x = _corner_pad(x, patch, patch)
    x = F.conv2d(x, sd[p + '.projection.weight'], sd[p + '.projection.bias'], stride=patch)
    hw = (x.shape[2], x.shape[3])
    x = x.flatten(2).transpose(1, 2)
    return _ln(sd, p + '.norm', x), hw


def patch_merging(sd: SD, p: str, x: Tensor, hw, stride: int):
    """mmdet PatchMerging [upstream]: Unfold(k=2, stride) channel order (C, kh, kw) → LN(4C) → Linear(4C→2C)."""
    b, l, c = x.shape
    h, w = hw
    x = x.view(b, h, w, c).permute(0, 3, 1, 2)
    x = _corner_pad(x, 2, stride)
    h, w = x.shape[-2:]
    x = F.unfold(x, kernel_size=2, dilation=1, padding=0, stride=stride)
    oh = (h - (2 - 1) - 1) // stride + 1
    ow = (w - (2 - 1) - 1) // stride + 1
    x = x.transpose(1, 2)
    x = _ln(sd, p + '.norm', x)
    x = F.linear(x, sd[p + '.reduction.weight'])
    return x, (oh, ow)


def swin_forward(cfg, sd: SD, x: Tensor, prefix: str = BB) -> List[Tensor]:
    """CustomSwinTransformer.forward, swin.py:745-774."""
    x, hw = patch_embed(sd, prefix + 'patch_embed', x, cfg.patch_size)
    if cfg.use_abs_emb:
        ape = sd[prefix + 'absolute_pos_embed']
        w_, h_ = ape.shape[2:4]                                               # swin.py:750 (deliberate swap)
        if hw[0] != h_ or hw[1] != w_:
            pe = F.interpolate(ape, size=hw, mode='bicubic', align_corners=False).flatten(2).transpose(1, 2)
        else:
            pe = ape.flatten(2).transpose(1, 2)
        x = x + pe
    outs = []
    n_stage = len(cfg.depths)
    for i in range(n_stage):
        sp = f'{prefix}stages.{i}'
        for j in range(cfg.depths[i]):
            x = swin_block(sd, f'{sp}.blocks.{j}', x, hw, cfg.num_heads[i], cfg.window_size, shift=(j % 2 == 1))
        out, out_hw = x, hw
        if i < n_stage - 1:
            x, hw = patch_merging(sd, sp + '.downsample', x, hw, cfg.strides[i + 1])
        out = _ln(sd, f'{prefix}norm{i}', out)
        c = out.shape[-1]
        outs.append(out.view(-1, out_hw[0], out_hw[1], c).permute(0, 3, 1, 2).contiguous())
    return outs


# --------------------------------------------------------------------------------------
# A10  MSDeformAttnPixelDecoder (mmdet 3.0.0) + MultiScaleDeformableAttention (mmcv 2.0.0) [upstream]
# --------------------------------------------------------------------------------------
def sine_pos_enc(b: int, h: int, w: int, num_feats: int, temperature: float = 10000.0,
                 scale: float = 2 * math.pi, eps: float = 1e-6, offset: float = 0.0) -> Tensor:
    """mmdet SinePositionalEncoding(normalize=True) on an all-False mask [upstream]."""
    not_mask = torch.ones((b, h, w), dtype=torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(b, h, w, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(b, h, w, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def ms_deform_attn_core(value: Tensor, spatial_shapes: Sequence[Tuple[int, int]], sampling_locations: Tensor,
                        attention_weights: Tensor) -> Tensor:
    """mmcv ``multi_scale_deformable_attn_pytorch`` [upstream]: bilinear ``grid_sample`` (zeros padding,
    align_corners=False) of each level, weighted sum.  value (B, Nv, H, D); loc (B, Nq, H, L, P, 2);
    w (B, Nq, H, L, P) → (B, Nq, H*D)."""
    bs, _, nh, d = value.shape
    _, nq, _, nl, npnt, _ = sampling_locations.shape
    value_list = value.split([h * w for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * nh, d, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * nh, 1, nq, nl * npnt)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, nh * d, nq)
    return out.transpose(1, 2).contiguous()


def ms_deform_attn(sd: SD, p: str, query: Tensor, query_pos: Tensor, reference_points: Tensor,
                   spatial_shapes, nh: int, nl: int, npnt: int) -> Tensor:
    """mmcv MultiScaleDeformableAttention.forward (batch_first, value = un-positioned query) [upstream]."""
    identity = query
    value = query
    q = query + query_pos
    bs, nq, c = q.shape
    value = _lin(sd, p + '.value_proj', value).view(bs, nq, nh, -1)
    off = _lin(sd, p + '.sampling_offsets', q).view(bs, nq, nh, nl, npnt, 2)
    aw = _lin(sd, p + '.attention_weights', q).view(bs, nq, nh, nl * npnt).softmax(-1).view(bs, nq, nh, nl, npnt)
    ss = torch.tensor(spatial_shapes, dtype=torch.long)
    normalizer = torch.stack([ss[..., 1], ss[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = ms_deform_attn_core(value, spatial_shapes, loc, aw)
    return _lin(sd, p + '.output_proj', out) + identity


def _conv_gn(sd: SD, p: str, x: Tensor, groups: int = 32, padding: int = 0, relu: bool = False) -> Tensor:
    """mmcv ConvModule(conv → GN [→ ReLU]) [upstream]; keys ``conv`` / ``gn``."""
    x = F.conv2d(x, sd[p + '.conv.weight'], sd.get(p + '.conv.bias'), padding=padding)
    x = F.group_norm(x, groups, sd[p + '.gn.weight'], sd[p + '.gn.bias'], 1e-5)
    return F.relu(x) if relu else x


def pixel_decoder_forward(cfg, sd: SD, feats: List[Tensor], prefix: str = HEAD + 'pixel_decoder.'):
    """mmdet MSDeformAttnPixelDecoder.forward [upstream] (cfg at mask_bev_panoptic_head.py:119-146)."""
    bs = feats[0].shape[0]
    n_in, nl = len(feats), cfg.pd_levels
    strides = [4, 8, 16, 32][:n_in]                                            # mask_bev_panoptic_head.py:111
    enc_in, pos_list, shapes, refs = [], [], [], []
    for i in range(nl):
        lvl = n_in - i - 1
        feat = feats[lvl]
        proj = _conv_gn(sd, f'{prefix}input_convs.{i}', feat)
        h, w = feat.shape[-2:]
        pos = sine_pos_enc(bs, h, w, cfg.head_feat // 2)
        lvl_pos = sd[prefix + 'level_encoding.weight'][i].view(1, -1, 1, 1) + pos
        # MlvlPointGenerator.single_level_grid_priors(offset=0.5) / (w*stride, h*stride)
        sx = (torch.arange(0, w) + 0.5) * strides[lvl]
        sy = (torch.arange(0, h) + 0.5) * strides[lvl]
        xx = sx.repeat(h)
        yy = sy.view(-1, 1).repeat(1, w).view(-1)
        ref = torch.stack([xx, yy], dim=-1)
        factor = torch.tensor([[w, h]]) * strides[lvl]
        ref = ref / factor
        enc_in.append(proj.flatten(2).permute(0, 2, 1))
        pos_list.append(lvl_pos.flatten(2).permute(0, 2, 1))
        shapes.append((h, w))
        refs.append(ref)
    query = torch.cat(enc_in, dim=1)
    qpos = torch.cat(pos_list, dim=1)
    ref = torch.cat(refs, dim=0)[None, :, None].repeat(bs, 1, nl, 1)
    for l in range(cfg.pd_layers):
        lp = f'{prefix}encoder.layers.{l}'
        query = ms_deform_attn(sd, lp + '.self_attn', query, qpos, ref, shapes, cfg.pd_heads, nl, cfg.pd_points)
        query = _ln(sd, lp + '.norms.0', query)
        query = ffn(sd, lp + '.ffn', query, act='relu')
        query = _ln(sd, lp + '.norms.1', query)
    memory = query.permute(0, 2, 1)
    outs = list(torch.split(memory, [h * w for h, w in shapes], dim=-1))
    outs = [x.reshape(bs, -1, shapes[i][0], shapes[i][1]) for i, x in enumerate(outs)]
    for i in range(n_in - nl - 1, -1, -1):
        cur = _conv_gn(sd, f'{prefix}lateral_convs.{i}', feats[i])
        y = cur + F.interpolate(outs[-1], size=cur.shape[-2:], mode='bilinear', align_corners=False)
        y = _conv_gn(sd, f'{prefix}output_convs.{i}', y, padding=1, relu=True)
        outs.append(y)
    multi_scale = outs[:cfg.pd_levels]
    mask_feature = F.conv2d(outs[-1], sd[prefix + 'mask_feature.weight'], sd[prefix + 'mask_feature.bias'])
    return mask_feature, multi_scale


# --------------------------------------------------------------------------------------
# A11/A12  Mask2Former head forward (in-repo: mask2former_head.py:428-562)
# --------------------------------------------------------------------------------------
def mha(sd: SD, p: str, query: Tensor, key: Tensor, value: Tensor, query_pos, key_pos, attn_mask, nh: int) -> Tensor:
    """mmcv MultiheadAttention(batch_first=True) → nn.MultiheadAttention [upstream]: pos added to q/k only;
    bool mask True = blocked; returns identity + out."""
    identity = query
    q = query + query_pos if query_pos is not None else query
    k = key + key_pos if key_pos is not None else key
    q, k, v = q.transpose(0, 1), k.transpose(0, 1), value.transpose(0, 1)
    e = q.shape[-1]
    out = F.multi_head_attention_forward(
        q, k, v, e, nh, sd[p + '.attn.in_proj_weight'], sd[p + '.attn.in_proj_bias'], None, None, False, 0.0,
        sd[p + '.attn.out_proj.weight'], sd[p + '.attn.out_proj.bias'], training=False, key_padding_mask=None,
        need_weights=True, attn_mask=attn_mask)[0]
    return identity + out.transpose(0, 1)


def forward_head(cfg, sd: SD, decoder_out: Tensor, mask_feature: Tensor, target_size, prefix: str = HEAD):
    """Mask2FormerHead._forward_head, mask2former_head.py:428-472."""
    x = _ln(sd, prefix + 'transformer_decoder.post_norm', decoder_out)
    cls_pred = _lin(sd, prefix + 'cls_embed', x)
    m = F.relu(_lin(sd, prefix + 'mask_embed.0', x))
    m = F.relu(_lin(sd, prefix + 'mask_embed.2', m))
    m = _lin(sd, prefix + 'mask_embed.4', m)
    mask_pred = torch.einsum('bqc,bchw->bqhw', m, mask_feature)
    attn_mask = F.interpolate(mask_pred, target_size, mode='bilinear', align_corners=False)
    attn_mask = attn_mask.flatten(2).unsqueeze(1).repeat((1, cfg.dec_heads, 1, 1)).flatten(0, 1)
    attn_mask = (attn_mask.sigmoid() < 0.5).detach()
    return cls_pred, mask_pred, attn_mask


def head_forward(cfg, sd: SD, feats: List[Tensor], prefix: str = HEAD, return_parts: bool = False, pixel_decoder=None):
    """Mask2FormerHead.forward, mask2former_head.py:474-562 → (cls_list, mask_list, [None]*n).  ``pixel_decoder``:
    optional callable feats -> (mask_features, memories) in place of the restated MSDeformAttnPixelDecoder (the second
    head fixture pins this function against the reference with a stand-in that is independent of this file)."""
    bs = feats[0].shape[0]
    mask_features, memories = (pixel_decoder(feats) if pixel_decoder is not None
                               else pixel_decoder_forward(cfg, sd, feats, prefix + 'pixel_decoder.'))
    nl = cfg.pd_levels
    dec_in, dec_pos = [], []
    for i in range(nl):
        x = memories[i].flatten(2).permute(0, 2, 1)                          # decoder_input_projs = Identity
        x = x + sd[prefix + 'level_embed.weight'][i].view(1, 1, -1)
        h, w = memories[i].shape[-2:]
        pos = sine_pos_enc(bs, h, w, cfg.head_feat // 2).flatten(2).permute(0, 2, 1)
        dec_in.append(x)
        dec_pos.append(pos)
    query_feat = sd[prefix + 'query_feat.weight'].unsqueeze(0).repeat((bs, 1, 1))
    query_embed = sd[prefix + 'query_embed.weight'].unsqueeze(0).repeat((bs, 1, 1))
    cls_list, mask_list = [], []
    cls_pred, mask_pred, attn_mask = forward_head(cfg, sd, query_feat, mask_features, memories[0].shape[-2:], prefix)
    cls_list.append(cls_pred)
    mask_list.append(mask_pred)
    for i in range(cfg.dec_layers):
        lvl = i % nl
        attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False   # mask2former_head.py:538-539
        lp = f'{prefix}transformer_decoder.layers.{i}'
        # Mask2FormerTransformerDecoderLayer [upstream]: cross → LN → self → LN → FFN → LN
        q = mha(sd, lp + '.cross_attn', query_feat, dec_in[lvl], dec_in[lvl], query_embed, dec_pos[lvl], attn_mask,
                cfg.dec_heads)
        q = _ln(sd, lp + '.norms.0', q)
        q = mha(sd, lp + '.self_attn', q, q, q, query_embed, query_embed, None, cfg.dec_heads)
        q = _ln(sd, lp + '.norms.1', q)
        q = ffn(sd, lp + '.ffn', q, act='relu')
        query_feat = _ln(sd, lp + '.norms.2', q)
        cls_pred, mask_pred, attn_mask = forward_head(cfg, sd, query_feat, mask_features,
                                                      memories[(i + 1) % nl].shape[-2:], prefix)
        cls_list.append(cls_pred)
        mask_list.append(mask_pred)
    heights = [None for _ in cls_list]
    if return_parts:
        return cls_list, mask_list, heights, dict(mask_features=mask_features, memories=memories)
    return cls_list, mask_list, heights


def model_forward(cfg, sd: SD, point_clouds: Sequence[Tensor], training: bool = True):
    """MaskBevModule.forward, mask_bev_module.py:174-178."""
    x = encoder_forward(cfg, sd, point_clouds, training)
    x = swin_forward(cfg, sd, x)
    return head_forward(cfg, sd, x)


# --------------------------------------------------------------------------------------
# A13  loss  (mask2former_head.py:154-232, 246-298, 326-426 + mmdet 3.0.0 losses/matcher [upstream])
# --------------------------------------------------------------------------------------
class PointSource:
    """Supplier of the uniform random sampling points of the loss, in the reference's draw order:
    per decoder output: B x rand(1, P, 2) (matcher, mask2former_head.py:191), then rand(G, 3P, 2) and
    rand(G, P - int(0.75 P), 2) (importance sampling).  CPU generator so product and oracle can share it."""

    def __init__(self, seed: Optional[int] = 0):
        # seed=None → the global torch RNG (what the reference itself draws from)
        self.gen = None if seed is None else torch.Generator().manual_seed(seed)

    def rand(self, *shape) -> Tensor:
        return torch.rand(*shape) if self.gen is None else torch.rand(*shape, generator=self.gen)


def point_sample(inp: Tensor, points: Tensor) -> Tensor:
    """mmcv point_sample [upstream]: grid_sample at 2p-1, align_corners=False. inp (N,C,H,W), points (N,P,2)."""
    out = F.grid_sample(inp, 2.0 * points.unsqueeze(2) - 1.0, align_corners=False)
    return out.squeeze(3)


def match_cost(cfg, cls_score: Tensor, mask_pts_pred: Tensor, gt_labels: Tensor, gt_pts: Tensor) -> Tensor:
    """mmdet ClassificationCost(2) + CrossEntropyLossCost(sigmoid, 5) + DiceCost(pred_act, eps 1, 5) [upstream]."""
    cls_cost = -cls_score.softmax(-1)[:, gt_labels] * 2.0
    p = mask_pts_pred.flatten(1).float()
    g = gt_pts.flatten(1).float()
    n = p.shape[1]
    pos = F.binary_cross_entropy_with_logits(p, torch.ones_like(p), reduction='none')
    neg = F.binary_cross_entropy_with_logits(p, torch.zeros_like(p), reduction='none')
    bce = (torch.einsum('nc,mc->nm', pos, g) + torch.einsum('nc,mc->nm', neg, 1 - g)) / n * 5.0
    ps = p.sigmoid()
    numerator = 2 * torch.einsum('nc,mc->nm', ps, g)
    denominator = ps.sum(-1)[:, None] + g.sum(-1)[None, :]
    dice = (1 - (numerator + 1.0) / (denominator + 1.0)) * 5.0
    return cls_cost + bce + dice


def get_targets_single(cfg, cls_score: Tensor, mask_pred: Tensor, gt_labels: Tensor, gt_masks: Tensor, pts: PointSource):
    """Mask2FormerHead._get_targets_single, mask2former_head.py:154-232."""
    from scipy.optimize import linear_sum_assignment
    nq, ng = cls_score.shape[0], gt_labels.shape[0]
    coords = pts.rand(1, cfg.num_points, 2)
    mp = point_sample(mask_pred.unsqueeze(1), coords.repeat(nq, 1, 1)).squeeze(1)
    gp = point_sample(gt_masks.unsqueeze(1).float(), coords.repeat(ng, 1, 1)).squeeze(1)
    cost = match_cost(cfg, cls_score, mp, gt_labels, gp).detach().cpu()
    rows, cols = linear_sum_assignment(cost)
    assigned = torch.zeros(nq, dtype=torch.long)
    assigned[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
    pos_inds = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
    pos_gt = assigned[pos_inds] - 1
    labels = gt_labels.new_full((nq,), cfg.num_classes, dtype=torch.long)
    labels[pos_inds] = gt_labels[pos_gt]
    mask_targets = gt_masks[pos_gt]
    mask_weights = mask_pred.new_zeros((nq,))
    mask_weights[pos_inds] = 1.0
    return labels, mask_targets, mask_weights, nq       # avg_factor of MaskPseudoSampler = num_pos + num_neg


def _weight_reduce_mean(loss: Tensor, avg_factor) -> Tensor:
    return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)      # mmdet weight_reduce_loss [upstream]


def loss_single(cfg, cls_scores: Tensor, mask_preds: Tensor, labels_gt: Tensor, masks_gt: Tensor, pts: PointSource,
                world_size: int = 1):
    """Mask2FormerHead._loss_by_feat_single, mask2former_head.py:326-426."""
    b = cls_scores.size(0)
    t = [get_targets_single(cfg, cls_scores[i], mask_preds[i], labels_gt[i], masks_gt[i], pts) for i in range(b)]
    labels = torch.stack([x[0] for x in t], 0).flatten(0, 1)
    mask_targets = torch.cat([x[1] for x in t], 0)
    mask_weights = torch.stack([x[2] for x in t], 0)
    avg_factor = sum(x[3] for x in t)
    class_weight = cls_scores.new_tensor(cfg.class_weight)
    ce = F.cross_entropy(cls_scores.flatten(0, 1), labels, weight=class_weight, reduction='none')
    loss_cls = 2.0 * _weight_reduce_mean(ce, class_weight[labels].sum())
    num_total_masks = max(float(avg_factor), 1.0)           # reduce_mean over 1 rank, mask2former_head.py:388-389
    mp = mask_preds[mask_weights > 0]
    if mask_targets.shape[0] == 0:
        return loss_cls, mp.sum(), mp.sum()
    with torch.no_grad():
        g = mp.shape[0]
        n_samp = int(cfg.num_points * cfg.oversample_ratio)
        coords = pts.rand(g, n_samp, 2)
        logits = point_sample(mp.unsqueeze(1), coords)
        unc = -torch.abs(logits)
        n_unc = int(cfg.importance_sample_ratio * cfg.num_points)
        n_rand = cfg.num_points - n_unc
        idx = torch.topk(unc[:, 0, :], k=n_unc, dim=1)[1]
        idx = idx + (n_samp * torch.arange(g, dtype=torch.long))[:, None]
        coords = coords.view(-1, 2)[idx.view(-1), :].view(g, n_unc, 2)
        if n_rand > 0:
            coords = torch.cat((coords, pts.rand(g, n_rand, 2)), dim=1)
        tgt = point_sample(mask_targets.unsqueeze(1).float(), coords).squeeze(1)
    pred = point_sample(mp.unsqueeze(1), coords).squeeze(1)
    # DiceLoss(use_sigmoid, activate, naive_dice, eps=1) x5
    ps = pred.sigmoid().flatten(1)
    tg = tgt.flatten(1).float()
    a = torch.sum(ps * tg, 1)
    d = (2 * a + 1.0) / (torch.sum(ps, 1) + torch.sum(tg, 1) + 1.0)
    loss_dice = 5.0 * _weight_reduce_mean(1 - d, num_total_masks)
    # CrossEntropyLoss(use_sigmoid) x5
    bce = F.binary_cross_entropy_with_logits(pred.reshape(-1), tgt.reshape(-1).float(), reduction='none')
    loss_mask = 5.0 * _weight_reduce_mean(bce, num_total_masks * cfg.num_points)
    return loss_cls, loss_mask, loss_dice


def loss_dict(cfg, cls_list, mask_list, labels_gt: Tensor, masks_gt: Tensor, pts: Optional[PointSource] = None):
    """Mask2FormerHead.loss, mask2former_head.py:246-298 (height terms are the int 0, :386)."""
    pts = pts or PointSource(0)
    res = [loss_single(cfg, c, m, labels_gt, masks_gt, pts) for c, m in zip(cls_list, mask_list)]
    out = dict(loss_cls=res[-1][0], loss_mask=res[-1][1], loss_dice=res[-1][2], loss_height=0)
    for i, (lc, lm, ld) in enumerate(res[:-1]):
        out[f'd{i}.loss_cls'], out[f'd{i}.loss_mask'], out[f'd{i}.loss_dice'], out[f'd{i}.loss_height'] = lc, lm, ld, 0
    return out


def total_loss(d) -> Tensor:
    return sum(v for k, v in d.items() if 'loss' in k)                     # mask_bev_module.py:193-195


# --------------------------------------------------------------------------------------
# random weights with the reference's state_dict layout (for fixtures / standalone runs)
# --------------------------------------------------------------------------------------
def make_state_dict(cfg, seed: int = 0, scale: float = 1.0) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def rn(*shape, std=0.05):
        return torch.randn(*shape, generator=g) * std * scale

    def lin(p, o, i, bias=True, std=None):
        sd[p + '.weight'] = rn(o, i, std=std or (1.0 / math.sqrt(i)))
        if bias:
            sd[p + '.bias'] = rn(o, std=0.05)

    def norm(p, c):
        sd[p + '.weight'] = 1.0 + rn(c, std=0.1)
        sd[p + '.bias'] = rn(c, std=0.1)

    # encoder
    if getattr(cfg, 'encoding_type', 'vanilla') == 'fourier':
        g_, m_ = cfg.fourier_group, 4 // cfg.fourier_group
        pe = ENC + '_pos_encoder.'
        sd[pe + 'Wr.weight'] = rn(16, m_, std=1.0)                                   # F_dim 32, gamma 1 (:56-58)
        lin(pe + 'mlp.0', 32, 32)
        lin(pe + 'mlp.2', 128 // g_, 32)
    cin = getattr(cfg, 'pfn_in', cfg.pc_dim) + 7
    chans = [cin] + cfg.feat_channels
    for i in range(len(cfg.feat_channels)):
        last = i == len(cfg.feat_channels) - 1
        units = chans[i + 1] if last else chans[i + 1] // 2
        p = f'{ENC}_voxel_encoder.pfn_layers.{i}.'
        sd[p + 'linear.weight'] = rn(units, chans[i], std=1.0 / math.sqrt(chans[i]))
        norm(p + 'norm', units)
        sd[p + 'norm.running_mean'] = torch.zeros(units)
        sd[p + 'norm.running_var'] = torch.ones(units)
        sd[p + 'norm.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    c_enc = cfg.feat_channels[-1]
    sd[ENC + '_layer_norm.weight'] = 1.0 + rn(c_enc, cfg.ny, cfg.nx, std=0.1)
    sd[ENC + '_layer_norm.bias'] = rn(c_enc, cfg.ny, cfg.nx, std=0.1)
    # backbone
    e = cfg.embed_dim
    sd[BB + 'patch_embed.projection.weight'] = rn(e, c_enc, cfg.patch_size, cfg.patch_size,
                                                  std=1.0 / math.sqrt(c_enc * cfg.patch_size ** 2))
    sd[BB + 'patch_embed.projection.bias'] = rn(e)
    norm(BB + 'patch_embed.norm', e)
    if cfg.use_abs_emb:
        pr, pc = (cfg.nx, cfg.ny) if not cfg.swap_dims else (cfg.ny, cfg.nx)          # swin.py:588-597
        sd[BB + 'absolute_pos_embed'] = rn(1, e, pr // cfg.patch_size, pc // cfg.patch_size, std=0.02)
    ws = cfg.window_size
    c = e
    for i, depth in enumerate(cfg.depths):
        for j in range(depth):
            p = f'{BB}stages.{i}.blocks.{j}'
            norm(p + '.norm1', c)
            sd[p + '.attn.w_msa.relative_position_bias_table'] = rn((2 * ws - 1) ** 2, cfg.num_heads[i], std=0.2)
            sd[p + '.attn.w_msa.relative_position_index'] = rel_position_index(ws)
            lin(p + '.attn.w_msa.qkv', 3 * c, c)
            lin(p + '.attn.w_msa.proj', c, c)
            norm(p + '.norm2', c)
            lin(p + '.ffn.layers.0.0', cfg.mlp_ratio * c, c)
            lin(p + '.ffn.layers.1', c, cfg.mlp_ratio * c)
        norm(f'{BB}norm{i}', c)
        if i < len(cfg.depths) - 1:
            norm(f'{BB}stages.{i}.downsample.norm', 4 * c)
            lin(f'{BB}stages.{i}.downsample.reduction', 2 * c, 4 * c, bias=False)
            c *= 2
    # head
    fch, och = cfg.head_feat, cfg.head_out
    in_ch = [e * 2 ** i for i in range(len(cfg.depths))]
    pd = HEAD + 'pixel_decoder.'
    n_in, nl = len(in_ch), cfg.pd_levels
    for i in range(nl):
        ci = in_ch[n_in - i - 1]
        sd[f'{pd}input_convs.{i}.conv.weight'] = rn(fch, ci, 1, 1, std=1.0 / math.sqrt(ci))
        sd[f'{pd}input_convs.{i}.conv.bias'] = rn(fch)
        norm(f'{pd}input_convs.{i}.gn', fch)
    sd[pd + 'level_encoding.weight'] = rn(nl, fch, std=1.0)
    for l in range(cfg.pd_layers):
        p = f'{pd}encoder.layers.{l}'
        lin(p + '.self_attn.sampling_offsets', cfg.pd_heads * nl * cfg.pd_points * 2, fch, std=0.02)
        sd[p + '.self_attn.sampling_offsets.bias'] = rn(cfg.pd_heads * nl * cfg.pd_points * 2, std=1.5)
        lin(p + '.self_attn.attention_weights', cfg.pd_heads * nl * cfg.pd_points, fch)
        lin(p + '.self_attn.value_proj', fch, fch)
        lin(p + '.self_attn.output_proj', fch, fch)
        lin(p + '.ffn.layers.0.0', cfg.pd_ffn, fch)
        lin(p + '.ffn.layers.1', fch, cfg.pd_ffn)
        norm(p + '.norms.0', fch)
        norm(p + '.norms.1', fch)
    for i in range(n_in - nl):
        sd[f'{pd}lateral_convs.{i}.conv.weight'] = rn(fch, in_ch[i], 1, 1, std=1.0 / math.sqrt(in_ch[i]))
        norm(f'{pd}lateral_convs.{i}.gn', fch)
        sd[f'{pd}output_convs.{i}.conv.weight'] = rn(fch, fch, 3, 3, std=1.0 / math.sqrt(9 * fch))
        norm(f'{pd}output_convs.{i}.gn', fch)
    sd[pd + 'mask_feature.weight'] = rn(och, fch, 1, 1, std=1.0 / math.sqrt(fch))
    sd[pd + 'mask_feature.bias'] = rn(och)
    td = HEAD + 'transformer_decoder.'
    for l in range(cfg.dec_layers):
        p = f'{td}layers.{l}'
        for a in ('cross_attn', 'self_attn'):
            sd[f'{p}.{a}.attn.in_proj_weight'] = rn(3 * fch, fch, std=1.0 / math.sqrt(fch))
            sd[f'{p}.{a}.attn.in_proj_bias'] = rn(3 * fch)
            lin(f'{p}.{a}.attn.out_proj', fch, fch)
        lin(p + '.ffn.layers.0.0', cfg.dec_ffn, fch)
        lin(p + '.ffn.layers.1', fch, cfg.dec_ffn)
        for k in range(3):
            norm(f'{p}.norms.{k}', fch)
    norm(td + 'post_norm', fch)
    sd[HEAD + 'query_embed.weight'] = rn(cfg.num_queries, fch, std=1.0)
    sd[HEAD + 'query_feat.weight'] = rn(cfg.num_queries, fch, std=1.0)
    sd[HEAD + 'level_embed.weight'] = rn(nl, fch, std=1.0)
    lin(HEAD + 'cls_embed', cfg.num_classes + 1, fch)
    lin(HEAD + 'mask_embed.0', fch, fch)
    lin(HEAD + 'mask_embed.2', fch, fch)
    lin(HEAD + 'mask_embed.4', och, fch)
    return sd
