"""``MaskBevBackbone`` — same constructor / forward as
/root/reference: mask_bev/models/backbones/mask_bev_backbone.py:6-64 (Swin-T-like, depths 2-2-6-2,
heads 3-6-12-24, MLP ratio 4, no dropout)."""
from __future__ import annotations

from torch import nn

from .swin import CustomSwinTransformer


class MaskBevBackbone(nn.Module):
    def __init__(self, pseudo_img_size, in_channels: int, embded_dims: int, patch_size: int, window_size: int,
                 strides, use_abs_enc: bool, swap_dims: bool = False, backbone_overwrites=None):
        super().__init__()
        config = self._get_config(pseudo_img_size, in_channels, embded_dims, patch_size, window_size, strides,
                                  use_abs_enc, swap_dims)
        config.update(backbone_overwrites or {})
        self._backbone = CustomSwinTransformer(**config)
        self._backbone.init_weights()

    def forward(self, x, cut=None):
        """(B, C, ny, nx) → 4 maps (B, C_i, ny/S_i, nx/S_i).  ``cut``: see ``CustomSwinTransformer.forward``."""
        return self._backbone(x) if cut is None else self._backbone(x, cut=cut)

    @staticmethod
    def _get_config(pretrain_img_size, in_channels, embed_dims, patch_size, window_size, strides, use_abs_pos_embed,
                    swap_dims):
        return dict(pretrain_img_size=tuple(pretrain_img_size), in_channels=in_channels, embed_dims=embed_dims,
                    patch_size=patch_size, window_size=window_size, mlp_ratio=4, depths=(2, 2, 6, 2),
                    num_heads=(3, 6, 12, 24), strides=tuple(strides), out_indices=(0, 1, 2, 3), qkv_bias=True,
                    qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0,
                    use_abs_pos_embed=use_abs_pos_embed, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN'),
                    with_cp=False, init_cfg=None, swap_dims=swap_dims)
