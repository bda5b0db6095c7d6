"""``MaskBevPanopticHead`` — constructor / forward / loss of
/root/reference: mask_bev/models/head/mask_bev_panoptic_head.py:15-215."""
from __future__ import annotations

from typing import List

from torch import nn

from .registry import MODELS


class MaskBevPanopticHead(nn.Module):
    def __init__(self, in_channels: List[int], feat_channels: int, out_channels: int, num_queries: int,
                 num_classes: int, reverse_class_weights: bool = False, predict_height: bool = False):
        super().__init__()
        config = self._get_config(num_classes, 0, num_queries, in_channels, feat_channels, out_channels,
                                  reverse_class_weights)
        self._num_classes = num_classes
        self._predict_height = predict_height
        self._panoptic_head = MODELS.build(config, predict_height=predict_height)
        self._panoptic_head.init_weights()

    def forward(self, x):
        return self._panoptic_head.forward(x, None)

    def loss(self, cls, masks, label_gt, masks_gt, heights_pred=None, heights_gt=None):
        return self._panoptic_head.loss(cls, masks, label_gt, masks_gt, None, heights_pred, heights_gt)

    def update_mAP_metrics(self, layer_index: int, pred_cls, pred_masks, labels_gt, masks_gt, cls_metric=None,
                           map_metric=None, mIoU_metric=None):
        """mask_bev_panoptic_head.py:34-96 (mask_bev_amd/metrics.py): the assignment of the loss just evaluated is reused,
        the matched mask IoU runs on K15, and ``map_metric`` (torchmetrics' COCO mask mAP in the reference) is a
        :class:`~mask_bev_amd.metrics.MaskMeanAveragePrecision` fed with device-side pairwise IoUs."""
        from . import metrics
        metrics.update_metrics(self._panoptic_head, layer_index, pred_cls, pred_masks, labels_gt, masks_gt, cls_metric,
                               mIoU_metric, map_metric)

    @staticmethod
    def _get_config(num_things_classes, num_stuff_classes, num_queries, in_channels, head_feat_channels,
                    head_out_channels, reverse_class_weights):
        """Same dictionary as mask_bev_panoptic_head.py:98-215 (type strings included)."""
        num_classes = num_things_classes + num_stuff_classes
        class_weights = [1.0] * num_classes + [0.1]
        if reverse_class_weights:
            class_weights = list(reversed(class_weights))
        levels, heads = 3, 8
        return dict(
            type='Mask2FormerHead', in_channels=in_channels, strides=[4, 8, 16, 32], feat_channels=head_feat_channels,
            out_channels=head_out_channels, num_things_classes=num_things_classes,
            num_stuff_classes=num_stuff_classes, num_queries=num_queries, num_transformer_feat_level=levels,
            align_corners=False,
            pixel_decoder=dict(
                type='mmdet.MSDeformAttnPixelDecoder', num_outs=levels, norm_cfg=dict(type='GN', num_groups=32),
                act_cfg=dict(type='ReLU'),
                encoder=dict(num_layers=6, layer_cfg=dict(
                    self_attn_cfg=dict(embed_dims=head_feat_channels, num_heads=heads, num_levels=levels,
                                       num_points=4, im2col_step=64, dropout=0.0, batch_first=True),
                    ffn_cfg=dict(embed_dims=head_feat_channels, feedforward_channels=1024, num_fcs=2, ffn_drop=0.0,
                                 act_cfg=dict(type='ReLU', inplace=True)))),
                positional_encoding=dict(num_feats=head_feat_channels // 2, normalize=True)),
            enforce_decoder_input_project=False,
            positional_encoding=dict(num_feats=head_feat_channels // 2, normalize=True),
            transformer_decoder=dict(return_intermediate=True, num_layers=9, layer_cfg=dict(
                self_attn_cfg=dict(embed_dims=head_feat_channels, num_heads=heads, attn_drop=0.0, proj_drop=0.0,
                                   batch_first=True),
                cross_attn_cfg=dict(embed_dims=head_feat_channels, num_heads=heads, attn_drop=0.0, proj_drop=0.0,
                                    batch_first=True),
                ffn_cfg=dict(embed_dims=head_feat_channels, feedforward_channels=2048, num_fcs=2,
                             act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0.0, add_identity=True))),
            loss_cls=dict(type='mmdet.CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0, reduction='mean',
                          class_weight=class_weights),
            loss_mask=dict(type='mmdet.CrossEntropyLoss', use_sigmoid=True, reduction='mean', loss_weight=5.0),
            loss_dice=dict(type='mmdet.DiceLoss', use_sigmoid=True, activate=True, reduction='mean', naive_dice=True,
                           eps=1.0, loss_weight=5.0),
            train_cfg=dict(num_points=12544, oversample_ratio=3.0, importance_sample_ratio=0.75,
                           assigner=dict(type='mmdet.HungarianAssigner', match_costs=[
                               dict(type='mmdet.ClassificationCost', weight=2.0),
                               dict(type='mmdet.CrossEntropyLossCost', weight=5.0, use_sigmoid=True),
                               dict(type='mmdet.DiceCost', weight=5.0, pred_act=True, eps=1.0)]),
                           sampler=dict(type='mmdet.MaskPseudoSampler')),
            test_cfg=dict(mode='whole'))
