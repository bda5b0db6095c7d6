"""Tiny registry resolving the mm* type strings of the reference configs to this package's classes
(SURVEY.md §8b "Registry / import names to expose").  The GPU box has none of mmcv / mmdet / mmdet3d."""
from __future__ import annotations

from typing import Any, Callable, Dict


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules: Dict[str, Any] = {}

    def register_module(self, name: str = None, module: Any = None):
        def deco(cls):
            self._modules[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key: str):
        k = key.split('.', 1)[1] if key.startswith(('mmdet.', 'mmdet3d.', 'mmcv.')) else key
        if k not in self._modules:
            raise KeyError(f'{key} is not registered in {self.name}')
        return self._modules[k]

    def build(self, cfg: Dict[str, Any], **default_args):
        cfg = dict(cfg)
        cls = self.get(cfg.pop('type'))
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        return cls(**cfg)


MODELS = Registry('models')
TASK_UTILS = Registry('task_utils')


def _populate():
    from .encoders import PillarFeatureNet, PointPillarsScatter, Voxelization
    from .mask2former_head import Mask2FormerHead, Mask2FormerTransformerDecoder, MSDeformAttnPixelDecoder
    from .swin import CustomSwinTransformer
    for cls in (Voxelization, PillarFeatureNet, PointPillarsScatter, CustomSwinTransformer, Mask2FormerHead,
                MSDeformAttnPixelDecoder, Mask2FormerTransformerDecoder):
        MODELS.register_module(module=cls)
    # loss / matcher type strings of mask_bev_panoptic_head.py:177-214 are folded into Mask2FormerHead.loss;
    # they resolve to descriptors so that configs naming them stay valid.
    for name in ('CrossEntropyLoss', 'DiceLoss'):
        MODELS.register_module(name=name, module=dict)
    for name in ('HungarianAssigner', 'ClassificationCost', 'CrossEntropyLossCost', 'DiceCost', 'MaskPseudoSampler'):
        TASK_UTILS.register_module(name=name, module=dict)


_populate()
