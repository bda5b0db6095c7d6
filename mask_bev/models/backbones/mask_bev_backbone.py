"""/root/reference: mask_bev/models/backbones/mask_bev_backbone.py:8-64."""
from mask_bev_amd.backbone import MaskBevBackbone  # noqa: F401
