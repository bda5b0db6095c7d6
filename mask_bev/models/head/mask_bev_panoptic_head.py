"""/root/reference: mask_bev/models/head/mask_bev_panoptic_head.py:14-215."""
from mask_bev_amd.head import MaskBevPanopticHead  # noqa: F401
