"""/root/reference: mask_bev/models/networks/mask2former_head/mask2former_head.py:22-562."""
from mask_bev_amd.mask2former_head import Mask2FormerHead  # noqa: F401
