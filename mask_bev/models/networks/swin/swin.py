"""/root/reference: mask_bev/models/networks/swin/swin.py:22-774."""
from mask_bev_amd.swin import (CustomSwinTransformer, ShiftWindowMSA, SwinBlock, SwinBlockSequence,  # noqa: F401
                               WindowMSA)
