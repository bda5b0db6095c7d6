"""/root/reference: mask_bev/models/encoders/mask_bev_encoders.py:15-123."""
from mask_bev_amd.encoders import EncodingType, MaskBevEncoder  # noqa: F401
