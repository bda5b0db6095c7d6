"""``mask_bev.mask_bev_module.MaskBevModule`` (/root/reference: mask_bev/mask_bev_module.py:34) → mask_bev_amd."""
from mask_bev_amd.mask_bev_module import MaskBevModule  # noqa: F401
from mask_bev_amd.training_types import LrSchedulerType, OptimizerType  # noqa: F401
