"""/root/reference: mask_bev/models/training_types.py:1-13."""
from mask_bev_amd.training_types import LrSchedulerType, OptimizerType  # noqa: F401
