"""String enums of the YAML keys ``optimiser_type`` / ``lr_schedulers_type``
(/root/reference: mask_bev/models/training_types.py:1-13; values appear in configs/training/**.yml)."""


class OptimizerType:
    ADAM = 'adam'
    LAMB = 'lamb'
    SGD = 'sgd'
    ADAM_W = 'adam_w'


class LrSchedulerType:
    STEP = 'step'
    REDUCE_ON_PLATEAU = 'plateau'
    COSINE = 'cosine'
    POLY = 'poly'
