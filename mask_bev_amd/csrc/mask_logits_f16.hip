// K7 for IEEE-half activations: the same source with lo16_t = _Float16 (common.hpp).
#define MBV_H16 1
#include "mask_logits.hip"
