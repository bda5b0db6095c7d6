// AdamW arithmetic shared by K11's optimizer pass (optim.hip) and K3's backward with the update of the (C, ny, nx)
// LayerNorm affine fused into it (scatter_layernorm.hip): ONE definition, so the two paths are bit-identical.
// Follows torch/optim/adamw.py's single-tensor update order (mask_bev/mask_bev_module.py:131-166 configures it).
#pragma once
#include "common.hpp"

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, grad_scale;
  int decoupled, zero_grad;
  int shadow_kind;                 // MBV_DT_BF16 / MBV_DT_F16: storage of the weight shadow
  const float* loss_scale;         // device scalar (nullable): gradients arrive multiplied by it (fp16 loss scaling)
  const int* skip;                 // device flag (nullable): non-zero = the gradient held inf / nan, skip the update
  const int* applied;              // device count of updates APPLIED so far (nullable): the bias corrections then use
                                   //   t = *applied + 1 — torch.amp.GradScaler skips optimizer.step() on an overflow, so
                                   //   Adam's step count must not advance there (a host count would)
};

__device__ __forceinline__ unsigned short shadow_bits(float p, int kind) {
  return kind == MBV_DT_F16 ? __builtin_bit_cast(unsigned short, (_Float16)p) : f32_to_bf16_rne(p);
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  // contraction pinned OFF: optim.hip is built with hipcc's default (fused multiply-adds where the compiler likes),
  // scatter_layernorm.hip with -ffp-contract=off (its LayerNorm sums are order-sensitive) — the update must not depend on
  // which unit runs it; unfused multiply / add is also what torch's single-tensor AdamW computes
#pragma clang fp contract(off)
  g *= a.grad_scale;
  if (a.decoupled) {
    p *= 1.0f - a.lr * a.weight_decay;            // param.mul_(1 - lr * wd)
  } else if (a.weight_decay != 0.0f) {
    g += a.weight_decay * p;                      // Adam: L2 term folded into the gradient
  }
  m += (g - m) * (1.0f - a.beta1);                // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + (1.0f - a.beta2) * g * g;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) / a.bias_correction2_sqrt + a.eps;
  p -= (a.lr / a.bias_correction1) * (m / denom);
}
