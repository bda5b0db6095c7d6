// K13 — row sums of the point-sampled mask losses (dice + BCE) and their gradient, one pass each.
//
// Replaces the elementwise chain of mmdet's DiceLoss (naive_dice, eps 1) and CrossEntropyLoss(use_sigmoid) as
// called on the sampled points at mask_bev/models/networks/mask2former_head/mask2former_head.py:406-424:
// sigmoid, product, three row reductions, binary_cross_entropy_with_logits and its reduction — ten passes over
// the (10·B·Q, 12 544) f32 point logits forward and about as many backward (≈ 1.9 ms per step measured) —
// by one read of (logits, targets) forward and one read + one write backward.  HBM-bound; no MFMA.
//
//   fwd:  out[r] = ( Σ σ(x)·t,  Σ σ(x),  Σ t,  Σ bce(x, t) ),   bce = max(x, 0) − x·t + log1p(exp(−|x|))
//   bwd:  dx = σ(1 − σ)·(g0·t + g1) + g3·(σ − t)      (g2, the gradient of Σ t, does not reach x)
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void acc_point(float x, float t, float& s_it, float& s_p, float& s_t, float& s_b) {
  // the forward was VALU-bound on libm's log1pf and the IEEE division (≈ 70 instructions per point, 185 us per
  // launch against ≈ 70 us of HBM time): 1 + e lies in (1, 2], where v_log_f32 / v_rcp_f32 are good to ≈ 1e-7
  // absolute — invisible in sums of 12 544 O(1) terms
  const float e = __expf(-fabsf(x));                  // in (0, 1]
  const float inv = mbv_rcp(1.0f + e);
  const float sig = x >= 0.f ? inv : e * inv;
  s_it += sig * t;
  s_p += sig;
  s_t += t;
  s_b += fmaxf(x, 0.f) - x * t + mbv_ln(1.0f + e);
}

__global__ void __launch_bounds__(kThreads) k_mask_loss_rows_fwd(const float* __restrict__ x, const float* __restrict__ t,
                                                                 int p, float* __restrict__ out) {
  __shared__ float red[4][kThreads / 64];
  const long row = blockIdx.x;
  const float* xr = x + row * p;
  const float* tr = t + row * p;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const bool vec = (p & 3) == 0 && ((reinterpret_cast<size_t>(xr) | reinterpret_cast<size_t>(tr)) & 15) == 0;
  if (vec) {
    for (int i = threadIdx.x * 4; i < p; i += kThreads * 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + i);
      const float4 tv = *reinterpret_cast<const float4*>(tr + i);
      acc_point(xv.x, tv.x, s0, s1, s2, s3);
      acc_point(xv.y, tv.y, s0, s1, s2, s3);
      acc_point(xv.z, tv.z, s0, s1, s2, s3);
      acc_point(xv.w, tv.w, s0, s1, s2, s3);
    }
  } else {
    for (int i = threadIdx.x; i < p; i += kThreads) acc_point(xr[i], tr[i], s0, s1, s2, s3);
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; red[3][wave] = s3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float s = 0.f;
    for (int w = 0; w < kThreads / 64; ++w) s += red[threadIdx.x][w];
    out[row * 4 + threadIdx.x] = s;
  }
}

__device__ __forceinline__ float grad_point(float x, float t, float g0, float g1, float g3) {
  const float e = __expf(-fabsf(x));
  const float inv = mbv_rcp(1.0f + e);                // argument in (1, 2]
  const float sig = x >= 0.f ? inv : e * inv;
  return sig * (1.0f - sig) * (g0 * t + g1) + g3 * (sig - t);
}

__global__ void __launch_bounds__(kThreads) k_mask_loss_rows_bwd(const float* __restrict__ x, const float* __restrict__ t,
                                                                 const float* __restrict__ g, int p,
                                                                 float* __restrict__ dx) {
  const long row = blockIdx.x;
  const float* xr = x + row * p;
  const float* tr = t + row * p;
  float* dr = dx + row * p;
  const float g0 = g[row * 4 + 0], g1 = g[row * 4 + 1], g3 = g[row * 4 + 3];
  const bool vec = (p & 3) == 0 &&
                   ((reinterpret_cast<size_t>(xr) | reinterpret_cast<size_t>(tr) | reinterpret_cast<size_t>(dr)) & 15) == 0;
  if (vec) {
    for (int i = threadIdx.x * 4; i < p; i += kThreads * 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + i);
      const float4 tv = *reinterpret_cast<const float4*>(tr + i);
      float4 d;
      d.x = grad_point(xv.x, tv.x, g0, g1, g3);
      d.y = grad_point(xv.y, tv.y, g0, g1, g3);
      d.z = grad_point(xv.z, tv.z, g0, g1, g3);
      d.w = grad_point(xv.w, tv.w, g0, g1, g3);
      *reinterpret_cast<float4*>(dr + i) = d;
    }
  } else {
    for (int i = threadIdx.x; i < p; i += kThreads) dr[i] = grad_point(xr[i], tr[i], g0, g1, g3);
  }
}

// ---- the dice / BCE algebra on the row sums, per decoder output (round 6: one launch instead of ~ 10 ATen launches forward and
// ~ 9 backward).  Workgroup d owns the g = rows / outputs rows of decoder output d:
//   den = S1 + S2 + 1, dice = (2 S0 + 1) / den;  loss_dice[d] = c_dice sum(1 - dice);  loss_mask[d] = c_mask sum(S3)
// and leaves, per row, the gradient of (loss_dice[d], loss_mask[d]) with respect to the row's sums for UNIT upstream gradients:
//   coef[row] = (-2 c_dice / den,  c_dice dice / den,  c_mask)      (d/dS0, d/dS1 = d/dS2, d/dS3)
// which k_mask_loss_rows_bwd2 scales by the upstream gradients of its decoder output — the backward needs no glue launch.
__global__ void __launch_bounds__(256) k_dice_bce_reduce(const float* __restrict__ sums, int g, float c_dice, float c_mask,
                                                         float* __restrict__ loss_dice, float* __restrict__ loss_mask,
                                                         float* __restrict__ coef) {
  __shared__ double red[2][4];
  const int d = blockIdx.x;
  double a_d = 0.0, a_m = 0.0;
  for (int i = threadIdx.x; i < g; i += 256) {
    const long row = (long)d * g + i;
    const float4 s = *reinterpret_cast<const float4*>(sums + row * 4);
    const float den = s.y + s.z + 1.0f;
    const float dice = (2.0f * s.x + 1.0f) / den;
    a_d += (double)(1.0f - dice);
    a_m += (double)s.w;
    coef[row * 3 + 0] = -2.0f * c_dice / den;
    coef[row * 3 + 1] = c_dice * dice / den;
    coef[row * 3 + 2] = c_mask;
  }
  a_d = wave_sum_d(a_d);
  a_m = wave_sum_d(a_m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a_d; red[1][wave] = a_m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss_dice[d] = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) * (double)c_dice);
    loss_mask[d] = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) * (double)c_mask);
  }
}

// dx of the rows kernel with the per-row gradient assembled on the fly: (g0, g1, g3) = (gd coef0, gd coef1, gm coef2) with
// gd / gm the upstream gradients of the row's decoder output (element strides sd / sm: 0 for a broadcast scalar)
__global__ void __launch_bounds__(kThreads) k_mask_loss_rows_bwd2(const float* __restrict__ x, const float* __restrict__ t,
                                                                  const float* __restrict__ coef,
                                                                  const float* __restrict__ g_dice, int sd,
                                                                  const float* __restrict__ g_mask, int sm, int g, int p,
                                                                  float* __restrict__ dx) {
  const long row = blockIdx.x;
  const long d = row / g;
  const float gd = g_dice ? g_dice[d * sd] : 0.f, gm = g_mask ? g_mask[d * sm] : 0.f;
  const float* xr = x + row * p;
  const float* tr = t + row * p;
  float* dr = dx + row * p;
  const float g0 = gd * coef[row * 3 + 0], g1 = gd * coef[row * 3 + 1], g3 = gm * coef[row * 3 + 2];
  const bool vec = (p & 3) == 0 &&
                   ((reinterpret_cast<size_t>(xr) | reinterpret_cast<size_t>(tr) | reinterpret_cast<size_t>(dr)) & 15) == 0;
  if (vec) {
    for (int i = threadIdx.x * 4; i < p; i += kThreads * 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + i);
      const float4 tv = *reinterpret_cast<const float4*>(tr + i);
      float4 o;
      o.x = grad_point(xv.x, tv.x, g0, g1, g3);
      o.y = grad_point(xv.y, tv.y, g0, g1, g3);
      o.z = grad_point(xv.z, tv.z, g0, g1, g3);
      o.w = grad_point(xv.w, tv.w, g0, g1, g3);
      *reinterpret_cast<float4*>(dr + i) = o;
    }
  } else {
    for (int i = threadIdx.x; i < p; i += kThreads) dr[i] = grad_point(xr[i], tr[i], g0, g1, g3);
  }
}

// Matching-cost terms of the point-sampled mask logits x (rows, P), one pass:
//   terms[0] = softplus(-x)  (BCE against 1),  terms[1] = softplus(-x) + x  (BCE against 0),  terms[2] = sigmoid(x)
// laid out (groups, 3, Q, P) so that one batched GEMM against the sampled ground truth (groups, P, G) yields all three
// cost matrices, plus the row sums of terms[1] and terms[2] (the "against 0" constant and the dice denominator).
// Replaces five elementwise passes and two reductions of mmdet's CrossEntropyLossCost / DiceCost as used at
// mask2former_head.py:199-205.
__global__ void __launch_bounds__(kThreads) k_match_terms(const float* __restrict__ x, int q, int p, int ones_row,
                                                          float* __restrict__ terms, float* __restrict__ sums) {
  __shared__ float red[2][kThreads / 64];
  const long row = blockIdx.x;                      // = group * q + query
  const long group = row / q, query = row - group * q;
  const float* xr = x + row * p;
  // ones_row: a group has 3 q + 1 rows, the last one all ones — the same GEMM against the sampled ground truth then also
  // returns its row sums (the dice denominator's target part), which used to be a 200 MB reduction launch of its own
  float* tg = terms + group * (3L * q + (ones_row ? 1 : 0)) * p;
  float* t0 = tg + (0L * q + query) * p;
  float* t1 = tg + (1L * q + query) * p;
  float* t2 = tg + (2L * q + query) * p;
  if (ones_row && query == 0)
    for (int i = threadIdx.x; i < p; i += kThreads) tg[3L * q * p + i] = 1.0f;
  float s1 = 0.f, s2 = 0.f;
  auto point = [&](float v, float& pos, float& neg, float& sig) {
    const float e = __expf(-fabsf(v));
    const float inv = 1.0f / (1.0f + e);
    sig = v >= 0.f ? inv : e * inv;
    pos = fmaxf(-v, 0.f) + log1pf(e);               // softplus(-x)
    neg = pos + v;                                  // softplus(x)
    s1 += neg;
    s2 += sig;
  };
  const bool vec = (p & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(terms) & 15) == 0;
  if (vec) {                                        // one 16-byte read, three 16-byte writes per 4 points
    for (int i = threadIdx.x * 4; i < p; i += kThreads * 4) {
      const float4 v = *reinterpret_cast<const float4*>(xr + i);
      float4 a, b, c;
      point(v.x, a.x, b.x, c.x);
      point(v.y, a.y, b.y, c.y);
      point(v.z, a.z, b.z, c.z);
      point(v.w, a.w, b.w, c.w);
      *reinterpret_cast<float4*>(t0 + i) = a;
      *reinterpret_cast<float4*>(t1 + i) = b;
      *reinterpret_cast<float4*>(t2 + i) = c;
    }
  } else {
    for (int i = threadIdx.x; i < p; i += kThreads) point(xr[i], t0[i], t1[i], t2[i]);
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float s = 0.f;
    for (int w = 0; w < kThreads / 64; ++w) s += red[threadIdx.x][w];
    sums[row * 2 + threadIdx.x] = s;
  }
}

// Matching cost of every (query, ground-truth column) pair from the batched GEMM's products (mask2former_head.py:199-205
// with mmdet's ClassificationCost 2.0, CrossEntropyLossCost(use_sigmoid) 5.0, DiceCost(pred_act, eps 1) 5.0):
//   cost = -2 softmax(cls)[label] + 5 (pos.t + sum(neg) - neg.t) / P + 5 (1 - (2 sig.t + 1) / (sum(sig) + sum(t) + 1))
// prod (groups, 3 q + 1, G) = [pos; neg; sig; ones] . t, sums (groups, q, 2) = [sum neg, sum sig]; group = (decoder
// output, image) with the image index fastest.  One launch instead of a softmax, a gather and ~ 20 element-wise ATen launches.
__global__ void __launch_bounds__(256) k_match_cost(const float* __restrict__ cls, const int64_t* __restrict__ labels,
                                                    const float* __restrict__ prod, const float* __restrict__ sums,
                                                    long total, int q, int g, int k1, int batch, float inv_points,
                                                    float* __restrict__ cost) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % g);
  const long qi = (i / g) % q, grp = i / ((long)g * q);
  const int b = (int)(grp % batch);
  const float* c = cls + (grp * q + qi) * k1;
  float mx = c[0];
  for (int k = 1; k < k1; ++k) mx = fmaxf(mx, c[k]);
  float den = 0.f;
  for (int k = 0; k < k1; ++k) den += __expf(c[k] - mx);
  const int lab = (int)labels[(long)b * g + col];
  const float prob = __expf(c[lab] - mx) / den;
  const float* pg = prod + grp * (3L * q + 1) * g;
  const float pos = pg[(0L * q + qi) * g + col], neg = pg[(1L * q + qi) * g + col], sig = pg[(2L * q + qi) * g + col];
  const float tsum = pg[3L * q * g + col];
  const float s_neg = sums[(grp * q + qi) * 2], s_sig = sums[(grp * q + qi) * 2 + 1];
  const float cls_cost = -prob * 2.0f;
  const float bce = (pos + s_neg - neg) * inv_points;
  const float dice = 1.0f - (2.0f * sig + 1.0f) / (s_sig + tsum + 1.0f);
  cost[i] = cls_cost + 5.0f * bce + 5.0f * dice;
}

// Class-weighted cross entropy of one decoder output per workgroup (mmdet CrossEntropyLoss with class_weight,
// reduction by avg_factor = sum of the class weights of the targets: mask2former_head.py:393-404).  Row n = (image, query)
// takes the label of its assigned ground-truth column, or `num_classes` (no object) when unassigned.
//   fwd: loss[d] = lw * sum_n w[y_n] (-log softmax(x_n)[y_n]) / (sum_n w[y_n] + eps);  wsum[d] = sum_n w[y_n]
//   bwd: dx[n][k] = g[d] lw w[y_n] (softmax(x_n)[k] - [k == y_n]) / (wsum[d] + eps)
template <bool BWD>
__global__ void __launch_bounds__(256) k_cls_loss(const float* __restrict__ cls, const int32_t* __restrict__ assigned,
                                                  const int64_t* __restrict__ labels, const float* __restrict__ cw,
                                                  int rows, int q, int g, int k1, float lw, float eps,
                                                  float* __restrict__ loss, float* __restrict__ wsum,
                                                  const float* __restrict__ gout, float* __restrict__ dcls) {
  __shared__ double red[2][4];
  const int d = blockIdx.x;
  double s_l = 0.0, s_w = 0.0;
  float scale = 0.f;
  if constexpr (BWD) scale = gout[d] * lw / (wsum[d] + eps);
  for (int n = threadIdx.x; n < rows; n += 256) {
    const float* x = cls + ((long)d * rows + n) * k1;
    const int a = assigned[(long)d * rows + n];
    const int y = a >= 0 ? (int)labels[(long)(n / q) * g + a] : k1 - 1;
    float mx = x[0];
    for (int k = 1; k < k1; ++k) mx = fmaxf(mx, x[k]);
    float den = 0.f;
    for (int k = 0; k < k1; ++k) den += __expf(x[k] - mx);
    const float w = cw[y];
    if constexpr (BWD) {
      float* dx = dcls + ((long)d * rows + n) * k1;
      for (int k = 0; k < k1; ++k) dx[k] = scale * w * (__expf(x[k] - mx) / den - (k == y ? 1.f : 0.f));
    } else {
      s_l += (double)(w * (mx + __logf(den) - x[y]));
      s_w += (double)w;
    }
  }
  if constexpr (!BWD) {
    s_l = wave_sum_d(s_l);
    s_w = wave_sum_d(s_w);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s_l; red[1][wave] = s_w; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double l = red[0][0] + red[0][1] + red[0][2] + red[0][3], w = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      wsum[d] = (float)w;
      loss[d] = (float)((double)lw * l / (w + (double)eps));
    }
  }
}

}  // namespace

extern "C" int mbv_match_cost(const float* cls, const int64_t* labels, const float* prod, const float* row_sums,
                              int64_t groups, int32_t queries, int32_t targets, int32_t classes_plus_one, int32_t batch,
                              int32_t points, float* cost, void* stream) {
  if (groups < 0 || queries <= 0 || targets <= 0 || classes_plus_one <= 0 || batch <= 0 || points <= 0) return MBV_ERR_BAD_ARG;
  if (groups == 0) return MBV_OK;
  if (!cls || !labels || !prod || !row_sums || !cost || groups % batch) return MBV_ERR_BAD_ARG;
  const long total = (long)groups * queries * targets;
  hipLaunchKernelGGL(k_match_cost, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cls, labels,
                     prod, row_sums, total, queries, targets, classes_plus_one, batch, 1.0f / (float)points, cost);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_cls_loss_fwd(const float* cls, const int32_t* assigned, const int64_t* labels, const float* class_weight,
                                int32_t outputs, int32_t batch, int32_t queries, int32_t targets, int32_t classes_plus_one,
                                float loss_weight, float eps, float* loss, float* weight_sum, void* stream) {
  if (outputs < 0 || batch <= 0 || queries <= 0 || targets <= 0 || classes_plus_one <= 0) return MBV_ERR_BAD_ARG;
  if (outputs == 0) return MBV_OK;
  if (!cls || !assigned || !labels || !class_weight || !loss || !weight_sum) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_cls_loss<false>, dim3((unsigned)outputs), dim3(256), 0, (hipStream_t)stream, cls, assigned, labels,
                     class_weight, batch * queries, queries, targets, classes_plus_one, loss_weight, eps, loss, weight_sum,
                     (const float*)nullptr, (float*)nullptr);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_cls_loss_bwd(const float* cls, const int32_t* assigned, const int64_t* labels, const float* class_weight,
                                const float* weight_sum, const float* grad_loss, int32_t outputs, int32_t batch,
                                int32_t queries, int32_t targets, int32_t classes_plus_one, float loss_weight, float eps,
                                float* grad_cls, void* stream) {
  if (outputs < 0 || batch <= 0 || queries <= 0 || targets <= 0 || classes_plus_one <= 0) return MBV_ERR_BAD_ARG;
  if (outputs == 0) return MBV_OK;
  if (!cls || !assigned || !labels || !class_weight || !weight_sum || !grad_loss || !grad_cls) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_cls_loss<true>, dim3((unsigned)outputs), dim3(256), 0, (hipStream_t)stream, cls, assigned, labels,
                     class_weight, batch * queries, queries, targets, classes_plus_one, loss_weight, eps, (float*)nullptr,
                     const_cast<float*>(weight_sum), grad_loss, grad_cls);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_mask_loss_rows_fwd(const float* logits, const float* targets, int64_t rows, int32_t points,
                                      float* out_sums, void* stream) {
  if (rows < 0 || points <= 0) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!logits || !targets || !out_sums) return MBV_ERR_BAD_ARG;
  if (rows > 0x7fffffffL) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_mask_loss_rows_fwd, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     points, out_sums);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_mask_loss_rows_bwd(const float* logits, const float* targets, const float* grad_sums, int64_t rows,
                                      int32_t points, float* grad_logits, void* stream) {
  if (rows < 0 || points <= 0) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!logits || !targets || !grad_sums || !grad_logits) return MBV_ERR_BAD_ARG;
  if (rows > 0x7fffffffL) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_mask_loss_rows_bwd, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     grad_sums, points, grad_logits);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_dice_bce_reduce(const float* sums, int64_t rows, int32_t outputs, float c_dice, float c_mask,
                                   float* loss_dice, float* loss_mask, float* coef, void* stream) {
  if (rows < 0 || outputs <= 0 || rows % outputs) return MBV_ERR_BAD_ARG;
  if (!sums || !loss_dice || !loss_mask || !coef || (reinterpret_cast<size_t>(sums) & 15)) return MBV_ERR_BAD_ARG;
  if (rows / outputs > 0x7fffffffL) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_dice_bce_reduce, dim3((unsigned)outputs), dim3(256), 0, (hipStream_t)stream, sums,
                     (int)(rows / outputs), c_dice, c_mask, loss_dice, loss_mask, coef);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_mask_loss_rows_bwd_coef(const float* logits, const float* targets, const float* coef, const float* grad_dice,
                                           int32_t stride_dice, const float* grad_mask, int32_t stride_mask, int64_t rows,
                                           int32_t outputs, int32_t points, float* grad_logits, void* stream) {
  if (rows < 0 || points <= 0 || outputs <= 0 || rows % outputs || stride_dice < 0 || stride_mask < 0) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!logits || !targets || !coef || !grad_logits) return MBV_ERR_BAD_ARG;
  if (rows > 0x7fffffffL) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_mask_loss_rows_bwd2, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     coef, grad_dice, stride_dice, grad_mask, stride_mask, (int)(rows / outputs), points, grad_logits);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_match_cost_terms(const float* logits, int64_t groups, int32_t queries, int32_t points,
                                    int32_t ones_row, float* terms, float* row_sums, void* stream) {
  if (groups < 0 || queries <= 0 || points <= 0) return MBV_ERR_BAD_ARG;
  if (groups == 0) return MBV_OK;
  if (!logits || !terms || !row_sums) return MBV_ERR_BAD_ARG;
  if (groups * queries > 0x7fffffffL) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_match_terms, dim3((unsigned)(groups * queries)), dim3(kThreads), 0, (hipStream_t)stream, logits,
                     queries, points, ones_row ? 1 : 0, terms, row_sums);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
