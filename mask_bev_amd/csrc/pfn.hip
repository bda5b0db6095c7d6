// K2b — PillarFeatureNet layers on real points only: the per-pillar (non-GEMM) parts, forward and backward.
//
// Replaces the PFNLayer stack of mmdet3d PillarFeatureNet (Linear(no bias) → BatchNorm1d(eps 1e-3, momentum
// 0.01) → ReLU → max over the 32 point slots → concat [x, max]) reached from MaskBevEncoder.encode
// (mask_bev/models/encoders/mask_bev_encoders.py:70-72,119-120).  The reference evaluates it on the zero-padded
// (V, 32, C) tensor; here a pillar holds its n real rows (compact, contiguous, `row_start`) plus ONE
// representative padded row with multiplicity P - n (SURVEY.md §7 "Padded-row algebra"):
//   y[r]    = W_a a_prev[r] + t[v],  t[v] = W_b max_prev[v]          (the two GEMMs run on hipBLASLt, f32)
//   y_pad[v] = W_a a_pad_prev[v] + t[v]
//   batch statistics over all V*P rows: real rows + (P - n) copies of the padded row
//   a = relu(bn(y)), a_pad = relu(bn(y_pad)), max[v] = max(max_r a[r], a_pad[v] if n < P)
// One wavefront walks one pillar at a time with lane = channel, so every row access is a contiguous
// 256 B (64 channels) / 512 B (128 channels) segment; the channel sums stay in registers across pillars and
// leave the workgroup once.  Backward carries, for the padded rows, the SUM over the P - n copies (all maps
// are linear in the gradient), which reproduces the dense BatchNorm backward exactly.
#include "common.hpp"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlocks = 1024;     // persistent-style grid: one set of channel atomics per workgroup

// CPL = channels per lane (1 for 64 channels, 2 for 128)
template <int CPL>
__global__ void __launch_bounds__(256) k_pfn_stats(float* __restrict__ Y, const float* __restrict__ T,
                                                   float* __restrict__ Ypad, const int32_t* __restrict__ row_start,
                                                   const int32_t* __restrict__ num_points, int V, int P, int U,
                                                   double* __restrict__ sums) {
  constexpr int UMAX = 64 * CPL;
  __shared__ double red[kWavesPerBlock][2 * UMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s[CPL], q[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) { s[k] = 0.f; q[k] = 0.f; }
  double ds[CPL], dq[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) { ds[k] = 0.0; dq[k] = 0.0; }
  int since_flush = 0;
  for (int v = blockIdx.x * kWavesPerBlock + wave; v < V; v += gridDim.x * kWavesPerBlock) {
    const int n = num_points[v];
    const int64_t rs = row_start[v];
    const float mult = (float)(P - n);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      const float t = T ? T[(int64_t)v * U + c] : 0.f;
      for (int j = 0; j < n; ++j) {
        float y = Y[(rs + j) * U + c];
        if (T) { y += t; Y[(rs + j) * U + c] = y; }
        s[k] += y;
        q[k] += y * y;
      }
      float yp = Ypad[(int64_t)v * U + c];
      if (T) { yp += t; Ypad[(int64_t)v * U + c] = yp; }
      s[k] += mult * yp;
      q[k] += mult * yp * yp;
    }
    if (++since_flush == 64) {      // bound the f32 partial sums: fold into f64 every 64 pillars
#pragma unroll
      for (int k = 0; k < CPL; ++k) { ds[k] += s[k]; dq[k] += q[k]; s[k] = 0.f; q[k] = 0.f; }
      since_flush = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    red[wave][lane + 64 * k] = ds[k] + s[k];
    red[wave][UMAX + lane + 64 * k] = dq[k] + q[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * UMAX; i += blockDim.x) {
    const int half = i / UMAX, c = i - half * UMAX;
    if (c < U) atomicAdd(&sums[half * U + c], red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// mean / var → scale, shift (+ running-stat update); one workgroup
__global__ void k_pfn_finalize(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float eps, float momentum, int training,
                               float* __restrict__ running_mean, float* __restrict__ running_var, int U,
                               float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                               float* __restrict__ rstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= U) return;
  float mean, var;
  if (training) {
    const double m = sums[c] / count;
    double vv = sums[U + c] / count - m * m;
    if (vv < 0.0) vv = 0.0;
    mean = (float)m;
    var = (float)vv;
    const double unbiased = vv * (count / (count > 1.0 ? count - 1.0 : 1.0));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float rstd = rsqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  mean_out[c] = mean;
  rstd_out[c] = rstd;
}

template <int CPL>
__global__ void __launch_bounds__(256) k_pfn_apply_max(const float* __restrict__ Y, const float* __restrict__ Ypad,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const int32_t* __restrict__ row_start,
                                                       const int32_t* __restrict__ num_points, int V, int P, int U,
                                                       float* __restrict__ A, float* __restrict__ Apad,
                                                       float* __restrict__ M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sc[CPL], sh[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
    sc[k] = c < U ? scale[c] : 0.f;
    sh[k] = c < U ? shift[c] : 0.f;
  }
  for (int v = blockIdx.x * kWavesPerBlock + wave; v < V; v += gridDim.x * kWavesPerBlock) {
    const int n = num_points[v];
    const int64_t rs = row_start[v];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      const float ap = fmaxf(Ypad[(int64_t)v * U + c] * sc[k] + sh[k], 0.f);
      float mx = n < P ? ap : -INFINITY;
      for (int j = 0; j < n; ++j) {
        const float a = fmaxf(Y[(rs + j) * U + c] * sc[k] + sh[k], 0.f);
        if (A) A[(rs + j) * U + c] = a;
        mx = fmaxf(mx, a);
      }
      if (Apad) Apad[(int64_t)v * U + c] = ap;
      M[(int64_t)v * U + c] = mx;
    }
  }
}

// backward 1: route dM through the max (first maximal row, padded row last — torch.max's first-index rule on
// the dense tensor), add the direct gradients, apply relu', accumulate the BatchNorm backward sums.
// dA (K, U) may be null (last layer); it is overwritten with dz.  SApad (V, U) may be null.
template <int CPL>
__global__ void __launch_bounds__(256) k_pfn_bwd_route(const float* __restrict__ Y, const float* __restrict__ Ypad,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       float* __restrict__ DZ /* in: dA or garbage, out: dz */,
                                                       int has_dA, const float* __restrict__ SApad,
                                                       const float* __restrict__ dM,
                                                       const int32_t* __restrict__ row_start,
                                                       const int32_t* __restrict__ num_points, int V, int P, int U,
                                                       float* __restrict__ DZpad, double* __restrict__ sums) {
  constexpr int UMAX = 64 * CPL;
  __shared__ double red[kWavesPerBlock][2 * UMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sc[CPL], sh[CPL], mu[CPL], rs_[CPL];
  double s1[CPL], s2[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < U;
    sc[k] = ok ? scale[c] : 0.f; sh[k] = ok ? shift[c] : 0.f; mu[k] = ok ? mean[c] : 0.f; rs_[k] = ok ? rstd[c] : 0.f;
    s1[k] = 0.0; s2[k] = 0.0;
  }
  for (int v = blockIdx.x * kWavesPerBlock + wave; v < V; v += gridDim.x * kWavesPerBlock) {
    const int n = num_points[v];
    const int64_t rs = row_start[v];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      const float yp = Ypad[(int64_t)v * U + c];
      const float ap = fmaxf(yp * sc[k] + sh[k], 0.f);
      // arg-max (first maximal real row; the padded rows sit after the real rows in the dense tensor)
      float mx = -INFINITY;
      int arg = -1;
      for (int j = 0; j < n; ++j) {
        const float a = fmaxf(Y[(rs + j) * U + c] * sc[k] + sh[k], 0.f);
        if (a > mx) { mx = a; arg = j; }
      }
      const bool pad_wins = (n < P) && (ap > mx);
      const float dm = dM[(int64_t)v * U + c];
      float f1 = 0.f, f2 = 0.f;
      for (int j = 0; j < n; ++j) {
        const float y = Y[(rs + j) * U + c];
        const float a = y * sc[k] + sh[k];
        float g = has_dA ? DZ[(rs + j) * U + c] : 0.f;
        if (!pad_wins && j == arg) g += dm;
        const float dz = a > 0.f ? g : 0.f;
        DZ[(rs + j) * U + c] = dz;
        f1 += dz;
        f2 += dz * (y - mu[k]) * rs_[k];
      }
      float gp = SApad ? SApad[(int64_t)v * U + c] : 0.f;
      if (pad_wins) gp += dm;
      const float dzp = (yp * sc[k] + sh[k]) > 0.f ? gp : 0.f;
      DZpad[(int64_t)v * U + c] = dzp;
      f1 += dzp;
      f2 += dzp * (yp - mu[k]) * rs_[k];
      s1[k] += f1;
      s2[k] += f2;
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    red[wave][lane + 64 * k] = s1[k];
    red[wave][UMAX + lane + 64 * k] = s2[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * UMAX; i += blockDim.x) {
    const int half = i / UMAX, c = i - half * UMAX;
    if (c < U) atomicAdd(&sums[half * U + c], red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// backward 2: BatchNorm backward per row (in place dz → dy), summed padded-row gradient, dt[v] = sum of both
template <int CPL>
__global__ void __launch_bounds__(256) k_pfn_bwd_bn(const float* __restrict__ Y, const float* __restrict__ Ypad,
                                                    float* __restrict__ DZ, float* __restrict__ DZpad,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const double* __restrict__ sums,
                                                    double count, int training,
                                                    const int32_t* __restrict__ row_start,
                                                    const int32_t* __restrict__ num_points, int V, int P, int U,
                                                    float* __restrict__ dT) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mu[CPL], rs_[CPL], gs[CPL], c1[CPL], c2[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < U;
    mu[k] = ok ? mean[c] : 0.f; rs_[k] = ok ? rstd[c] : 0.f; gs[k] = ok ? gamma[c] * rstd[c] : 0.f;
    c1[k] = (training && ok) ? (float)(sums[c] / count) : 0.f;
    c2[k] = (training && ok) ? (float)(sums[U + c] / count) : 0.f;
  }
  for (int v = blockIdx.x * kWavesPerBlock + wave; v < V; v += gridDim.x * kWavesPerBlock) {
    const int n = num_points[v];
    const int64_t rs = row_start[v];
    const float mult = (float)(P - n);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      float acc = 0.f;
      for (int j = 0; j < n; ++j) {
        const float xh = (Y[(rs + j) * U + c] - mu[k]) * rs_[k];
        const float dy = gs[k] * (DZ[(rs + j) * U + c] - c1[k] - xh * c2[k]);
        DZ[(rs + j) * U + c] = dy;
        acc += dy;
      }
      const float xhp = (Ypad[(int64_t)v * U + c] - mu[k]) * rs_[k];
      const float dyp = gs[k] * (DZpad[(int64_t)v * U + c] - mult * c1[k] - mult * xhp * c2[k]);
      DZpad[(int64_t)v * U + c] = dyp;
      if (dT) dT[(int64_t)v * U + c] = acc + dyp;
    }
  }
}

int grid_for(int64_t v) {
  const int64_t need = (v + kWavesPerBlock - 1) / kWavesPerBlock;
  return (int)(need < kBlocks ? (need > 0 ? need : 1) : kBlocks);
}

}  // namespace

#define MBV_PFN_DISPATCH(KERNEL, ...)                                                                  \
  if (units <= 64) hipLaunchKernelGGL((KERNEL<1>), dim3(grid_for(num_pillars)), dim3(256), 0, stream, __VA_ARGS__); \
  else hipLaunchKernelGGL((KERNEL<2>), dim3(grid_for(num_pillars)), dim3(256), 0, stream, __VA_ARGS__);

static int pfn_check(int64_t num_pillars, int32_t units, int32_t max_points) {
  if (num_pillars < 0 || max_points <= 0 || units <= 0) return MBV_ERR_BAD_ARG;
  if (units > 128) return MBV_ERR_UNSUPPORTED;
  return MBV_OK;
}

extern "C" int mbv_pfn_stats(float* y, const float* t, float* y_pad, const int32_t* row_start,
                             const int32_t* num_points, int64_t num_pillars, int32_t units, int32_t max_points,
                             double* sums, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (int rc = pfn_check(num_pillars, units, max_points)) return rc;
  if (!sums) return MBV_ERR_BAD_ARG;
  MBV_CHECK_HIP(mbv_fill_async(sums, 0, sizeof(double) * 2 * units, stream));
  if (num_pillars == 0) return MBV_OK;
  if (!y || !y_pad || !row_start || !num_points) return MBV_ERR_BAD_ARG;
  MBV_PFN_DISPATCH(k_pfn_stats, y, t, y_pad, row_start, num_points, (int)num_pillars, max_points, units, sums)
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_pfn_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                                   float momentum, int32_t training, float* running_mean, float* running_var,
                                   int32_t units, float* scale, float* shift, float* mean, float* rstd,
                                   void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (units <= 0 || count <= 0) return MBV_ERR_BAD_ARG;
  if (!sums || !gamma || !beta || !running_mean || !running_var || !scale || !shift || !mean || !rstd)
    return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_pfn_finalize, dim3((units + 63) / 64), dim3(64), 0, stream, sums, count, gamma, beta, eps,
                     momentum, training, running_mean, running_var, units, scale, shift, mean, rstd);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_pfn_apply_max(const float* y, const float* y_pad, const float* scale, const float* shift,
                                 const int32_t* row_start, const int32_t* num_points, int64_t num_pillars,
                                 int32_t units, int32_t max_points, float* a, float* a_pad, float* m,
                                 void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (int rc = pfn_check(num_pillars, units, max_points)) return rc;
  if (num_pillars == 0) return MBV_OK;
  if (!y || !y_pad || !scale || !shift || !row_start || !num_points || !m) return MBV_ERR_BAD_ARG;
  MBV_PFN_DISPATCH(k_pfn_apply_max, y, y_pad, scale, shift, row_start, num_points, (int)num_pillars, max_points, units,
                   a, a_pad, m)
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_pfn_bwd_route(const float* y, const float* y_pad, const float* scale, const float* shift,
                                 const float* mean, const float* rstd, float* dz, int32_t has_da,
                                 const float* sum_da_pad, const float* dm, const int32_t* row_start,
                                 const int32_t* num_points, int64_t num_pillars, int32_t units, int32_t max_points,
                                 float* dz_pad, double* sums, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (int rc = pfn_check(num_pillars, units, max_points)) return rc;
  if (!sums) return MBV_ERR_BAD_ARG;
  MBV_CHECK_HIP(mbv_fill_async(sums, 0, sizeof(double) * 2 * units, stream));
  if (num_pillars == 0) return MBV_OK;
  if (!y || !y_pad || !scale || !shift || !mean || !rstd || !dz || !dm || !row_start || !num_points || !dz_pad)
    return MBV_ERR_BAD_ARG;
  MBV_PFN_DISPATCH(k_pfn_bwd_route, y, y_pad, scale, shift, mean, rstd, dz, has_da, sum_da_pad, dm, row_start,
                   num_points, (int)num_pillars, max_points, units, dz_pad, sums)
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_pfn_bwd_bn(const float* y, const float* y_pad, float* dz, float* dz_pad, const float* mean,
                              const float* rstd, const float* gamma, const double* sums, double count,
                              int32_t training, const int32_t* row_start, const int32_t* num_points,
                              int64_t num_pillars, int32_t units, int32_t max_points, float* dt, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (int rc = pfn_check(num_pillars, units, max_points)) return rc;
  if (num_pillars == 0) return MBV_OK;
  if (!y || !y_pad || !dz || !dz_pad || !mean || !rstd || !gamma || !sums || !row_start || !num_points)
    return MBV_ERR_BAD_ARG;
  MBV_PFN_DISPATCH(k_pfn_bwd_bn, y, y_pad, dz, dz_pad, mean, rstd, gamma, sums, count, training, row_start, num_points,
                   (int)num_pillars, max_points, units, dt)
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
