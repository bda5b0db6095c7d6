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
constexpr int kBlocks = 2048;     // persistent-style grid: one set of channel atomics per workgroup
constexpr int kChunk = 8;         // rows of a pillar in flight per wave
constexpr int kChunk128 = 8;      // ... of the 4-channel lane map at 128 units (4 — fewer registers, three waves per SIMD — measured 4-8 % slower)

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
  // Rows are read in chunks of kChunk with all their loads issued before the first use (and before the stores
  // that follow, which the compiler must otherwise order against later loads of Y): a pillar holds ≈ 5 rows, so
  // the walk costs about one memory round trip per pillar instead of one per row and channel half.  The header
  // (n, row_start) of the wave's next pillar is fetched while the current one is processed.
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  int n_next = v < V ? num_points[v] : 0;
  int64_t rs_next = v < V ? (int64_t)row_start[v] : 0;
  for (; v < V; v += vstride) {
    const int n = n_next;
    const int64_t rs = rs_next;
    if (v + vstride < V) { n_next = num_points[v + vstride]; rs_next = row_start[v + vstride]; }
    const float mult = (float)(P - n);
    float t[CPL], yp[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      t[k] = (T && c < U) ? T[(int64_t)v * U + c] : 0.f;
      yp[k] = c < U ? Ypad[(int64_t)v * U + c] : 0.f;
    }
    for (int j0 = 0; j0 < n; j0 += kChunk) {
      float y[kChunk][CPL];
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          y[j][k] = (j0 + j < n && c < U) ? Y[(rs + j0 + j) * U + c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (j0 + j < n && c < U) {
            float yy = y[j][k];
            if (T) { yy += t[k]; Y[(rs + j0 + j) * U + c] = yy; }
            s[k] += yy;
            q[k] += yy * yy;
          }
        }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      float ypp = yp[k];
      if (T) { ypp += t[k]; Ypad[(int64_t)v * U + c] = ypp; }
      s[k] += mult * ypp;
      q[k] += mult * ypp * ypp;
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

// Σ over the rows of (nparts, 2 U) partial sums → one row, in a fixed order: a workgroup takes 32 columns, 32 part-lanes each
// (lane j adds the parts congruent to j modulo 32), then the lanes' sums meet in LDS.
__global__ void __launch_bounds__(1024) k_pfn_fold_parts(const double* __restrict__ parts, int nparts, int U2,
                                                         double* __restrict__ out) {
  __shared__ double red[32][33];
  const int c = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  double a = 0.0;
  if (col < U2)
    for (int p = j; p < nparts; p += 32) a += parts[(long)p * U2 + col];
  red[j][c] = a;
  __syncthreads();
  if (j == 0 && col < U2) {
    double s = 0.0;
    for (int k = 0; k < 32; ++k) s += red[k][c];
    out[col] = s;
  }
}

// mean / var → scale, shift (+ running-stat update); one workgroup
__global__ void k_pfn_finalize(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float eps, float momentum, int training,
                               float* __restrict__ running_mean, float* __restrict__ running_var, int U,
                               float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                               float* __restrict__ rstd_out, int nparts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= U) return;
  float mean, var;
  if (training) {
    double s0 = 0.0, s1 = 0.0;                     // nparts rows of (2 U) partial sums, added in a fixed order
    for (int p = 0; p < nparts; ++p) { s0 += sums[(long)p * 2 * U + c]; s1 += sums[(long)p * 2 * U + U + c]; }
    const double m = s0 / count;
    double vv = s1 / count - m * m;
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
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  int n_next = v < V ? num_points[v] : 0;
  int64_t rs_next = v < V ? (int64_t)row_start[v] : 0;
  for (; v < V; v += vstride) {
    const int n = n_next;
    const int64_t rs = rs_next;
    if (v + vstride < V) { n_next = num_points[v + vstride]; rs_next = row_start[v + vstride]; }
    float ap[CPL], mx[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      ap[k] = c < U ? fmaxf(Ypad[(int64_t)v * U + c] * sc[k] + sh[k], 0.f) : 0.f;
      mx[k] = n < P ? ap[k] : -INFINITY;
    }
    for (int j0 = 0; j0 < n; j0 += kChunk) {
      float y[kChunk][CPL];
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          y[j][k] = (j0 + j < n && c < U) ? Y[(rs + j0 + j) * U + c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (j0 + j < n && c < U) {
            const float a = fmaxf(y[j][k] * sc[k] + sh[k], 0.f);
            if (A) A[(rs + j0 + j) * U + c] = a;
            mx[k] = fmaxf(mx[k], a);
          }
        }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      if (Apad) Apad[(int64_t)v * U + c] = ap[k];
      M[(int64_t)v * U + c] = mx[k];
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
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  int n_next = v < V ? num_points[v] : 0;
  int64_t rs_next = v < V ? (int64_t)row_start[v] : 0;
  for (; v < V; v += vstride) {
    const int n = n_next;
    const int64_t rs = rs_next;
    if (v + vstride < V) { n_next = num_points[v + vstride]; rs_next = row_start[v + vstride]; }
    float yp[CPL], ap[CPL], dm[CPL], gp[CPL], mx[CPL];
    int arg[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      const bool ok = c < U;
      yp[k] = ok ? Ypad[(int64_t)v * U + c] : 0.f;
      dm[k] = ok ? dM[(int64_t)v * U + c] : 0.f;
      gp[k] = (ok && SApad) ? SApad[(int64_t)v * U + c] : 0.f;
      ap[k] = fmaxf(yp[k] * sc[k] + sh[k], 0.f);
      mx[k] = -INFINITY;
      arg[k] = -1;
    }
    // pass 1: arg-max (first maximal real row; the padded rows sit after the real rows in the dense tensor).
    // Pillars of up to kChunk rows (almost all of them) keep their rows in registers for pass 2.
    float y0[kChunk][CPL];
    for (int j0 = 0; j0 < n; j0 += kChunk) {
      float y[kChunk][CPL];
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          y[j][k] = (j0 + j < n && c < U) ? Y[(rs + j0 + j) * U + c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          if (j0 == 0) y0[j][k] = y[j][k];
          if (j0 + j < n) {
            const float a = fmaxf(y[j][k] * sc[k] + sh[k], 0.f);
            if (a > mx[k]) { mx[k] = a; arg[k] = j0 + j; }
          }
        }
    }
    bool pad_wins[CPL];
    float f1[CPL], f2[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) { pad_wins[k] = (n < P) && (ap[k] > mx[k]); f1[k] = 0.f; f2[k] = 0.f; }
    // pass 2: dz = relu'(a) * (dA + routed dM), BatchNorm-backward sums
    for (int j0 = 0; j0 < n; j0 += kChunk) {
      float y[kChunk][CPL], g[kChunk][CPL];
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          const bool ok = j0 + j < n && c < U;
          y[j][k] = j0 == 0 ? y0[j][k] : (ok ? Y[(rs + j0 + j) * U + c] : 0.f);
          g[j][k] = (ok && has_dA) ? DZ[(rs + j0 + j) * U + c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (j0 + j < n && c < U) {
            const float yy = y[j][k];
            const float a = yy * sc[k] + sh[k];
            float gg = g[j][k];
            if (!pad_wins[k] && j0 + j == arg[k]) gg += dm[k];
            const float dz = a > 0.f ? gg : 0.f;
            DZ[(rs + j0 + j) * U + c] = dz;
            f1[k] += dz;
            f2[k] += dz * (yy - mu[k]) * rs_[k];
          }
        }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      float gpp = gp[k];
      if (pad_wins[k]) gpp += dm[k];
      const float dzp = (yp[k] * sc[k] + sh[k]) > 0.f ? gpp : 0.f;
      DZpad[(int64_t)v * U + c] = dzp;
      f1[k] += dzp;
      f2[k] += dzp * (yp[k] - mu[k]) * rs_[k];
      s1[k] += f1[k];
      s2[k] += f2[k];
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
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  int n_next = v < V ? num_points[v] : 0;
  int64_t rs_next = v < V ? (int64_t)row_start[v] : 0;
  for (; v < V; v += vstride) {
    const int n = n_next;
    const int64_t rs = rs_next;
    if (v + vstride < V) { n_next = num_points[v + vstride]; rs_next = row_start[v + vstride]; }
    const float mult = (float)(P - n);
    float acc[CPL], ypv[CPL], dzpv[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      acc[k] = 0.f;
      ypv[k] = c < U ? Ypad[(int64_t)v * U + c] : 0.f;
      dzpv[k] = c < U ? DZpad[(int64_t)v * U + c] : 0.f;
    }
    for (int j0 = 0; j0 < n; j0 += kChunk) {
      float y[kChunk][CPL], dz[kChunk][CPL];
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          const bool ok = j0 + j < n && c < U;
          y[j][k] = ok ? Y[(rs + j0 + j) * U + c] : 0.f;
          dz[j][k] = ok ? DZ[(rs + j0 + j) * U + c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (j0 + j < n && c < U) {
            const float xh = (y[j][k] - mu[k]) * rs_[k];
            const float dy = gs[k] * (dz[j][k] - c1[k] - xh * c2[k]);
            DZ[(rs + j0 + j) * U + c] = dy;
            acc[k] += dy;
          }
        }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= U) continue;
      const float xhp = (ypv[k] - mu[k]) * rs_[k];
      const float dyp = gs[k] * (dzpv[k] - mult * c1[k] - mult * xhp * c2[k]);
      DZpad[(int64_t)v * U + c] = dyp;
      if (dT) dT[(int64_t)v * U + c] = acc[k] + dyp;
    }
  }
}

// backward 2 with a lane = 4 ADJACENT channels: a wave instruction covers 64 / (U / 4) whole rows (4 at 64 units, 2 at 128) as
// 16-byte accesses instead of one row as 4-byte accesses — a quarter of the memory instructions for the same bytes (these walks
// are bound by vector-memory instructions in flight).  Row loads are unconditional on a clamped row index; the row
// groups' partial sums of a pillar meet through two shuffles.  U4 = U / 4 in {8, 16, 32}.
template <int U4, int CH>
__global__ void __launch_bounds__(256) k_pfn_bwd_bn_v4(const float* __restrict__ Y, const float* __restrict__ Ypad,
                                                       float* __restrict__ DZ, float* __restrict__ DZpad,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const double* __restrict__ sums,
                                                       double count, int training,
                                                       const int32_t* __restrict__ row_start,
                                                       const int32_t* __restrict__ num_points, int V, int P,
                                                       float* __restrict__ dT) {
  constexpr int U = 4 * U4, RPW = 64 / U4, NI = CH / RPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % U4, rg = lane / U4, c = 4 * sub;
  const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs4 = *reinterpret_cast<const float4*>(rstd + c);
  const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
  const float4 gs = make_float4(gm.x * rs4.x, gm.y * rs4.y, gm.z * rs4.z, gm.w * rs4.w);
  float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f), c2 = c1;
  if (training) {
    c1 = make_float4((float)(sums[c] / count), (float)(sums[c + 1] / count), (float)(sums[c + 2] / count), (float)(sums[c + 3] / count));
    c2 = make_float4((float)(sums[U + c] / count), (float)(sums[U + c + 1] / count), (float)(sums[U + c + 2] / count),
                     (float)(sums[U + c + 3] / count));
  }
  const float4* Y4 = reinterpret_cast<const float4*>(Y);
  float4* DZ4 = reinterpret_cast<float4*>(DZ);
  const float4* YP4 = reinterpret_cast<const float4*>(Ypad);
  float4* DP4 = reinterpret_cast<float4*>(DZpad);
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  if (v >= V) return;
  // Two pillars deep: while pillar v is computed and stored, the first 8 rows (and the padded row) of the wave's NEXT pillar
  // are already in flight — the compiler cannot hoist them above this pillar's stores itself (DZ is updated in place) — and the
  // header (n, row_start) of the pillar after that is requested.  Every load is unconditional on a clamped index.
  auto load_rows = [&](int64_t rs, int n, int j0, float4 (&y)[NI], float4 (&dz)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = j0 + i * RPW + rg;
      const int64_t at = (rs + (r < n ? r : (n > 0 ? n - 1 : 0))) * U4 + sub;
      y[i] = Y4[at];
      dz[i] = DZ4[at];
    }
  };
  const int vlast = V - 1;
  int n = num_points[v];
  int64_t rs = row_start[v];
  int v1 = v + vstride < V ? v + vstride : vlast;
  int n1 = num_points[v1];
  int64_t rs1 = row_start[v1];
  float4 y[NI], dz[NI], yn[NI], dzn[NI];
  load_rows(rs, n, 0, y, dz);
  float4 ypv = YP4[(int64_t)v * U4 + sub], dzpv = DP4[(int64_t)v * U4 + sub];
  for (; v < V; v += vstride) {
    // header two pillars ahead, rows + padded row one pillar ahead
    const int v2 = v + 2 * vstride < V ? v + 2 * vstride : vlast;
    const int n2 = num_points[v2];
    const int64_t rs2 = row_start[v2];
    load_rows(rs1, n1, 0, yn, dzn);
    const float4 ypn = YP4[(int64_t)v1 * U4 + sub], dzpn = DP4[(int64_t)v1 * U4 + sub];
    const float mult = (float)(P - n);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < n; j0 += CH) {
      if (j0) load_rows(rs, n, j0, y, dz);               // pillars of more than 8 rows: the later chunks as they come
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = j0 + i * RPW + rg;
        if (r < n) {
          float4 dy;
          dy.x = gs.x * (dz[i].x - c1.x - (y[i].x - mu.x) * rs4.x * c2.x);
          dy.y = gs.y * (dz[i].y - c1.y - (y[i].y - mu.y) * rs4.y * c2.y);
          dy.z = gs.z * (dz[i].z - c1.z - (y[i].z - mu.z) * rs4.z * c2.z);
          dy.w = gs.w * (dz[i].w - c1.w - (y[i].w - mu.w) * rs4.w * c2.w);
          DZ4[(rs + r) * U4 + sub] = dy;
          acc.x += dy.x; acc.y += dy.y; acc.z += dy.z; acc.w += dy.w;
        }
      }
    }
#pragma unroll
    for (int o = U4; o < 64; o <<= 1) {
      acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
      acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (rg == 0) {
      float4 dyp;
      dyp.x = gs.x * (dzpv.x - mult * c1.x - mult * (ypv.x - mu.x) * rs4.x * c2.x);
      dyp.y = gs.y * (dzpv.y - mult * c1.y - mult * (ypv.y - mu.y) * rs4.y * c2.y);
      dyp.z = gs.z * (dzpv.z - mult * c1.z - mult * (ypv.z - mu.z) * rs4.z * c2.z);
      dyp.w = gs.w * (dzpv.w - mult * c1.w - mult * (ypv.w - mu.w) * rs4.w * c2.w);
      DP4[(int64_t)v * U4 + sub] = dyp;
      if (dT) *reinterpret_cast<float4*>(dT + (int64_t)v * U + c) = make_float4(acc.x + dyp.x, acc.y + dyp.y, acc.z + dyp.z, acc.w + dyp.w);
    }
    n = n1; rs = rs1; n1 = n2; rs1 = rs2; v1 = v2;
#pragma unroll
    for (int i = 0; i < NI; ++i) { y[i] = yn[i]; dz[i] = dzn[i]; }
    ypv = ypn; dzpv = dzpn;
  }
}

// apply + max with the 4-channel lane map of k_pfn_bwd_bn_v4 (a wave instruction = 64 / U4 whole rows, 16-byte accesses,
// the wave's next pillar in flight while this one is computed); the row groups' maxima meet through shuffles.
template <int U4, int CH>
__global__ void __launch_bounds__(256) k_pfn_apply_max_v4(const float* __restrict__ Y, const float* __restrict__ Ypad,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const int32_t* __restrict__ row_start,
                                                          const int32_t* __restrict__ num_points, int V, int P,
                                                          float* __restrict__ A, float* __restrict__ Apad,
                                                          float* __restrict__ M) {
  constexpr int RPW = 64 / U4, NI = CH / RPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % U4, rg = lane / U4, c = 4 * sub;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
  const float4* Y4 = reinterpret_cast<const float4*>(Y);
  const float4* YP4 = reinterpret_cast<const float4*>(Ypad);
  float4* A4 = reinterpret_cast<float4*>(A);
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  if (v >= V) return;
  auto load_rows = [&](int64_t rs, int n, int j0, float4 (&y)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = j0 + i * RPW + rg;
      y[i] = Y4[(rs + (r < n ? r : (n > 0 ? n - 1 : 0))) * U4 + sub];
    }
  };
  auto act = [&](const float4& y) {
    return make_float4(fmaxf(y.x * sc.x + sh.x, 0.f), fmaxf(y.y * sc.y + sh.y, 0.f), fmaxf(y.z * sc.z + sh.z, 0.f),
                       fmaxf(y.w * sc.w + sh.w, 0.f));
  };
  const int vlast = V - 1;
  int n = num_points[v];
  int64_t rs = row_start[v];
  int v1 = v + vstride < V ? v + vstride : vlast;
  int n1 = num_points[v1];
  int64_t rs1 = row_start[v1];
  float4 y[NI], yn[NI];
  load_rows(rs, n, 0, y);
  float4 ypv = YP4[(int64_t)v * U4 + sub];
  for (; v < V; v += vstride) {
    const int v2 = v + 2 * vstride < V ? v + 2 * vstride : vlast;
    const int n2 = num_points[v2];
    const int64_t rs2 = row_start[v2];
    load_rows(rs1, n1, 0, yn);
    const float4 ypn = YP4[(int64_t)v1 * U4 + sub];
    const float4 ap = act(ypv);
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int j0 = 0; j0 < n; j0 += CH) {
      if (j0) load_rows(rs, n, j0, y);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = j0 + i * RPW + rg;
        if (r < n) {
          const float4 a = act(y[i]);
          if (A) A4[(rs + r) * U4 + sub] = a;
          mx.x = fmaxf(mx.x, a.x); mx.y = fmaxf(mx.y, a.y); mx.z = fmaxf(mx.z, a.z); mx.w = fmaxf(mx.w, a.w);
        }
      }
    }
#pragma unroll
    for (int o = U4; o < 64; o <<= 1) {
      mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
      mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
    }
    if (rg == 0) {
      if (n < P) { mx.x = fmaxf(mx.x, ap.x); mx.y = fmaxf(mx.y, ap.y); mx.z = fmaxf(mx.z, ap.z); mx.w = fmaxf(mx.w, ap.w); }
      if (Apad) *reinterpret_cast<float4*>(Apad + (int64_t)v * (4 * U4) + c) = ap;
      *reinterpret_cast<float4*>(M + (int64_t)v * (4 * U4) + c) = mx;
    }
    n = n1; rs = rs1; n1 = n2; rs1 = rs2; v1 = v2;
#pragma unroll
    for (int i = 0; i < NI; ++i) y[i] = yn[i];
    ypv = ypn;
  }
}

// backward 1 (route dM through the max, relu', BatchNorm-backward sums) with the same lane map and pillar pipeline.  The
// arg-max of a channel is the FIRST maximal real row (torch.max's first-index rule on the dense tensor): inside a lane rows
// arrive in ascending order (strict >), across the row groups a tie goes to the smaller row index.
template <int U4, int CH>
__global__ void __launch_bounds__(256) k_pfn_bwd_route_v4(const float* __restrict__ Y, const float* __restrict__ Ypad,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ DZ, int has_dA, const float* __restrict__ SApad,
                                                          const float* __restrict__ dM,
                                                          const int32_t* __restrict__ row_start,
                                                          const int32_t* __restrict__ num_points, int V, int P,
                                                          float* __restrict__ DZpad, double* __restrict__ sums) {
  constexpr int U = 4 * U4, RPW = 64 / U4, NI = CH / RPW;
  __shared__ double red[kWavesPerBlock][2 * U];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % U4, rg = lane / U4, c = 4 * sub;
  const float4 sc4 = *reinterpret_cast<const float4*>(scale + c), sh4 = *reinterpret_cast<const float4*>(shift + c);
  const float4 mu4 = *reinterpret_cast<const float4*>(mean + c), rs4 = *reinterpret_cast<const float4*>(rstd + c);
  const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
  const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rsd[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  const float4* Y4 = reinterpret_cast<const float4*>(Y);
  float4* DZ4 = reinterpret_cast<float4*>(DZ);
  const float4* YP4 = reinterpret_cast<const float4*>(Ypad);
  const float4* DM4 = reinterpret_cast<const float4*>(dM);
  const float4* SA4 = reinterpret_cast<const float4*>(SApad ? SApad : dM);       // some readable address when absent
  const int vstride = gridDim.x * kWavesPerBlock;
  int v = blockIdx.x * kWavesPerBlock + wave;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (v < V) {
    auto load_rows = [&](int64_t rs, int n, int j0, float4 (&y)[NI], float4 (&g)[NI], bool want_g) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = j0 + i * RPW + rg;
        const int64_t at = (rs + (r < n ? r : (n > 0 ? n - 1 : 0))) * U4 + sub;
        y[i] = Y4[at];
        if (want_g) g[i] = DZ4[at];
      }
    };
    const bool dA = has_dA != 0;
    const int vlast = V - 1;
    int n = num_points[v];
    int64_t rs = row_start[v];
    int v1 = v + vstride < V ? v + vstride : vlast;
    int n1 = num_points[v1];
    int64_t rs1 = row_start[v1];
    float4 y[NI], g[NI], yn[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) g[i] = zero4;
    load_rows(rs, n, 0, y, g, false);
    for (; v < V; v += vstride) {
      const int v2 = v + 2 * vstride < V ? v + 2 * vstride : vlast;
      const int n2 = num_points[v2];
      const int64_t rs2 = row_start[v2];
      // (this pillar's padded row, dM and dA_pad are asked for here, not a pillar ahead: they are first used behind pass 1,
      // and the twelve registers of a second set cost the kernel a wave per SIMD)
      const float4 ypv = YP4[(int64_t)v * U4 + sub], dmv = DM4[(int64_t)v * U4 + sub], gpv = SA4[(int64_t)v * U4 + sub];
      if (dA) {                                          // dA of THIS pillar: used in pass 2, in flight during pass 1
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = i * RPW + rg;
          g[i] = DZ4[(rs + (r < n ? r : (n > 0 ? n - 1 : 0))) * U4 + sub];
        }
      }
      {
        float4 gdummy[NI];
        load_rows(rs1, n1, 0, yn, gdummy, false);        // the next pillar's rows of y
      }
      const float yp[4] = {ypv.x, ypv.y, ypv.z, ypv.w}, dm[4] = {dmv.x, dmv.y, dmv.z, dmv.w};
      const float gp[4] = {SApad ? gpv.x : 0.f, SApad ? gpv.y : 0.f, SApad ? gpv.z : 0.f, SApad ? gpv.w : 0.f};
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int arg[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
      // pass 1: arg-max over the real rows (the first chunk stays in registers for pass 2)
      {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = i * RPW + rg;
          if (r < n) {
            const float yy[4] = {y[i].x, y[i].y, y[i].z, y[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float a = fmaxf(yy[k] * sc[k] + sh[k], 0.f);
              if (a > mx[k]) { mx[k] = a; arg[k] = r; }
            }
          }
        }
        for (int j0 = CH; j0 < n; j0 += CH) {
          float4 yl[NI], gl[NI];
          load_rows(rs, n, j0, yl, gl, false);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int r = j0 + i * RPW + rg;
            if (r < n) {
              const float yy[4] = {yl[i].x, yl[i].y, yl[i].z, yl[i].w};
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float a = fmaxf(yy[k] * sc[k] + sh[k], 0.f);
                if (a > mx[k]) { mx[k] = a; arg[k] = r; }
              }
            }
          }
        }
      }
#pragma unroll
      for (int o = U4; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float om = __shfl_xor(mx[k], o, 64);
          const int oa = __shfl_xor(arg[k], o, 64);
          if (om > mx[k] || (om == mx[k] && oa < arg[k])) { mx[k] = om; arg[k] = oa; }
        }
      }
      bool pad_wins[4];
      float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) pad_wins[k] = (n < P) && (fmaxf(yp[k] * sc[k] + sh[k], 0.f) > mx[k]);
      // pass 2: dz = relu'(a) * (dA + routed dM), BatchNorm-backward sums
      for (int j0 = 0; j0 < n; j0 += CH) {
        if (j0) load_rows(rs, n, j0, y, g, dA);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = j0 + i * RPW + rg;
          if (r < n) {
            const float yy[4] = {y[i].x, y[i].y, y[i].z, y[i].w};
            const float gi[4] = {g[i].x, g[i].y, g[i].z, g[i].w};
            float dz[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float a = yy[k] * sc[k] + sh[k];
              float gg = dA ? gi[k] : 0.f;
              if (!pad_wins[k] && r == arg[k]) gg += dm[k];
              dz[k] = a > 0.f ? gg : 0.f;
              f1[k] += dz[k];
              f2[k] += dz[k] * (yy[k] - mu[k]) * rsd[k];
            }
            DZ4[(rs + r) * U4 + sub] = make_float4(dz[0], dz[1], dz[2], dz[3]);
          }
        }
      }
      if (rg == 0) {                                   // the padded row: once per pillar
        float dzp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float gpp = gp[k];
          if (pad_wins[k]) gpp += dm[k];
          dzp[k] = (yp[k] * sc[k] + sh[k]) > 0.f ? gpp : 0.f;
          f1[k] += dzp[k];
          f2[k] += dzp[k] * (yp[k] - mu[k]) * rsd[k];
        }
        *reinterpret_cast<float4*>(DZpad + (int64_t)v * U + c) = make_float4(dzp[0], dzp[1], dzp[2], dzp[3]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[k] += f1[k]; s2[k] += f2[k]; }
      n = n1; rs = rs1; n1 = n2; rs1 = rs2; v1 = v2;
#pragma unroll
      for (int i = 0; i < NI; ++i) y[i] = yn[i];
    }
  }
  // the row groups' channel sums meet through shuffles, the waves' in LDS; one set of channel atomics per workgroup
#pragma unroll
  for (int o = U4; o < 64; o <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[k] += __shfl_xor(s1[k], o, 64); s2[k] += __shfl_xor(s2[k], o, 64); }
  if (rg == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[wave][c + k] = s1[k]; red[wave][U + c + k] = s2[k]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * U; i += blockDim.x) atomicAdd(&sums[i], red[0][i] + red[1][i] + red[2][i] + red[3][i]);
}

int grid_for(int64_t v) {
  const int64_t need = (v + kWavesPerBlock - 1) / kWavesPerBlock;
  return (int)(need < kBlocks ? (need > 0 ? need : 1) : kBlocks);
}

}  // namespace

// The 4-channel lane map (k_pfn_*_v4) takes 32 / 64 / 128 units and 16-byte aligned tensors (null = absent).
template <typename... Ps>
static bool pfn_v4_ok(int32_t units, Ps... ptrs) {
  size_t bits = 0;
  ((bits |= reinterpret_cast<size_t>(ptrs)), ...);
  return (bits & 15) == 0 && (units == 32 || units == 64 || units == 128);
}

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
                     momentum, training, running_mean, running_var, units, scale, shift, mean, rstd, 1);
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
  if (pfn_v4_ok(units, y, y_pad, scale, shift, a, a_pad, m)) {
    const dim3 grid(grid_for(num_pillars)), block(256);
#define MBV_V4(U4, CH) hipLaunchKernelGGL((k_pfn_apply_max_v4<U4, CH>), grid, block, 0, stream, y, y_pad, scale, shift, row_start, \
                                      num_points, (int)num_pillars, max_points, a, a_pad, m)
    if (units == 32) MBV_V4(8, 8); else if (units == 64) MBV_V4(16, 8); else MBV_V4(32, kChunk128);
#undef MBV_V4
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
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
  if (pfn_v4_ok(units, y, y_pad, scale, shift, mean, rstd, dz, sum_da_pad, dm, dz_pad)) {
    const dim3 grid(grid_for(num_pillars)), block(256);
#define MBV_V4(U4, CH) hipLaunchKernelGGL((k_pfn_bwd_route_v4<U4, CH>), grid, block, 0, stream, y, y_pad, scale, shift, mean, rstd, dz, \
                                      has_da, sum_da_pad, dm, row_start, num_points, (int)num_pillars, max_points, dz_pad, sums)
    if (units == 32) MBV_V4(8, 8); else if (units == 64) MBV_V4(16, 8); else MBV_V4(32, kChunk128);
#undef MBV_V4
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
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
  if (pfn_v4_ok(units, y, y_pad, dz, dz_pad, mean, rstd, gamma, dt)) {
    const dim3 grid(grid_for(num_pillars)), block(256);
    if (units == 32)
      hipLaunchKernelGGL((k_pfn_bwd_bn_v4<8, 8>), grid, block, 0, stream, y, y_pad, dz, dz_pad, mean, rstd, gamma, sums, count,
                         training, row_start, num_points, (int)num_pillars, max_points, dt);
    else if (units == 64)
      hipLaunchKernelGGL((k_pfn_bwd_bn_v4<16, 8>), grid, block, 0, stream, y, y_pad, dz, dz_pad, mean, rstd, gamma, sums, count,
                         training, row_start, num_points, (int)num_pillars, max_points, dt);
    else
      hipLaunchKernelGGL((k_pfn_bwd_bn_v4<32, kChunk128>), grid, block, 0, stream, y, y_pad, dz, dz_pad, mean, rstd, gamma, sums, count,
                         training, row_start, num_points, (int)num_pillars, max_points, dt);
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
  MBV_PFN_DISPATCH(k_pfn_bwd_bn, y, y_pad, dz, dz_pad, mean, rstd, gamma, sums, count, training, row_start, num_points,
                   (int)num_pillars, max_points, units, dt)
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// ---- BatchNorm statistics as a STREAM ------------------------------------------------------------------------------------
// With the pillar term already inside y (K2c's gathered addend), Σ y and Σ y² over the real rows no longer need the pillar
// structure: a column-sum pass at streaming rate (k_pfn_stats walks pillar by pillar, lane = channel, and rewrites y: 1.1-2.2
// TB/s).  A thread owns 4 adjacent channels; f32 partials are folded into f64 every 64 rows; every block leaves one row of
// (2 U) doubles — added in a fixed order by k_pfn_finalize, no atomics.  The last blocks take the padded rows instead: pillar p's
// representative row y_pad[p] (+ t[p], stored back) counts P - n_p times.
constexpr int kStatRowBlocks = 512, kStatPadBlocks = 256, kStatParts = kStatRowBlocks + kStatPadBlocks;

__global__ void __launch_bounds__(256) k_pfn_colstats(const float* __restrict__ Y, long K, float* __restrict__ Ypad,
                                                      const float* __restrict__ T, const int* __restrict__ num_points,
                                                      long V, int U, int P, double* __restrict__ parts) {
  __shared__ double red[2][256][4];
  const int u4 = U >> 2;                                  // threads per row
  const int sub = threadIdx.x % u4, lanes_rows = 256 / u4, rsub = threadIdx.x / u4;
  const bool active = rsub < lanes_rows;
  double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  int since = 0;
  auto fold = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) { ds[k] += s[k]; dq[k] += q[k]; s[k] = 0.f; q[k] = 0.f; }
    since = 0;
  };
  if ((int)blockIdx.x < kStatRowBlocks) {
    const long per = (K + kStatRowBlocks - 1) / kStatRowBlocks;
    const long r0 = (long)blockIdx.x * per, r1 = r0 + per < K ? r0 + per : K;
    if (active) {
      for (long r = r0 + rsub; r < r1; r += 4L * lanes_rows) {          // four rows in flight per thread
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long rr = r + (long)j * lanes_rows;
          v[j] = rr < r1 ? *reinterpret_cast<const float4*>(Y + rr * U + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[0] += v[j].x; s[1] += v[j].y; s[2] += v[j].z; s[3] += v[j].w;
          q[0] += v[j].x * v[j].x; q[1] += v[j].y * v[j].y; q[2] += v[j].z * v[j].z; q[3] += v[j].w * v[j].w;
        }
        if (++since == 16) fold();
      }
    }
  } else {
    const int pb = (int)blockIdx.x - kStatRowBlocks;
    const long per = (V + kStatPadBlocks - 1) / kStatPadBlocks;
    const long p0 = (long)pb * per, p1 = p0 + per < V ? p0 + per : V;
    if (active) {
      for (long p = p0 + rsub; p < p1; p += 4L * lanes_rows) {             // four pillars in flight per thread
        float4 v[4], t[4];
        int np[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long pp = p + (long)j * lanes_rows;
          const bool in = pp < p1;
          v[j] = in ? *reinterpret_cast<const float4*>(Ypad + pp * U + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
          t[j] = (in && T) ? *reinterpret_cast<const float4*>(T + pp * U + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
          np[j] = in ? num_points[pp] : P;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long pp = p + (long)j * lanes_rows;
          v[j].x += t[j].x; v[j].y += t[j].y; v[j].z += t[j].z; v[j].w += t[j].w;
          if (T && pp < p1) *reinterpret_cast<float4*>(Ypad + pp * U + 4 * sub) = v[j];
          const float mult = (float)(P - np[j]);
          s[0] += mult * v[j].x; s[1] += mult * v[j].y; s[2] += mult * v[j].z; s[3] += mult * v[j].w;
          q[0] += mult * v[j].x * v[j].x; q[1] += mult * v[j].y * v[j].y; q[2] += mult * v[j].z * v[j].z;
          q[3] += mult * v[j].w * v[j].w;
        }
        if (++since == 16) fold();
      }
    }
  }
  fold();
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][threadIdx.x][k] = ds[k]; red[1][threadIdx.x][k] = dq[k]; }
  __syncthreads();
  // the threads of one channel group: sub, sub + u4, ...: channel c = 4 sub + k
  for (int i = threadIdx.x; i < 2 * U; i += 256) {
    const int half = i / U, c = i - half * U, sb = c >> 2, k = c & 3;
    double a = 0.0;
    for (int j = 0; j < lanes_rows; ++j) a += red[half][j * u4 + sb][k];
    parts[(long)blockIdx.x * 2 * U + i] = a;
  }
}

// ---- the whole forward of the PFN layers behind ONE call ------------------------------------------------------------------
// The layers' forward is 8-9 launches each (two or three Linears, statistics, BatchNorm finalisation, apply + max) and runs in
// the eager section in front of the captured step, where the host's time per launch IS step time (0.9 ms of host work for
// 0.4 ms of kernels, scratch/time_encoder_host.py): issued from here it is one boundary crossing.  Every tensor of the
// forward — and everything the backward wants — lives in ONE caller-provided f32 workspace at the offsets
// mbv_pfn_forward_layout reports (11 per layer, in floats, -1 = the layer has no such tensor):
//   0 y (K, U)   1 y_pad (V, U)   2 t (V, U)   3 sums (2 U doubles)   4 scale   5 shift   6 mean   7 rstd (U each)
//   8 a (K, U)   9 a_pad (V, U)   10 m (V, U)
extern "C" int mbv_skinny_gemm_f32(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n, int32_t ldw,
                                   int32_t weight_is_nk, void* stream);

static int64_t pfn_align(int64_t floats) { return (floats + 63) / 64 * 64; }        // 256-byte pieces

extern "C" int64_t mbv_pfn_forward_layout(int64_t num_rows, int64_t num_pillars, const int32_t* units, int32_t num_layers,
                                          int64_t* offsets) {
  if (num_rows < 0 || num_pillars < 0 || !units || num_layers <= 0) return -1;
  int64_t off = 0;
  for (int l = 0; l < num_layers; ++l) {
    const int64_t u = units[l];
    if (u <= 0) return -1;
    const bool last = l == num_layers - 1;
    const int64_t sizes[11] = {num_rows * u, num_pillars * u, l > 0 ? num_pillars * u : -1, (int64_t)(kStatParts + 1) * 4 * u, u, u, u, u,
                               last ? -1 : num_rows * u, last ? -1 : num_pillars * u, num_pillars * u};
    for (int j = 0; j < 11; ++j) {
      if (offsets) offsets[l * 11 + j] = sizes[j] < 0 ? -1 : off;
      if (sizes[j] >= 0) off += pfn_align(sizes[j]);
    }
  }
  return off;
}

extern "C" int mbv_skinny_gemm_f32_addrows(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n,
                                           int32_t ldw, const float* add, const int64_t* index, void* stream);

extern "C" int mbv_pfn_forward(const float* rows, int32_t in_features, const int32_t* row_start, const int32_t* num_points,
                               const int64_t* row_pillar, int64_t num_rows, int64_t num_pillars, int32_t max_points,
                               const float* const* weights,
                               const float* const* gammas, const float* const* betas, float* const* running_means,
                               float* const* running_vars, const int32_t* units, int32_t num_layers, float eps,
                               float momentum, int32_t training, float* workspace, int64_t workspace_floats, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_layers <= 0 || num_layers > 8 || !units || !weights || !gammas || !betas || !running_means || !running_vars)
    return MBV_ERR_BAD_ARG;
  if (num_rows <= 0 || num_pillars <= 0 || max_points <= 0 || in_features <= 0) return MBV_ERR_BAD_ARG;
  if (!rows || !row_start || !num_points || !workspace) return MBV_ERR_BAD_ARG;
  int64_t offs[8 * 11];
  const int64_t need = mbv_pfn_forward_layout(num_rows, num_pillars, units, num_layers, offs);
  if (need < 0) return MBV_ERR_BAD_ARG;
  if (workspace_floats < need) return MBV_ERR_WORKSPACE;
  const double count = (double)num_pillars * (double)max_points;
  const float* a_prev = rows;
  const float* apad_prev = nullptr;
  const float* m_prev = nullptr;
  int c_prev = in_features;
  for (int l = 0; l < num_layers; ++l) {
    const int u = units[l];
    if (int rc = pfn_check(num_pillars, u, max_points)) return rc;
    if (!weights[l] || !gammas[l] || !betas[l] || !running_means[l] || !running_vars[l]) return MBV_ERR_BAD_ARG;
    const int64_t* o = offs + l * 11;
    float* y = workspace + o[0];
    float* ypad = workspace + o[1];
    float* t = o[2] >= 0 ? workspace + o[2] : nullptr;
    double* sums = reinterpret_cast<double*>(workspace + o[3]);
    float *scale = workspace + o[4], *shift = workspace + o[5], *mean = workspace + o[6], *rstd = workspace + o[7];
    float* a = o[8] >= 0 ? workspace + o[8] : nullptr;
    float* apad = o[9] >= 0 ? workspace + o[9] : nullptr;
    float* m = workspace + o[10];
    const int ldw = l == 0 ? c_prev : 2 * c_prev;          // layers behind the first read [a | max]: (U, 2 C)
    const bool stream_stats = row_pillar != nullptr && (u & 3) == 0 && 256 % (u >> 2) == 0;
    if (l == 0) {
      if (int rc = mbv_skinny_gemm_f32(a_prev, weights[l], y, num_rows, c_prev, u, ldw, 1, stream_)) return rc;
      MBV_CHECK_HIP(mbv_fill_async(ypad, 0, sizeof(float) * (size_t)num_pillars * u, stream));      // W . 0
    } else {
      // the pillar term t = W_b . max first: with the streamed statistics y leaves K2c with t[pillar(row)] already added
      if (int rc = mbv_skinny_gemm_f32(m_prev, weights[l] + c_prev, t, num_pillars, c_prev, u, ldw, 1, stream_)) return rc;
      if (stream_stats) {
        if (int rc = mbv_skinny_gemm_f32_addrows(a_prev, weights[l], y, num_rows, c_prev, u, ldw, t, row_pillar, stream_)) return rc;
      } else {
        if (int rc = mbv_skinny_gemm_f32(a_prev, weights[l], y, num_rows, c_prev, u, ldw, 1, stream_)) return rc;
      }
      if (int rc = mbv_skinny_gemm_f32(apad_prev, weights[l], ypad, num_pillars, c_prev, u, ldw, 1, stream_)) return rc;
    }
    if (stream_stats) {
      hipLaunchKernelGGL(k_pfn_colstats, dim3(kStatParts), dim3(256), 0, stream, y, (long)num_rows, ypad, t, num_points,
                         (long)num_pillars, u, max_points, sums);
      MBV_CHECK_LAUNCH();
      // parts → one row (stored behind the parts), then the usual finalisation
      double* total = sums + (size_t)kStatParts * 2 * u;
      hipLaunchKernelGGL(k_pfn_fold_parts, dim3((2 * u + 31) / 32), dim3(1024), 0, stream, sums, kStatParts, 2 * u, total);
      MBV_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_pfn_finalize, dim3((u + 63) / 64), dim3(64), 0, stream, total, count, gammas[l], betas[l], eps, momentum,
                         training, running_means[l], running_vars[l], u, scale, shift, mean, rstd, 1);
      MBV_CHECK_LAUNCH();
    } else {
      if (int rc = mbv_pfn_stats(y, t, ypad, row_start, num_points, num_pillars, u, max_points, sums, stream_)) return rc;
      if (int rc = mbv_pfn_bn_finalize(sums, count, gammas[l], betas[l], eps, momentum, training, running_means[l],
                                       running_vars[l], u, scale, shift, mean, rstd, stream_))
        return rc;
    }
    if (int rc = mbv_pfn_apply_max(y, ypad, scale, shift, row_start, num_points, num_pillars, u, max_points, a, apad, m,
                                   stream_))
      return rc;
    a_prev = a; apad_prev = apad; m_prev = m; c_prev = u;
  }
  return MBV_OK;
}
