// K18 — GroupNorm of NCHW maps, fused with what surrounds it in the pixel decoder's ConvModules.
//
// Replaces the `conv → GroupNorm(32) [→ ReLU]` tails of mmcv ConvModule as the pixel decoder builds them
// (/root/reference: mask_bev/models/head/mask_bev_panoptic_head.py:119-123 → mmdet MSDeformAttnPixelDecoder: input_convs,
// lateral_convs, output_convs) and the FPN step `lateral + interpolate(previous, bilinear)` between them.  Under
// autocast torch runs, per ConvModule, a 16-bit → f32 cast of the convolution output, three GroupNorm kernels (row
// moments, fused parameters, apply) that reach 0.8 TB/s on these maps, for the lateral step a 4x up-sampled f32 copy of
// the coarser level and a strided add over three 67 MB maps, an activation kernel and an f32 → 16-bit cast for the next
// convolution — ≈ 0.55 ms of a 29 ms step at (4, 256, 128, 128); here it is two passes over the convolution output:
//
//   stats:  per (sample, group, split) partial  Σx, Σx²  in f64             (x read once, 16-byte accesses)
//   apply:  y = (x − mean) · rstd · γ_c + β_c  [+ bilinear(previous level)]  [ReLU]   stored in the consumer's dtype
//
// and the backward is two more (per-plane sums of dy and dy·x̂, then dx with the group terms folded in; dγ, dβ
// accumulated straight into the parameter arena).  The up-sampling follows ATen's upsample_bilinear2d
// (align_corners = False: source index max(scale · (dst + 0.5) − 0.5, 0), neighbour clamped at the border).
// HBM-bound streaming kernels; statistics in f64 (biased variance E[x²] − mean², as torch's GroupNorm).
#include "common.hpp"

namespace {

__device__ __forceinline__ float h16_to_f32(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
__device__ __forceinline__ unsigned f32_to_h16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }

// 4 consecutive elements; kind = MBV_DT_F32 / MBV_DT_BF16 / MBV_DT_F16 storage (block-uniform)
__device__ __forceinline__ float4 ld4(const void* base, int kind, long elem) {
  if (kind == MBV_DT_F32) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
  const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
  if (kind == MBV_DT_BF16)
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  return make_float4(h16_to_f32(u.x & 0xffffu), h16_to_f32(u.x >> 16), h16_to_f32(u.y & 0xffffu), h16_to_f32(u.y >> 16));
}
__device__ __forceinline__ float ld1(const void* base, int kind, long elem) {
  if (kind == MBV_DT_F32) return reinterpret_cast<const float*>(base)[elem];
  const unsigned short u = reinterpret_cast<const unsigned short*>(base)[elem];
  return kind == MBV_DT_BF16 ? __uint_as_float((unsigned)u << 16) : h16_to_f32(u);
}
__device__ __forceinline__ void st4(void* base, int kind, long elem, float4 v) {
  if (kind == MBV_DT_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem) = v;
    return;
  }
  uint2 u;
  if (kind == MBV_DT_BF16) {
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
  } else {
    u.x = f32_to_h16(v.x) | (f32_to_h16(v.y) << 16);
    u.y = f32_to_h16(v.z) | (f32_to_h16(v.w) << 16);
  }
  *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + elem) = u;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {      // 256 threads; every thread gets the sum
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// A group of an NCHW map is one contiguous run of n = (C / G) * H * W elements.  Block (bg, s) sums chunk s of it.
__global__ void __launch_bounds__(256) k_gn_stats(const void* __restrict__ x, int kind, long n, int splits,
                                                  double* __restrict__ partial) {
  __shared__ double red[4];
  const long bg = blockIdx.x / splits;
  const int s = blockIdx.x - (int)(bg * splits);
  const long nvec = n >> 2, per = (nvec + splits - 1) / splits;
  const long v0 = s * per, v1 = v0 + per < nvec ? v0 + per : nvec;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  long v = v0 + threadIdx.x;
  for (; v + 256 < v1; v += 512) {                   // two independent 16-byte loads in flight per thread
    const float4 a = ld4(x, kind, bg * n + 4 * v), b = ld4(x, kind, bg * n + 4 * (v + 256));
    s1[0] += (a.x + a.y) + (a.z + a.w); s2[0] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    s1[1] += (b.x + b.y) + (b.z + b.w); s2[1] += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
  }
  if (v < v1) {
    const float4 a = ld4(x, kind, bg * n + 4 * v);
    s1[0] += (a.x + a.y) + (a.z + a.w); s2[0] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  const double t1 = block_sum_d((double)s1[0] + (double)s1[1], red);
  const double t2 = block_sum_d((double)s2[0] + (double)s2[1], red);
  if (threadIdx.x == 0) {
    partial[((long)blockIdx.x) * 2] = t1;
    partial[((long)blockIdx.x) * 2 + 1] = t2;
  }
}

struct GnArgs {
  const void* x; int x_kind;
  const double* partial; int splits;           // forward: the stats kernel's partial sums
  const float* mean; const float* rstd;        // backward: the saved statistics (B * G)
  const float* gamma; const float* beta;
  float eps;
  int C, H, W, G;
  const void* add; int add_kind; int ah, aw;   // (B, C, ah, aw) map added after bilinear up-sampling to (H, W); null = none
  int relu;
  void* y; int y_kind;
  float* mean_out; float* rstd_out;
};

// ATen's source index of a destination pixel (align_corners = False)
__device__ __forceinline__ void up_axis(float scale, int dst, int in_size, int& i0, int& step, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  step = i0 < in_size - 1 ? 1 : 0;
  l1 = src - (float)i0;
}

// grid = (B * C planes) x chunks; a thread owns 4 consecutive pixels of its plane per iteration
__global__ void __launch_bounds__(256) k_gn_apply(const GnArgs a, int chunks) {
  const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
  const int c = plane % a.C, b = plane / a.C;
  const int cpg = a.C / a.G, g = c / cpg;
  const long hw = (long)a.H * a.W;
  double t1 = 0.0, t2 = 0.0;
  const double* p = a.partial + ((long)(b * a.G + g) * a.splits) * 2;
  for (int s = 0; s < a.splits; ++s) { t1 += p[2 * s]; t2 += p[2 * s + 1]; }
  const double n = (double)cpg * (double)hw;
  const double mean_d = t1 / n;
  double var = t2 / n - mean_d * mean_d;
  var = var < 0.0 ? 0.0 : var;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  if (chunk == 0 && c == g * cpg && threadIdx.x == 0) {
    a.mean_out[b * a.G + g] = mean;
    a.rstd_out[b * a.G + g] = rstd;
  }
  const float ga = a.gamma[c] * rstd, be = a.beta[c] - mean * ga;         // y = x * ga + be
  const long base = (long)plane * hw;
  const long nvec = hw >> 2, per = (nvec + chunks - 1) / chunks;
  const long v0 = chunk * per, v1 = v0 + per < nvec ? v0 + per : nvec;
  const float sh = a.add ? (float)a.ah / (float)a.H : 0.f, sw = a.add ? (float)a.aw / (float)a.W : 0.f;
  const long abase = (long)plane * a.ah * a.aw;
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const float4 x = ld4(a.x, a.x_kind, base + 4 * v);
    float o[4] = {x.x * ga + be, x.y * ga + be, x.z * ga + be, x.w * ga + be};
    if (a.add) {                                          // W % 4 == 0: the four pixels share a row
      const int oy = (int)((4 * v) / a.W), ox = (int)((4 * v) - (long)oy * a.W);
      int y0, ys; float ly;
      up_axis(sh, oy, a.ah, y0, ys, ly);
      const long r0 = abase + (long)y0 * a.aw, r1 = r0 + (long)ys * a.aw;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int x0, xs; float lx;
        up_axis(sw, ox + j, a.aw, x0, xs, lx);
        const float v00 = ld1(a.add, a.add_kind, r0 + x0), v01 = ld1(a.add, a.add_kind, r0 + x0 + xs);
        const float v10 = ld1(a.add, a.add_kind, r1 + x0), v11 = ld1(a.add, a.add_kind, r1 + x0 + xs);
        o[j] += (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
      }
    }
    if (a.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = o[j] > 0.f ? o[j] : 0.f;
    }
    st4(a.y, a.y_kind, base + 4 * v, make_float4(o[0], o[1], o[2], o[3]));
  }
}

struct GnBwdArgs {
  const void* dy; int dy_kind;
  const void* x; int x_kind;
  const float* mean; const float* rstd; const float* gamma; const float* beta;
  int C, H, W, G, relu, B;
  float* plane_sums;                           // (B * C, 2): Σ dy', Σ dy' x̂ per plane (dy' = dy masked by the ReLU gate)
  void* dx; int dx_kind;
  float* dgamma; float* dbeta; int accumulate;
};

// one block per (b, c) plane
__global__ void __launch_bounds__(256) k_gn_bwd_sums(const GnBwdArgs a) {
  __shared__ double red[4];
  const int plane = blockIdx.x, c = plane % a.C, b = plane / a.C;
  const int cpg = a.C / a.G, g = c / cpg;
  const float mean = a.mean[b * a.G + g], rstd = a.rstd[b * a.G + g];
  const float ga = a.gamma[c];
  const float ga_f = ga * rstd, be_f = a.beta[c] - mean * ga_f;       // the forward's y = x * ga_f + be_f (same arithmetic)
  const long hw = (long)a.H * a.W, base = (long)plane * hw, nvec = hw >> 2;
  float s1 = 0.f, s2 = 0.f;
  for (long v = threadIdx.x; v < nvec; v += 256) {
    const float4 dy = ld4(a.dy, a.dy_kind, base + 4 * v), x = ld4(a.x, a.x_kind, base + 4 * v);
    const float xv[4] = {x.x, x.y, x.z, x.w};
    const float xh[4] = {(x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd};
    float d[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.relu && !(xv[j] * ga_f + be_f > 0.f)) d[j] = 0.f;
      s1 += d[j];
      s2 += d[j] * xh[j];
    }
  }
  const double t1 = block_sum_d((double)s1, red), t2 = block_sum_d((double)s2, red);
  if (threadIdx.x == 0) {
    a.plane_sums[2 * (long)plane] = (float)t1;
    a.plane_sums[2 * (long)plane + 1] = (float)t2;
  }
}

// grid = planes x chunks:  dx = rstd (dy' γ_c − m1 − x̂ m2),  m1 = Σ_{c in g} γ_c Σdy' / n,  m2 = Σ_{c in g} γ_c Σdy'x̂ / n
__global__ void __launch_bounds__(256) k_gn_bwd_dx(const GnBwdArgs a, int chunks) {
  const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
  const int c = plane % a.C, b = plane / a.C;
  const int cpg = a.C / a.G, g = c / cpg;
  const float mean = a.mean[b * a.G + g], rstd = a.rstd[b * a.G + g];
  const float ga = a.gamma[c];
  const float ga_f = ga * rstd, be_f = a.beta[c] - mean * ga_f;
  const long hw = (long)a.H * a.W;
  float m1 = 0.f, m2 = 0.f;
  for (int j = 0; j < cpg; ++j) {
    const int cc = g * cpg + j;
    const float gj = a.gamma[cc];
    m1 += gj * a.plane_sums[2 * ((long)b * a.C + cc)];
    m2 += gj * a.plane_sums[2 * ((long)b * a.C + cc) + 1];
  }
  const float inv_n = 1.f / ((float)cpg * (float)hw);
  m1 *= inv_n; m2 *= inv_n;
  if (b == 0 && chunk == 0 && threadIdx.x == 0) {     // dγ_c = Σ_b Σ dy' x̂,  dβ_c = Σ_b Σ dy'
    float dg = 0.f, db = 0.f;
    for (int bb = 0; bb < a.B; ++bb) {
      db += a.plane_sums[2 * ((long)bb * a.C + c)];
      dg += a.plane_sums[2 * ((long)bb * a.C + c) + 1];
    }
    a.dgamma[c] = a.accumulate ? a.dgamma[c] + dg : dg;
    a.dbeta[c] = a.accumulate ? a.dbeta[c] + db : db;
  }
  const long base = (long)plane * hw;
  const long nvec = hw >> 2, per = (nvec + chunks - 1) / chunks;
  const long v0 = chunk * per, v1 = v0 + per < nvec ? v0 + per : nvec;
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const float4 dy = ld4(a.dy, a.dy_kind, base + 4 * v), x = ld4(a.x, a.x_kind, base + 4 * v);
    const float xv[4] = {x.x, x.y, x.z, x.w};
    const float xh[4] = {(x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd};
    float d[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.relu && !(xv[j] * ga_f + be_f > 0.f)) d[j] = 0.f;
      d[j] = rstd * (d[j] * ga - m1 - xh[j] * m2);
    }
    st4(a.dx, a.dx_kind, base + 4 * v, make_float4(d[0], d[1], d[2], d[3]));
  }
}

int gn_splits(int64_t batch, int groups, int64_t n) {
  // ≈ 1024 blocks over the launch, at least 4096 elements per block
  int64_t s = 1024 / (batch * groups > 0 ? batch * groups : 1);
  const int64_t cap = n / 4096;
  if (s > cap) s = cap;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

int gn_chunks(int64_t planes, int64_t hw) {
  int64_t ch = 2048 / (planes > 0 ? planes : 1);
  const int64_t cap = hw / 4096;
  if (ch > cap) ch = cap;
  return ch < 1 ? 1 : (int)ch;
}

bool kind_ok(int k) { return k == MBV_DT_F32 || k == MBV_DT_BF16 || k == MBV_DT_F16; }

// d(added map) of the fused FPN step: the adjoint of ATen's upsample_bilinear2d (align_corners = False) in GATHER form.
// A thread owns one pixel (Y, X) of the coarse (ah, aw) plane and walks the fine pixels whose two source taps can touch
// it — src(y) = scale (y + 0.5) - 0.5 in (Y - 1, Y + 1) — evaluating up_axis exactly as the forward did: no atomics, every
// output written once, the fine gradient read ~ once from HBM (its 4 x 4 neighbourhoods overlap in L1 / L2).  ATen's
// scatter kernel for the same adjoint took 78 us on the step's (4, 256, 128, 128) -> (64, 64) map, + 8 us for the cast.
__global__ void __launch_bounds__(256) k_upsample_bilinear_bwd(const void* __restrict__ gy, int gy_kind, long planes,
                                                               int H, int W, int ah, int aw, void* __restrict__ out,
                                                               int out_kind) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)ah * aw;
  if (idx >= planes * per) return;
  const long plane = idx / per;
  const int Y = (int)((idx - plane * per) / aw), X = (int)(idx - plane * per - (long)Y * aw);
  const float sh = (float)ah / (float)H, sw = (float)aw / (float)W;
  // fine rows / columns whose source position lies in (Y - 1, Y + 1): a superset by one on each side, filtered below
  int y_lo = (int)floorf(((float)Y - 0.5f) / sh - 0.5f) - 1, y_hi = (int)ceilf(((float)Y + 1.5f) / sh - 0.5f) + 1;
  int x_lo = (int)floorf(((float)X - 0.5f) / sw - 0.5f) - 1, x_hi = (int)ceilf(((float)X + 1.5f) / sw - 0.5f) + 1;
  y_lo = y_lo < 0 ? 0 : y_lo; x_lo = x_lo < 0 ? 0 : x_lo;
  y_hi = y_hi > H - 1 ? H - 1 : y_hi; x_hi = x_hi > W - 1 ? W - 1 : x_hi;
  const long base = plane * (long)H * W;
  float acc = 0.f;
  constexpr int KW = 8;
  if (y_hi - y_lo < KW && x_hi - x_lo < KW) {
    // the adjoint is separable: the row and column weights of the window once per thread (≤ 8 + 8 evaluations of up_axis
    // instead of one per visited pixel), then every load of a window row before its first use
    float wy[KW], wx[KW];
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      int p0, ps; float l;
      const int y = y_lo + i, x = x_lo + i;
      up_axis(sh, y <= y_hi ? y : y_hi, ah, p0, ps, l);
      wy[i] = y <= y_hi ? (p0 == Y ? 1.f - l : 0.f) + (p0 + ps == Y ? l : 0.f) : 0.f;
      up_axis(sw, x <= x_hi ? x : x_hi, aw, p0, ps, l);
      wx[i] = x <= x_hi ? (p0 == X ? 1.f - l : 0.f) + (p0 + ps == X ? l : 0.f) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      if (wy[i] == 0.f) continue;                       // (also rows beyond y_hi)
      const long row = base + (long)(y_lo + i) * W;
      float g[KW];
#pragma unroll
      for (int j = 0; j < KW; ++j) g[j] = wx[j] != 0.f ? ld1(gy, gy_kind, row + x_lo + j) : 0.f;
      float r = 0.f;
#pragma unroll
      for (int j = 0; j < KW; ++j) r += wx[j] * g[j];
      acc += wy[i] * r;
    }
    if (out_kind == MBV_DT_F32) reinterpret_cast<float*>(out)[idx] = acc;
    else reinterpret_cast<unsigned short*>(out)[idx] = out_kind == MBV_DT_BF16 ? f32_to_bf16_rne(acc) : (unsigned short)f32_to_h16(acc);
    return;
  }
  for (int y = y_lo; y <= y_hi; ++y) {
    int y0, ys; float ly;
    up_axis(sh, y, ah, y0, ys, ly);
    const float wy = (y0 == Y ? 1.f - ly : 0.f) + (y0 + ys == Y ? ly : 0.f);     // (ys == 0: both taps are y0 — weights add up)
    if (wy == 0.f) continue;
    for (int x = x_lo; x <= x_hi; ++x) {
      int x0, xs; float lx;
      up_axis(sw, x, aw, x0, xs, lx);
      const float wx = (x0 == X ? 1.f - lx : 0.f) + (x0 + xs == X ? lx : 0.f);
      if (wx != 0.f) acc += wy * wx * ld1(gy, gy_kind, base + (long)y * W + x);
    }
  }
  if (out_kind == MBV_DT_F32) reinterpret_cast<float*>(out)[idx] = acc;
  else reinterpret_cast<unsigned short*>(out)[idx] = out_kind == MBV_DT_BF16 ? f32_to_bf16_rne(acc) : (unsigned short)f32_to_h16(acc);
}

}  // namespace

extern "C" int mbv_upsample_bilinear_bwd(const void* grad_out, int32_t grad_dtype, int64_t planes, int32_t h, int32_t w,
                                         int32_t in_h, int32_t in_w, void* grad_in, int32_t in_dtype, void* stream) {
  if (planes < 0 || h <= 0 || w <= 0 || in_h <= 0 || in_w <= 0) return MBV_ERR_BAD_ARG;
  if (planes == 0) return MBV_OK;
  if (!grad_out || !grad_in || grad_dtype < 0 || grad_dtype > 2 || in_dtype < 0 || in_dtype > 2) return MBV_ERR_BAD_ARG;
  const long total = planes * (long)in_h * in_w;
  if (total > 0x7fffffffL * 256) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_upsample_bilinear_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out, grad_dtype, (long)planes, h, w, in_h, in_w, grad_in, in_dtype);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_groupnorm_supported(int32_t channels, int32_t groups, int32_t h, int32_t w) {
  return (channels > 0 && groups > 0 && channels % groups == 0 && h > 0 && w > 0 && ((int64_t)h * w) % 4 == 0) ? 1 : 0;
}

extern "C" size_t mbv_groupnorm_workspace_bytes(int64_t batch, int32_t channels, int32_t groups, int32_t h, int32_t w) {
  if (batch <= 0 || groups <= 0 || channels <= 0) return 0;
  const int64_t n = (int64_t)(channels / groups) * h * w;
  return (size_t)batch * groups * gn_splits(batch, groups, n) * 2 * sizeof(double);
}

extern "C" int mbv_groupnorm_fwd(const void* x, int32_t x_dtype, int64_t batch, int32_t channels, int32_t h, int32_t w,
                                 int32_t groups, const float* gamma, const float* beta, float eps, const void* add,
                                 int32_t add_dtype, int32_t add_h, int32_t add_w, int32_t relu, void* y, int32_t y_dtype,
                                 float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream) {
  if (batch < 0) return MBV_ERR_BAD_ARG;
  if (batch == 0) return MBV_OK;
  if (!mbv_groupnorm_supported(channels, groups, h, w)) return MBV_ERR_UNSUPPORTED;
  if (!x || !gamma || !beta || !y || !mean || !rstd || !kind_ok(x_dtype) || !kind_ok(y_dtype)) return MBV_ERR_BAD_ARG;
  if (add && (!kind_ok(add_dtype) || add_h <= 0 || add_w <= 0)) return MBV_ERR_BAD_ARG;
  if (add && (w & 3)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) return MBV_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < mbv_groupnorm_workspace_bytes(batch, channels, groups, h, w)) return MBV_ERR_WORKSPACE;
  if (batch * channels > 0x7fffffffLL / 64) return MBV_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)(channels / groups) * h * w;
  const int splits = gn_splits(batch, groups, n);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(k_gn_stats, dim3((unsigned)(batch * groups * splits)), dim3(256), 0, st, x, x_dtype, (long)n, splits,
                     partial);
  MBV_CHECK_LAUNCH();
  GnArgs a{x, x_dtype, partial, splits, nullptr, nullptr, gamma, beta, eps, channels, h, w, groups, add, add_dtype, add_h,
           add_w, relu, y, y_dtype, mean, rstd};
  const int chunks = gn_chunks(batch * channels, (int64_t)h * w);
  hipLaunchKernelGGL(k_gn_apply, dim3((unsigned)(batch * channels * chunks)), dim3(256), 0, st, a, chunks);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// dx in dx_dtype; dgamma / dbeta (channels) f32 stored, or added to when accumulate != 0; plane_sums: (batch * channels * 2)
// floats of scratch.  With relu != 0 the gate is recomputed from x (y > 0  <=>  x̂ γ + β > 0).  The gradient of an
// up-sampled addend is dy itself (the caller hands it to the up-sampling's backward).
extern "C" int mbv_groupnorm_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int64_t batch, int32_t channels,
                                 int32_t h, int32_t w, int32_t groups, int32_t relu, void* dx, int32_t dx_dtype,
                                 float* dgamma, float* dbeta, int32_t accumulate, float* plane_sums, void* stream) {
  if (batch < 0) return MBV_ERR_BAD_ARG;
  if (batch == 0) return MBV_OK;
  if (!mbv_groupnorm_supported(channels, groups, h, w)) return MBV_ERR_UNSUPPORTED;
  if (!dy || !x || !mean || !rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !plane_sums) return MBV_ERR_BAD_ARG;
  if (!kind_ok(dy_dtype) || !kind_ok(x_dtype) || !kind_ok(dx_dtype)) return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(dx)) & 15) return MBV_ERR_UNSUPPORTED;
  if (batch * channels > 0x7fffffffLL / 64) return MBV_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  GnBwdArgs a{dy, dy_dtype, x, x_dtype, mean, rstd, gamma, beta, channels, h, w, groups, relu, (int)batch, plane_sums, dx,
              dx_dtype, dgamma, dbeta, accumulate};
  hipLaunchKernelGGL(k_gn_bwd_sums, dim3((unsigned)(batch * channels)), dim3(256), 0, st, a);
  MBV_CHECK_LAUNCH();
  const int chunks = gn_chunks(batch * channels, (int64_t)h * w);
  hipLaunchKernelGGL(k_gn_bwd_dx, dim3((unsigned)(batch * channels * chunks)), dim3(256), 0, st, a, chunks);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
