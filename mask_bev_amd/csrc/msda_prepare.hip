// K16 — sampling locations and attention weights of the pixel decoder's deformable attention, fused.
//
// Replaces the element-wise chain between the two projections of the query and the sampling op in mmcv's
// MultiScaleDeformableAttention.forward (configured at mask_bev/models/head/mask_bev_panoptic_head.py:127-136, run
// inside `self.pixel_decoder(x)`, mask_bev/models/networks/mask2former_head/mask2former_head.py:500):
//
//     attention_weights = attention_weights.softmax(-1)                       (over the L*P samples of a head)
//     offset_normalizer = stack([W_l, H_l])
//     sampling_locations = reference_points + sampling_offsets / offset_normalizer
//
// — under autocast eight small launches forward (casts, softmax, stack, div, add) and as many backward, 12 times
// per step.  One thread owns a (batch, query, head): 2*L*P offsets and L*P logits in, as many locations / weights
// out, all contiguous per thread row.  Rounding follows the torch composition: with bf16 projections the quotient
// offset / normalizer is a bf16 value (the dtype of the division's result) before it is added to the f32 reference
// point, the softmax is evaluated in f32, and both gradients return as bf16.
#include "common.hpp"

namespace {

constexpr int kMaxLP = 16;

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<unsigned short>(const unsigned short* p) { return __uint_as_float((unsigned)*p << 16); }
template <typename T>
__device__ __forceinline__ void stf(T* p, float v);
template <>
__device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void stf<unsigned short>(unsigned short* p, float v) { *p = f32_to_bf16_rne(v); }
template <typename T>
__device__ __forceinline__ float round_like(float v);     // the value as the tensor dtype would hold it
template <>
__device__ __forceinline__ float round_like<float>(float v) { return v; }
template <>
__device__ __forceinline__ float round_like<unsigned short>(float v) { return __uint_as_float((unsigned)f32_to_bf16_rne(v) << 16); }

// 4 consecutive elements of a row (rows start 8-byte aligned for bf16, 16-byte aligned for f32 when L*P % 4 == 0)
template <typename T>
__device__ __forceinline__ void ld4(const T* p, float* v);
template <>
__device__ __forceinline__ void ld4<float>(const float* p, float* v) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <>
__device__ __forceinline__ void ld4<unsigned short>(const unsigned short* p, float* v) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void st4(T* p, const float* v);
template <>
__device__ __forceinline__ void st4<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void st4<unsigned short>(unsigned short* p, const float* v) {
  uint2 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}

// IEEE half storage (dtype flag MBV_DT_F16)
template <>
__device__ __forceinline__ float ldf<_Float16>(const _Float16* p) { return (float)*p; }
template <>
__device__ __forceinline__ void stf<_Float16>(_Float16* p, float v) { *p = (_Float16)v; }
template <>
__device__ __forceinline__ float round_like<_Float16>(float v) { return (float)(_Float16)v; }
template <>
__device__ __forceinline__ void ld4<_Float16>(const _Float16* p, float* v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 q = *reinterpret_cast<const h4*>(p);
  v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
}
template <>
__device__ __forceinline__ void st4<_Float16>(_Float16* p, const float* v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 q;
  q[0] = (_Float16)v[0]; q[1] = (_Float16)v[1]; q[2] = (_Float16)v[2]; q[3] = (_Float16)v[3];
  *reinterpret_cast<h4*>(p) = q;
}

struct Norm { float w[8], h[8]; };

template <typename T>
__global__ void __launch_bounds__(256) k_msda_prepare_fwd(const T* __restrict__ off, const T* __restrict__ logit,
                                                          const float* __restrict__ ref, Norm nrm, long rows,
                                                          int num_query, int heads, int levels, int points,
                                                          float* __restrict__ loc, float* __restrict__ attn,
                                                          long ld_off, long ld_logit, const float* __restrict__ b_off,
                                                          const float* __restrict__ b_logit) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const int lp = levels * points;
  const long tok = i / heads;
  const int head = (int)(i - tok * heads);
  const int n = (int)(tok % num_query);
  const float rx = ref[2 * n], ry = ref[2 * n + 1];
  const T* o = off + tok * ld_off + (long)head * lp * 2;
  const T* a = logit + tok * ld_logit + (long)head * lp;
  // the projections' biases, when the GEMM in front left them out (one GEMM for both projections has no single bias vector
  // of the library's epilogue type): added here in f32
  const float* bo = b_off ? b_off + (long)head * lp * 2 : nullptr;
  const float* ba = b_logit ? b_logit + (long)head * lp : nullptr;
  float v[kMaxLP];
  float m = -INFINITY;
  const bool vec = (lp & 3) == 0;               // uniform: whole rows move as 8 / 16-byte pieces
  if (vec) {
#pragma unroll
    for (int k = 0; k < kMaxLP; k += 4) {
      if (k < lp) ld4<T>(a + k, v + k);
      else v[k] = v[k + 1] = v[k + 2] = v[k + 3] = -INFINITY;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kMaxLP; ++k) v[k] = k < lp ? ldf<T>(a + k) : -INFINITY;
  }
  if (ba) {
#pragma unroll
    for (int k = 0; k < kMaxLP; ++k)
      if (k < lp) v[k] += ba[k];
  }
#pragma unroll
  for (int k = 0; k < kMaxLP; ++k) m = fmaxf(m, v[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxLP; ++k) {
    v[k] = k < lp ? expf(v[k] - m) : 0.f;
    s += v[k];
  }
  const float inv = 1.f / s;
  float* lo = loc + i * lp * 2;
  float* ao = attn + i * lp;
  if (vec) {
#pragma unroll
    for (int k = 0; k < kMaxLP; k += 4) {
      if (k < lp) {
        const float w4[4] = {v[k] * inv, v[k + 1] * inv, v[k + 2] * inv, v[k + 3] * inv};
        st4<float>(ao + k, w4);
        float o8[8], r8[8];
        ld4<T>(o + 2 * k, o8);
        ld4<T>(o + 2 * k + 4, o8 + 4);
        if (bo) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] += bo[2 * k + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int l = (k + e) / points;
          r8[2 * e] = rx + round_like<T>(o8[2 * e] / nrm.w[l]);
          r8[2 * e + 1] = ry + round_like<T>(o8[2 * e + 1] / nrm.h[l]);
        }
        st4<float>(lo + 2 * k, r8);
        st4<float>(lo + 2 * k + 4, r8 + 4);
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < kMaxLP; ++k) {
    if (k < lp) {
      const int l = k / points;
      ao[k] = v[k] * inv;
      lo[2 * k] = rx + round_like<T>((ldf<T>(o + 2 * k) + (bo ? bo[2 * k] : 0.f)) / nrm.w[l]);
      lo[2 * k + 1] = ry + round_like<T>((ldf<T>(o + 2 * k + 1) + (bo ? bo[2 * k + 1] : 0.f)) / nrm.h[l]);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_msda_prepare_bwd(const float* __restrict__ g_loc,
                                                          const float* __restrict__ g_attn,
                                                          const float* __restrict__ attn, Norm nrm, long rows,
                                                          int levels, int points, T* __restrict__ g_off,
                                                          T* __restrict__ g_logit, int heads, long ld_off,
                                                          long ld_logit) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const int lp = levels * points;
  const float* gl = g_loc + i * lp * 2;
  const float* ga = g_attn + i * lp;
  const float* aw = attn + i * lp;
  float a[kMaxLP], g[kMaxLP];
  float dot = 0.f;
  const bool vec = (lp & 3) == 0;
  if (vec) {
#pragma unroll
    for (int k = 0; k < kMaxLP; k += 4) {
      if (k < lp) {
        ld4<float>(aw + k, a + k);
        ld4<float>(ga + k, g + k);
      } else {
        a[k] = a[k + 1] = a[k + 2] = a[k + 3] = 0.f;
        g[k] = g[k + 1] = g[k + 2] = g[k + 3] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < kMaxLP; ++k) {
      a[k] = k < lp ? aw[k] : 0.f;
      g[k] = k < lp ? ga[k] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < kMaxLP; ++k) dot += a[k] * g[k];
  // (batch, query) row strides: the two gradients may be columns of one wider matrix; the heads stay packed in a row
  const long bn = i / heads;
  const int hh = (int)(i - bn * heads);
  T* go = g_off + bn * ld_off + (long)hh * lp * 2;
  T* gq = g_logit + bn * ld_logit + (long)hh * lp;
  if (vec) {
#pragma unroll
    for (int k = 0; k < kMaxLP; k += 4) {
      if (k < lp) {
        const float q4[4] = {a[k] * (g[k] - dot), a[k + 1] * (g[k + 1] - dot), a[k + 2] * (g[k + 2] - dot),
                             a[k + 3] * (g[k + 3] - dot)};
        st4<T>(gq + k, q4);
        float l8[8], o8[8];
        ld4<float>(gl + 2 * k, l8);
        ld4<float>(gl + 2 * k + 4, l8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int l = (k + e) / points;
          o8[2 * e] = round_like<T>(l8[2 * e]) / nrm.w[l];
          o8[2 * e + 1] = round_like<T>(l8[2 * e + 1]) / nrm.h[l];
        }
        st4<T>(go + 2 * k, o8);
        st4<T>(go + 2 * k + 4, o8 + 4);
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < kMaxLP; ++k) {
    if (k < lp) {
      const int l = k / points;
      stf<T>(gq + k, a[k] * (g[k] - dot));                    // softmax backward
      // the quotient was a tensor of dtype T: its incoming gradient is a T value before the division's backward
      stf<T>(go + 2 * k, round_like<T>(gl[2 * k]) / nrm.w[l]);
      stf<T>(go + 2 * k + 1, round_like<T>(gl[2 * k + 1]) / nrm.h[l]);
    }
  }
}

bool fill_norm(const int64_t* shapes_host, int levels, Norm& n) {
  if (!shapes_host || levels < 1 || levels > 8) return false;
  for (int l = 0; l < 8; ++l) { n.w[l] = 1.f; n.h[l] = 1.f; }
  for (int l = 0; l < levels; ++l) {
    const int64_t h = shapes_host[2 * l], w = shapes_host[2 * l + 1];
    if (h <= 0 || w <= 0) return false;
    n.h[l] = (float)h; n.w[l] = (float)w;
  }
  return true;
}

// The two 16-bit GEMM inputs of the query side in one pass over the f32 token map: x_lo = lo(x) (value projection) and
// q_lo = lo(x + pos) (offset / weight projections), pos (pos_rows, C) broadcast over the batch.  torch ran a cast and
// a type-converting add (a non-vectorised kernel: 18 us for 22 MB).  A thread owns 4 consecutive channels.
__global__ void __launch_bounds__(256) k_msda_query_inputs(const float* __restrict__ x, const float* __restrict__ pos,
                                                           long n4, long pos_n4, int f16, unsigned short* __restrict__ x_lo,
                                                           unsigned short* __restrict__ q_lo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4*>(x)[i];
  const float4 p = reinterpret_cast<const float4*>(pos)[i % pos_n4];
  const float q[4] = {a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
  const float v[4] = {a.x, a.y, a.z, a.w};
  unsigned short xo[4], qo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (f16) {
      xo[e] = __builtin_bit_cast(unsigned short, (_Float16)v[e]);
      qo[e] = __builtin_bit_cast(unsigned short, (_Float16)q[e]);
    } else {
      xo[e] = f32_to_bf16_rne(v[e]);
      qo[e] = f32_to_bf16_rne(q[e]);
    }
  }
  reinterpret_cast<uint2*>(x_lo)[i] = make_uint2(xo[0] | ((unsigned)xo[1] << 16), xo[2] | ((unsigned)xo[3] << 16));
  reinterpret_cast<uint2*>(q_lo)[i] = make_uint2(qo[0] | ((unsigned)qo[1] << 16), qo[2] | ((unsigned)qo[3] << 16));
}

}  // namespace

extern "C" int mbv_msda_prepare_supported(int32_t num_levels, int32_t num_points) {
  return num_levels >= 1 && num_levels <= 8 && num_points >= 1 && num_levels * num_points <= kMaxLP;
}

extern "C" int mbv_msda_prepare_fwd_ld(const void* offsets, int64_t ld_offsets, const void* logits, int64_t ld_logits,
                                       const float* bias_offsets, const float* bias_logits, int32_t is_bf16,
                                       const float* ref_points, const int64_t* spatial_shapes_host, int32_t batch,
                                       int32_t num_query, int32_t num_heads, int32_t num_levels, int32_t num_points,
                                       float* loc, float* attn, void* stream) {
  if (batch <= 0 || num_query <= 0 || num_heads <= 0) return MBV_ERR_BAD_ARG;
  if (ld_offsets < (int64_t)num_heads * num_levels * num_points * 2 || ld_logits < (int64_t)num_heads * num_levels * num_points)
    return MBV_ERR_BAD_ARG;
  if (!mbv_msda_prepare_supported(num_levels, num_points)) return MBV_ERR_UNSUPPORTED;
  {   // the vector path reads 4 elements at a time: rows and the two base pointers must keep that alignment
    const size_t es = is_bf16 ? 2 : 4;
    if ((num_levels * num_points) % 4 == 0 &&
        (((reinterpret_cast<size_t>(offsets) | reinterpret_cast<size_t>(logits)) & (4 * es - 1)) != 0 ||
         (ld_offsets % 4) != 0 || (ld_logits % 4) != 0))
      return MBV_ERR_BAD_ARG;
  }
  if (!offsets || !logits || !ref_points || !loc || !attn) return MBV_ERR_BAD_ARG;
  Norm n;
  if (!fill_norm(spatial_shapes_host, num_levels, n)) return MBV_ERR_BAD_ARG;
  const long rows = (long)batch * num_query * num_heads;
  const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  if (is_bf16 == MBV_DT_F16)
    hipLaunchKernelGGL(k_msda_prepare_fwd<_Float16>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(offsets), reinterpret_cast<const _Float16*>(logits),
                       ref_points, n, rows, num_query, num_heads, num_levels, num_points, loc, attn, (long)ld_offsets,
                       (long)ld_logits, bias_offsets, bias_logits);
  else if (is_bf16)
    hipLaunchKernelGGL(k_msda_prepare_fwd<unsigned short>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(offsets), reinterpret_cast<const unsigned short*>(logits),
                       ref_points, n, rows, num_query, num_heads, num_levels, num_points, loc, attn, (long)ld_offsets,
                       (long)ld_logits, bias_offsets, bias_logits);
  else
    hipLaunchKernelGGL(k_msda_prepare_fwd<float>, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(offsets), reinterpret_cast<const float*>(logits), ref_points, n,
                       rows, num_query, num_heads, num_levels, num_points, loc, attn, (long)ld_offsets, (long)ld_logits,
                       bias_offsets, bias_logits);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_msda_prepare_fwd(const void* offsets, const void* logits, int32_t is_bf16, const float* ref_points,
                                    const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query,
                                    int32_t num_heads, int32_t num_levels, int32_t num_points, float* loc, float* attn,
                                    void* stream) {
  const int64_t lp = (int64_t)num_levels * num_points;
  if (!offsets || !logits) return MBV_ERR_BAD_ARG;
  return mbv_msda_prepare_fwd_ld(offsets, num_heads * lp * 2, logits, num_heads * lp, nullptr, nullptr, is_bf16,
                                 ref_points, spatial_shapes_host, batch, num_query, num_heads, num_levels, num_points, loc,
                                 attn, stream);
}

extern "C" int mbv_msda_prepare_bwd_ld(const float* grad_loc, const float* grad_attn, const float* attn,
                                       const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query,
                                       int32_t num_heads, int32_t num_levels, int32_t num_points, int32_t out_bf16,
                                       void* grad_offsets, int64_t ld_offsets, void* grad_logits, int64_t ld_logits,
                                       void* stream) {
  if (batch <= 0 || num_query <= 0 || num_heads <= 0) return MBV_ERR_BAD_ARG;
  // row strides are given per (batch, query) row of H*L*P*2 / H*L*P elements; inside a row the heads stay packed
  if (ld_offsets < (int64_t)num_heads * num_levels * num_points * 2 || ld_logits < (int64_t)num_heads * num_levels * num_points)
    return MBV_ERR_BAD_ARG;
  if (!mbv_msda_prepare_supported(num_levels, num_points)) return MBV_ERR_UNSUPPORTED;
  if (!grad_loc || !grad_attn || !attn || !grad_offsets || !grad_logits) return MBV_ERR_BAD_ARG;
  Norm n;
  if (!fill_norm(spatial_shapes_host, num_levels, n)) return MBV_ERR_BAD_ARG;
  const long rows = (long)batch * num_query * num_heads;
  const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  if (out_bf16 == MBV_DT_F16)
    hipLaunchKernelGGL(k_msda_prepare_bwd<_Float16>, grid, block, 0, (hipStream_t)stream, grad_loc, grad_attn, attn, n,
                       rows, num_levels, num_points, reinterpret_cast<_Float16*>(grad_offsets),
                       reinterpret_cast<_Float16*>(grad_logits), num_heads, (long)ld_offsets, (long)ld_logits);
  else if (out_bf16)
    hipLaunchKernelGGL(k_msda_prepare_bwd<unsigned short>, grid, block, 0, (hipStream_t)stream, grad_loc, grad_attn,
                       attn, n, rows, num_levels, num_points, reinterpret_cast<unsigned short*>(grad_offsets),
                       reinterpret_cast<unsigned short*>(grad_logits), num_heads, (long)ld_offsets, (long)ld_logits);
  else
    hipLaunchKernelGGL(k_msda_prepare_bwd<float>, grid, block, 0, (hipStream_t)stream, grad_loc, grad_attn, attn, n,
                       rows, num_levels, num_points, reinterpret_cast<float*>(grad_offsets),
                       reinterpret_cast<float*>(grad_logits), num_heads, (long)ld_offsets, (long)ld_logits);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_msda_prepare_bwd(const float* grad_loc, const float* grad_attn, const float* attn,
                                    const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query,
                                    int32_t num_heads, int32_t num_levels, int32_t num_points, int32_t out_bf16,
                                    void* grad_offsets, void* grad_logits, void* stream) {
  const int64_t lp = (int64_t)num_levels * num_points;
  return mbv_msda_prepare_bwd_ld(grad_loc, grad_attn, attn, spatial_shapes_host, batch, num_query, num_heads,
                                 num_levels, num_points, out_bf16, grad_offsets, num_heads * lp * 2, grad_logits,
                                 num_heads * lp, stream);
}

// x (rows, C) f32, pos (pos_rows, C) f32 with rows % pos_rows == 0 (pos repeats every pos_rows rows: the batch
// broadcast of the pixel decoder's positional encoding, mask_bev_panoptic_head.py:127-136 → mmcv
// MultiScaleDeformableAttention.forward `query = query + query_pos`); x_lo = lo(x), q_lo = lo(x + pos) in `dtype`
// (MBV_DT_BF16 / MBV_DT_F16).  C % 4 == 0, 16-byte aligned pointers.
extern "C" int mbv_msda_query_inputs(const float* x, const float* pos, int64_t rows, int64_t pos_rows, int32_t C,
                                     int32_t dtype, void* x_lo, void* q_lo, void* stream) {
  if (rows < 0 || pos_rows <= 0 || C <= 0 || (C & 3)) return MBV_ERR_BAD_ARG;
  if (dtype != MBV_DT_BF16 && dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!x || !pos || !x_lo || !q_lo || rows % pos_rows) return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(pos)) & 15) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(x_lo) | reinterpret_cast<size_t>(q_lo)) & 7) return MBV_ERR_UNSUPPORTED;
  const long n4 = rows * (C / 4), pn4 = pos_rows * (C / 4);
  hipLaunchKernelGGL(k_msda_query_inputs, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, pos,
                     n4, pn4, dtype == MBV_DT_F16 ? 1 : 0, reinterpret_cast<unsigned short*>(x_lo),
                     reinterpret_cast<unsigned short*>(q_lo));
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
