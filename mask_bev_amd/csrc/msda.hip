// K5 — multi-scale deformable attention (pixel decoder), forward and backward, gfx950.
//
// Replaces mmcv `ms_deform_attn_forward/backward` (or its `grid_sample` fallback) reached from the
// pixel decoder configured at mask_bev/models/head/mask_bev_panoptic_head.py:127-136 and executed inside
// `self.pixel_decoder(x)` at mask_bev/models/networks/mask2former_head/mask2former_head.py:500.
//
// Layout: value (B, N, H, D) with D contiguous, so the D lanes that share a (query, head) read one
// contiguous D*4-byte segment per bilinear corner (128 B at D = 32: two such segments per wave
// instruction, the full-rate shape for both plain loads and f32 atomics on gfx950).
// Forward is a gather (L2-resident value maps); backward scatters grad_value with f32 atomics
// (1.06 GB of adds per layer at B=4 → bounded by the ≈1.3 TB/s chip-wide atomic rate)
// and reduces grad_location / grad_weight over the D lanes with in-wave shuffles.
#include "common.hpp"

namespace {

struct Corner {
  int off[4];    // element offset of the 4 corners within the level's (h*w, H, D) slab, -1 = outside
  float wgt[4];  // bilinear weights
  float lh, lw;  // fractional parts (for the location gradient)
};

// mmcv `ms_deform_attn_im2col_bilinear`: pixel = loc * size - 0.5, zero padding outside
__device__ __forceinline__ bool bilinear_setup(float loc_x, float loc_y, int h, int w, int stride_pix, Corner& c) {
  const float him = loc_y * (float)h - 0.5f;
  const float wim = loc_x * (float)w - 0.5f;
  if (!(him > -1.f && wim > -1.f && him < (float)h && wim < (float)w)) return false;
  const int hl = (int)floorf(him), wl = (int)floorf(wim);
  const int hh = hl + 1, wh = wl + 1;
  const float lh = him - (float)hl, lw = wim - (float)wl;
  const float uh = 1.f - lh, uw = 1.f - lw;
  c.lh = lh;
  c.lw = lw;
  c.wgt[0] = uh * uw;
  c.wgt[1] = uh * lw;
  c.wgt[2] = lh * uw;
  c.wgt[3] = lh * lw;
  c.off[0] = (hl >= 0 && wl >= 0) ? (hl * w + wl) * stride_pix : -1;
  c.off[1] = (hl >= 0 && wh <= w - 1) ? (hl * w + wh) * stride_pix : -1;
  c.off[2] = (hh <= h - 1 && wl >= 0) ? (hh * w + wl) * stride_pix : -1;
  c.off[3] = (hh <= h - 1 && wh <= w - 1) ? (hh * w + wh) * stride_pix : -1;
  return true;
}

// one thread per (b, q, head, channel); D = head_dim lanes cooperate on a (b, q, head)
__global__ void __launch_bounds__(256) k_msda_fwd(const float* __restrict__ value,
                                                  const int64_t* __restrict__ shapes,
                                                  const int64_t* __restrict__ level_start,
                                                  const float* __restrict__ loc, const float* __restrict__ attn,
                                                  int64_t total, int num_value, int heads, int dim, int levels,
                                                  int num_query, int points, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % dim);
  int64_t t = idx / dim;
  const int hd = (int)(t % heads);
  t /= heads;                       // t = b * num_query + q
  const int b = (int)(t / num_query);
  const int stride_pix = heads * dim;
  const float* vb = value + (int64_t)b * num_value * stride_pix + hd * dim + d;
  const int64_t lw_base = (t * heads + hd) * levels * points;   // index into attn; loc is 2x that
  float acc = 0.f;
  for (int l = 0; l < levels; ++l) {
    const int h = (int)shapes[l * 2], w = (int)shapes[l * 2 + 1];
    const float* vl = vb + level_start[l] * stride_pix;
    for (int p = 0; p < points; ++p) {
      const int64_t k = lw_base + l * points + p;
      const float lx = loc[k * 2], ly = loc[k * 2 + 1], aw = attn[k];
      Corner c;
      if (bilinear_setup(lx, ly, h, w, stride_pix, c)) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c.off[j] >= 0) v += c.wgt[j] * vl[c.off[j]];
        acc += aw * v;
      }
    }
  }
  out[idx] = acc;   // (B, Nq, H*D): idx already is ((b*Nq+q)*H+hd)*D+d
}

__global__ void __launch_bounds__(256) k_msda_bwd(const float* __restrict__ grad_out,
                                                  const float* __restrict__ value,
                                                  const int64_t* __restrict__ shapes,
                                                  const int64_t* __restrict__ level_start,
                                                  const float* __restrict__ loc, const float* __restrict__ attn,
                                                  int64_t total, int num_value, int heads, int dim, int levels,
                                                  int num_query, int points, float* __restrict__ grad_value,
                                                  float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // total is a multiple of dim and dim divides 64, so a (b, q, head) group never straddles the tail
  const bool live = idx < total;
  const int64_t sidx = live ? idx : total - 1;
  const int d = (int)(sidx % dim);
  int64_t t = sidx / dim;
  const int hd = (int)(t % heads);
  t /= heads;
  const int b = (int)(t / num_query);
  const int stride_pix = heads * dim;
  const int64_t vbase = (int64_t)b * num_value * stride_pix + hd * dim + d;
  const int64_t lw_base = (t * heads + hd) * levels * points;
  const float go = live ? grad_out[sidx] : 0.f;
  for (int l = 0; l < levels; ++l) {
    const int h = (int)shapes[l * 2], w = (int)shapes[l * 2 + 1];
    const int64_t lbase = vbase + level_start[l] * stride_pix;
    for (int p = 0; p < points; ++p) {
      const int64_t k = lw_base + l * points + p;
      const float lx = loc[k * 2], ly = loc[k * 2 + 1], aw = attn[k];
      float g_w = 0.f, g_x = 0.f, g_y = 0.f;
      Corner c;
      if (bilinear_setup(lx, ly, h, w, stride_pix, c)) {
        const float tg = go * aw;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = 0.f;
          if (c.off[j] >= 0) {
            v[j] = value[lbase + c.off[j]];
            if (live) atomicAdd(grad_value + lbase + c.off[j], c.wgt[j] * tg);
          }
        }
        const float uh = 1.f - c.lh, uw = 1.f - c.lw;
        const float val = c.wgt[0] * v[0] + c.wgt[1] * v[1] + c.wgt[2] * v[2] + c.wgt[3] * v[3];
        const float gh = -uw * v[0] - c.lw * v[1] + uw * v[2] + c.lw * v[3];
        const float gw = -uh * v[0] + uh * v[1] - c.lh * v[2] + c.lh * v[3];
        g_w = go * val;
        g_x = (float)w * gw * tg;
        g_y = (float)h * gh * tg;
      }
      // reduce over the `dim` lanes of this (b, q, head); dim is a power of two <= 64
      for (int o = dim >> 1; o > 0; o >>= 1) {
        g_w += __shfl_xor(g_w, o, 64);
        g_x += __shfl_xor(g_x, o, 64);
        g_y += __shfl_xor(g_y, o, 64);
      }
      if (live && d == 0) {
        grad_attn[k] = g_w;
        grad_loc[k * 2] = g_x;
        grad_loc[k * 2 + 1] = g_y;
      }
    }
  }
}

bool pow2_le64(int d) { return d > 0 && d <= 64 && (d & (d - 1)) == 0; }

}  // namespace

extern "C" int mbv_ms_deform_attn_fwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                      const float* sampling_loc, const float* attn_weight, int32_t batch,
                                      int32_t num_value, int32_t num_heads, int32_t head_dim, int32_t num_levels,
                                      int32_t num_query, int32_t num_points, float* out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_value <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0 || num_points <= 0)
    return MBV_ERR_BAD_ARG;
  if (!pow2_le64(head_dim)) return MBV_ERR_UNSUPPORTED;
  if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !out) return MBV_ERR_BAD_ARG;
  const int64_t total = (int64_t)batch * num_query * num_heads * head_dim;
  hipLaunchKernelGGL(k_msda_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, value, spatial_shapes,
                     level_start, sampling_loc, attn_weight, total, num_value, num_heads, head_dim, num_levels,
                     num_query, num_points, out);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_ms_deform_attn_bwd(const float* grad_out, const float* value, const int64_t* spatial_shapes,
                                      const int64_t* level_start, const float* sampling_loc,
                                      const float* attn_weight, int32_t batch, int32_t num_value, int32_t num_heads,
                                      int32_t head_dim, int32_t num_levels, int32_t num_query, int32_t num_points,
                                      float* grad_value, float* grad_loc, float* grad_attn, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_value <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0 || num_points <= 0)
    return MBV_ERR_BAD_ARG;
  if (!pow2_le64(head_dim)) return MBV_ERR_UNSUPPORTED;
  if (!grad_out || !value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !grad_value ||
      !grad_loc || !grad_attn)
    return MBV_ERR_BAD_ARG;
  MBV_CHECK_HIP(mbv_fill_async(grad_value, 0, sizeof(float) * (size_t)batch * num_value * num_heads * head_dim, stream));
  const int64_t total = (int64_t)batch * num_query * num_heads * head_dim;
  hipLaunchKernelGGL(k_msda_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, grad_out, value,
                     spatial_shapes, level_start, sampling_loc, attn_weight, total, num_value, num_heads, head_dim,
                     num_levels, num_query, num_points, grad_value, grad_loc, grad_attn);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
