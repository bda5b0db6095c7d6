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

// XCD-aware block order (blocks b and b + 8 share an XCD and its L2): every XCD walks a CONTIGUOUS eighth of the block
// index.  The gather kernels below index (batch, query, head) linearly and neighbouring queries sample neighbouring
// value pixels: round-robin, every XCD touched every band of every value map and fetched it for itself (PMC: 135 MB
// per call for a 22 MB map); contiguous, an XCD's L2 sees half a sample's queries and their band of one map.
__device__ __forceinline__ int64_t xcd_block(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, x = bid & 7, s = bid >> 3;
  return (int64_t)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s);
}

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

// Forward, 4 channels per lane: the bilinear set-up of a (query, head, level, point) sample is the same for every
// channel, so with one channel per lane it is recomputed by 32 lanes; here a lane owns 4 adjacent channels (one
// 16-byte piece of the 128-byte pixel row) and 8 lanes cover a head — a quarter of the threads and of the
// set-up instructions for the same gathers, issued as 16-byte loads.
// 4 adjacent channels of a value row in its storage type VK (MBV_DT_F32 / _BF16 / _F16): 16 or 8 bytes per gather.  The
// 16-bit forms halve the bytes every bilinear tap pulls through L2 — what the forward and the location / weight gradient are
// bound by (48 taps of 128 B per (query, head) in f32); under 16-bit autocast the reference's value IS a 16-bit Linear output.
template <int VK>
__device__ __forceinline__ float4 value4(const void* __restrict__ base, int64_t elem) {
  if constexpr (VK == MBV_DT_F32) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
  } else {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
    if constexpr (VK == MBV_DT_BF16)
      return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                         __uint_as_float(u.y & 0xffff0000u));
    return make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(u.x & 0xffffu)),
                       (float)__builtin_bit_cast(_Float16, (unsigned short)(u.x >> 16)),
                       (float)__builtin_bit_cast(_Float16, (unsigned short)(u.y & 0xffffu)),
                       (float)__builtin_bit_cast(_Float16, (unsigned short)(u.y >> 16)));
  }
}

template <int VK>
__global__ void __launch_bounds__(256) k_msda_fwd_v4(const void* __restrict__ value,
                                                     const int64_t* __restrict__ shapes,
                                                     const int64_t* __restrict__ level_start,
                                                     const float* __restrict__ loc, const float* __restrict__ attn,
                                                     int64_t total4, int num_value, int heads, int dim, int levels,
                                                     int num_query, int points, float* __restrict__ out) {
  const int64_t idx = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int d4n = dim >> 2;
  const int d4 = (int)(idx % d4n);
  int64_t t = idx / d4n;
  const int hd = (int)(t % heads);
  t /= heads;                       // t = b * num_query + q
  const int b = (int)(t / num_query);
  const int stride_pix = heads * dim;
  const int64_t vb = (int64_t)b * num_value * stride_pix + hd * dim + d4 * 4;
  const int64_t lw_base = (t * heads + hd) * levels * points;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < levels; ++l) {
    const int h = (int)shapes[l * 2], w = (int)shapes[l * 2 + 1];
    const int64_t vl = vb + level_start[l] * stride_pix;
    for (int p = 0; p < points; ++p) {
      const int64_t k = lw_base + l * points + p;
      const float lx = loc[k * 2], ly = loc[k * 2 + 1], aw = attn[k];
      Corner c;
      if (bilinear_setup(lx, ly, h, w, stride_pix, c)) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c.off[j] >= 0) {
            const float4 x = value4<VK>(value, vl + c.off[j]);
            v.x += c.wgt[j] * x.x; v.y += c.wgt[j] * x.y; v.z += c.wgt[j] * x.z; v.w += c.wgt[j] * x.w;
          }
        }
        acc.x += aw * v.x; acc.y += aw * v.y; acc.z += aw * v.z; acc.w += aw * v.w;
      }
    }
  }
  *reinterpret_cast<float4*>(out + ((t * heads + hd) * dim + d4 * 4)) = acc;
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

// ---------------------------------------------------------------------------------------------------------
// Backward, banded form (self-attention over the multi-scale token map: num_query == num_value, queries in the
// same level-major raster order as the values).  One workgroup owns a BAND of rows of one level's (h, w) map of
// one (batch, head) as f64 accumulators in LDS (128 KB: 16 rows of a 32-wide level, 8 rows of a 64-wide one) and
// processes the queries whose reference row falls into that band (contiguous query ranges, one per query level).
// A bilinear corner inside the band is a ds_add_f64 (≈ 10 cycles per wave instruction; ds_add_f32 would be ≈ 190,
// scratch/ubench/lds_atomic.hip); a corner that a large offset carries outside the band is a direct global f32
// atomic.  Every (query, point) is handled exactly once, so the result does not depend on how well the
// reference-row heuristic matches the learned offsets — only the share of global atomics does.  The band leaves as
// one flush in the full-rate atomic shape.  Global float atomics drop from 1.06 GB per launch to the flushes
// (≈ 100 MB) plus the out-of-band corners.
// sum over each aligned group of 32 lanes on the DPP network (no LDS crossbar traffic next to the LDS atomics):
// row_shr 1/2/4/8 inside the 16-lane rows, row_bcast:15 into the odd rows; lanes 31 and 63 hold the group sums.
__device__ __forceinline__ float group32_sum_dpp(float v) {
#define MBV_DPP_ADD(ctrl, rmask)                                                                              \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true));
  MBV_DPP_ADD(0x111, 0xf)   // row_shr:1
  MBV_DPP_ADD(0x112, 0xf)   // row_shr:2
  MBV_DPP_ADD(0x114, 0xf)   // row_shr:4
  MBV_DPP_ADD(0x118, 0xf)   // row_shr:8
  MBV_DPP_ADD(0x142, 0xa)   // row_bcast:15 (rows 1 and 3 receive lane 15 of rows 0 and 2)
#undef MBV_DPP_ADD
  return v;
}

struct MsdaBands {
  int levels;
  int h[8], w[8], lstart[8], band_rows[8], bands[8], chunks[8], block_begin[9];
};

__global__ void __launch_bounds__(1024) k_msda_bwd_banded(const float* __restrict__ grad_out,
                                                         const float* __restrict__ value,
                                                         const float* __restrict__ loc, const float* __restrict__ attn,
                                                         MsdaBands cfg, int num_value, int heads, int dim, int points,
                                                         float* __restrict__ grad_value, float* __restrict__ grad_loc,
                                                         float* __restrict__ grad_attn) {
  __shared__ double map[16384];
  int level = 0;
  while (level + 1 < cfg.levels && (int)blockIdx.x >= cfg.block_begin[level + 1]) ++level;
  const int local = blockIdx.x - cfg.block_begin[level];
  const int chunks = cfg.chunks[level], bands = cfg.bands[level];
  const int chunk = local % chunks, band = (local / chunks) % bands, bh = local / (chunks * bands);
  const int hd = bh % heads, b = bh / heads;
  const int h = cfg.h[level], w = cfg.w[level], lstart = cfg.lstart[level], levels = cfg.levels;
  const int R0 = band * cfg.band_rows[level], R1 = min(h, R0 + cfg.band_rows[level]);
  const int map_n = (R1 - R0) * w * dim;
  for (int i = threadIdx.x; i < map_n; i += 1024) map[i] = 0.0;
  // queries whose reference row ((2 y + 1) h_l) / (2 h_q) lies in [R0, R1): one contiguous range per query level
  int rbeg[8], rend[8], total = 0;
#pragma unroll
  for (int lq = 0; lq < 8; ++lq) {
    rbeg[lq] = rend[lq] = 0;
    if (lq < levels) {
      const int hq = cfg.h[lq], wq = cfg.w[lq];
      auto first_row = [&](int R) {          // smallest y with reference row >= R
        const int num = 2 * hq * R - h;
        int y = num <= 0 ? 0 : (num + 2 * h - 1) / (2 * h);
        return y > hq ? hq : y;
      };
      rbeg[lq] = cfg.lstart[lq] + first_row(R0) * wq;
      rend[lq] = cfg.lstart[lq] + (R1 >= h ? hq : first_row(R1)) * wq;
      total += rend[lq] - rbeg[lq];
    }
  }
  const int per = (total + chunks - 1) / chunks;
  const int i0 = min(total, chunk * per), i1 = min(total, i0 + per);
  const int d = threadIdx.x % dim, slot = threadIdx.x / dim, slots = 1024 / dim;
  const int stride_pix = heads * dim;
  const int64_t slab = ((int64_t)b * num_value + lstart) * stride_pix + hd * dim + d;     // value / grad_value base
  __syncthreads();
  // one query per slot and iteration, its points U at a time: the loads of all U points, then their 4U value
  // gathers, are issued together (16 waves per CU have to hide two dependent L2 round trips per query)
  constexpr int U = 4;
  const int nq_mine = i1 - i0;
  for (int n0 = 0; n0 < nq_mine; n0 += slots) {              // wave-uniform trip count
    const int n = n0 + slot;
    const bool qlive = n < nq_mine;
    int qi = i0 + (qlive ? n : nq_mine - 1);
    int q = 0;
#pragma unroll
    for (int lq = 0; lq < 8; ++lq) {         // position qi of the concatenated ranges -> query index
      const int cnt = rend[lq] - rbeg[lq];
      if (qi >= 0 && qi < cnt) q = rbeg[lq] + qi;
      qi -= cnt;
    }
    const int64_t qh = ((int64_t)b * num_value + q) * heads + hd;
    const float go = grad_out[qh * dim + d];
    for (int p0 = 0; p0 < points; p0 += U) {
      bool live[U];
      int64_t kk[U];
      float lx[U], ly[U], aw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        live[u] = qlive && p0 + u < points;
        kk[u] = (qh * levels + level) * points + (p0 + u < points ? p0 + u : points - 1);
        lx[u] = loc[kk[u] * 2];
        ly[u] = loc[kk[u] * 2 + 1];
        aw[u] = attn[kk[u]];
      }
      Corner c[U];
      bool inside[U];
      float v[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        inside[u] = bilinear_setup(lx[u], ly[u], h, w, 1, c[u]);              // offsets in pixels
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[u][j] = 0.f;
          if (inside[u] && c[u].off[j] >= 0) v[u][j] = value[slab + (int64_t)c[u].off[j] * stride_pix];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float g_w = 0.f, g_x = 0.f, g_y = 0.f;
        if (inside[u]) {
          const float tg = go * aw[u];
          if (live[u]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int off = c[u].off[j];
              if (off >= 0) {
                const int rel = off - R0 * w;
                if (rel >= 0 && rel < (R1 - R0) * w) atomicAdd(&map[rel * dim + d], (double)(c[u].wgt[j] * tg));
                else atomicAdd(grad_value + slab + (int64_t)off * stride_pix, c[u].wgt[j] * tg);
              }
            }
          }
          const float uh = 1.f - c[u].lh, uw = 1.f - c[u].lw;
          const float val = c[u].wgt[0] * v[u][0] + c[u].wgt[1] * v[u][1] + c[u].wgt[2] * v[u][2] + c[u].wgt[3] * v[u][3];
          const float gh = -uw * v[u][0] - c[u].lw * v[u][1] + uw * v[u][2] + c[u].lw * v[u][3];
          const float gw = -uh * v[u][0] + uh * v[u][1] - c[u].lh * v[u][2] + c[u].lh * v[u][3];
          g_w = go * val;
          g_x = (float)w * gw * tg;
          g_y = (float)h * gh * tg;
        }
        int writer = 0;                      // lane of the dim-group that ends up holding the sums
        if (dim == 32) {
          g_w = group32_sum_dpp(g_w);
          g_x = group32_sum_dpp(g_x);
          g_y = group32_sum_dpp(g_y);
          writer = 31;
        } else {
          for (int o = dim >> 1; o > 0; o >>= 1) {
            g_w += __shfl_xor(g_w, o, 64);
            g_x += __shfl_xor(g_x, o, 64);
            g_y += __shfl_xor(g_y, o, 64);
          }
        }
        if (live[u] && d == writer) {
          grad_attn[kk[u]] = g_w;
          grad_loc[kk[u] * 2] = g_x;
          grad_loc[kk[u] * 2 + 1] = g_y;
        }
      }
    }
  }
  __syncthreads();
  const int64_t gbase = ((int64_t)b * num_value + lstart + R0 * w) * stride_pix + hd * dim;
  for (int i = threadIdx.x; i < map_n; i += 1024) {
    const double v = map[i];
    if (v != 0.0) atomicAdd(grad_value + gbase + (int64_t)(i / dim) * stride_pix + (i % dim), (float)v);
  }
}

// sum over each aligned group of 8 lanes, result in EVERY lane of the group
__device__ __forceinline__ float group8_allsum(float v) {
#define MBV_DPP_ADD(ctrl, bmask)                                                                              \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, bmask, true));
  MBV_DPP_ADD(0xB1, 0xf)    // + lane ^ 1
  MBV_DPP_ADD(0x4E, 0xf)    // + lane ^ 2
#undef MBV_DPP_ADD
  const float hi = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xa, true));  // row_shr:4
  const float lo = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, true));  // row_shl:4
  return v + hi + lo;       // banks 1, 3 receive the quad below, banks 0, 2 the quad above
}


// ---------------------------------------------------------------------------------------------------------
// Backward with no global atomics at all (head_dim = 32, every level map <= 4096 pixels), two independent parts:
//
//  location / weight part — d(sampling location), d(attention weight): a pure gather like the forward.  8 lanes share a
//      (query, head), a lane owns 4 adjacent channels (16-byte gathers of the value rows), the two reductions over the
//      32 channels are three DPP steps.
//  value part — d(value): one workgroup owns the WHOLE (h, w) map of one level for 4 of the 32 channels of
//      one (batch, head) as f64 LDS accumulators in planar [channel][pixel] order (64 x 64 x 4 x 8 B = 128 KB at the
//      finest level), one lane per query: a corner is four ds_add_f64 whose 64 lanes (64 consecutive queries, so
//      mostly consecutive pixels) fall on consecutive 8-byte slots.  Nothing can land outside the map, so there are
//      no out-of-band global atomics (160 us of the banded kernel's 400), every element of grad_value is written
//      exactly once by plain stores (no zero fill, no flush atomics), and the value map is not read here at all.
//      The price is that the 8 channel groups each redo the bilinear set-up of every sample (ALU the banded kernel
//      left idle).
// d(value) of ONE level: workgroup = (batch, head, 4-channel group); see the comment above.  Launched once per level so
// that the dynamic LDS is that level's map (128 KB only at the finest level; the coarser levels' workgroups then share
// CUs).  P = 4: the four sampling points of a (query, head, level) are one 32-byte + one 16-byte load.
template <int P>
__global__ void __launch_bounds__(1024) k_msda_bwd_value(const float* __restrict__ grad_out,
                                                        const float* __restrict__ loc, const float* __restrict__ attn,
                                                        int level, int levels, int h, int w, int lstart, int num_value,
                                                        int num_query, int heads, int points_rt,
                                                        float* __restrict__ grad_value) {
  constexpr int dim = 32, CG = 4;
  const int points = P > 0 ? P : points_rt;
  extern __shared__ __attribute__((aligned(16))) double map[];      // [CG][h * w]
  // blocks b and b + 8 share an XCD: with batch * heads a multiple of 8 the eight channel groups of a (batch, head),
  // which read the same grad_out rows, locations and weights, land on ONE XCD's L2 (speed only)
  const int nbh = gridDim.x >> 3;
  const int split = blockIdx.x / nbh, bhid = blockIdx.x - split * nbh;
  const int hd = bhid % heads, b = bhid / heads;
  const int npix = h * w, stride_pix = heads * dim;
  for (int i = threadIdx.x; i < CG * npix; i += 1024) map[i] = 0.0;
  __syncthreads();
  // P == 4: the loads of the thread's NEXT query (gradient row, 4 locations, 4 weights: 64 B) are issued before the
  // current query's 64 LDS adds, so that their round trip runs underneath them — the 16 waves of the workgroup share one
  // LDS-atomic pipe and otherwise fall into lockstep: everybody waits for memory, then everybody queues adds
  float4 n_go = make_float4(0.f, 0.f, 0.f, 0.f), n_l01 = n_go, n_l23 = n_go, n_a4 = n_go;
  if constexpr (P == 4) {
    if ((int)threadIdx.x < num_query) {
      const int64_t qh0 = ((int64_t)b * num_query + threadIdx.x) * heads + hd;
      const int64_t kb0 = (qh0 * levels + level) * 4;
      n_go = *reinterpret_cast<const float4*>(grad_out + qh0 * dim + split * CG);
      n_l01 = *reinterpret_cast<const float4*>(loc + kb0 * 2);
      n_l23 = *reinterpret_cast<const float4*>(loc + kb0 * 2 + 4);
      n_a4 = *reinterpret_cast<const float4*>(attn + kb0);
    }
  }
  for (int q = threadIdx.x; q < num_query; q += 1024) {
    const int64_t qh = ((int64_t)b * num_query + q) * heads + hd;
    float4 go;
    float4 c_l01, c_l23, c_a4;
    if constexpr (P == 4) {
      go = n_go; c_l01 = n_l01; c_l23 = n_l23; c_a4 = n_a4;
      if (q + 1024 < num_query) {
        const int64_t qh1 = ((int64_t)b * num_query + q + 1024) * heads + hd;
        const int64_t kb1 = (qh1 * levels + level) * 4;
        n_go = *reinterpret_cast<const float4*>(grad_out + qh1 * dim + split * CG);
        n_l01 = *reinterpret_cast<const float4*>(loc + kb1 * 2);
        n_l23 = *reinterpret_cast<const float4*>(loc + kb1 * 2 + 4);
        n_a4 = *reinterpret_cast<const float4*>(attn + kb1);
      }
    } else {
      go = *reinterpret_cast<const float4*>(grad_out + qh * dim + split * CG);
    }
    const int64_t kb = (qh * levels + level) * points;
    constexpr int U = P > 0 ? P : 1;
    for (int p0 = 0; p0 < points; p0 += U) {
      float lx[U], ly[U], aw[U];
      if constexpr (P == 4) {
        const float4 l01 = c_l01;
        const float4 l23 = c_l23;
        const float4 a4 = c_a4;
        lx[0] = l01.x; ly[0] = l01.y; lx[1] = l01.z; ly[1] = l01.w;
        lx[2] = l23.x; ly[2] = l23.y; lx[3] = l23.z; ly[3] = l23.w;
        aw[0] = a4.x; aw[1] = a4.y; aw[2] = a4.z; aw[3] = a4.w;
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          lx[u] = loc[(kb + p0 + u) * 2];
          ly[u] = loc[(kb + p0 + u) * 2 + 1];
          aw[u] = attn[kb + p0 + u];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Corner c;
        if (!bilinear_setup(lx[u], ly[u], h, w, 1, c)) continue;          // offsets in pixels
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int off = c.off[j];
          if (off < 0) continue;
          const float wj = c.wgt[j] * aw[u];
          atomicAdd(&map[off], (double)(wj * go.x));
          atomicAdd(&map[npix + off], (double)(wj * go.y));
          atomicAdd(&map[2 * npix + off], (double)(wj * go.z));
          atomicAdd(&map[3 * npix + off], (double)(wj * go.w));
        }
      }
    }
  }
  __syncthreads();
  float* gv = grad_value + ((int64_t)b * num_value + lstart) * stride_pix + hd * dim + split * CG;
  for (int i = threadIdx.x; i < npix; i += 1024)
    *reinterpret_cast<float4*>(gv + (int64_t)i * stride_pix) =
        make_float4((float)map[i], (float)map[npix + i], (float)map[2 * npix + i], (float)map[3 * npix + i]);
}


// ---------------------------------------------------------------------------------------------------------
// d(value) with PACKED FIXED-POINT LDS accumulators (round 3; 16-bit compute modes).
//
// The f64 form above is bound by the LDS atomic pipe: one `ds_add_f64` wave instruction costs 9-16 clocks whatever
// its active lanes and carries ONE f32 addend per lane — 16 instructions per sample for a 4-channel group.  Integer
// LDS atomics cost the same per instruction (`ds_add_u64` ≈ 9 clocks, scratch/ubench/lds_atomic.hip), and an integer
// sum can carry TWO addends: a contribution w·g of two adjacent channels is rounded to two signed 32-bit fixed-point
// numbers and added as one 64-bit word `(hi << 32) + lo` (two's complement: a negative `lo` borrows from `hi`, and the
// borrow is undone exactly when the word is split again, as long as both sums stay inside 32 bits).  Half the atomic
// instructions, the cheaper instruction, half the LDS per channel — so one launch holds all three levels (the
// finest level's blocks take 2 channels = one 32 KB plane, the coarser levels' 4 channels = two planes) with every
// block resident at once — and integer addition is associative: the result does not depend on the order in which the
// waves arrive (bit-reproducible, like the f64 form).
//
// Range / precision: a block first takes L1 = sum over its queries of (attention mass of the (query, head) on this
// level) x (largest |grad_out| among the block's channels) from the rows it is going to read anyway.  The bilinear
// weight of a corner is in [0, 1] and a sample puts at most one corner on a pixel, so |sum at a pixel| <= L1 for ANY
// sampling pattern and any weights; with the scale 2^(29 - e), L1 < 2^e, the 32-bit halves cannot wrap (one bit of
// margin for the roundings).  Every addend is rounded to 2^(e - 30): with 5 376 queries and unit-variance gradients
// about 4e-6 — three decimal digits finer than the bf16 / fp16 GEMM operands on either side of this kernel.  The f32
// compute mode keeps the f64 accumulators.
//
// Output: written straight in the caller's dtype (f32, or the 16-bit type of the matrix whose first columns are
// d(value) — ops._MSDAQuerySide's [d value | d offsets | d logits]) with the caller's row stride.
struct MsdaFxArgs {
  const float* grad_out;
  const float* loc;
  const float* attn;
  // stream-ordered copies written by k_msda_bwd_relayout (see there): what a block reads is CONTIGUOUS in them
  const float* go_t;     // [B][H][8][Nq][4]      grad_out by 4-channel group
  const float* loc_t;    // [B][H][L][Nq][4][2]   sampling locations by level
  const float* attn_t;   // [B][H][L][Nq][4]      attention weights by level
  const float* l1_part;  // [B][H][tiles][L][8]   per 64-query tile: sum of (attention mass) x (max |g| of the group)
  int q_tiles;
  void* out;
  int64_t out_ld;        // elements between consecutive (b, n) rows of `out`
  int out_dt;            // MBV_DT_F32 / BF16 / F16
  int levels, num_value, num_query, heads, frac_bits, batch;
  int h[8], w[8], lstart[8];
  int planes[8];         // u64 planes per block of that level: 1 (2 channels) or 2 (4 channels)
  int block_begin[9];    // first block of each level's range
};

__device__ __forceinline__ unsigned long long fx_pack(float a, float b) {      // a -> low half, b -> high half
  const int lo = __float2int_rn(a), hi = __float2int_rn(b);
  return ((unsigned long long)(unsigned)(hi + (lo >> 31)) << 32) | (unsigned)lo;
}

// two adjacent channels in the output's dtype (DT = MBV_DT_F32 / BF16 / F16), one 8- or 4-byte store
template <int DT>
__device__ __forceinline__ void fx_store2(void* base, int64_t elem, float a, float b) {
  if constexpr (DT == MBV_DT_F32) {
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(base) + elem) = make_float2(a, b);
  } else {
    unsigned ua, ub;
    if constexpr (DT == MBV_DT_BF16) {
      ua = f32_to_bf16_rne(a); ub = f32_to_bf16_rne(b);
    } else {
      ua = __builtin_bit_cast(unsigned short, (_Float16)a); ub = __builtin_bit_cast(unsigned short, (_Float16)b);
    }
    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(base) + elem) = ua | (ub << 16);
  }
}

template <int PLANES, int DT>
__device__ __forceinline__ void msda_value_fx_block(const MsdaFxArgs& A, int level, int local, unsigned long long* map) {
  constexpr int dim = 32, CH = 2 * PLANES, NT = 512;
  const int nbh = A.batch * A.heads;
  // block -> (group, batch*head): the groups of one (batch, head) sit at ids equal modulo 8, i.e. on one XCD / L2
  const int split = local / nbh, bhid = local - split * nbh;
  const int hd = bhid % A.heads, b = bhid / A.heads;
  const int h = A.h[level], w = A.w[level], npix = h * w;
  const int num_query = A.num_query, heads = A.heads, levels = A.levels;
  for (int i = threadIdx.x; i < PLANES * npix; i += NT) map[i] = 0ull;
  // Range: |sum at a pixel| <= L1 := sum over queries of (this level's attention mass of the (query, head)) x (largest
  // |grad_out| among this block's channels) — every addend is (bilinear weight <= 1) x (attention weight) x g, and a
  // sample puts at most one corner on a pixel.  The block computes L1 from the rows it is going to read anyway and
  // scales by 2^(29 - e), L1 < 2^e: the 32-bit halves cannot wrap (one bit of margin for the roundings), whatever the
  // sampling pattern, and typical inputs get 4-5 bits more than the a-priori bound num_query * max|g| would leave.
  __shared__ float red[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // L1 = the sum of the relayout pass's per-tile partial sums for this (batch, head, level, 4-channel group)
  const int grp4 = PLANES == 2 ? split : split >> 1;       // a 2-channel block uses its 4-channel group's (larger) bound
  float l1 = 0.f;
  for (int t = threadIdx.x; t < A.q_tiles; t += NT)
    l1 += A.l1_part[((((int64_t)b * heads + hd) * A.q_tiles + t) * levels + level) * 8 + grp4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o);
  if (lane == 0) red[wave] = l1;
  __syncthreads();
  l1 = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) l1 += red[i];
  l1 *= 1.0001f;                                       // the f32 sum itself is rounded
  if (!(l1 == l1) || l1 > 3.0e38f) l1 = 3.0e38f;      // NaN / inf gradients: no meaningful scale, keep the kernel defined
  int e = 0;
  if (l1 > 0.f) (void)frexpf(l1, &e);                 // l1 = f * 2^e, f in [0.5, 1): l1 < 2^e
  const float scale = ldexpf(1.f, 29 - e), inv_scale = ldexpf(1.f, e - 29);
  // lane = query: the 64 lanes of an atomic instruction are 64 consecutive queries, i.e. mostly consecutive pixels on
  // consecutive 8-byte slots — conflict-free banks.  (Spreading the lanes QS queries apart to avoid same-pixel meetings
  // was measured: 131 -> 169 us, the scattered slots collide on banks far more often than neighbours meet on a pixel.)
  // The next query's 64 bytes are requested before the current query's adds.
  float4 n_go = make_float4(0.f, 0.f, 0.f, 0.f), n_l01 = n_go, n_l23 = n_go, n_a4 = n_go;
  // consecutive lanes read consecutive 16 / 32 / 16-byte records: full cache lines (from the original tensors a lane's
  // 16 bytes of grad_out sit 1 KB from its neighbour's, its locations 768 B: 64 lines per load instruction, and the
  // texture-address unit works per line — the kernel was bound by that, not by the LDS atomics)
  const float* go_b = A.go_t + ((((int64_t)b * heads + hd) * 8 + grp4) * num_query) * 4 + (PLANES == 2 ? 0 : (split & 1) * 2);
  const float* loc_b = A.loc_t + ((((int64_t)b * heads + hd) * levels + level) * num_query) * 8;
  const float* attn_b = A.attn_t + ((((int64_t)b * heads + hd) * levels + level) * num_query) * 4;
  auto load_q = [&](int q, float4& go, float4& l01, float4& l23, float4& a4) {
    if constexpr (PLANES == 2) {
      go = *reinterpret_cast<const float4*>(go_b + (int64_t)q * 4);
    } else {
      const float2 g = *reinterpret_cast<const float2*>(go_b + (int64_t)q * 4);
      go = make_float4(g.x, g.y, 0.f, 0.f);
    }
    l01 = *reinterpret_cast<const float4*>(loc_b + (int64_t)q * 8);
    l23 = *reinterpret_cast<const float4*>(loc_b + (int64_t)q * 8 + 4);
    a4 = *reinterpret_cast<const float4*>(attn_b + (int64_t)q * 4);
  };
  if ((int)threadIdx.x < num_query) load_q(threadIdx.x, n_go, n_l01, n_l23, n_a4);
  for (int q = threadIdx.x; q < num_query; q += NT) {
    float4 go = n_go;
    const float4 l01 = n_l01, l23 = n_l23, a4 = n_a4;
    if (q + NT < num_query) load_q(q + NT, n_go, n_l01, n_l23, n_a4);
    go.x *= scale; go.y *= scale; go.z *= scale; go.w *= scale;
    const float lx[4] = {l01.x, l01.z, l23.x, l23.z}, ly[4] = {l01.y, l01.w, l23.y, l23.w};
    const float aw[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      Corner c;
      if (!bilinear_setup(lx[u], ly[u], h, w, 1, c)) continue;          // offsets in pixels
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int off = c.off[j];
        if (off < 0) continue;
        const float wj = c.wgt[j] * aw[u];
        atomicAdd(&map[off], fx_pack(wj * go.x, wj * go.y));
        if constexpr (PLANES == 2) atomicAdd(&map[npix + off], fx_pack(wj * go.z, wj * go.w));
      }
    }
  }
  __syncthreads();
  const int64_t gv = ((int64_t)b * A.num_value + A.lstart[level]) * A.out_ld + hd * dim + split * CH;
  for (int i = threadIdx.x; i < npix; i += NT) {
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) {
      const long long v = (long long)map[pl * npix + i];
      const int lo = (int)(unsigned)(v & 0xffffffffll);
      const int hi = (int)((v - (long long)lo) >> 32);
      fx_store2<DT>(A.out, gv + (int64_t)i * A.out_ld + 2 * pl, (float)lo * inv_scale, (float)hi * inv_scale);
    }
  }
}

// The same forward for num_points == 4 without a branch between a tap's address and its load: out-of-range taps read element 0
// of the level's slab with weight 0, so a level's 4 x 4 gathers (and, before them, its 4 locations and weights as three
// 16-byte loads) are all in flight before the first one is consumed — the branchy form waited for a (location → address →
// gather) round trip per sample, 12 in a row: it is bound by that latency, not by bytes (a 16-bit value map changed nothing).
struct Taps4 {
  int off[4];
  float wgt[4];
};
__device__ __forceinline__ void taps_clamped(float loc_x, float loc_y, float aw, int h, int w, int stride_pix, Taps4& t) {
  const float him = loc_y * (float)h - 0.5f;
  const float wim = loc_x * (float)w - 0.5f;
  const bool in = him > -1.f && wim > -1.f && him < (float)h && wim < (float)w;
  const int hl = (int)floorf(him), wl = (int)floorf(wim);
  const int hh = hl + 1, wh = wl + 1;
  const float lh = him - (float)hl, lw = wim - (float)wl;
  const float uh = 1.f - lh, uw = 1.f - lw;
  const bool t0 = in && hl >= 0 && wl >= 0, t1 = in && hl >= 0 && wh <= w - 1;
  const bool t2 = in && hh <= h - 1 && wl >= 0, t3 = in && hh <= h - 1 && wh <= w - 1;
  t.off[0] = t0 ? (hl * w + wl) * stride_pix : 0;
  t.off[1] = t1 ? (hl * w + wh) * stride_pix : 0;
  t.off[2] = t2 ? (hh * w + wl) * stride_pix : 0;
  t.off[3] = t3 ? (hh * w + wh) * stride_pix : 0;
  t.wgt[0] = t0 ? aw * uh * uw : 0.f;
  t.wgt[1] = t1 ? aw * uh * lw : 0.f;
  t.wgt[2] = t2 ? aw * lh * uw : 0.f;
  t.wgt[3] = t3 ? aw * lh * lw : 0.f;
}

template <int VK>
__global__ void __launch_bounds__(256) k_msda_fwd_p4(const void* __restrict__ value, const int64_t* __restrict__ shapes,
                                                     const int64_t* __restrict__ level_start,
                                                     const float* __restrict__ loc, const float* __restrict__ attn,
                                                     int64_t total4, int num_value, int heads, int dim, int levels,
                                                     int num_query, float* __restrict__ out) {
  const int64_t idx = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int d4n = dim >> 2;
  const int d4 = (int)(idx % d4n);
  int64_t t = idx / d4n;
  const int hd = (int)(t % heads);
  t /= heads;                       // t = b * num_query + q
  const int b = (int)(t / num_query);
  const int stride_pix = heads * dim;
  const int64_t vb = (int64_t)b * num_value * stride_pix + hd * dim + d4 * 4;
  const int64_t lw_base = (t * heads + hd) * levels * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < levels; ++l) {
    const int h = (int)shapes[l * 2], w = (int)shapes[l * 2 + 1];
    const int64_t vl = vb + level_start[l] * stride_pix;
    const float4 xy0 = *reinterpret_cast<const float4*>(loc + (lw_base + l * 4) * 2);
    const float4 xy1 = *reinterpret_cast<const float4*>(loc + (lw_base + l * 4) * 2 + 4);
    const float4 aw = *reinterpret_cast<const float4*>(attn + lw_base + l * 4);
    Taps4 tp[4];
    taps_clamped(xy0.x, xy0.y, aw.x, h, w, stride_pix, tp[0]);
    taps_clamped(xy0.z, xy0.w, aw.y, h, w, stride_pix, tp[1]);
    taps_clamped(xy1.x, xy1.y, aw.z, h, w, stride_pix, tp[2]);
    taps_clamped(xy1.z, xy1.w, aw.w, h, w, stride_pix, tp[3]);
    float4 x[16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[p * 4 + j] = value4<VK>(value, vl + tp[p].off[j]);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float wj = tp[p].wgt[j];
        acc.x += wj * x[p * 4 + j].x; acc.y += wj * x[p * 4 + j].y;
        acc.z += wj * x[p * 4 + j].z; acc.w += wj * x[p * 4 + j].w;
      }
  }
  *reinterpret_cast<float4*>(out + ((t * heads + hd) * dim + d4 * 4)) = acc;
}

// Re-layout pass of the packed value-gradient kernel: one block = (batch, head, 64 consecutive queries).  Reads the
// block's grad_out (64 x 128 B), locations (64 x L x 32 B) and weights (64 x L x 16 B) in full lines, turns them through
// LDS and writes them where the value kernel's blocks read CONTIGUOUS streams: grad_out by 4-channel group, locations /
// weights by level.  It also leaves, per (batch, head, tile, level, group), the tile's share of the range bound L1.
// num_points == 4, head_dim == 32, levels <= 8.
__global__ void __launch_bounds__(256) k_msda_bwd_relayout(const float* __restrict__ grad_out,
                                                           const float* __restrict__ loc, const float* __restrict__ attn,
                                                           int num_query, int heads, int levels, int q_tiles,
                                                           float* __restrict__ go_t, float* __restrict__ loc_t,
                                                           float* __restrict__ attn_t, float* __restrict__ l1_part) {
  __shared__ __attribute__((aligned(16))) float s_go[64 * 32];          // [q][32 ch]
  __shared__ __attribute__((aligned(16))) float s_loc[64 * 8 * 8];      // [q][l][8]   (levels <= 8)
  __shared__ __attribute__((aligned(16))) float s_at[64 * 8 * 4];       // [q][l][4]
  __shared__ float s_sum[8 * 8];                                         // [l][group]
  const int tile = blockIdx.x % q_tiles;
  const int bh = blockIdx.x / q_tiles;
  const int hd = bh % heads, b = bh / heads;
  const int q0 = tile * 64;
  const int tid = threadIdx.x;
  if (tid < 64) s_sum[tid] = 0.f;
  // grad_out: 64 queries x 8 pieces of 16 B
  for (int p = tid; p < 64 * 8; p += 256) {
    const int q = p >> 3, c = p & 7;
    const int qq = q0 + q < num_query ? q0 + q : num_query - 1;
    *reinterpret_cast<float4*>(s_go + q * 32 + c * 4) =
        *reinterpret_cast<const float4*>(grad_out + (((int64_t)b * num_query + qq) * heads + hd) * 32 + c * 4);
  }
  const int lp = levels * 2;                           // 16-byte pieces of locations per (query, head): L x 4 points x 2 / 4
  for (int p = tid; p < 64 * lp; p += 256) {
    const int q = p / lp, c = p - q * lp;
    const int qq = q0 + q < num_query ? q0 + q : num_query - 1;
    *reinterpret_cast<float4*>(s_loc + q * 64 + c * 4) =
        *reinterpret_cast<const float4*>(loc + ((((int64_t)b * num_query + qq) * heads + hd) * levels) * 8 + c * 4);
  }
  for (int p = tid; p < 64 * levels; p += 256) {
    const int q = p / levels, l = p - q * levels;
    const int qq = q0 + q < num_query ? q0 + q : num_query - 1;
    *reinterpret_cast<float4*>(s_at + q * 32 + l * 4) =
        *reinterpret_cast<const float4*>(attn + ((((int64_t)b * num_query + qq) * heads + hd) * levels + l) * 4);
  }
  __syncthreads();
  // grad_out by group: piece (g, q)
  for (int p = tid; p < 8 * 64; p += 256) {
    const int g = p >> 6, q = p & 63;
    if (q0 + q < num_query)
      *reinterpret_cast<float4*>(go_t + ((((int64_t)b * heads + hd) * 8 + g) * num_query + q0 + q) * 4) =
          *reinterpret_cast<const float4*>(s_go + q * 32 + g * 4);
  }
  for (int p = tid; p < levels * 64 * 2; p += 256) {   // locations by level: piece (l, q, half)
    const int l = p / 128, r = p - l * 128, q = r >> 1, hf = r & 1;
    if (q0 + q < num_query)
      *reinterpret_cast<float4*>(loc_t + ((((int64_t)b * heads + hd) * levels + l) * num_query + q0 + q) * 8 + hf * 4) =
          *reinterpret_cast<const float4*>(s_loc + q * 64 + l * 8 + hf * 4);
  }
  for (int p = tid; p < levels * 64; p += 256) {       // weights by level: piece (l, q)
    const int l = p >> 6, q = p & 63;
    if (q0 + q < num_query)
      *reinterpret_cast<float4*>(attn_t + ((((int64_t)b * heads + hd) * levels + l) * num_query + q0 + q) * 4) =
          *reinterpret_cast<const float4*>(s_at + q * 32 + l * 4);
  }
  // range-bound partial sums: (query, group) pairs, 64 x 8 = 512 over 256 threads; LDS f32 sums of <= 64 addends
  for (int p = tid; p < 64 * 8; p += 256) {
    const int q = p & 63, g = p >> 6;
    const bool live = q0 + q < num_query;              // (no early exit: the whole wave takes part in the shuffles)
    const float4 gv = *reinterpret_cast<const float4*>(s_go + q * 32 + g * 4);
    const float m = fmaxf(fmaxf(fabsf(gv.x), fabsf(gv.y)), fmaxf(fabsf(gv.z), fabsf(gv.w)));
    for (int l = 0; l < levels; ++l) {
      const float4 a = *reinterpret_cast<const float4*>(s_at + q * 32 + l * 4);
      float t = live ? m * (fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w)) : 0.f;
      // the 64 queries of a group are the 64 lanes of one wave: reduce in the wave, one LDS add per (level, group)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if ((tid & 63) == 0) s_sum[l * 8 + g] = t;       // one wave per (g): no race (p = tid + 256 * it -> g = tid / 64 + 4 it)
    }
  }
  __syncthreads();
  if (tid < levels * 8) {
    const int l = tid >> 3, g = tid & 7;
    l1_part[((((int64_t)b * heads + hd) * q_tiles + tile) * levels + l) * 8 + g] = s_sum[l * 8 + g];
  }
}

template <int DT>
__global__ void __launch_bounds__(512) k_msda_bwd_value_fx(const MsdaFxArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long fxmap[];
  int level = 0;                                     // block ranges: block_begin[l] .. block_begin[l + 1]
  for (int l = 1; l < A.levels; ++l)
    if ((int)blockIdx.x >= A.block_begin[l]) level = l;
  const int local = (int)blockIdx.x - A.block_begin[level];
  if (A.planes[level] == 2)
    msda_value_fx_block<2, DT>(A, level, local, fxmap);
  else
    msda_value_fx_block<1, DT>(A, level, local, fxmap);
}

// d(location), d(weight): 8 lanes per (query, head), 4 channels per lane
template <int VK>
__global__ void __launch_bounds__(256) k_msda_bwd_locattn(const float* __restrict__ grad_out,
                                                          const void* __restrict__ value,
                                                          const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ level_start,
                                                          const float* __restrict__ loc, const float* __restrict__ attn,
                                                          int64_t total8, int num_value, int heads, int levels,
                                                          int num_query, int points, float* __restrict__ grad_loc,
                                                          float* __restrict__ grad_attn) {
  constexpr int dim = 32;
  const int stride_pix = heads * dim;
  const int64_t idx = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const bool live = idx < total8;                     // total8 is a multiple of 8: a group of 8 lanes is all in or out
  const int64_t sidx = live ? idx : total8 - 1;
  const int d4 = (int)(sidx & 7);
  int64_t t = sidx >> 3;
  const int hd = (int)(t % heads);
  t /= heads;                                         // t = b * num_query + q
  const int b = (int)(t / num_query);
  const int64_t vb = (int64_t)b * num_value * stride_pix + hd * dim + d4 * 4;
  const int64_t lw_base = (t * heads + hd) * levels * points;
  const float4 go = *reinterpret_cast<const float4*>(grad_out + (t * heads + hd) * dim + d4 * 4);
  float keep_w = 0.f, keep_x = 0.f, keep_y = 0.f;
  for (int l = 0; l < levels; ++l) {
    const int h = (int)shapes[l * 2], w = (int)shapes[l * 2 + 1];
    const int64_t vl = vb + level_start[l] * stride_pix;
    for (int p = 0; p < points; ++p) {
      const int64_t k = lw_base + l * points + p;
      const float lx = loc[k * 2], ly = loc[k * 2 + 1], aw = attn[k];
      float g_w = 0.f, g_x = 0.f, g_y = 0.f;
      Corner c;
      if (bilinear_setup(lx, ly, h, w, stride_pix, c)) {
        float s4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s4[j] = 0.f;
          if (c.off[j] >= 0) {
            const float4 x = value4<VK>(value, vl + c.off[j]);
            s4[j] = go.x * x.x + go.y * x.y + go.z * x.z + go.w * x.w;
          }
        }
        const float uh = 1.f - c.lh, uw = 1.f - c.lw;
        g_w = c.wgt[0] * s4[0] + c.wgt[1] * s4[1] + c.wgt[2] * s4[2] + c.wgt[3] * s4[3];
        const float gh = -uw * s4[0] - c.lw * s4[1] + uw * s4[2] + c.lw * s4[3];
        const float gw = -uh * s4[0] + uh * s4[1] - c.lh * s4[2] + c.lh * s4[3];
        g_x = (float)w * gw * aw;
        g_y = (float)h * gh * aw;
      }
      g_w = group8_allsum(g_w);
      g_x = group8_allsum(g_x);
      g_y = group8_allsum(g_y);
      // lane d4 of the group keeps sample s when s % 8 == d4: the group's stores are then runs of 8 consecutive
      // samples (32 B of weights, 64 B of locations) instead of one 4-byte store per sample
      const int sidx8 = l * points + p;
      if ((sidx8 & 7) == d4) { keep_w = g_w; keep_x = g_x; keep_y = g_y; }
      if ((sidx8 & 7) == 7 || (l == levels - 1 && p == points - 1)) {
        const int first = sidx8 & ~7;
        if (live && first + d4 <= sidx8) {
          grad_attn[lw_base + first + d4] = keep_w;
          *reinterpret_cast<float2*>(grad_loc + (lw_base + first + d4) * 2) = make_float2(keep_x, keep_y);
        }
      }
    }
  }
}

// The 4-point form of the kernel above with a level's 16 gathers in flight together (cf. k_msda_fwd_p4).
struct TapsG {
  int off[4];
  float ok[4];       // 1 / 0: the tap exists
  float lh, lw;
  float in;          // 1 / 0: the sample lies inside the padded map at all
};
__device__ __forceinline__ void taps_grad(float loc_x, float loc_y, int h, int w, int stride_pix, TapsG& t) {
  const float him = loc_y * (float)h - 0.5f;
  const float wim = loc_x * (float)w - 0.5f;
  const bool in = him > -1.f && wim > -1.f && him < (float)h && wim < (float)w;
  const int hl = (int)floorf(him), wl = (int)floorf(wim);
  const int hh = hl + 1, wh = wl + 1;
  t.lh = him - (float)hl;
  t.lw = wim - (float)wl;
  t.in = in ? 1.f : 0.f;
  const bool t0 = in && hl >= 0 && wl >= 0, t1 = in && hl >= 0 && wh <= w - 1;
  const bool t2 = in && hh <= h - 1 && wl >= 0, t3 = in && hh <= h - 1 && wh <= w - 1;
  t.off[0] = t0 ? (hl * w + wl) * stride_pix : 0;
  t.off[1] = t1 ? (hl * w + wh) * stride_pix : 0;
  t.off[2] = t2 ? (hh * w + wl) * stride_pix : 0;
  t.off[3] = t3 ? (hh * w + wh) * stride_pix : 0;
  t.ok[0] = t0 ? 1.f : 0.f; t.ok[1] = t1 ? 1.f : 0.f; t.ok[2] = t2 ? 1.f : 0.f; t.ok[3] = t3 ? 1.f : 0.f;
}

template <int VK>
__global__ void __launch_bounds__(256) k_msda_bwd_locattn_p4(const float* __restrict__ grad_out,
                                                             const void* __restrict__ value,
                                                             const int64_t* __restrict__ shapes,
                                                             const int64_t* __restrict__ level_start,
                                                             const float* __restrict__ loc, const float* __restrict__ attn,
                                                             int64_t total8, int num_value, int heads, int levels,
                                                             int num_query, float* __restrict__ grad_loc,
                                                             float* __restrict__ grad_attn) {
  constexpr int dim = 32, points = 4;
  const int stride_pix = heads * dim;
  const int64_t idx = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const bool live = idx < total8;                     // total8 is a multiple of 8: a group of 8 lanes is all in or out
  const int64_t sidx = live ? idx : total8 - 1;
  const int d4 = (int)(sidx & 7);
  int64_t t = sidx >> 3;
  const int hd = (int)(t % heads);
  t /= heads;                                         // t = b * num_query + q
  const int b = (int)(t / num_query);
  const int64_t vb = (int64_t)b * num_value * stride_pix + hd * dim + d4 * 4;
  const int64_t lw_base = (t * heads + hd) * levels * points;
  const float4 go = *reinterpret_cast<const float4*>(grad_out + (t * heads + hd) * dim + d4 * 4);
  float keep_w = 0.f, keep_x = 0.f, keep_y = 0.f;
  for (int l = 0; l < levels; ++l) {
    const int h = (int)shapes[l * 2], w = (int)shapes[l * 2 + 1];
    const int64_t vl = vb + level_start[l] * stride_pix;
    const float4 xy0 = *reinterpret_cast<const float4*>(loc + (lw_base + l * 4) * 2);
    const float4 xy1 = *reinterpret_cast<const float4*>(loc + (lw_base + l * 4) * 2 + 4);
    const float4 aw4 = *reinterpret_cast<const float4*>(attn + lw_base + l * 4);
    const float awp[4] = {aw4.x, aw4.y, aw4.z, aw4.w};
    TapsG tp[4];
    taps_grad(xy0.x, xy0.y, h, w, stride_pix, tp[0]);
    taps_grad(xy0.z, xy0.w, h, w, stride_pix, tp[1]);
    taps_grad(xy1.x, xy1.y, h, w, stride_pix, tp[2]);
    taps_grad(xy1.z, xy1.w, h, w, stride_pix, tp[3]);
    float4 x[16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[p * 4 + j] = value4<VK>(value, vl + tp[p].off[j]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float s4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = x[p * 4 + j];
        s4[j] = tp[p].ok[j] * (go.x * v.x + go.y * v.y + go.z * v.z + go.w * v.w);
      }
      const float lh = tp[p].lh, lw = tp[p].lw, uh = 1.f - lh, uw = 1.f - lw;
      float g_w = uh * uw * s4[0] + uh * lw * s4[1] + lh * uw * s4[2] + lh * lw * s4[3];
      const float gh = -uw * s4[0] - lw * s4[1] + uw * s4[2] + lw * s4[3];
      const float gw = -uh * s4[0] + uh * s4[1] - lh * s4[2] + lh * s4[3];
      float g_x = (float)w * gw * awp[p] * tp[p].in;
      float g_y = (float)h * gh * awp[p] * tp[p].in;
      g_w = group8_allsum(g_w * tp[p].in);
      g_x = group8_allsum(g_x);
      g_y = group8_allsum(g_y);
      const int sidx8 = l * points + p;
      if ((sidx8 & 7) == d4) { keep_w = g_w; keep_x = g_x; keep_y = g_y; }
      if ((sidx8 & 7) == 7 || (l == levels - 1 && p == points - 1)) {
        const int first = sidx8 & ~7;
        if (live && first + d4 <= sidx8) {
          grad_attn[lw_base + first + d4] = keep_w;
          *reinterpret_cast<float2*>(grad_loc + (lw_base + first + d4) * 2) = make_float2(keep_x, keep_y);
        }
      }
    }
  }
}

struct MsdaLevels {
  int levels;
  int h[8], w[8], lstart[8];
};

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
  if ((head_dim & 3) == 0 && ((reinterpret_cast<size_t>(value) | reinterpret_cast<size_t>(out)) & 15) == 0) {
    const int64_t total4 = total / 4;
    if (num_points == 4 && ((reinterpret_cast<size_t>(sampling_loc) | reinterpret_cast<size_t>(attn_weight)) & 15) == 0) {
      hipLaunchKernelGGL(k_msda_fwd_p4<MBV_DT_F32>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream,
                         (const void*)value, spatial_shapes, level_start, sampling_loc, attn_weight, total4, num_value,
                         num_heads, head_dim, num_levels, num_query, out);
      MBV_CHECK_LAUNCH();
      return MBV_OK;
    }
    hipLaunchKernelGGL(k_msda_fwd_v4<MBV_DT_F32>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream,
                       (const void*)value, spatial_shapes, level_start, sampling_loc, attn_weight, total4, num_value,
                       num_heads, head_dim, num_levels, num_query, num_points, out);
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
  hipLaunchKernelGGL(k_msda_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, value, spatial_shapes,
                     level_start, sampling_loc, attn_weight, total, num_value, num_heads, head_dim, num_levels,
                     num_query, num_points, out);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// Forward with the value map in 16 bits (value_dtype MBV_DT_BF16 / _F16; MBV_DT_F32 = the entry above): head_dim % 4 == 0.
extern "C" int mbv_ms_deform_attn_fwd_v(const void* value, int32_t value_dtype, const int64_t* spatial_shapes,
                                        const int64_t* level_start, const float* sampling_loc, const float* attn_weight,
                                        int32_t batch, int32_t num_value, int32_t num_heads, int32_t head_dim,
                                        int32_t num_levels, int32_t num_query, int32_t num_points, float* out, void* stream_) {
  if (value_dtype == MBV_DT_F32)
    return mbv_ms_deform_attn_fwd(reinterpret_cast<const float*>(value), spatial_shapes, level_start, sampling_loc,
                                  attn_weight, batch, num_value, num_heads, head_dim, num_levels, num_query, num_points, out,
                                  stream_);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_value <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0 || num_points <= 0)
    return MBV_ERR_BAD_ARG;
  if (value_dtype != MBV_DT_BF16 && value_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  if (!pow2_le64(head_dim) || (head_dim & 3)) return MBV_ERR_UNSUPPORTED;
  if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !out) return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(value) & 7) || (reinterpret_cast<size_t>(out) & 15)) return MBV_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)batch * num_query * num_heads * head_dim / 4;
  const dim3 grid((unsigned)((total4 + 255) / 256)), block(256);
  if (num_points == 4 && ((reinterpret_cast<size_t>(sampling_loc) | reinterpret_cast<size_t>(attn_weight)) & 15) == 0) {
    if (value_dtype == MBV_DT_BF16)
      hipLaunchKernelGGL(k_msda_fwd_p4<MBV_DT_BF16>, grid, block, 0, stream, value, spatial_shapes, level_start, sampling_loc,
                         attn_weight, total4, num_value, num_heads, head_dim, num_levels, num_query, out);
    else
      hipLaunchKernelGGL(k_msda_fwd_p4<MBV_DT_F16>, grid, block, 0, stream, value, spatial_shapes, level_start, sampling_loc,
                         attn_weight, total4, num_value, num_heads, head_dim, num_levels, num_query, out);
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
  if (value_dtype == MBV_DT_BF16)
    hipLaunchKernelGGL(k_msda_fwd_v4<MBV_DT_BF16>, grid, block, 0, stream, value, spatial_shapes, level_start, sampling_loc,
                       attn_weight, total4, num_value, num_heads, head_dim, num_levels, num_query, num_points, out);
  else
    hipLaunchKernelGGL(k_msda_fwd_v4<MBV_DT_F16>, grid, block, 0, stream, value, spatial_shapes, level_start, sampling_loc,
                       attn_weight, total4, num_value, num_heads, head_dim, num_levels, num_query, num_points, out);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// d(location), d(weight) alone (the part-2 kernel of the split backward) with the value map in f32 / bf16 / fp16: head_dim 32.
extern "C" int mbv_ms_deform_attn_bwd_locattn(const float* grad_out, const void* value, int32_t value_dtype,
                                              const int64_t* spatial_shapes, const int64_t* level_start,
                                              const float* sampling_loc, const float* attn_weight, int32_t batch,
                                              int32_t num_value, int32_t num_heads, int32_t head_dim, int32_t num_levels,
                                              int32_t num_query, int32_t num_points, float* grad_loc, float* grad_attn,
                                              void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_value <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0 || num_points <= 0)
    return MBV_ERR_BAD_ARG;
  if (head_dim != 32) return MBV_ERR_UNSUPPORTED;
  if (!grad_out || !value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !grad_loc || !grad_attn)
    return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(grad_out) & 15) || (reinterpret_cast<size_t>(value) & (value_dtype == MBV_DT_F32 ? 15 : 7)))
    return MBV_ERR_UNSUPPORTED;
  const int64_t total8 = (int64_t)batch * num_query * num_heads * 8;
  const dim3 grid((unsigned)((total8 + 255) / 256)), block(256);
  const bool p4 = num_points == 4 && ((reinterpret_cast<size_t>(sampling_loc) | reinterpret_cast<size_t>(attn_weight)) & 15) == 0;
#define MBV_LOCATTN(VK)                                                                                                  \
  if (p4)                                                                                                                \
    hipLaunchKernelGGL(k_msda_bwd_locattn_p4<VK>, grid, block, 0, stream, grad_out, value, spatial_shapes, level_start,  \
                       sampling_loc, attn_weight, total8, num_value, num_heads, num_levels, num_query, grad_loc,         \
                       grad_attn);                                                                                       \
  else                                                                                                                   \
    hipLaunchKernelGGL(k_msda_bwd_locattn<VK>, grid, block, 0, stream, grad_out, value, spatial_shapes, level_start,     \
                       sampling_loc, attn_weight, total8, num_value, num_heads, num_levels, num_query, num_points,       \
                       grad_loc, grad_attn)
  if (value_dtype == MBV_DT_F32) { MBV_LOCATTN(MBV_DT_F32); }
  else if (value_dtype == MBV_DT_BF16) { MBV_LOCATTN(MBV_DT_BF16); }
  else if (value_dtype == MBV_DT_F16) { MBV_LOCATTN(MBV_DT_F16); }
  else return MBV_ERR_BAD_ARG;
#undef MBV_LOCATTN
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_ms_deform_attn_bwd_split(int32_t head_dim, int32_t num_levels, const int64_t* spatial_shapes_host) {
  if (!spatial_shapes_host || head_dim != 32 || num_levels <= 0 || num_levels > 8) return 0;
  for (int l = 0; l < num_levels; ++l) {
    const int64_t h = spatial_shapes_host[2 * l], w = spatial_shapes_host[2 * l + 1];
    if (h <= 0 || w <= 0 || h * w > 4096) return 0;
  }
  return 1;
}

extern "C" int mbv_ms_deform_attn_bwd(const float* grad_out, const float* value, const int64_t* spatial_shapes,
                                      const int64_t* level_start, const float* sampling_loc,
                                      const float* attn_weight, int32_t batch, int32_t num_value, int32_t num_heads,
                                      int32_t head_dim, int32_t num_levels, int32_t num_query, int32_t num_points,
                                      const int64_t* spatial_shapes_host, float* grad_value, float* grad_loc,
                                      float* grad_attn, int32_t part, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_value <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0 || num_points <= 0)
    return MBV_ERR_BAD_ARG;
  if (!pow2_le64(head_dim)) return MBV_ERR_UNSUPPORTED;
  if (!grad_out || !value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight) return MBV_ERR_BAD_ARG;
  // an output may be null when `part` leaves it out (1 = value only, 2 = location / weight only)
  if ((!grad_value && ((part & 3) != 2)) || ((!grad_loc || !grad_attn) && ((part & 3) != 1))) return MBV_ERR_BAD_ARG;
  // part: bit 0 = d(value), bit 1 = d(location, weight); bits 2.. = optional mask of the levels whose d(value) this
  // call produces (0 = all) — a caller may put the levels' launches on different streams
  const int level_mask = part >> 2;
  part &= 3;
  if (part < 1 || level_mask < 0 || level_mask > 255 || (level_mask && part != 1)) return MBV_ERR_BAD_ARG;
  {
    // no global atomics (see k_msda_bwd_value); maps of more than 4096 pixels take the banded form below
    bool fits = mbv_ms_deform_attn_bwd_split(head_dim, num_levels, spatial_shapes_host) != 0 &&
                ((reinterpret_cast<size_t>(grad_out) | reinterpret_cast<size_t>(value) |
                  reinterpret_cast<size_t>(grad_value) | reinterpret_cast<size_t>(sampling_loc) |
                  reinterpret_cast<size_t>(attn_weight)) & 15) == 0;
    MsdaLevels lv;
    lv.levels = num_levels;
    int lstart = 0, max_pix = 0;
    for (int l = 0; fits && l < num_levels; ++l) {
      const int h = (int)spatial_shapes_host[2 * l], w = (int)spatial_shapes_host[2 * l + 1];
      if (h <= 0 || w <= 0 || h * w > 4096) { fits = false; break; }
      lv.h[l] = h; lv.w[l] = w; lv.lstart[l] = lstart;
      lstart += h * w;
      if (h * w > max_pix) max_pix = h * w;
    }
    if (fits && lstart == num_value) {
      static bool attr_done = false;      // idempotent attribute of the code objects, not library state
      if (!attr_done && max_pix * 32 > 65536) {
        MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msda_bwd_value<4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msda_bwd_value<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr_done = true;
      }
      const int which = part;
      for (int l = num_levels - 1; l >= 0 && (which & 1); --l) {      // finest (longest) level first
        if (level_mask && !((level_mask >> l) & 1)) continue;
        const dim3 grid((unsigned)(batch * num_heads * 8)), block(1024);
        const size_t lds = (size_t)lv.h[l] * lv.w[l] * 4 * sizeof(double);
        if (num_points == 4)
          hipLaunchKernelGGL(k_msda_bwd_value<4>, grid, block, lds, stream, grad_out, sampling_loc, attn_weight, l,
                             num_levels, lv.h[l], lv.w[l], lv.lstart[l], num_value, num_query, num_heads, num_points,
                             grad_value);
        else
          hipLaunchKernelGGL(k_msda_bwd_value<0>, grid, block, lds, stream, grad_out, sampling_loc, attn_weight, l,
                             num_levels, lv.h[l], lv.w[l], lv.lstart[l], num_value, num_query, num_heads, num_points,
                             grad_value);
        MBV_CHECK_LAUNCH();
      }
      if (which & 2) {
        const int64_t total8 = (int64_t)batch * num_query * num_heads * 8;
        if (num_points == 4 && ((reinterpret_cast<size_t>(sampling_loc) | reinterpret_cast<size_t>(attn_weight)) & 15) == 0)
          hipLaunchKernelGGL(k_msda_bwd_locattn_p4<MBV_DT_F32>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, stream,
                             grad_out, (const void*)value, spatial_shapes, level_start, sampling_loc, attn_weight, total8,
                             num_value, num_heads, num_levels, num_query, grad_loc, grad_attn);
        else
          hipLaunchKernelGGL(k_msda_bwd_locattn<MBV_DT_F32>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, stream,
                             grad_out, (const void*)value, spatial_shapes, level_start, sampling_loc, attn_weight, total8,
                             num_value, num_heads, num_levels, num_query, num_points, grad_loc, grad_attn);
        MBV_CHECK_LAUNCH();
      }
      return MBV_OK;
    }
  }
  if (part != 3) return MBV_ERR_UNSUPPORTED;
  MBV_CHECK_HIP(mbv_fill_async(grad_value, 0, sizeof(float) * (size_t)batch * num_value * num_heads * head_dim, stream));
  if (spatial_shapes_host && num_query == num_value && num_levels <= 8) {
    // banded form: needs the level shapes on the host to size the bands, and the self-attention query order
    MsdaBands cfg;
    cfg.levels = num_levels;
    const int bh = batch * num_heads;
    int lstart = 0, blocks = 0;
    bool ok = true;
    for (int l = 0; l < num_levels; ++l) {
      const int h = (int)spatial_shapes_host[2 * l], w = (int)spatial_shapes_host[2 * l + 1];
      const int rows = 16384 / (w * head_dim);
      if (h <= 0 || w <= 0 || rows < 1) { ok = false; break; }
      cfg.h[l] = h; cfg.w[l] = w; cfg.lstart[l] = lstart;
      cfg.band_rows[l] = rows < h ? rows : h;
      cfg.bands[l] = (h + cfg.band_rows[l] - 1) / cfg.band_rows[l];
      int chunks = 256 / (bh * cfg.bands[l]);
      cfg.chunks[l] = chunks < 1 ? 1 : chunks;
      cfg.block_begin[l] = blocks;
      blocks += bh * cfg.bands[l] * cfg.chunks[l];
      lstart += h * w;
    }
    if (ok && lstart == num_value) {
      cfg.block_begin[num_levels] = blocks;
      hipLaunchKernelGGL(k_msda_bwd_banded, dim3((unsigned)blocks), dim3(1024), 0, stream, grad_out, value, sampling_loc,
                         attn_weight, cfg, num_value, num_heads, head_dim, num_points, grad_value, grad_loc, grad_attn);
      MBV_CHECK_LAUNCH();
      return MBV_OK;
    }
  }
  const int64_t total = (int64_t)batch * num_query * num_heads * head_dim;
  hipLaunchKernelGGL(k_msda_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, grad_out, value,
                     spatial_shapes, level_start, sampling_loc, attn_weight, total, num_value, num_heads, head_dim,
                     num_levels, num_query, num_points, grad_value, grad_loc, grad_attn);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_ms_deform_attn_bwd_value_packed_supported(int32_t head_dim, int32_t num_levels, int32_t num_points,
                                                             int32_t num_query, const int64_t* spatial_shapes_host) {
  if (!spatial_shapes_host || head_dim != 32 || num_points != 4 || num_levels <= 0 || num_levels > 8) return 0;
  if (num_query <= 0 || num_query > (1 << 18)) return 0;            // >= 12 fractional bits
  // One u64 plane (two channels) of a level's map per block in dynamic LDS: <= 16 384 pixels (128 KB; round 6 — until then
  // 4 096, the f64 split form's limit, which sent the 128 x 128 level of the 1024 x 1024 BEV configuration to the banded f64
  // kernel).  The location / weight part that follows is mbv_ms_deform_attn_bwd_locattn (gathers: no map-size limit).
  for (int l = 0; l < num_levels; ++l) {
    const int64_t h = spatial_shapes_host[2 * l], w = spatial_shapes_host[2 * l + 1];
    if (h <= 0 || w <= 0 || h * w > 16384) return 0;
  }
  return 1;
}

extern "C" size_t mbv_ms_deform_attn_bwd_value_packed_workspace_bytes(int32_t batch, int32_t num_heads, int32_t num_levels,
                                                                      int32_t num_query) {
  if (batch <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0) return 0;
  const size_t bhq = (size_t)batch * num_heads * num_query, q_tiles = ((size_t)num_query + 63) / 64;
  return mbv_align_up(bhq * 32 * 4, 256) + mbv_align_up(bhq * num_levels * 8 * 4, 256) +
         mbv_align_up(bhq * num_levels * 4 * 4, 256) + mbv_align_up((size_t)batch * num_heads * q_tiles * num_levels * 8 * 4, 256);
}

extern "C" int mbv_ms_deform_attn_bwd_value_packed(const float* grad_out, const float* sampling_loc,
                                                   const float* attn_weight, int32_t batch, int32_t num_value,
                                                   int32_t num_heads, int32_t head_dim, int32_t num_levels,
                                                   int32_t num_query, int32_t num_points,
                                                   const int64_t* spatial_shapes_host, void* grad_value,
                                                   int32_t out_dtype, int64_t out_ld, void* workspace,
                                                   size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_value <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0) return MBV_ERR_BAD_ARG;
  if (!grad_out || !sampling_loc || !attn_weight || !grad_value || !spatial_shapes_host) return MBV_ERR_BAD_ARG;
  if (!mbv_ms_deform_attn_bwd_value_packed_supported(head_dim, num_levels, num_points, num_query, spatial_shapes_host))
    return MBV_ERR_UNSUPPORTED;
  if (out_dtype != MBV_DT_F32 && out_dtype != MBV_DT_BF16 && out_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  const int esz = out_dtype == MBV_DT_F32 ? 4 : 2;
  if (out_ld < (int64_t)num_heads * head_dim || (out_ld * esz) % 4 != 0 || (reinterpret_cast<size_t>(grad_value) & 7) != 0 ||
      ((reinterpret_cast<size_t>(grad_out) | reinterpret_cast<size_t>(sampling_loc) |
        reinterpret_cast<size_t>(attn_weight)) & 15) != 0)
    return MBV_ERR_BAD_ARG;
  if (out_dtype == MBV_DT_F32 && (out_ld % 2) != 0) return MBV_ERR_BAD_ARG;
  if (!workspace || (reinterpret_cast<size_t>(workspace) & 15) ||
      workspace_bytes < mbv_ms_deform_attn_bwd_value_packed_workspace_bytes(batch, num_heads, num_levels, num_query))
    return MBV_ERR_BAD_ARG;
  MsdaFxArgs A;
  const int q_tiles = (num_query + 63) / 64;
  {
    MbvCarver cw(workspace);
    const size_t bhq = (size_t)batch * num_heads * num_query;
    float* go_t = cw.take<float>(bhq * 32);
    float* loc_t = cw.take<float>(bhq * num_levels * 8);
    float* attn_t = cw.take<float>(bhq * num_levels * 4);
    float* l1p = cw.take<float>((size_t)batch * num_heads * q_tiles * num_levels * 8);
    A.go_t = go_t; A.loc_t = loc_t; A.attn_t = attn_t; A.l1_part = l1p; A.q_tiles = q_tiles;
    hipLaunchKernelGGL(k_msda_bwd_relayout, dim3((unsigned)(batch * num_heads * q_tiles)), dim3(256), 0, stream, grad_out,
                       sampling_loc, attn_weight, num_query, num_heads, num_levels, q_tiles, go_t, loc_t, attn_t, l1p);
    MBV_CHECK_LAUNCH();
  }
  A.grad_out = grad_out; A.loc = sampling_loc; A.attn = attn_weight; A.out = grad_value; A.out_ld = out_ld;
  A.out_dt = out_dtype; A.levels = num_levels; A.num_value = num_value; A.num_query = num_query; A.heads = num_heads;
  A.batch = batch;
  int bits = 0;
  while ((1 << bits) < num_query) ++bits;
  A.frac_bits = 30 - bits;
  int lstart = 0, max_pix_bytes = 0;
  const int nbh = batch * num_heads;
  for (int l = 0; l < num_levels; ++l) {
    A.h[l] = (int)spatial_shapes_host[2 * l]; A.w[l] = (int)spatial_shapes_host[2 * l + 1]; A.lstart[l] = lstart;
    lstart += A.h[l] * A.w[l];
  }
  if (lstart != num_value) return MBV_ERR_BAD_ARG;
  // a level with a large map gets one plane (2 channels) per block, so that every block fits 32 KB of LDS and all
  // blocks of the launch (4 per CU) are resident at once
  int blocks = 0;
  for (int l = 0; l < num_levels; ++l) {
    const int npix = A.h[l] * A.w[l];
    A.planes[l] = npix * 16 <= 32768 ? 2 : 1;
    const int bytes = A.planes[l] * npix * 8;
    if (bytes > max_pix_bytes) max_pix_bytes = bytes;
    A.block_begin[l] = blocks;
    blocks += nbh * (32 / (2 * A.planes[l]));
  }
  A.block_begin[num_levels] = blocks;
  const dim3 grid((unsigned)blocks), block(512);
  if (max_pix_bytes > 65536) {           // a large level's plane: more than the default dynamic-LDS limit
    static bool attr_done = false;       // idempotent attribute of the code objects, not library state
    if (!attr_done) {
      MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msda_bwd_value_fx<MBV_DT_F32>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msda_bwd_value_fx<MBV_DT_BF16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msda_bwd_value_fx<MBV_DT_F16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      attr_done = true;
    }
  }
  if (out_dtype == MBV_DT_F32)
    hipLaunchKernelGGL(k_msda_bwd_value_fx<MBV_DT_F32>, grid, block, (size_t)max_pix_bytes, stream, A);
  else if (out_dtype == MBV_DT_BF16)
    hipLaunchKernelGGL(k_msda_bwd_value_fx<MBV_DT_BF16>, grid, block, (size_t)max_pix_bytes, stream, A);
  else
    hipLaunchKernelGGL(k_msda_bwd_value_fx<MBV_DT_F16>, grid, block, (size_t)max_pix_bytes, stream, A);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
