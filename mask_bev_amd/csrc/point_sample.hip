// K8 — indexed bilinear point sampling of mask maps (the loss / matcher path), forward and backward.
//
// Replaces mmcv `point_sample` (= F.grid_sample, align_corners=False, zero padding) as used by the reference's
// loss and Hungarian targets (mask_bev/models/networks/mask2former_head/mask2former_head.py:194-200,402-410),
// together with the tensor gathers that feed it (`mask_preds[mask_weights > 0]`, `gt_masks[pos_assigned_gt_inds]`,
// :227,:393 — a 420 MB copy of GT masks per decoder output at 512x512 / 100 queries / batch 4).
// out[g][p] = bilinear(src[src_index[g]], coords[coord_index[g]][p]); nothing is gathered or copied.
//
// Backward: every sampled map receives 4 * P scattered adds.  Random global float atomics run at ~0.08 TB/s
// on gfx950 (one lane per row, MI355X_MICROARCH.md "Global float atomics"); instead ONE workgroup owns one
// map, accumulates its whole (H, W) gradient tile in LDS (64 KB at 128 x 128) with LDS atomics and writes it
// out once with coalesced stores.  Maps larger than the LDS tile fall back to global atomics.
#include "common.hpp"
#include "bilinear.hpp"

namespace {

constexpr int kTileFloats = 16384;   // 64 KB LDS tile: 128 x 128 mask logits
constexpr int kBandTileFloats = 35840;   // 140 KB of dynamic LDS for the row bands of a larger map (forward)

__global__ void __launch_bounds__(256) k_point_sample_fwd(const float* __restrict__ src,
                                                          const int32_t* __restrict__ src_index,
                                                          const float* __restrict__ coords,
                                                          const int32_t* __restrict__ coord_index, int P, int H, int W,
                                                          float* __restrict__ out) {
  const int g = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* s = src + (int64_t)src_index[g] * H * W;
  const float* c = coords + ((int64_t)coord_index[g] * P + p) * 2;
  Bil b;
  bil_setup(c[0], c[1], H, W, b);
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (b.o[j] >= 0) v += b.w[j] * s[b.o[j]];
  out[(int64_t)g * P + p] = v;
}

// A map of hw floats → LDS.  Eight 16-byte loads per thread are issued before the first LDS store: written as a plain
// load / store loop the compiler waits for every load in turn (the trip count is a run-time value), i.e. eight
// dependent memory round trips per 64 KB map instead of one.
__device__ __forceinline__ void stage_map(float* tile, const float* __restrict__ s, int hw) {
  const int nt = blockDim.x;
  if ((hw & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* t4 = reinterpret_cast<float4*>(tile);
    const int n4 = hw >> 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * nt) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * nt;
        v[u] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * nt;
        if (i < n4) t4[i] = v[u];
      }
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += nt) tile[i] = s[i];
  }
}

// forward, one workgroup per row: the whole (H, W) map is staged in LDS with coalesced loads, then sampled from
// LDS — instead of 4 random 4-byte HBM/L2 gathers per point
__global__ void __launch_bounds__(512) k_point_sample_fwd_lds(const float* __restrict__ src,
                                                              const int32_t* __restrict__ src_index,
                                                              const float* __restrict__ coords,
                                                              const int32_t* __restrict__ coord_index, int P, int H,
                                                              int W, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[kTileFloats];
  const int g = blockIdx.x;
  const int hw = H * W;
  const float* s = src + (int64_t)src_index[g] * hw;
  stage_map(tile, s, hw);
  __syncthreads();
  const float2* c = reinterpret_cast<const float2*>(coords + (int64_t)coord_index[g] * P * 2);
  float* o = out + (int64_t)g * P;
  // batches of 4 points per thread: their coordinate loads are in flight together (one memory round trip per batch)
  for (int p0 = threadIdx.x; p0 < P; p0 += 4 * (int)blockDim.x) {
    float2 xy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      xy[u] = p < P ? c[p] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      if (p < P) {
        Bil b;
        bil_setup(xy[u].x, xy[u].y, H, W, b);
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (b.o[j] >= 0) v += b.w[j] * tile[b.o[j]];
        o[p] = v;
      }
    }
  }
}

// The same for a map larger than one tile (round 6: 256 x 256 logits): workgroup (g, band) stages image rows
// [y0, y0 + band_rows] of the map — band_rows + 1 rows: the lower taps of its last row — and samples the points whose
// UPPER tap row, clamped to [0, H - 1], lies in [y0, y0 + band_rows): every point belongs to exactly one band.  The
// bilinear arithmetic is bil_setup's on the whole map (same weights, same sum order: bit-identical to the gather form).
__global__ void __launch_bounds__(1024) k_point_sample_fwd_bands(const float* __restrict__ src,
                                                                const int32_t* __restrict__ src_index,
                                                                const float* __restrict__ coords,
                                                                const int32_t* __restrict__ coord_index, int P, int H,
                                                                int W, int bands, int band_rows, float* __restrict__ out) {
  // dynamic LDS, up to kBandTileFloats (140 KB): every band walks ALL the map's coordinates (8 bytes a point, 300 KB per
  // over-sampled row), so FEW bands matter more than two workgroups per CU — 256 x 256 logits are 2 bands, not 5
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int g = (int)blockIdx.x / bands, band = (int)blockIdx.x - g * bands;
  const int y0 = band * band_rows;
  const int y_end = (y0 + band_rows + 1) < H ? (y0 + band_rows + 1) : H;       // rows staged: [y0, y_end)
  const int lo = y0 * W, hw = (y_end - y0) * W;
  const float* s = src + (int64_t)src_index[g] * H * W + lo;
  if ((lo & 3) == 0) {
    stage_map(tile, s, hw);
  } else {
    for (int i = threadIdx.x; i < hw; i += blockDim.x) tile[i] = s[i];
  }
  __syncthreads();
  const float2* c = reinterpret_cast<const float2*>(coords + (int64_t)coord_index[g] * P * 2);
  float* o = out + (int64_t)g * P;
  for (int p0 = threadIdx.x; p0 < P; p0 += 4 * (int)blockDim.x) {
    float2 xy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      xy[u] = p < P ? c[p] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      if (p < P) {
        int yt = (int)floorf(xy[u].y * (float)H - 0.5f);
        yt = yt < 0 ? 0 : (yt > H - 1 ? H - 1 : yt);
        if (yt >= y0 && yt < y0 + band_rows) {
          Bil b;
          bil_setup(xy[u].x, xy[u].y, H, W, b);
          float v = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (b.o[j] >= 0) v += b.w[j] * tile[b.o[j] - lo];
          o[p] = v;
        }
      }
    }
  }
}

// binary maps packed 32 pixels / word (bit i of word k = pixel 32 k + i != 0)
__global__ void __launch_bounds__(256) k_pack_binary(const float* __restrict__ src, int64_t hw, int64_t words_per_map,
                                                     uint32_t* __restrict__ packed) {
  const int64_t map = blockIdx.y;
  const int64_t pix = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
  const bool bit = pix < hw && src[map * hw + pix] != 0.f;
  const unsigned long long m = __ballot(bit);
  const int lane = threadIdx.x & 63;
  const int64_t w0 = pix / 32;
  if (lane == 0 && w0 < words_per_map) packed[map * words_per_map + w0] = (uint32_t)(m & 0xffffffffull);
  if (lane == 32 && w0 < words_per_map) packed[map * words_per_map + w0] = (uint32_t)(m >> 32);
}

// Same packing, 16 pixels per thread: a lane reads four float4 (4 x 16 B, each wave instruction one contiguous KiB),
// turns each into a nibble, and the 8 lanes that share a word OR their nibbles together with three xor-shuffles.
// (One pixel per thread ran at 1.9 TB/s: 105 M threads for 420 MB.)  hw % 1024 == 0 (whole wave tiles per map).
__global__ void __launch_bounds__(256) k_pack_binary_v4(const float* __restrict__ src, int64_t hw, int64_t words_per_map,
                                                        uint32_t* __restrict__ packed) {
  const int64_t map = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // 1024 pixels per wave
  const int64_t p0 = wave * 1024;
  if (p0 >= hw) return;
  const float* s = src + map * hw + p0;
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(s + u * 256 + lane * 4);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    uint32_t w = ((v[u].x != 0.f) ? 1u : 0u) | ((v[u].y != 0.f) ? 2u : 0u) | ((v[u].z != 0.f) ? 4u : 0u) |
                 ((v[u].w != 0.f) ? 8u : 0u);
    w <<= 4 * (lane & 7);
    w |= __shfl_xor(w, 1, 64);
    w |= __shfl_xor(w, 2, 64);
    w |= __shfl_xor(w, 4, 64);
    if ((lane & 7) == 0) packed[map * words_per_map + (p0 + u * 256) / 32 + (lane >> 3)] = w;
  }
}

constexpr int kPackedWords = 32768;   // 128 KB LDS: up to 1024 x 1024 binary pixels

// WORDS = LDS words reserved for the packed map: 8192 (32 KB, a 512 x 512 mask: four workgroups per CU) or 32768
// (128 KB, up to 1024 x 1024: one per CU) — the occupancy, not the arithmetic, sets this kernel's time.
template <int WORDS>
__global__ void __launch_bounds__(512) k_point_sample_packed(const uint32_t* __restrict__ packed,
                                                             int64_t words_per_map,
                                                             const int32_t* __restrict__ src_index,
                                                             const float* __restrict__ coords,
                                                             const int32_t* __restrict__ coord_index, int P, int H,
                                                             int W, float* __restrict__ out) {
  __shared__ uint32_t bits[WORDS];
  const int g = blockIdx.x;
  const uint32_t* s = packed + (int64_t)src_index[g] * words_per_map;
  for (int i = threadIdx.x; i < (int)words_per_map; i += blockDim.x) bits[i] = s[i];
  __syncthreads();
  const float2* c = reinterpret_cast<const float2*>(coords + (int64_t)coord_index[g] * P * 2);
  float* o = out + (int64_t)g * P;
  for (int p0 = threadIdx.x; p0 < P; p0 += 4 * (int)blockDim.x) {       // 4 coordinate loads in flight per thread
    float2 xy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      xy[u] = p < P ? c[p] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      if (p < P) {
        Bil b;
        bil_setup(xy[u].x, xy[u].y, H, W, b);
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (b.o[j] >= 0 && ((bits[b.o[j] >> 5] >> (b.o[j] & 31)) & 1u)) v += b.w[j];
        o[p] = v;
      }
    }
  }
}

// `bands` > 1 (round 6: maps larger than one LDS tile — the 256 x 256 mask logits of the 1024 x 1024 BEV configuration):
// workgroup (map g, band) owns `band_rows` image rows of the map, walks ALL the map's points and keeps the taps that fall
// on its rows.  The walk is repeated per band (coordinates are 8 bytes a point), the adds are not: 12 000 maps x 12 544
// points were 600 M global f32 atomics (28 ms per step); this form accumulates in LDS like the one-tile case.
__global__ void __launch_bounds__(1024) k_point_sample_bwd_lds(const float* __restrict__ grad_out,
                                                               const int32_t* __restrict__ src_index,
                                                               const float* __restrict__ coords,
                                                               const int32_t* __restrict__ coord_index, int P, int H,
                                                               int W, void* __restrict__ grad_src, int out_kind,
                                                               int perm_outer, int perm_inner, int perm_rows, int bands,
                                                               int band_rows) {
  // f64 accumulators: on gfx950 an LDS ds_add_f32 wave instruction takes ≈ 192 cycles (the lanes are serialised),
  // ds_add_f64 / ds_add_u64 ≈ 9-16 (scratch/ubench/lds_atomic.hip) — the f32 form of this kernel ran 1.0 ms, bound
  // by exactly that.  128 KB for a 128 x 128 map; the sum is also order-independent to f32 precision.
  __shared__ double tile[kTileFloats];
  const int g = (int)blockIdx.x / bands, band = (int)blockIdx.x - g * bands;
  const int y0 = band * band_rows;
  const int rows_here = (H - y0) < band_rows ? (H - y0) : band_rows;
  const int lo = y0 * W, hw = rows_here * W;            // this workgroup's pixels: [lo, lo + hw) of the map
  for (int i = threadIdx.x; i < hw; i += blockDim.x) tile[i] = 0.0;
  __syncthreads();
  const float2* c = reinterpret_cast<const float2*>(coords + (int64_t)coord_index[g] * P * 2);
  const float* go = grad_out + (int64_t)g * P;
  for (int p0 = threadIdx.x; p0 < P; p0 += 4 * (int)blockDim.x) {       // 4 + 4 loads in flight per thread
    float2 xy[4];
    float gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      xy[u] = p < P ? c[p] : make_float2(0.f, 0.f);
      gv[u] = p < P ? go[p] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p0 + u * (int)blockDim.x < P) {
        Bil b;
        bil_setup(xy[u].x, xy[u].y, H, W, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned o = (unsigned)(b.o[j] - lo);      // (a missing tap, -1, and every pixel outside the band: >= hw)
          if (b.o[j] >= 0 && o < (unsigned)hw) atomicAdd(&tile[o], (double)(b.w[j] * gv[u]));
        }
      }
    }
  }
  __syncthreads();
  // destination row: the source map's own, or — perm_outer > 0 — with its two leading axes exchanged: map
  // (o, n, r) of an (outer, inner, rows) stack goes to row (n, o, r) (the stacked decoder outputs (D, B, Q) stored sample-major)
  int64_t row = src_index[g];
  if (perm_outer > 0) {
    const int64_t r = row % perm_rows, on = row / perm_rows;
    const int64_t o = on / perm_inner, n = on - o * perm_inner;
    row = (n * perm_outer + o) * perm_rows + r;
  }
  const int64_t map_hw = (int64_t)H * W;
  if (out_kind == MBV_DT_F32) {
    float* dst = reinterpret_cast<float*>(grad_src) + row * map_hw + lo;
    if ((hw & 3) == 0 && (lo & 3) == 0 && (map_hw & 3) == 0) {
      for (int i = threadIdx.x * 4; i < hw; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(dst + i) = make_float4((float)tile[i], (float)tile[i + 1], (float)tile[i + 2], (float)tile[i + 3]);
    } else {
      for (int i = threadIdx.x; i < hw; i += blockDim.x) dst[i] = (float)tile[i];
    }
    return;
  }
  unsigned short* dst = reinterpret_cast<unsigned short*>(grad_src) + row * map_hw + lo;
  auto lo2 = [&](double a, double b2) -> unsigned {
    const float x = (float)a, y = (float)b2;
    if (out_kind == MBV_DT_F16)
      return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)y) << 16);
    return pack_bf16x2(x, y);
  };
  if ((hw & 3) == 0 && (lo & 3) == 0 && (map_hw & 3) == 0) {
    for (int i = threadIdx.x * 4; i < hw; i += blockDim.x * 4)
      *reinterpret_cast<uint2*>(dst + i) = make_uint2(lo2(tile[i], tile[i + 1]), lo2(tile[i + 2], tile[i + 3]));
  } else {
    for (int i = threadIdx.x; i < hw; i += blockDim.x) dst[i] = (unsigned short)(lo2(tile[i], 0.0) & 0xffffu);
  }
}

// band split of an (H, W) map for the LDS form: rows per workgroup (<= kTileFloats pixels) and the number of bands; 0 when
// one image row alone exceeds the tile
static inline int lds_band_rows(int H, int W, int* bands) {
  if (W > kTileFloats) { *bands = 0; return 0; }
  int br = kTileFloats / W;
  br = br > H ? H : br;
  *bands = (H + br - 1) / br;
  return br;
}

__global__ void __launch_bounds__(256) k_point_sample_bwd_atomic(const float* __restrict__ grad_out,
                                                                 const int32_t* __restrict__ src_index,
                                                                 const float* __restrict__ coords,
                                                                 const int32_t* __restrict__ coord_index, int P, int H,
                                                                 int W, float* __restrict__ grad_src) {
  const int g = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* c = coords + ((int64_t)coord_index[g] * P + p) * 2;
  Bil b;
  bil_setup(c[0], c[1], H, W, b);
  const float gv = grad_out[(int64_t)g * P + p];
  float* dst = grad_src + (int64_t)src_index[g] * H * W;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (b.o[j] >= 0) atomicAdd(dst + b.o[j], b.w[j] * gv);
}

}  // namespace

extern "C" int mbv_point_sample_fwd(const float* src, const int32_t* src_index, const float* coords,
                                    const int32_t* coord_index, int32_t num_rows, int32_t num_points, int32_t H,
                                    int32_t W, float* out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_rows < 0 || num_points <= 0 || H <= 0 || W <= 0) return MBV_ERR_BAD_ARG;
  if (num_rows == 0) return MBV_OK;
  if (!src || !src_index || !coords || !coord_index || !out) return MBV_ERR_BAD_ARG;
  if (num_rows > 65535) return MBV_ERR_UNSUPPORTED;
  if ((int64_t)H * W <= kTileFloats && (int64_t)num_points * 8 >= (int64_t)H * W) {
    hipLaunchKernelGGL(k_point_sample_fwd_lds, dim3(num_rows), dim3(512), 0, stream, src, src_index, coords,
                       coord_index, num_points, H, W, out);
  } else if (W + W <= kBandTileFloats && (int64_t)num_points * 8 >= (int64_t)H * W &&
             (H + kBandTileFloats / W - 2) / (kBandTileFloats / W - 1) <= 64) {
    // a map of several tiles sampled densely (the importance sampling of 256 x 256 logits): row bands through LDS
    int band_rows = kBandTileFloats / W - 1;                   // + 1 staged row for the lower taps
    const int bands = (H + band_rows - 1) / band_rows;
    band_rows = (H + bands - 1) / bands;                       // equal bands (the last one is not a sliver)
    const size_t lds = (size_t)(band_rows + 1) * W * sizeof(float);
    static bool attr_done = false;       // idempotent attribute of the code object, not library state
    if (!attr_done) {
      MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_point_sample_fwd_bands),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kBandTileFloats * (int)sizeof(float)));
      attr_done = true;
    }
    hipLaunchKernelGGL(k_point_sample_fwd_bands, dim3((unsigned)num_rows * (unsigned)bands), dim3(1024), lds, stream, src,
                       src_index, coords, coord_index, num_points, H, W, bands, band_rows, out);
  } else {
    hipLaunchKernelGGL(k_point_sample_fwd, dim3((num_points + 255) / 256, num_rows), dim3(256), 0, stream, src,
                       src_index, coords, coord_index, num_points, H, W, out);
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int64_t mbv_packed_mask_words(int32_t H, int32_t W) {
  if (H <= 0 || W <= 0) return 0;
  return ((int64_t)H * W + 63) / 64 * 2;   // whole 64-pixel groups (one wavefront ballot each)
}

extern "C" int mbv_pack_binary_masks(const float* src, int64_t num_maps, int32_t H, int32_t W, uint32_t* packed,
                                     void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_maps < 0 || H <= 0 || W <= 0) return MBV_ERR_BAD_ARG;
  if (num_maps == 0) return MBV_OK;
  if (!src || !packed) return MBV_ERR_BAD_ARG;
  if (num_maps > 65535) return MBV_ERR_UNSUPPORTED;
  const int64_t hw = (int64_t)H * W, words = mbv_packed_mask_words(H, W);
  if (hw % 1024 == 0 && (reinterpret_cast<size_t>(src) & 15) == 0)
    hipLaunchKernelGGL(k_pack_binary_v4, dim3((unsigned)((hw / 1024 + 3) / 4), (unsigned)num_maps), dim3(256), 0, stream,
                       src, hw, words, packed);
  else
    hipLaunchKernelGGL(k_pack_binary, dim3((unsigned)((words * 32 + 255) / 256), (unsigned)num_maps), dim3(256), 0,
                       stream, src, hw, words, packed);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_point_sample_packed_fwd(const uint32_t* packed, const int32_t* src_index, const float* coords,
                                           const int32_t* coord_index, int32_t num_rows, int32_t num_points,
                                           int32_t H, int32_t W, float* out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_rows < 0 || num_points <= 0 || H <= 0 || W <= 0) return MBV_ERR_BAD_ARG;
  if (num_rows == 0) return MBV_OK;
  if (!packed || !src_index || !coords || !coord_index || !out) return MBV_ERR_BAD_ARG;
  const int64_t words = mbv_packed_mask_words(H, W);
  if (words > kPackedWords) return MBV_ERR_UNSUPPORTED;
  if (words <= 8192)
    hipLaunchKernelGGL(k_point_sample_packed<8192>, dim3(num_rows), dim3(512), 0, stream, packed, words, src_index,
                       coords, coord_index, num_points, H, W, out);
  else
    hipLaunchKernelGGL(k_point_sample_packed<kPackedWords>, dim3(num_rows), dim3(512), 0, stream, packed, words,
                       src_index, coords, coord_index, num_points, H, W, out);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_point_sample_bwd_stack(const float* grad_out, const int32_t* src_index, const float* coords,
                                          const int32_t* coord_index, int32_t num_rows, int32_t num_points, int32_t H,
                                          int32_t W, int32_t outer, int32_t inner, int32_t rows, void* grad_src,
                                          int32_t out_dtype, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_rows <= 0 || num_points <= 0 || H <= 0 || W <= 0 || outer <= 0 || inner <= 0 || rows <= 0) return MBV_ERR_BAD_ARG;
  if (!grad_out || !src_index || !coords || !coord_index || !grad_src) return MBV_ERR_BAD_ARG;
  if (out_dtype != MBV_DT_F32 && out_dtype != MBV_DT_BF16 && out_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  // every map of the stack is sampled exactly once (no zero fill), in the LDS-tile form (larger maps: in row bands)
  int bands = 0;
  const int band_rows = lds_band_rows(H, W, &bands);
  if ((int64_t)num_rows != (int64_t)outer * inner * rows || num_rows > 65535 || bands == 0 || bands > 64)
    return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_point_sample_bwd_lds, dim3((unsigned)num_rows * (unsigned)bands), dim3(1024), 0, stream, grad_out,
                     src_index, coords, coord_index, num_points, H, W, grad_src, out_dtype, outer, inner, rows, bands,
                     band_rows);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_point_sample_bwd(const float* grad_out, const int32_t* src_index, const float* coords,
                                    const int32_t* coord_index, int32_t num_rows, int32_t num_points, int32_t H,
                                    int32_t W, int64_t num_src_maps, float* grad_src, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_rows < 0 || num_points <= 0 || H <= 0 || W <= 0 || num_src_maps < 0) return MBV_ERR_BAD_ARG;
  if (!grad_src && num_src_maps > 0) return MBV_ERR_BAD_ARG;
  // The LDS form stores every sampled map's whole tile (src_index holds no duplicates): when every map is sampled —
  // num_rows == num_src_maps, the loss with all queries matched — nothing is left for the zero fill (262 MB, 35 us)
  int bands = 0;
  const int band_rows = lds_band_rows(H, W, &bands);
  const bool lds_form = bands >= 1 && bands <= 64;      // one LDS tile, or up to 64 row bands of the map
  if (!(lds_form && num_rows == num_src_maps && num_rows > 0))
    MBV_CHECK_HIP(mbv_fill_async(grad_src, 0, sizeof(float) * (size_t)num_src_maps * H * W, stream));
  if (num_rows == 0) return MBV_OK;
  if (!grad_out || !src_index || !coords || !coord_index) return MBV_ERR_BAD_ARG;
  if (num_rows > 65535) return MBV_ERR_UNSUPPORTED;
  if (lds_form) {
    hipLaunchKernelGGL(k_point_sample_bwd_lds, dim3((unsigned)num_rows * (unsigned)bands), dim3(1024), 0, stream, grad_out,
                       src_index, coords, coord_index, num_points, H, W, grad_src, MBV_DT_F32, 0, 0, 0, bands, band_rows);
  } else {
    hipLaunchKernelGGL(k_point_sample_bwd_atomic, dim3((num_points + 255) / 256, num_rows), dim3(256), 0, stream,
                       grad_out, src_index, coords, coord_index, num_points, H, W, grad_src);
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
