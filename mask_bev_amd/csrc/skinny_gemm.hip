// K2c — streaming f32 GEMM for the PillarFeatureNet's Linear layers.
//
// mmdet3d `PFNLayer.linear` (`nn.Linear(in, out, bias=False)`, /root/reference: mask_bev/models/encoders/mask_bev_encoders.py:
// 70-72 → mmdet3d PillarFeatureNet, restated at oracle/maskbev_oracle.py:233-253) on the compact rows of the real points:
//   Y (M, N) = X (M, C) . W^T      forward           (W (N, C), row stride ldw: the [a | max] halves of a layer's weight are views)
//   dX (M, N) = dY (M, C) . W      data gradient     (W (C, N))
// with M = 440 668 rows at the bench batch and C, N <= 128: 100-340 MB streamed for 0.6-7 GFLOP, exact f32.  The library's f32
// kernels ran these at 2.4-3.3 TB/s (47-104 us each, 0.6 ms per step).  Here a wave owns 32-row tiles of X: the tile goes through a
// wave-private LDS image (no workgroup barrier anywhere), the whole of W sits in the wave's registers as MFMA B fragments for the
// whole launch, and the products are v_mfma_f32_32x32x2_f32 (exact f32).  The contraction index is permuted — lane half h takes
// k in [h C/2, (h + 1) C/2) — so that a lane's A operands of four consecutive steps are ONE 16-byte LDS read; the next tile's
// global loads are issued before the current tile's products.  Bounds: the f32 MFMA rate (64 -> 128: 46 us) or HBM (10 -> 64: 22 us).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;

// CPH = padded contraction length / 2 (a multiple of 4), NTW = 32-column tiles per wave
// ADD: y[row] += add[index[row]] on the way out (the PFNLayer's W_b . max(pillar) term: one (pillars, N) product gathered per row
// instead of a second pass over y that adds it)
template <int CPH, int NTW, bool WT, bool VEC, bool ADD = false>     // VEC: c == 2 CPH exactly (16-byte tile loads, shifts for the row / column split); WT: B(k, n) = W[n * ldw + k] (forward), else W[k * ldw + n] (data gradient)
__global__ void __launch_bounds__(64 * kWaves) k_skinny_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, long m, int c, int n, int ldw,
                                                            const float* __restrict__ add = nullptr,
                                                            const int64_t* __restrict__ index = nullptr) {
  extern __shared__ float lds[];
  constexpr int CP = 2 * CPH;
  constexpr int S = (CP > NTW * 32 ? CP : NTW * 32) + 4;  // row stride (words): conflict-free 16-byte reads; the image also
                                                          // turns the 32 x (32 NTW) result tile into whole rows on its way out
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float* img = lds + wave * 32 * S;
  const int n0 = blockIdx.y * NTW * 32;

  // B fragments of the wave's columns, for the whole launch
  float b[CPH][NTW];
  if (WT && VEC && (ldw & 3) == 0 && (reinterpret_cast<size_t>(w) & 15) == 0) {
    // a lane's CPH contraction indices are consecutive in its weight row: 16-byte loads (a quarter of the requests — with 2 048
    // waves fetching the same 32 KB, the dword form of this prologue cost as much as the products of a short launch)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = n0 + t * 32 + r;
#pragma unroll
      for (int s4 = 0; s4 < CPH / 4; ++s4) {
        const float4 v = col < n ? *reinterpret_cast<const float4*>(w + (long)col * ldw + h * CPH + 4 * s4)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        b[4 * s4][t] = v.x; b[4 * s4 + 1][t] = v.y; b[4 * s4 + 2][t] = v.z; b[4 * s4 + 3][t] = v.w;
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < CPH; ++s) {
      const int k = h * CPH + s;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = n0 + t * 32 + r;
        b[s][t] = (k < c && col < n) ? (WT ? w[(long)col * ldw + k] : w[(long)k * ldw + col]) : 0.f;
      }
    }
  }
  // zero the image once: the pad columns (k >= c) stay zero
  for (int i = lane; i < 32 * S; i += 64) img[i] = 0.f;

  const long tiles = (m + 31) / 32;
  const long stride = (long)gridDim.x * kWaves;
  constexpr bool vec = VEC;
  constexpr int NV = VEC ? CP / 8 : 1;                    // float4 per lane of a full 32 x CP tile
  constexpr int NS = VEC ? 1 : CP / 2;                    // scalar form (c % 4 != 0): 32 c / 64 values per lane
  float4 pre[NV];
  float pres[NS];
  auto fetch = [&](long tile) {
    const long base = tile * 32 * (long)c, end = m * (long)c;
    if constexpr (vec) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const long g = base + 4L * (i * 64 + lane);       // float4 index inside the tile's 32 x CP block
        pre[i] = g + 3 < end ? *reinterpret_cast<const float4*>(x + g) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int f = i * 64 + lane;
        const long g = base + f;
        pres[i] = (f < 32 * c && g < end) ? x[g] : 0.f;
      }
    }
  };
  auto stage = [&]() {
    if constexpr (vec) {
      constexpr int per_row = CP / 4;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int f = i * 64 + lane;
        *reinterpret_cast<float4*>(img + (f / per_row) * S + 4 * (f % per_row)) = pre[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int f = i * 64 + lane;
        if (f < 32 * c) {
          const int row = f / c, q = f - row * c;
          img[row * S + q] = pres[i];
        }
      }
    }
  };

  long tile = (long)blockIdx.x * kWaves + wave;
  if (tile < tiles) fetch(tile);
  for (; tile < tiles; tile += stride) {
    __builtin_amdgcn_wave_barrier();
    stage();
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the image is written
    if (tile + stride < tiles) fetch(tile + stride);
    // (ADD) the addend row of tile row (lane & 31): rows of a tile are handed round by shuffles on the way out, so every
    // gathered load below depends on nothing but this one
    long my_index = 0;
    if constexpr (ADD) my_index = tile * 32 + (lane & 31) < m ? index[tile * 32 + (lane & 31)] : 0;
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // the lane's contraction indices are h * CPH + [0, CPH): with the permuted order its A operands are contiguous.
    // (the image holds row-major [row][k]; half 1's block starts CPH words into the row)
    float4 a_cur = *reinterpret_cast<const float4*>(img + r * S + h * CPH);
#pragma unroll
    for (int j = 0; j < CPH / 4; ++j) {
      const float4 a4 = a_cur;
      if (j + 1 < CPH / 4) a_cur = *reinterpret_cast<const float4*>(img + r * S + h * CPH + 4 * (j + 1));   // one read ahead
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b[4 * j + s][t], acc[t], 0, 0, 0);
    }
    // the result leaves as whole 16-byte pieces of its rows: 32 x (32 NTW) floats through the wave's image (its x tile is
    // consumed) — 8 NTW stores per lane instead of 16 NTW scattered ones, few enough for the next tile's loads, issued before
    // them, to be waited for without draining the stores too
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) img[((i & 3) + 8 * (i >> 2) + 4 * h) * S + t * 32 + r] = acc[t][i];
    __builtin_amdgcn_wave_barrier();
    const long row0 = tile * 32;
    constexpr int per_out = NTW * 8;                    // float4 per result row
    if constexpr (ADD) {
      // all gathered addend pieces of the tile first (NTW * 4 independent 16-byte loads per lane), then the stores
      float4 tadd[NTW * 4];
#pragma unroll
      for (int i = 0; i < NTW * 4; ++i) {
        const int f = i * 64 + lane, row = f / per_out, q = f % per_out;
        const int col = n0 + 4 * q;
        const int lo = __shfl((int)(my_index & 0xffffffffL), row), hi = __shfl((int)(my_index >> 32), row);
        const long pil = ((long)hi << 32) | (unsigned)lo;
        tadd[i] = (row0 + row < m && col < n) ? *reinterpret_cast<const float4*>(add + pil * n + col)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NTW * 4; ++i) {
        const int f = i * 64 + lane, row = f / per_out, q = f % per_out;
        const int col = n0 + 4 * q;
        if (row0 + row < m && col < n) {
          float4 v = *reinterpret_cast<const float4*>(img + row * S + 4 * q);
          v.x += tadd[i].x; v.y += tadd[i].y; v.z += tadd[i].z; v.w += tadd[i].w;
          *reinterpret_cast<float4*>(y + (row0 + row) * n + col) = v;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NTW * 4; ++i) {
        const int f = i * 64 + lane, row = f / per_out, q = f % per_out;
        const int col = n0 + 4 * q;
        if (row0 + row < m && col < n)
          *reinterpret_cast<float4*>(y + (row0 + row) * n + col) = *reinterpret_cast<const float4*>(img + row * S + 4 * q);
      }
    }
    if constexpr (!vec) {                               // the pad columns of the x image must read as zero again
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < 32 * S; i += 64) img[i] = 0.f;
    }
  }
}

template <int CPH, int NTW>
int launch(int wt, const float* x, const float* w, float* y, long m, int c, int n, int ldw, hipStream_t st,
           const float* add = nullptr, const int64_t* index = nullptr) {
  const long tiles = (m + 31) / 32;
  long bx = (tiles + kWaves - 1) / kWaves;
  if (bx > 512) bx = 512;
  const dim3 grid((unsigned)bx, (unsigned)((n + NTW * 32 - 1) / (NTW * 32))), block(64 * kWaves);
  const size_t lds = (size_t)kWaves * 32 * ((2 * CPH > NTW * 32 ? 2 * CPH : NTW * 32) + 4) * sizeof(float);
  const bool vec = c == 2 * CPH && (reinterpret_cast<size_t>(x) & 15) == 0;
  if (add) {                                             // (forward form only)
    if (!wt) return MBV_ERR_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL((k_skinny_f32<CPH, NTW, true, true, true>), grid, block, lds, st, x, w, y, m, c, n, ldw, add, index);
    else hipLaunchKernelGGL((k_skinny_f32<CPH, NTW, true, false, true>), grid, block, lds, st, x, w, y, m, c, n, ldw, add, index);
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
  if (wt && vec) hipLaunchKernelGGL((k_skinny_f32<CPH, NTW, true, true>), grid, block, lds, st, x, w, y, m, c, n, ldw, nullptr, nullptr);
  else if (wt) hipLaunchKernelGGL((k_skinny_f32<CPH, NTW, true, false>), grid, block, lds, st, x, w, y, m, c, n, ldw, nullptr, nullptr);
  else if (vec) hipLaunchKernelGGL((k_skinny_f32<CPH, NTW, false, true>), grid, block, lds, st, x, w, y, m, c, n, ldw, nullptr, nullptr);
  else hipLaunchKernelGGL((k_skinny_f32<CPH, NTW, false, false>), grid, block, lds, st, x, w, y, m, c, n, ldw, nullptr, nullptr);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

}  // namespace

// The largest dynamic LDS a launch of this file asks for (kWaves x 32 x (128 + 4) floats = 67 584 B for the <64, 2>, <32, 4> and
// <8, 4> shapes): more than the 64 KB some devices allow a workgroup (gfx950: 160 KB) — such a device gets "unsupported", and
// the callers keep the library GEMM, instead of a failed launch.
static bool skinny_lds_fits() {
  static int limit = -1;                 // an attribute of the device, queried once
  if (limit < 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess)
      v = 65536;
    limit = v;
  }
  return (size_t)limit >= (size_t)kWaves * 32 * (128 + 4) * sizeof(float);
}

extern "C" int mbv_skinny_gemm_f32_supported(int64_t m, int32_t contraction, int32_t out_cols) {
  return m > 0 && contraction >= 1 && contraction <= 128 && out_cols >= 32 && out_cols <= 128 && out_cols % 32 == 0 &&
         m * (int64_t)(contraction > out_cols ? contraction : out_cols) < 0x7fffffffffLL && skinny_lds_fits();
}

static int skinny_dispatch(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n, int32_t ldw,
                           int32_t weight_is_nk, const float* add, const int64_t* index, hipStream_t st) {
  if (c <= 16) return n <= 64 ? launch<8, 2>(weight_is_nk, x, w, y, m, c, n, ldw, st, add, index)
                              : launch<8, 4>(weight_is_nk, x, w, y, m, c, n, ldw, st, add, index);
  if (c <= 64) return n <= 64 ? launch<32, 2>(weight_is_nk, x, w, y, m, c, n, ldw, st, add, index)
                              : launch<32, 4>(weight_is_nk, x, w, y, m, c, n, ldw, st, add, index);
  return launch<64, 2>(weight_is_nk, x, w, y, m, c, n, ldw, st, add, index);
}

// y (m, n) = x (m, c) . w^T + add[index[row]]: the forward form with a gathered row addend (add (*, n) f32, index (m) i64).
extern "C" int mbv_skinny_gemm_f32_addrows(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n,
                                           int32_t ldw, const float* add, const int64_t* index, void* stream) {
  if (m < 0 || c <= 0 || n <= 0 || ldw <= 0) return MBV_ERR_BAD_ARG;
  if (m == 0) return MBV_OK;
  if (!mbv_skinny_gemm_f32_supported(m, c, n)) return MBV_ERR_UNSUPPORTED;
  if (!x || !w || !y || !add || !index) return MBV_ERR_BAD_ARG;
  if ((n & 3) || ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(add)) & 15)) return MBV_ERR_UNSUPPORTED;
  return skinny_dispatch(x, w, y, m, c, n, ldw, 1, add, index, (hipStream_t)stream);
}

// weight_is_nk != 0: y (m, n) = x (m, c) . w^T with w (n, c), row stride ldw;  == 0: y (m, n) = x (m, c) . w with w (c, n), row stride ldw.
extern "C" int mbv_skinny_gemm_f32(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n, int32_t ldw,
                                   int32_t weight_is_nk, void* stream) {
  if (m < 0 || c <= 0 || n <= 0 || ldw <= 0) return MBV_ERR_BAD_ARG;
  if (m == 0) return MBV_OK;
  if (!mbv_skinny_gemm_f32_supported(m, c, n)) return MBV_ERR_UNSUPPORTED;
  if (!x || !w || !y) return MBV_ERR_BAD_ARG;
  if ((n & 3) || (reinterpret_cast<size_t>(y) & 15)) return MBV_ERR_UNSUPPORTED;
  // register budget: CPH x NTW weight fragments <= 128; 64-column groups for the long contractions
  return skinny_dispatch(x, w, y, m, c, n, ldw, weight_is_nk, nullptr, nullptr, (hipStream_t)stream);
}
