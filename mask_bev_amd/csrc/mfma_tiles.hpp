// MFMA tile helpers shared by the attention kernels (K4 window attention, K6 decoder attention).
// Fragment maps: cdna_hip_programming.md §3 (32x32x16 bf16 and 32x32x2 f32; C/D layout col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace mbv_tiles {

// 16-bit operands are this build's lo16_t (common.hpp): bf16, or IEEE half in the `_f16` companions
typedef __attribute__((__vector_size__(8 * sizeof(lo16_t)))) lo16_t bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(lo16_t)))) lo16_t bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int NPAD = 128;   // rows of a staged [token][d] image: 4 blocks of 32
constexpr int NBLK = 4;

template <bool BF16, int D>
struct Lay {
  using T = std::conditional_t<BF16, lo16_t, float>;
  static constexpr int RS = BF16 ? D + 8 : D + 1;          // row stride of [token][d] images (elements)
  static constexpr int TROWS = D < 32 ? 32 : D;             // rows of [d][token] images (zero rows beyond D)
  static constexpr int TS = NPAD + 8;                       // row stride of [d][token] images
  static constexpr int ROW_IMG = NPAD * RS;
  static constexpr int T_IMG = TROWS * TS;
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(lo16_t v) { return (float)v; }

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef MBV_H16
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// accumulator register i of a 32x32 tile, lane half h  ->  row inside the 32-row block
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- MFMA tile helpers ------------------------------------------------------------------------------
// acc(32x32) += Arows[a0 + r][:] . Brows[b0 + r][:]^T   (contraction over d), both [row][d] images.
// Result layout: col = lane&31 <-> B row, acc_row(i, h) <-> A row.
template <bool BF16, int D>
__device__ __forceinline__ void mma_rows(const typename Lay<BF16, D>::T* __restrict__ a_img, int a0,
                                         const typename Lay<BF16, D>::T* __restrict__ b_img, int b0, f32x16& acc) {
  using L = Lay<BF16, D>;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  if constexpr (BF16) {
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_img + (a0 + r) * L::RS + 16 * ks + 8 * h);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_img + (b0 + r) * L::RS + 16 * ks + 8 * h);
      acc = mfma16(a, b, acc);
    }
  } else {
#pragma unroll 8
    for (int ks = 0; ks < D / 2; ++ks) {
      const float a = a_img[(a0 + r) * L::RS + 2 * ks + h];
      const float b = b_img[(b0 + r) * L::RS + 2 * ks + h];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
}

// out(32 x 32 cols [cb]) += X^T . M  where X is an accumulator tile (rows = contraction index k0 + acc_row,
// cols = lane = output row) and M holds the other operand: bf16 -> [d][token] image, f32 -> [token][d] image.
template <bool BF16, int D>
__device__ __forceinline__ void mma_acc_operand(const f32x16& x, const typename Lay<BF16, D>::T* __restrict__ m_img,
                                                int k0, int cb, f32x16& out) {
  using L = Lay<BF16, D>;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  if constexpr (BF16) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = (lo16_t)x[8 * s + j];
      // element j of half h is contraction index 16 s + 8 (j >> 2) + 4 h + (j & 3)
      const typename L::T* p = m_img + (r + 32 * cb) * L::TS + k0 + 16 * s + 4 * h;
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 8);
      bf16x8 b;
#pragma unroll
      for (int j = 0; j < 4; ++j) { b[j] = lo[j]; b[4 + j] = hi[j]; }
      out = mfma16(a, b, out);
    }
  } else {
    const int col = r + 32 * cb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float b = col < D ? m_img[(k0 + acc_row(i, h)) * L::RS + col] : 0.f;
      out = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], b, out, 0, 0, 0);
    }
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

}  // namespace mbv_tiles
