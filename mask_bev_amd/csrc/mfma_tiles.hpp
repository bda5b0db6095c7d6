// MFMA tile helpers shared by the attention kernels (K4 window attention, K6 decoder attention).
// Fragment maps: cdna_hip_programming.md §3 (32x32x16 bf16 and 32x32x2 f32; C/D layout col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace mbv_tiles {

// 16-bit operands are this build's lo16_t (common.hpp): bf16, or IEEE half in the `_f16` companions
typedef __attribute__((__vector_size__(8 * sizeof(lo16_t)))) lo16_t bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int NPAD = 128;   // rows of a staged [token][d] image: 4 blocks of 32
constexpr int NBLK = 4;

// f32 operands: padded-row [token][d] images (row stride D + 1 words: conflict-free scalar reads).  16-bit operands use
// the swizzled images below; `Lay<BF16, D>::T` only names the element type for code shared by both paths.
template <bool BF16, int D>
struct Lay {
  using T = std::conditional_t<BF16, lo16_t, float>;
  static constexpr int RS = D + 1;                          // row stride of the f32 [token][d] images (elements)
  static constexpr int ROW_IMG = NPAD * RS;
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

// ---- exact-f32 MFMA tile helpers (v_mfma_f32_32x32x2_f32) ---------------------------------------------------------
// acc(32x32) += Arows[a0 + r][:] . Brows[b0 + r][:]^T   (contraction over d), both padded-row [row][d] images.
// Result layout: col = lane&31 <-> B row, acc_row(i, h) <-> A row.
template <bool BF16, int D>
__device__ __forceinline__ void mma_rows(const float* __restrict__ a_img, int a0, const float* __restrict__ b_img, int b0,
                                         f32x16& acc) {
  static_assert(!BF16, "16-bit operands use mma_rows_swz");
  using L = Lay<false, D>;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll 8
  for (int ks = 0; ks < D / 2; ++ks) {
    const float a = a_img[(a0 + r) * L::RS + 2 * ks + h];
    const float b = b_img[(b0 + r) * L::RS + 2 * ks + h];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}

// out(32 x 32 cols [cb]) += X^T . M  where X is an accumulator tile (rows = contraction index k0 + acc_row,
// cols = lane = output row) and M the other operand's padded-row [token][d] image.
template <bool BF16, int D>
__device__ __forceinline__ void mma_acc_operand(const f32x16& x, const float* __restrict__ m_img, int k0, int cb,
                                                f32x16& out) {
  static_assert(!BF16, "16-bit operands use mma_acc_tr");
  using L = Lay<false, D>;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int col = r + 32 * cb;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float b = col < D ? m_img[(k0 + acc_row(i, h)) * L::RS + col] : 0.f;
    out = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], b, out, 0, 0, 0);
  }
}

// ---- 16-bit LDS images without padding or transposed copies -----------------------------------------------------
// A [token][d] image of a 16-bit operand is NPAD rows of exactly D elements; the 16-byte chunk c of row r sits at
// chunk c ^ ((r >> SH) & (D/8 - 1)), which makes the 16-byte MFMA operand reads of 16 consecutive rows hit 16
// different bank groups (the padded rows of mfma_tiles.hpp did the same at 12 % more LDS).  The [d][token] operand
// of the `X^T . M` products (P V, dS K, P^T dO, dS^T Q) is read from the SAME image with gfx950's transposing LDS
// read (`ds_read_b64_tr_b16`: a 16-lane group fetches a 4 x 16 block row-wise and receives it column-wise), so the
// transposed copies that the staging pass used to scatter into LDS two bytes at a time are gone: the backward's LDS
// footprint of K4 drops from 135 KB to 76 KB at D = 64 — two workgroups per CU instead of one — and the forward's to
// 52 KB (three instead of two); K6 uses the same images.
template <int D>
struct Swz {
  static constexpr int CH = D / 8;                                  // 16-byte chunks per row
  static constexpr int SH = D >= 64 ? 1 : (D == 32 ? 2 : 3);        // rows that share a swizzle value
  static constexpr int IMG = NPAD * D + 64;                         // + slack: at D = 16 a transposed read of the last
                                                                    //   rows runs 16 (discarded) columns past the image
  static __device__ __forceinline__ int chunk_off(int row, int c) {  // element offset of chunk c of `row`
    return row * D + ((c ^ ((row >> SH) & (CH - 1))) << 3);
  }
};

// acc(32x32) += A[a0 + r][:] . B[b0 + r][:]^T over d, both swizzled [token][d] images (cf. mbv_tiles::mma_rows)
template <int D>
__device__ __forceinline__ void mma_rows_swz(const lo16_t* __restrict__ a_img, int a0, const lo16_t* __restrict__ b_img,
                                             int b0, f32x16& acc) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_img + Swz<D>::chunk_off(a0 + r, 2 * ks + h));
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_img + Swz<D>::chunk_off(b0 + r, 2 * ks + h));
    acc = mfma16(a, b, acc);
  }
}

// out(32 x 32 cols [cb]) += X^T . M  (cf. mbv_tiles::mma_acc_operand): X an accumulator tile whose rows are the
// contraction index k0 + acc_row(i, h); M the swizzled [token][d] image, read column-wise by the transposing load.
// Lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p + 3 of a 4 x 16 block and receives column
// (lane & 15) of it; the groups cover column halves ((lane >> 4) & 1) and row halves (h) of the MFMA B operand, whose
// element j of half h is contraction index 16 s + 8 (j >> 2) + 4 h + (j & 3) — two reads, rows +0..3 and +8..11.
template <int D>
__device__ __forceinline__ void mma_acc_tr(const f32x16& x, const lo16_t* __restrict__ m_img, int k0, int cb,
                                           f32x16& out) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const int c = 4 * cb + 2 * ((lane >> 4) & 1) + (p >> 1);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (lo16_t)x[8 * s + j];
    const int row0 = k0 + 16 * s + 4 * h + q;
    const lo16_t* p0 = m_img + Swz<D>::chunk_off(row0, c) + 4 * (p & 1);
    const lo16_t* p1 = m_img + Swz<D>::chunk_off(row0 + 8, c) + 4 * (p & 1);
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)p0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)p1);
    const s16x8 t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    out = mfma16(a, __builtin_bit_cast(bf16x8, t), out);
  }
}

#ifdef MBV_H16
// ---- f32 operands as IEEE-half pairs (the split mode of K4; K20's arithmetic, csrc/gemm_f32s.hip) -------------------------
// An operand x * 2^e = hi + lo (hi = half(x 2^e), lo = half(x 2^e - hi): 22 significant bits) lives in TWO swizzled images;
// a product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 (the small terms first).
template <int D>
__device__ __forceinline__ void mma_rows_split(const lo16_t* __restrict__ a_hi, const lo16_t* __restrict__ a_lo, int a0,
                                               const lo16_t* __restrict__ b_hi, const lo16_t* __restrict__ b_lo, int b0,
                                               f32x16& acc) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const int oa = Swz<D>::chunk_off(a0 + r, 2 * ks + h), ob = Swz<D>::chunk_off(b0 + r, 2 * ks + h);
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a_hi + oa), al = *reinterpret_cast<const bf16x8*>(a_lo + oa);
    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b_hi + ob), bl = *reinterpret_cast<const bf16x8*>(b_lo + ob);
    acc = mfma16(ah, bl, acc);
    acc = mfma16(al, bh, acc);
    acc = mfma16(ah, bh, acc);
  }
}

// out += (X sx)^T . M  with X an accumulator tile (split in registers at the power-of-two scale sx) and M = m_hi + m_lo
template <int D>
__device__ __forceinline__ void mma_acc_tr_split(const f32x16& x, float sx, const lo16_t* __restrict__ m_hi,
                                                 const lo16_t* __restrict__ m_lo, int k0, int cb, f32x16& out) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const int c = 4 * cb + 2 * ((lane >> 4) & 1) + (p >> 1);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 ah, al;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float y = x[8 * s + j] * sx;
      ah[j] = (lo16_t)y;
      al[j] = (lo16_t)(y - (float)ah[j]);
    }
    const int row0 = k0 + 16 * s + 4 * h + q;
    const int o0 = Swz<D>::chunk_off(row0, c) + 4 * (p & 1), o1 = Swz<D>::chunk_off(row0 + 8, c) + 4 * (p & 1);
    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(m_hi + o0));
    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(m_hi + o1));
    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(m_lo + o0));
    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(m_lo + o1));
    const bf16x8 th = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
    const bf16x8 tl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
    out = mfma16(ah, tl, out);
    out = mfma16(al, th, out);
    out = mfma16(ah, th, out);
  }
}
#endif

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

}  // namespace mbv_tiles
