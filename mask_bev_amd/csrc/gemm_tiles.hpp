// Tile helpers shared by the MFMA GEMM translation units (gemm.hip: K17, 16-bit operands by LDS-DMA; gemm_f32s.hip: K20,
// f32 operands split into IEEE-half pairs at staging time): 16-bit LDS images of an operand tile and the fragment
// reads that feed v_mfma_f32_32x32x16_{bf16,f16} from them, buffer descriptors, the epilogue's scalar helpers, the
// XCD-aware work order.  Everything here is internal linkage (anonymous namespace of the including file).
#pragma once
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int STG_LD = 68;                        // f32 row stride of a wave's 32 x 64 epilogue tile
constexpr unsigned OOB = 0x80000000u;             // byte offset beyond any descriptor: the load returns zero

template <typename T>
struct Mma;
template <>
struct Mma<__bf16> {
  static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
  static __device__ __forceinline__ float to_f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
  // a plain cast: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN — MI355X_MICROARCH.md §Correctness)
  static __device__ __forceinline__ unsigned short from_f(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
  }
};
template <>
struct Mma<_Float16> {
  static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
  static __device__ __forceinline__ float to_f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
  static __device__ __forceinline__ unsigned short from_f(float f) {
    return __builtin_bit_cast(unsigned short, (_Float16)f);
  }
};

// Raw buffer descriptor (stride 0, range-checked: a byte offset at or beyond `bytes` reads as zero).
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r[0] = (int)(unsigned)b;
  r[1] = (int)(unsigned)((b >> 32) & 0xffffu);
  r[2] = (int)bytes;
  r[3] = 0x00020000;
  return r;
}

// One LDS-DMA wave-instruction: lane l's 16 bytes at byte offset voff + soff of the buffer land at LDS byte address
// lds_addr + 16 l.  Written in asm so that hipcc does not count it: the compiler otherwise waits vmcnt(0) in front of
// the next ds_read of the same __shared__ array (it cannot tell the two halves of the double buffer apart), which
// serialises the prefetch with the MFMAs it is meant to run under.  Completion is this kernel's own
// `s_waitcnt vmcnt(0)` in front of the barrier that publishes the tile; M0 is written in the statement that reads it
// (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(i32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(rs), "s"(soff), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
               : "memory", "m0");
}

__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)p);
}

// ---- LDS images ---------------------------------------------------------------------------------------------------
// KC image (contraction index contiguous): [128 rows][KB k] 16-bit.  KB = 64: 128-byte rows of 8 chunks of 16 B, chunk
// c of row r at physical chunk c ^ ((r >> 1) & 7) (two rows share a 256-byte bank row); KB = 32: 64-byte rows of 4
// chunks, chunk c at c ^ ((r >> 2) & 3) (four rows per bank row).  Either way the 16 rows x one chunk column that a
// ds_read_b128 lane group touches land in 16 different 16-byte slots.  One wave-instruction (1 KiB) fills 8 / 16 rows.
template <int KB>
__device__ __forceinline__ void kc_lane(int piece, int lane, int& r, int& c) {
  if (KB == 64) {
    r = 8 * piece + (lane >> 3);
    c = (lane & 7) ^ ((r >> 1) & 7);
  } else {
    r = 16 * piece + (lane >> 2);
    c = (lane & 3) ^ ((r >> 2) & 3);
  }
}

// KS image (contraction index strided): [KB k rows][128 cols] 16-bit, 256-byte rows of 16 chunks; chunk c of row r
// at physical chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3))  (cdna_hip_programming.md T10, image (b): conflict-free
// for the 32x32x16 transposed reads).  One wave-instruction fills 4 rows.
__device__ __forceinline__ int ks_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

__device__ __forceinline__ void ks_lane(int piece, int lane, int& r, int& c) {
  r = 4 * piece + (lane >> 4);
  c = (lane & 15) ^ ks_swz(r);
}

// Byte offset of lane `lane` of piece `local` (0 .. PPS-1) of sub-image `sub` of an operand tile whose first row /
// column is x0, for the K-step that starts at k0.  Every bound is checked: out-of-range elements get the OOB offset
// and arrive as zeros.
template <int KB, bool KS>
__device__ __forceinline__ unsigned piece_offset(int sub, int local, int x0, int x_total, int ld_bytes, int k0, int k_end,
                                                 int lane) {
  int r, c;
  if (KS) {
    ks_lane(local, lane, r, c);
    const int gk = k0 + r, gx = x0 + 128 * sub + 8 * c;
    return (gk < k_end && gx < x_total) ? (unsigned)gk * (unsigned)ld_bytes + (unsigned)gx * 2u : OOB;
  }
  kc_lane<KB>(local, lane, r, c);
  const int gx = x0 + 128 * sub + r, gk = k0 + 8 * c;
  return (gx < x_total && gk < k_end) ? (unsigned)gx * (unsigned)ld_bytes + (unsigned)gk * 2u : OOB;
}

// ---- LDS -> MFMA fragments ---------------------------------------------------------------------------------------
// 32x32x16 operand of k-step ks (16 k's): lane l (r = l & 31, h = l >> 5) holds element (row / col r, k = 8h + j).
template <int KB>
__device__ __forceinline__ uint4 frag_kc(const char* img, int r0, int ks, int lane) {
  const int row = r0 + (lane & 31), c = 2 * ks + (lane >> 5);
  if (KB == 64) return *reinterpret_cast<const uint4*>(img + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  return *reinterpret_cast<const uint4*>(img + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
}

__device__ __forceinline__ uint4 frag_ks(const char* img, int c0, int ks, int lane) {
  // 16-lane group g: columns c0 + 16 (g & 1) .. + 15, k rows 16 ks + 8 (g >> 1) + 4 s .. + 3 for s = 0, 1;
  // lane 4q + p of the group addresses row q, columns 4p .. 4p + 3 of the 4 x 16 block
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const int col = c0 + 16 * ((lane >> 4) & 1);
  const int ch = (col >> 3) + (p >> 1);
  const int rr = 16 * ks + 8 * (lane >> 5) + q;
  const char* p0 = img + rr * 256 + ((ch ^ ks_swz(rr)) << 4) + 8 * (p & 1);
  const int rr1 = rr + 4;
  const char* p1 = img + rr1 * 256 + ((ch ^ ks_swz(rr1)) << 4) + 8 * (p & 1);
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)p0);
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)p1);
  const s16x8 t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(uint4, t);
}

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__device__ __forceinline__ float gelu_cdf_parts(float z, float& dens) {
  // Phi(z) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 in erf), sharing exp(-z^2/2) with the density
  const float e = __expf(-0.5f * z * z);
  const float az = fabsf(z) * 0.70710678118654752f;
  const float t = mbv_rcp(1.f + 0.3275911f * az);            // argument in [1, inf)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float half_tail = 0.5f * poly * e;
  dens = 0.3989422804014327f * e;
  return z >= 0.f ? 1.f - half_tail : half_tail;
}

template <typename T>
__device__ __forceinline__ void store8(void* base, size_t o, const float (&v)[8], int out_f32) {
  if (out_f32) {
    float* d = reinterpret_cast<float*>(base) + o;
    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 u;
    u.x = Mma<T>::from_f(v[0]) | ((unsigned)Mma<T>::from_f(v[1]) << 16);
    u.y = Mma<T>::from_f(v[2]) | ((unsigned)Mma<T>::from_f(v[3]) << 16);
    u.z = Mma<T>::from_f(v[4]) | ((unsigned)Mma<T>::from_f(v[5]) << 16);
    u.w = Mma<T>::from_f(v[6]) | ((unsigned)Mma<T>::from_f(v[7]) << 16);
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + o) = u;
  }
}

template <typename T>
__device__ __forceinline__ void round8(float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = Mma<T>::to_f(Mma<T>::from_f(v[e]));
}

// OUT: 0 = STORE (C^T accumulators, row-major store through LDS with the epilogue), 1 = ATOMIC (f32 adds)
// XCD-aware block order (blocks b and b + 8 share an XCD): every XCD walks a contiguous range of the work index
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, s = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

}  // namespace
