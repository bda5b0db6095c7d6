// Absmax records: how the f32 kernels that compute on the 16-bit matrix pipe (K20 gemm_f32s.hip, K4's split mode in
// window_attn.hip) learn the range of an f32 tensor.  Internal linkage (anonymous namespace of the including file).
#pragma once
#include "common.hpp"

namespace {

// An absmax "word" is kAmaxSlots words (256 bytes): producers max-combine into slot (workgroup id mod kAmaxSlots) — tens
// of thousands of atomics on ONE address would serialise — and the consumer takes the maximum of the slots.
constexpr int kAmaxSlots = 64;

// 2^e with max * 2^e in [2^13, 2^14) from the bits of max|x| (exact; 1 for an all-zero operand), and its inverse.
__device__ __forceinline__ float pow2_scale(const unsigned* amax, float& inv) {
  inv = 1.f;
  if (!amax) return 1.f;
  unsigned bits = 0u;                             // (a uniform address: scalar loads and scalar max)
#pragma unroll
  for (int i = 0; i < kAmaxSlots; ++i) {
    const unsigned w = amax[i] & 0x7fffffffu;
    bits = bits > w ? bits : w;
  }
  if (bits == 0u) return 1.f;
  int e = 267 - (int)(bits >> 23);                // biased exponent of the scale: 127 + 13 - (exponent(max) - 127)
  e = e < 1 ? 1 : (e > 253 ? 253 : e);            // both the scale and its inverse stay normal numbers
  inv = __uint_as_float((unsigned)(254 - e) << 23);
  return __uint_as_float((unsigned)e << 23);
}


// The biased exponent of pow2_scale's result (127 = scale 1), for callers that combine scales of several tensors.
__device__ __forceinline__ int pow2_scale_exp(const unsigned* amax) {
  if (!amax) return 127;
  unsigned bits = 0u;
#pragma unroll
  for (int i = 0; i < kAmaxSlots; ++i) {
    const unsigned w = amax[i] & 0x7fffffffu;
    bits = bits > w ? bits : w;
  }
  if (bits == 0u) return 127;
  const int e = 267 - (int)(bits >> 23);
  return e < 1 ? 1 : (e > 253 ? 253 : e);
}

// 2^(e - 127) for a biased exponent clamped to the normal range
__device__ __forceinline__ float pow2_from_exp(int e) {
  e = e < 1 ? 1 : (e > 254 ? 254 : e);
  return __uint_as_float((unsigned)e << 23);
}

}  // namespace
