// Shared helpers for the gfx950 kernels of libmaskbev_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/maskbev_hip.h"

#define MBV_WAVE 64

// Launch check: a failed launch is returned to the caller as a positive hipError_t.
#define MBV_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

#define MBV_CHECK_HIP(expr)                        \
  do {                                             \
    hipError_t e__ = (expr);                       \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline size_t mbv_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Carves 256-byte aligned pieces out of a caller-provided workspace.
struct MbvCarver {
  char* base;
  size_t off;
  explicit MbvCarver(void* p) : base(reinterpret_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t count) {
    T* r = reinterpret_cast<T*>(base + off);
    off += mbv_align_up(count * sizeof(T), 256);
    return r;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
