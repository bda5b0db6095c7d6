// Shared helpers for the gfx950 kernels of libmaskbev_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/maskbev_hip.h"

#define MBV_WAVE 64

// ---- the 16-bit storage type of this translation unit ----------------------------------------------------------
// Every kernel file that touches 16-bit activations (K3, K4, K6, K7, K12, K13, the optimizer's weight shadow) is
// compiled twice: as it stands for bf16, and once more through its `<name>_f16.hip` companion, which defines MBV_H16
// and includes it, for IEEE half (`v_mfma_f32_32x32x16_f16`, `v_cvt_f16_f32`).  Kernels live in anonymous namespaces,
// so the two instantiations do not collide; the C entry points of the half build get an `_f16` suffix and hidden
// visibility, and the public entry point forwards to them when its dtype argument is MBV_DT_F16.
#define MBV_DT_F32 0
#define MBV_DT_BF16 1
#define MBV_DT_F16 2
#ifdef MBV_H16
typedef _Float16 lo16_t;
#define MBV_SYM(name) name##_f16
#define MBV_ENTRY extern "C" __attribute__((visibility("hidden")))
#else
typedef __bf16 lo16_t;
#define MBV_SYM(name) name
#define MBV_ENTRY extern "C"
#endif
// declaration of the half build's twin of an entry point (same signature), visible to the bf16 build only
#define MBV_F16_TWIN extern "C" __attribute__((visibility("hidden")))
// a dtype flag as the half build sees it: its own 16-bit type is "1"
#define MBV_LO_FLAG(x) ((x) == MBV_DT_F16 ? 1 : (x))

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

// v_rcp_f32 / v_log_f32 as they are (1 ulp), for arguments known to be normal numbers away from 0: `1.0f / x`, `__frcp_rn` and
// `__logf` compile to the IEEE division sequence (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) and to a logarithm
// with denormal scaling — 10 and 8 issue slots that a VALU-bound epilogue pays per element.
__device__ __forceinline__ float mbv_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float mbv_ln(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

static inline size_t mbv_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Zero / byte-fill device memory with a KERNEL.  hipMemsetAsync issued from inside this library was observed not
// to be re-executed when the enclosing stream capture is replayed as a HIP graph (gradients accumulated into
// stale pool memory from the second replay on); a kernel node always replays.
static __global__ void __launch_bounds__(256) mbv_k_fill_bytes(unsigned char* __restrict__ p, int value, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i + 16 <= n && (reinterpret_cast<size_t>(p) & 15) == 0) {
    const unsigned v4 = 0x01010101u * (unsigned)(value & 0xff);
    *reinterpret_cast<uint4*>(p + i) = make_uint4(v4, v4, v4, v4);
  } else {
    for (size_t j = i; j < n && j < i + 16; ++j) p[j] = (unsigned char)value;
  }
}

static inline hipError_t mbv_fill_async(void* ptr, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  const size_t threads = (bytes + 15) / 16;
  hipLaunchKernelGGL(mbv_k_fill_bytes, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<unsigned char*>(ptr), value, bytes);
  return hipGetLastError();
}

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

// f32 ↔ the bits of this build's 16-bit type (round to nearest even; half: overflow goes to ±inf)
__device__ __forceinline__ float lo16_to_f32(unsigned short h) {
#ifdef MBV_H16
  return (float)__builtin_bit_cast(_Float16, h);
#else
  return __uint_as_float((unsigned)h << 16);
#endif
}

// f32 → bf16 bits, round to nearest even (what a torch `.to(bfloat16)` does); NaN stays NaN.  gfx950 has the conversion
// as an instruction — v_cvt_pk_bf16_f32, two values per issue slot — where the integer form costs six per value.
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 mbv_bf2 __attribute__((ext_vector_type(2)));
  typedef float mbv_f2 __attribute__((ext_vector_type(2)));
  const mbv_f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mbv_bf2));
}
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ unsigned short f32_to_lo16(float f) {
#ifdef MBV_H16
  return __builtin_bit_cast(unsigned short, (_Float16)f);
#else
  return f32_to_bf16_rne(f);
#endif
}
