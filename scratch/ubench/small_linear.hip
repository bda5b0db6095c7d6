// Prototype: y (M, N) = x (M, K) . w (N, K)^T + bias for few-row f32 Linears (M = 400), exact-f32 MFMA straight from
// global memory; one 32 x 32 output tile per workgroup, the contraction split over its 4 waves (LDS reduction) and,
// for long K, over gridDim.z workgroups (f32 atomics onto a bias-initialised output).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

template <bool RELU>
__global__ void __launch_bounds__(256) k_small_nt(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                  int K, int kchunk) {
  __shared__ float red[3][32 * 33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int kb = (blockIdx.z * 4 + wave) * kchunk, ke = min(K, kb + kchunk);
  const int m = min(m0 + r, M - 1), n = min(n0 + r, N - 1);
  const float* xr = x + (size_t)m * K;
  const float* wr = w + (size_t)n * K;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = kb; k0 < ke; k0 += 32) {
    float4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = *reinterpret_cast<const float4*>(xr + k0 + 4 * u);
      b[u] = *reinterpret_cast<const float4*>(wr + k0 + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[u].y : a[u].x, h ? b[u].y : b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[u].w : a[u].z, h ? b[u].w : b[u].z, acc, 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave - 1][acc_row(i, h) * 33 + r] = acc[i];
  }
  __syncthreads();
  if (wave == 0) {
    const int col = n0 + r;
    const float bv = (bias && blockIdx.z == 0 && col < N) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = acc_row(i, h);
      float v = acc[i] + red[0][row * 33 + r] + red[1][row * 33 + r] + red[2][row * 33 + r] + bv;
      if (m0 + row < M && col < N) {
        if (gridDim.z > 1) atomicAdd(&y[(size_t)(m0 + row) * N + col], v);
        else y[(size_t)(m0 + row) * N + col] = RELU ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

extern "C" int small_nt(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu,
                        int ksplit, void* stream) {
  int kchunk = (K + 4 * ksplit - 1) / (4 * ksplit);
  kchunk = (kchunk + 31) / 32 * 32;
  dim3 grid((N + 31) / 32, (M + 31) / 32, ksplit);
  if (relu) hipLaunchKernelGGL(k_small_nt<true>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K, kchunk);
  else hipLaunchKernelGGL(k_small_nt<false>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K, kchunk);
  return (int)hipGetLastError();
}
