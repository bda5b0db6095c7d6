// micro-benchmark: cost of LDS atomic wave instructions on gfx950 (cycles per instruction, one wave per SIMD / CU load varied)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, int stride) {
  __shared__ float f[8192];
  __shared__ unsigned u[8192];
  __shared__ unsigned long long w[8192];
  __shared__ double dd[4096];
  for (int i = threadIdx.x; i < 8192; i += 256) { f[i] = 0.f; u[i] = 0; w[i] = 0; if (i < 4096) dd[i] = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx = (wave * 1024 + lane * stride) & 8191;
  long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) atomicAdd(&f[idx], 1.0f);
    if (MODE == 1) atomicAdd(&u[idx], 1u);
    if (MODE == 2) f[idx] += 1.0f;          // plain read-modify-write (not atomic)
    if (MODE == 3) atomicMax(&u[idx], (unsigned)i);
    if (MODE == 4) atomicAdd(&w[idx & 8191], 3ull);
    if (MODE == 5) atomicAdd(&dd[idx & 4095], 1.0);
    idx = (idx + 64 * stride) & 8191;
  }
  __syncthreads();
  long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0) / iters;
  if (threadIdx.x == 1) out[1000 + blockIdx.x] = f[5] + u[7] + (float)w[3] + (float)dd[9];
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  std::vector<float> h(4096);
  const char* names[6] = {"ds_add_f32", "ds_add_u32", "plain rmw", "ds_max_u32", "ds_add_u64", "ds_add_f64"};
  for (int stride : {1, 2}) {
    for (int mode = 0; mode < 6; ++mode) {
      for (int blocks : {256}) {
        auto run = [&](int it) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, it, stride);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, it, stride);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, it, stride);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, it, stride);
          if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, it, stride);
          if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, it, stride);
        };
        run(100); hipDeviceSynchronize(); run(4000); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
        printf("stride %2d  %-11s  %.1f clk per loop iteration (4 waves/CU issuing)\n", stride, names[mode], h[0]);
      }
    }
  }
  return 0;
}
