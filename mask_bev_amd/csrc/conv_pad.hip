// Layout turns around the in-tree 3 x 3 convolution (the pixel decoder's FPN output convolution, /root/reference:
// mask_bev/models/head/mask_bev_panoptic_head.py:119-146 — mmdet MSDeformAttnPixelDecoder.output_convs, a ConvModule
// (256 -> 256, 3 x 3, padding 1) on the (B, 256, 128, 128) stride-4 map): the convolution runs as ONE token-major product
// over k = (tap, channel) on a zero-bordered channels-last copy of the map (K20's GATHER 4, csrc/gemm_f32s.hip), so that
// the row of output position m for tap (dy, dx) is simply row m + (dy - 1)(W + 2) + (dx - 1) of the same matrix.
//
//   rows buffer: (G + B (H + 2)(W + 2) + G, C), G = W + 3 guard rows in front and behind; everything outside the interior
//   pixels is zero (the caller clears the buffer; k_pad_rows writes the interior only).
#include "common.hpp"

namespace {

template <typename T>
__global__ void __launch_bounds__(256) k_pad_rows(const T* __restrict__ src, T* __restrict__ dst, int C, int H, int W) {
  __shared__ T tile[64][33];
  const int x0 = blockIdx.x * 32, y = blockIdx.y, cblocks = (C + 63) / 64;
  const int b = blockIdx.z / cblocks, c0 = (blockIdx.z - b * cblocks) * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int c = c0 + ty + 8 * cc, x = x0 + tx;
    tile[ty + 8 * cc][tx] = (c < C && x < W) ? src[(((size_t)b * C + c) * H + y) * W + x] : (T)0;
  }
  __syncthreads();
  const int cl = threadIdx.x & 63, px = threadIdx.x >> 6;
  const size_t row0 = (size_t)(W + 3) + ((size_t)b * (H + 2) + y + 1) * (W + 2) + 1;
#pragma unroll
  for (int pp = 0; pp < 8; ++pp) {
    const int x = px + 4 * pp;
    if (x0 + x < W && c0 + cl < C) dst[(row0 + x0 + x) * C + c0 + cl] = tile[cl][x];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_unpad_rows(const T* __restrict__ src, T* __restrict__ dst, int C, int H, int W) {
  __shared__ T tile[64][33];
  const int x0 = blockIdx.x * 32, y = blockIdx.y, cblocks = (C + 63) / 64;
  const int b = blockIdx.z / cblocks, c0 = (blockIdx.z - b * cblocks) * 64;
  const int cl = threadIdx.x & 63, px = threadIdx.x >> 6;
  const size_t row0 = (size_t)(W + 3) + ((size_t)b * (H + 2) + y + 1) * (W + 2) + 1;
#pragma unroll
  for (int pp = 0; pp < 8; ++pp) {
    const int x = px + 4 * pp;
    tile[cl][x] = (x0 + x < W && c0 + cl < C) ? src[(row0 + x0 + x) * C + c0 + cl] : (T)0;
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int c = c0 + ty + 8 * cc, x = x0 + tx;
    if (c < C && x < W) dst[(((size_t)b * C + c) * H + y) * W + x] = tile[ty + 8 * cc][tx];
  }
}

}  // namespace

extern "C" int64_t mbv_conv_rows(int64_t batch, int64_t H, int64_t W) {
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  return 2 * (W + 3) + batch * (H + 2) * (W + 2);
}

// dst rows (mbv_conv_rows(batch, H, W), C), cleared by the caller, <- the interior pixels of src (batch, C, H, W); elem_size 2 or 4
extern "C" int mbv_conv_pad_rows(const void* src, void* dst, int64_t batch, int64_t C, int64_t H, int64_t W, int32_t elem_size,
                                 void* stream) {
  if (!src || !dst || batch <= 0 || C <= 0 || H <= 0 || W <= 0 || (elem_size != 2 && elem_size != 4)) return MBV_ERR_BAD_ARG;
  if (H > 65535 || batch * ((C + 63) / 64) > 65535) return MBV_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((W + 31) / 32), (unsigned)H, (unsigned)(batch * ((C + 63) / 64)));
  if (elem_size == 4)
    hipLaunchKernelGGL(k_pad_rows<float>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(src),
                       reinterpret_cast<float*>(dst), (int)C, (int)H, (int)W);
  else
    hipLaunchKernelGGL(k_pad_rows<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(src), reinterpret_cast<unsigned short*>(dst), (int)C, (int)H, (int)W);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// dst (batch, C, H, W) <- the interior pixels of the rows buffer src
extern "C" int mbv_conv_unpad_rows(const void* src, void* dst, int64_t batch, int64_t C, int64_t H, int64_t W, int32_t elem_size,
                                   void* stream) {
  if (!src || !dst || batch <= 0 || C <= 0 || H <= 0 || W <= 0 || (elem_size != 2 && elem_size != 4)) return MBV_ERR_BAD_ARG;
  if (H > 65535 || batch * ((C + 63) / 64) > 65535) return MBV_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((W + 31) / 32), (unsigned)H, (unsigned)(batch * ((C + 63) / 64)));
  if (elem_size == 4)
    hipLaunchKernelGGL(k_unpad_rows<float>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(src),
                       reinterpret_cast<float*>(dst), (int)C, (int)H, (int)W);
  else
    hipLaunchKernelGGL(k_unpad_rows<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(src), reinterpret_cast<unsigned short*>(dst), (int)C, (int)H, (int)W);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
