// K7 — per-query mask logits (MFMA contraction) and the boolean cross-attention mask derived from them.
//
// Replaces Mask2FormerHead._forward_head's `torch.einsum('bqc,bchw->bqhw', mask_embed, mask_feature)`
// (mask_bev/models/networks/mask2former_head/mask2former_head.py:459) and the mask pipeline that follows it
// (:460-470 bilinear resize → flatten → repeat over 8 heads → sigmoid() < 0.5, and the "row fully blocked →
// unblock" fix of :538-539).
//
// Contraction: out[b][q][p] = sum_c E[b][q][c] * F[b][c][p]  (M = Q <= 128 per workgroup row block, K = C,
// N = H*W pixels).  F is channel-major, i.e. the B operand is stored [K][N]; each workgroup stages a
// 128-pixel slab of F in LDS once (bf16: transposed to [pixel][c] so that MFMA B fragments are 16-byte reads;
// f32: natural [c][pixel], one float per lane) and every F byte is read from HBM exactly once.  E (<= 64 KB) is
// read from L2 by every workgroup.  bf16 uses v_mfma_f32_32x32x16_bf16, f32 uses v_mfma_f32_32x32x2_f32 (exact).
// The logits are HBM-write bound: B*Q*H*W*sizeof(T) bytes per call.
#include <type_traits>

#include "common.hpp"

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(lo16_t)))) lo16_t bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int PT = 128;    // pixels per workgroup
constexpr int QB = 128;    // queries per workgroup (4 waves x 32)

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef MBV_H16
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- bf16 --------------------------------------------------------------------------------------------
template <int KC, typename TOut>   // channels staged per pass; logits stored as bf16 or (the f32 accumulators) f32
__global__ void __launch_bounds__(256) k_mask_logits_bf16(const lo16_t* __restrict__ E, const lo16_t* __restrict__ F,
                                                          int Q, int C, int64_t HW, TOut* __restrict__ out) {
  constexpr int LD = KC + 8;
  __shared__ __attribute__((aligned(16))) lo16_t ft[PT * LD];      // [pixel][c]
  const int b = blockIdx.y, q0 = blockIdx.z * QB;
  const int64_t p0 = (int64_t)blockIdx.x * PT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const lo16_t* Fb = F + (int64_t)b * C * HW;
  const int q = q0 + 32 * wave + r;
  const lo16_t* Eq = E + ((int64_t)b * Q + (q < Q ? q : 0)) * C;
  f32x16 acc[4];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[pb][i] = 0.f;
  for (int kc = 0; kc < C; kc += KC) {
    if (kc) __syncthreads();
    // stage F[kc .. kc+KC)[p0 .. p0+PT) transposed: 8 consecutive pixels of one channel per thread-iteration
    for (int idx = threadIdx.x; idx < KC * (PT / 8); idx += blockDim.x) {
      const int c = idx / (PT / 8), px = (idx - c * (PT / 8)) * 8;
      lo16_t v[8];
      const bool c_ok = kc + c < C;
      if (c_ok && p0 + px + 7 < HW && (HW & 7) == 0) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(Fb + (int64_t)(kc + c) * HW + p0 + px);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = (c_ok && p0 + px + j < HW) ? Fb[(int64_t)(kc + c) * HW + p0 + px + j] : (lo16_t)0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ft[(px + j) * LD + c] = v[j];
    }
    __syncthreads();
    const int ksteps = (C - kc < KC ? C - kc : KC) / 16;
    for (int ks = 0; ks < ksteps; ++ks) {
      bf16x8 a;
      if (q < Q) {
        a = *reinterpret_cast<const bf16x8*>(Eq + kc + 16 * ks + 8 * h);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (lo16_t)0.f;
      }
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(&ft[(32 * pb + r) * LD + 16 * ks + 8 * h]);
        acc[pb] = mfma16(a, bfrag, acc[pb]);
      }
    }
  }
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int64_t p = p0 + 32 * pb + r;
    if (p >= HW) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = q0 + 32 * wave + acc_row(i, h);
      if (qq < Q) out[((int64_t)b * Q + qq) * HW + p] = (TOut)acc[pb][i];
    }
  }
}

// ---- f32 (exact) -------------------------------------------------------------------------------------
template <int KC>
__global__ void __launch_bounds__(256) k_mask_logits_f32(const float* __restrict__ E, const float* __restrict__ F,
                                                         int Q, int C, int64_t HW, float* __restrict__ out) {
  constexpr int LDF = PT + 1, LDE = KC + 1;
  __shared__ float fn[KC * LDF];      // [c][pixel]
  __shared__ float en[QB * LDE];      // [q][c]
  const int b = blockIdx.y, q0 = blockIdx.z * QB;
  const int64_t p0 = (int64_t)blockIdx.x * PT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float* Fb = F + (int64_t)b * C * HW;
  f32x16 acc[4];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[pb][i] = 0.f;
  for (int kc = 0; kc < C; kc += KC) {
    if (kc) __syncthreads();
    for (int idx = threadIdx.x; idx < KC * PT; idx += blockDim.x) {
      const int c = idx / PT, px = idx - c * PT;
      fn[c * LDF + px] = (kc + c < C && p0 + px < HW) ? Fb[(int64_t)(kc + c) * HW + p0 + px] : 0.f;
    }
    for (int idx = threadIdx.x; idx < QB * KC; idx += blockDim.x) {
      const int qq = idx / KC, c = idx - qq * KC;
      en[qq * LDE + c] = (q0 + qq < Q && kc + c < C) ? E[((int64_t)b * Q + q0 + qq) * C + kc + c] : 0.f;
    }
    __syncthreads();
    const int ksteps = (C - kc < KC ? C - kc : KC) / 2;
    for (int ks = 0; ks < ksteps; ++ks) {
      const float a = en[(32 * wave + r) * LDE + 2 * ks + h];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const float bb = fn[(2 * ks + h) * LDF + 32 * pb + r];
        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[pb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int64_t p = p0 + 32 * pb + r;
    if (p >= HW) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = q0 + 32 * wave + acc_row(i, h);
      if (qq < Q) out[((int64_t)b * Q + qq) * HW + p] = acc[pb][i];
    }
  }
}

// ---- attention mask ----------------------------------------------------------------------------------
__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const lo16_t* p) { return (float)*p; }

// one workgroup per (b, q): bilinear resize (align_corners=False) of the (H, W) logits to (h, w),
// blocked = sigmoid(v) < 0.5, and if every key would be blocked the row is un-blocked.
template <typename T>
__global__ void __launch_bounds__(256) k_attn_mask(const T* __restrict__ logits, int H, int W, int h, int w,
                                                   uint8_t* __restrict__ blocked) {
  __shared__ int cnt_s;
  const int64_t row = blockIdx.x;
  const T* src = logits + row * H * W;
  uint8_t* dst = blocked + row * h * w;
  const int L = h * w;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  if (threadIdx.x == 0) cnt_s = 0;
  __syncthreads();
  int cnt = 0;
  for (int t = threadIdx.x; t < L; t += blockDim.x) {
    const int oy = t / w, ox = t - oy * w;
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float v = (1.f - ly) * ((1.f - lx) * ld(src + y0 * W + x0) + lx * ld(src + y0 * W + x1)) +
                    ly * ((1.f - lx) * ld(src + y1 * W + x0) + lx * ld(src + y1 * W + x1));
    const bool blk = (1.f / (1.f + __expf(-v))) < 0.5f;
    dst[t] = blk ? 1 : 0;
    cnt += blk ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt_s, cnt);
  __syncthreads();
  if (cnt_s == L) {   // mask2former_head.py:538-539
    for (int t = threadIdx.x; t < L; t += blockDim.x) dst[t] = 0;
  }
}

}  // namespace

#ifndef MBV_H16
// the half build of this file (mask_logits_f16.hip); `is_bf16` = MBV_DT_F16 forwards there
MBV_F16_TWIN int mbv_mask_logits_fwd_f16(const void*, const void*, int32_t, int32_t, int32_t, int32_t, int64_t, void*,
                                         int32_t, void*);
MBV_F16_TWIN int mbv_attn_mask_from_logits_f16(const void*, int32_t, int64_t, int32_t, int32_t, int32_t, int32_t, uint8_t*,
                                               void*);
#endif

MBV_ENTRY int MBV_SYM(mbv_mask_logits_fwd)(const void* mask_embed, const void* mask_feature, int32_t is_bf16,
                                           int32_t batch, int32_t num_queries, int32_t channels, int64_t pixels,
                                           void* logits, int32_t logits_f32, void* stream_) {
#ifndef MBV_H16
  if (is_bf16 == MBV_DT_F16)
    return mbv_mask_logits_fwd_f16(mask_embed, mask_feature, 1, batch, num_queries, channels, pixels, logits, logits_f32,
                                   stream_);
#endif
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || num_queries <= 0 || channels <= 0 || pixels <= 0) return MBV_ERR_BAD_ARG;
  if (!mask_embed || !mask_feature || !logits) return MBV_ERR_BAD_ARG;
  const dim3 grid((unsigned)((pixels + PT - 1) / PT), batch, (num_queries + QB - 1) / QB), block(256);
  if (is_bf16) {
    if (channels % 16 != 0) return MBV_ERR_UNSUPPORTED;
    if (logits_f32)
      hipLaunchKernelGGL((k_mask_logits_bf16<256, float>), grid, block, 0, stream,
                         reinterpret_cast<const lo16_t*>(mask_embed), reinterpret_cast<const lo16_t*>(mask_feature),
                         num_queries, channels, pixels, reinterpret_cast<float*>(logits));
    else
      hipLaunchKernelGGL((k_mask_logits_bf16<256, lo16_t>), grid, block, 0, stream,
                         reinterpret_cast<const lo16_t*>(mask_embed), reinterpret_cast<const lo16_t*>(mask_feature),
                         num_queries, channels, pixels, reinterpret_cast<lo16_t*>(logits));
  } else {
    if (channels % 2 != 0) return MBV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_mask_logits_f32<128>, grid, block, 0, stream, reinterpret_cast<const float*>(mask_embed),
                       reinterpret_cast<const float*>(mask_feature), num_queries, channels, pixels,
                       reinterpret_cast<float*>(logits));
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

MBV_ENTRY int MBV_SYM(mbv_attn_mask_from_logits)(const void* logits, int32_t is_bf16, int64_t rows, int32_t H,
                                                 int32_t W, int32_t h, int32_t w, uint8_t* blocked, void* stream_) {
#ifndef MBV_H16
  if (is_bf16 == MBV_DT_F16) return mbv_attn_mask_from_logits_f16(logits, 1, rows, H, W, h, w, blocked, stream_);
#endif
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (rows < 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!logits || !blocked) return MBV_ERR_BAD_ARG;
  if (is_bf16)
    hipLaunchKernelGGL(k_attn_mask<lo16_t>, dim3((unsigned)rows), dim3(256), 0, stream,
                       reinterpret_cast<const lo16_t*>(logits), H, W, h, w, blocked);
  else
    hipLaunchKernelGGL(k_attn_mask<float>, dim3((unsigned)rows), dim3(256), 0, stream,
                       reinterpret_cast<const float*>(logits), H, W, h, w, blocked);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
