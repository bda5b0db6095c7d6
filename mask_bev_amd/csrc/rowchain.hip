// K19 — row-local stage chains for the transformer decoder's query side (B*Q = 400 tokens x 256 channels), gfx950.
//
// Replaces, per decoder layer, the ≈ 28 forward and ≈ 30 backward launches of the reference's
//   cross-attn out-proj + residual + LayerNorm -> self-attn q/k/v projections -> out-proj + residual + LayerNorm ->
//   FFN 256-2048-256 (ReLU) + residual + LayerNorm -> post-norm + class / mask-embed heads -> next layer's query projection
// (mask_bev/models/networks/mask2former_head/mask2former_head.py:535-560 and :428-472; mmdet
// Mask2FormerTransformerDecoderLayer / mmcv MultiheadAttention, FFN) — few-row GEMMs of 5-8 us each with the chip idle,
// LayerNorms, clamps, adds and casts — with ONE launch per stretch between two attention kernels.
//
// Everything on the query side between two attention calls is ROW-LOCAL: a workgroup owns 16 token rows and walks a
// small program of stages over them (mbv_rowchain_run, include/maskbev_hip.h): LOAD / STORE rows, GEMM against a weight
// matrix streamed straight from L2 into MFMA B fragments (the M <= 16 weight-streaming form: the operand is used once
// per workgroup, an LDS round trip would be pure overhead), LayerNorm forward / backward, ADD, column partial sums.
// Activations live in LDS "slots" of 16 x 256 f32 between stages; nothing of a chain touches HBM except what the
// backward pass or an attention kernel needs.  The program is passed BY VALUE in the kernel arguments (no upload, and a
// captured HIP graph bakes it in), built by the host side (mask_bev_amd/decoder_fused.py).
//
// GEMM operand types (`wdtype`): f32 weights -> exact-f32 MFMA (v_mfma_f32_16x16x4_f32), bf16 / fp16 weights -> the
// activations are rounded to that type as MFMA A fragments, f32 accumulation (v_mfma_f32_16x16x32_{bf16,f16}).
// Backward data gradients dX = dY W are NT products against TRANSPOSED weight copies (mbv_transpose_group, one launch
// per step for the whole decoder), so that the reduction index is the contiguous one in every stream.
#include "common.hpp"

namespace {

constexpr int RC_ROWS = 16, RC_MAXC = 256, RC_LD = RC_MAXC + 4, RC_SLOTS = 7, RC_NT = 512, RC_NW = RC_NT / 64;
constexpr int RC_MAX_STAGES = 64;

struct RowProgram {
  int num_stages, rows, q_mod, wdtype;
  float eps;
  int pad[3];
  MbvRowStage st[RC_MAX_STAGES];
};

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ float cvt_in(const void* p, int64_t i, int dt) {
  if (dt == MBV_DT_F32) return reinterpret_cast<const float*>(p)[i];
  const unsigned short h = reinterpret_cast<const unsigned short*>(p)[i];
  if (dt == MBV_DT_BF16) return __uint_as_float((unsigned)h << 16);
  return (float)__builtin_bit_cast(_Float16, h);
}

__device__ __forceinline__ unsigned short f32_to_16(float f, int dt) {
  if (dt == MBV_DT_BF16) return f32_to_bf16_rne(f);
  return __builtin_bit_cast(unsigned short, (_Float16)f);
}

// ---- stages ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_load(const MbvRowStage& s, float* slots, int row0, int rows, int q_mod) {
  float* dst = slots + s.dst * (RC_ROWS * RC_LD);
  const int dt = s.flags & 3, n = s.n;
  const int n4 = n >> 2;                                   // n % 4 == 0 (host-checked)
  for (int i = threadIdx.x; i < RC_ROWS * n4; i += RC_NT) {
    const int r = i / n4, c = (i - r * n4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int row = row0 + r;
    if (row < rows) {
      const int64_t o = (int64_t)row * s.ld + c;
      if (!s.p0) {                                         // base operand from a slot (e.g. x + positions)
        v = *reinterpret_cast<const float4*>(slots + s.src * (RC_ROWS * RC_LD) + r * RC_LD + c);
      } else if (dt == MBV_DT_F32) {
        v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(s.p0) + o);
      } else {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(s.p0) + o);
        if (dt == MBV_DT_BF16) {
          v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                          __uint_as_float(u.y & 0xffff0000u));
        } else {
          v = make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(u.x & 0xffff)),
                          (float)__builtin_bit_cast(_Float16, (unsigned short)(u.x >> 16)),
                          (float)__builtin_bit_cast(_Float16, (unsigned short)(u.y & 0xffff)),
                          (float)__builtin_bit_cast(_Float16, (unsigned short)(u.y >> 16)));
        }
      }
      if (s.p1) {                                          // + a second f32 operand, row index modulo q_mod (positions)
        const int prow = q_mod > 0 ? row % q_mod : row;
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(s.p1) + (int64_t)prow * s.ld2 + c);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    *reinterpret_cast<float4*>(dst + r * RC_LD + c) = v;
  }
}

__device__ __forceinline__ void st_store(const MbvRowStage& s, const float* slots, int row0, int rows) {
  const float* src = slots + s.src * (RC_ROWS * RC_LD);
  const int dt = s.flags & 3, n = s.n;
  const bool accum = (s.flags & MBV_RC_ACCUM) != 0;
  if ((n & 3) == 0) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < RC_ROWS * n4; i += RC_NT) {
      const int r = i / n4, c = (i - r * n4) * 4;
      const int row = row0 + r;
      if (row >= rows) continue;
      const float4 v = *reinterpret_cast<const float4*>(src + r * RC_LD + c);
      const int64_t o = (int64_t)row * s.ld + c;
      if (dt == MBV_DT_F32) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(const_cast<void*>(s.p0)) + o);
        if (accum) {
          const float4 old = *p;
          *p = make_float4(old.x + v.x, old.y + v.y, old.z + v.z, old.w + v.w);
        } else {
          *p = v;
        }
      } else {
        uint2 u;
        u.x = (unsigned)f32_to_16(v.x, dt) | ((unsigned)f32_to_16(v.y, dt) << 16);
        u.y = (unsigned)f32_to_16(v.z, dt) | ((unsigned)f32_to_16(v.w, dt) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(const_cast<void*>(s.p0)) + o) = u;
      }
    }
  } else {                                                 // narrow outputs (the class head: n = classes + 1)
    for (int i = threadIdx.x; i < RC_ROWS * n; i += RC_NT) {
      const int r = i / n, c = i - r * n;
      const int row = row0 + r;
      if (row >= rows) continue;
      const float v = src[r * RC_LD + c];
      const int64_t o = (int64_t)row * s.ld + c;
      if (dt == MBV_DT_F32) {
        float* p = reinterpret_cast<float*>(const_cast<void*>(s.p0)) + o;
        *p = accum ? *p + v : v;
      } else {
        reinterpret_cast<unsigned short*>(const_cast<void*>(s.p0))[o] = f32_to_16(v, dt);
      }
    }
  }
}

// dst[r][j] = act((ACCUM ? dst[r][j] : 0) + sum_k src[r][k] W[j][k] + bias[j]) [* (src2[r][j] > 0)], j < n.
// One wave owns 16-column output tiles t = wave, wave + 8, ...; its B fragments come straight from global memory
// (16 rows of W x 64 contiguous bytes per load instruction), all of a tile's loads issued before its first MFMA.
template <int WDT>
__device__ __forceinline__ void st_gemm(const MbvRowStage& s, float* slots) {
  const float* X = slots + s.src * (RC_ROWS * RC_LD);
  float* Y = slots + s.dst * (RC_ROWS * RC_LD);
  const float* Mk = (s.flags & MBV_RC_MASK) ? slots + s.src2 * (RC_ROWS * RC_LD) : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int n = s.n, k = s.k;
  const int tiles = (n + 15) >> 4;
  const bool accum = (s.flags & MBV_RC_ACCUM) != 0, relu = (s.flags & MBV_RC_RELU) != 0;
  const float* bias = reinterpret_cast<const float*>(s.p1);
  for (int t = wave; t < tiles; t += RC_NW) {
    const int col = t * 16 + m;
    const bool valid = col < n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (accum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = Y[(4 * g + i) * RC_LD + col];
    }
    if constexpr (WDT == MBV_DT_F32) {
      // k permuted consistently in A and B: MFMA j of a 16-wide k block sums k = kb + 4 g' + j, g' = 0..3
      const float* wrow = reinterpret_cast<const float*>(s.p0) + (int64_t)(valid ? col : 0) * s.ld + 4 * g;
      constexpr int KB = RC_MAXC / 16;
      float4 b[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        b[kb] = (valid && kb * 16 < k) ? *reinterpret_cast<const float4*>(wrow + kb * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb * 16 < k) {
          const float4 a = *reinterpret_cast<const float4*>(X + m * RC_LD + kb * 16 + 4 * g);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[kb].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[kb].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[kb].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[kb].w, acc, 0, 0, 0);
        }
      }
    } else {
      const unsigned short* wrow = reinterpret_cast<const unsigned short*>(s.p0) + (int64_t)(valid ? col : 0) * s.ld + 8 * g;
      constexpr int KB = RC_MAXC / 32;
      uint4 b[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        b[kb] = (valid && kb * 32 < k) ? *reinterpret_cast<const uint4*>(wrow + kb * 32) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb * 32 < k) {
          const float4 a0 = *reinterpret_cast<const float4*>(X + m * RC_LD + kb * 32 + 8 * g);
          const float4 a1 = *reinterpret_cast<const float4*>(X + m * RC_LD + kb * 32 + 8 * g + 4);
          if constexpr (WDT == MBV_DT_BF16) {
            bf16x8 a;
            a[0] = (__bf16)a0.x; a[1] = (__bf16)a0.y; a[2] = (__bf16)a0.z; a[3] = (__bf16)a0.w;
            a[4] = (__bf16)a1.x; a[5] = (__bf16)a1.y; a[6] = (__bf16)a1.z; a[7] = (__bf16)a1.w;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b[kb]), acc, 0, 0, 0);
          } else {
            f16x8 a;
            a[0] = (_Float16)a0.x; a[1] = (_Float16)a0.y; a[2] = (_Float16)a0.z; a[3] = (_Float16)a0.w;
            a[4] = (_Float16)a1.x; a[5] = (_Float16)a1.y; a[6] = (_Float16)a1.z; a[7] = (_Float16)a1.w;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(f16x8, b[kb]), acc, 0, 0, 0);
          }
        }
      }
    }
    if (valid) {
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[i] + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (Mk) v = Mk[(4 * g + i) * RC_LD + col] > 0.f ? v : 0.f;
        Y[(4 * g + i) * RC_LD + col] = v;
      }
    }
  }
}

// 32 lanes per row: dst = LN(src [+ src2]) * gamma + beta; optional: src <- the sum, stats (mean, rstd) to p2
__device__ __forceinline__ void st_ln(const MbvRowStage& s, float* slots, int row0, int rows, float eps) {
  float* A = slots + s.src * (RC_ROWS * RC_LD);
  const float* B = s.src2 >= 0 ? slots + s.src2 * (RC_ROWS * RC_LD) : nullptr;
  float* Y = slots + s.dst * (RC_ROWS * RC_LD);
  const int r = threadIdx.x >> 5, l = threadIdx.x & 31, n = s.n;
  const float* gamma = reinterpret_cast<const float*>(s.p0);
  const float* beta = reinterpret_cast<const float*>(s.p1);
  float v[RC_MAXC / 32];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    float x = 0.f;
    if (c < n) {
      x = A[r * RC_LD + c];
      if (B) x += B[r * RC_LD + c];
    }
    v[j] = x;
    sum += x;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)n;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    if (c < n) { const float d = v[j] - mean; var += d * d; }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
  const float rstd = rsqrtf(var / (float)n + eps);
  const bool save = (s.flags & MBV_RC_SAVE_SUM) != 0;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    if (c < n) {
      if (save) A[r * RC_LD + c] = v[j];
      Y[r * RC_LD + c] = (v[j] - mean) * rstd * gamma[c] + beta[c];
    }
  }
  if (s.p2 && l == 0 && row0 + r < rows) {
    float* st = reinterpret_cast<float*>(const_cast<void*>(s.p2)) + (int64_t)(row0 + r) * 2;
    st[0] = mean;
    st[1] = rstd;
  }
}

// dst = d(sum) of y = LN(sum) gamma + beta given g = src (dL/dy), sum = src2, stats p2; per-block partial
// d(gamma) / d(beta) rows to p1 [block][2 n] (rows beyond `rows` hold zero gradients: the LOAD stage zero-fills)
__device__ __forceinline__ void st_ln_bwd(const MbvRowStage& s, float* slots, float* red, int row0, int rows) {
  const float* G = slots + s.src * (RC_ROWS * RC_LD);
  const float* S = slots + s.src2 * (RC_ROWS * RC_LD);
  float* D = slots + s.dst * (RC_ROWS * RC_LD);
  const int r = threadIdx.x >> 5, l = threadIdx.x & 31, n = s.n;
  const float* gamma = reinterpret_cast<const float*>(s.p0);
  const bool live = row0 + r < rows;
  float mean = 0.f, rstd = 0.f;
  if (live) {
    const float* st = reinterpret_cast<const float*>(s.p2) + (int64_t)(row0 + r) * 2;
    mean = st[0];
    rstd = st[1];
  }
  float gw[RC_MAXC / 32], xh[RC_MAXC / 32], gr[RC_MAXC / 32];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    gw[j] = xh[j] = gr[j] = 0.f;
    if (c < n && live) {
      const float g = G[r * RC_LD + c];
      xh[j] = (S[r * RC_LD + c] - mean) * rstd;
      gr[j] = g;
      gw[j] = g * gamma[c];
      s1 += gw[j];
      s2 += gw[j] * xh[j];
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  const float inv = 1.f / (float)n;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    if (c < n) D[r * RC_LD + c] = live ? rstd * (gw[j] - inv * s1 - xh[j] * inv * s2) : 0.f;
  }
  // parameter gradients: column sums over this block's 16 rows through an LDS image [2][16][n] reused from `red`
  if (s.p1) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RC_MAXC / 32; ++j) {
      const int c = l + 32 * j;
      if (c < n) {
        red[r * RC_MAXC + c] = gr[j] * xh[j];
        red[RC_ROWS * RC_MAXC + r * RC_MAXC + c] = gr[j];
      }
    }
    __syncthreads();
    float* out = reinterpret_cast<float*>(const_cast<void*>(s.p1)) + (int64_t)blockIdx.x * 2 * n;
    for (int i = threadIdx.x; i < 2 * n; i += RC_NT) {
      const int which = i / n, c = i - which * n;
      float a = 0.f;
#pragma unroll
      for (int rr = 0; rr < RC_ROWS; ++rr) a += red[which * RC_ROWS * RC_MAXC + rr * RC_MAXC + c];
      out[i] = a;
    }
  }
}

__device__ __forceinline__ void st_add(const MbvRowStage& s, float* slots) {
  const float* A = slots + s.src * (RC_ROWS * RC_LD);
  const float* B = slots + s.src2 * (RC_ROWS * RC_LD);
  float* Y = slots + s.dst * (RC_ROWS * RC_LD);
  const int n4 = s.n >> 2;
  for (int i = threadIdx.x; i < RC_ROWS * n4; i += RC_NT) {
    const int r = i / n4, c = (i - r * n4) * 4;
    const float4 a = *reinterpret_cast<const float4*>(A + r * RC_LD + c);
    const float4 b = *reinterpret_cast<const float4*>(B + r * RC_LD + c);
    *reinterpret_cast<float4*>(Y + r * RC_LD + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// p0[block * ld + c] = sum over the block's rows of src[r][c]  (rows beyond `rows` are zero by construction)
__device__ __forceinline__ void st_colsum(const MbvRowStage& s, const float* slots) {
  const float* A = slots + s.src * (RC_ROWS * RC_LD);
  float* out = reinterpret_cast<float*>(const_cast<void*>(s.p0)) + (int64_t)blockIdx.x * s.ld;
  for (int c = threadIdx.x; c < s.n; c += RC_NT) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < RC_ROWS; ++r) a += A[r * RC_LD + c];
    out[c] = a;
  }
}

template <int WDT>
__global__ void __launch_bounds__(RC_NT) k_rowchain(const RowProgram P) {
  extern __shared__ __attribute__((aligned(16))) float rc_lds[];
  float* slots = rc_lds;
  float* red = rc_lds + RC_SLOTS * RC_ROWS * RC_LD;       // [2][16][256] scratch of the LayerNorm backward
  const int row0 = blockIdx.x * RC_ROWS;
  for (int i = 0; i < P.num_stages; ++i) {
    const MbvRowStage& s = P.st[i];
    switch (s.op) {
      case MBV_RC_LOAD: st_load(s, slots, row0, P.rows, P.q_mod); break;
      case MBV_RC_STORE: st_store(s, slots, row0, P.rows); break;
      case MBV_RC_GEMM: st_gemm<WDT>(s, slots); break;
      case MBV_RC_LN: st_ln(s, slots, row0, P.rows, P.eps); break;
      case MBV_RC_LN_BWD: st_ln_bwd(s, slots, red, row0, P.rows); break;
      case MBV_RC_ADD: st_add(s, slots); break;
      case MBV_RC_COLSUM: st_colsum(s, slots); break;
      default: break;
    }
    __syncthreads();
  }
}

// ---- grouped 2-D transpose (the decoder's weight matrices, once per step) ----------------------------------------
struct TrEntry {
  const void* src;
  void* dst;
  int rows, cols;          // src is (rows, cols) row-major, dst (cols, rows)
  int tile_begin;          // first 64 x 64 tile of this entry in the launch
  int tiles_c;
};
struct TrArgs {
  int n, esize;
  TrEntry e[MBV_TR_MAX];
};

template <typename T>
__global__ void __launch_bounds__(256) k_transpose_group(const TrArgs A) {
  __shared__ T tile[64][65];
  int ei = 0;
  for (int i = 1; i < A.n; ++i)
    if ((int)blockIdx.x >= A.e[i].tile_begin) ei = i;
  const TrEntry& e = A.e[ei];
  const int t = (int)blockIdx.x - e.tile_begin;
  const int tr = t / e.tiles_c, tc = t - tr * e.tiles_c;
  const T* src = reinterpret_cast<const T*>(e.src);
  T* dst = reinterpret_cast<T*>(e.dst);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = tr * 64 + j, c = tc * 64 + tx;
    if (r < e.rows && c < e.cols) tile[j][tx] = src[(int64_t)r * e.cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = tc * 64 + j, r = tr * 64 + tx;
    if (r < e.rows && c < e.cols) dst[(int64_t)c * e.rows + r] = tile[tx][j];
  }
}

}  // namespace

extern "C" int mbv_rowchain_max_stages(void) { return RC_MAX_STAGES; }
extern "C" int mbv_rowchain_slots(void) { return RC_SLOTS; }

extern "C" int mbv_rowchain_run(const MbvRowStage* stages, int32_t num_stages, int32_t rows, int32_t q_mod, float eps,
                                int32_t wdtype, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!stages || num_stages <= 0 || num_stages > RC_MAX_STAGES || rows <= 0 || q_mod < 0) return MBV_ERR_BAD_ARG;
  if (wdtype != MBV_DT_F32 && wdtype != MBV_DT_BF16 && wdtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  RowProgram P;
  P.num_stages = num_stages; P.rows = rows; P.q_mod = q_mod; P.wdtype = wdtype; P.eps = eps;
  P.pad[0] = P.pad[1] = P.pad[2] = 0;
  const int kq = wdtype == MBV_DT_F32 ? 16 : 32;          // reduction granularity of the MFMA k blocks
  const int wes = wdtype == MBV_DT_F32 ? 4 : 2;
  for (int i = 0; i < num_stages; ++i) {
    const MbvRowStage& s = stages[i];
    const int dt = s.flags & 3;
    auto slot_ok = [](int v) { return v >= 0 && v < RC_SLOTS; };
    switch (s.op) {
      case MBV_RC_LOAD:
        if (!slot_ok(s.dst) || (!s.p0 && (!slot_ok(s.src) || !s.p1)) || s.n <= 0 || s.n > RC_MAXC || (s.n & 3) || dt > 2 ||
            (s.p0 && ((s.ld & 3) || (reinterpret_cast<size_t>(s.p0) & (dt == MBV_DT_F32 ? 15 : 7)))) ||
            (s.p1 && ((reinterpret_cast<size_t>(s.p1) & 15) || (s.ld2 & 3))))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_STORE:
        if (!slot_ok(s.src) || !s.p0 || s.n <= 0 || s.n > RC_MAXC || dt > 2 || ((s.flags & MBV_RC_ACCUM) && dt != MBV_DT_F32))
          return MBV_ERR_BAD_ARG;
        if ((s.n & 3) == 0 && ((s.ld & 3) || (reinterpret_cast<size_t>(s.p0) & (dt == MBV_DT_F32 ? 15 : 7))))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_GEMM:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || s.dst == s.src || !s.p0 || s.n <= 0 || s.n > RC_MAXC || s.k <= 0 ||
            s.k > RC_MAXC || (s.k % kq) || ((int64_t)s.ld * wes) % 16 || (reinterpret_cast<size_t>(s.p0) & 15) ||
            ((s.flags & MBV_RC_MASK) && (!slot_ok(s.src2) || s.src2 == s.dst)))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_LN:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || (s.src2 >= 0 && !slot_ok(s.src2)) || !s.p0 || !s.p1 || s.n <= 0 ||
            s.n > RC_MAXC)
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_LN_BWD:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || !slot_ok(s.src2) || !s.p0 || !s.p2 || s.n <= 0 || s.n > RC_MAXC)
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_ADD:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || !slot_ok(s.src2) || s.n <= 0 || s.n > RC_MAXC || (s.n & 3))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_COLSUM:
        if (!slot_ok(s.src) || !s.p0 || s.n <= 0 || s.n > RC_MAXC || s.ld < s.n) return MBV_ERR_BAD_ARG;
        break;
      default:
        return MBV_ERR_BAD_ARG;
    }
    P.st[i] = s;
  }
  const size_t lds = (size_t)(RC_SLOTS * RC_ROWS * RC_LD + 2 * RC_ROWS * RC_MAXC) * sizeof(float);
  static bool attr_done = false;       // idempotent attribute of the code objects, not library state
  if (!attr_done) {
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowchain<MBV_DT_F32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowchain<MBV_DT_BF16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowchain<MBV_DT_F16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const dim3 grid((unsigned)((rows + RC_ROWS - 1) / RC_ROWS)), block(RC_NT);
  if (wdtype == MBV_DT_F32)
    hipLaunchKernelGGL(k_rowchain<MBV_DT_F32>, grid, block, lds, stream, P);
  else if (wdtype == MBV_DT_BF16)
    hipLaunchKernelGGL(k_rowchain<MBV_DT_BF16>, grid, block, lds, stream, P);
  else
    hipLaunchKernelGGL(k_rowchain<MBV_DT_F16>, grid, block, lds, stream, P);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_transpose_group(const void* const* src, void* const* dst, const int32_t* rows, const int32_t* cols,
                                   int32_t n, int32_t elem_size, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !dst || !rows || !cols || n < 0 || (elem_size != 2 && elem_size != 4)) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < n; base += MBV_TR_MAX) {
    TrArgs A;
    A.n = n - base < MBV_TR_MAX ? n - base : MBV_TR_MAX;
    A.esize = elem_size;
    int tiles = 0;
    for (int i = 0; i < A.n; ++i) {
      const int j = base + i;
      if (!src[j] || !dst[j] || rows[j] <= 0 || cols[j] <= 0) return MBV_ERR_BAD_ARG;
      A.e[i].src = src[j]; A.e[i].dst = dst[j]; A.e[i].rows = rows[j]; A.e[i].cols = cols[j];
      A.e[i].tile_begin = tiles;
      A.e[i].tiles_c = (cols[j] + 63) / 64;
      tiles += A.e[i].tiles_c * ((rows[j] + 63) / 64);
    }
    if (elem_size == 2)
      hipLaunchKernelGGL(k_transpose_group<unsigned short>, dim3((unsigned)tiles), dim3(256), 0, stream, A);
    else
      hipLaunchKernelGGL(k_transpose_group<float>, dim3((unsigned)tiles), dim3(256), 0, stream, A);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}
