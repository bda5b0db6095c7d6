// K12 — fused residual-add + LayerNorm over the channel axis of token-major activations, forward and backward.
//
// Replaces the `x + f(x)` → `nn.LayerNorm(C)` pairs of the Swin blocks (mask_bev/models/networks/swin/swin.py:
// 357-377), of the pixel-decoder layers and of the masked-attention decoder layers (post-LN; mmdet layers configured
// at mask_bev/models/head/mask_bev_panoptic_head.py:119-176).  Per (add, LN) pair torch runs an add kernel, the LN
// kernel and — under autocast — a cast of the LN output forward, and the LN input-gradient kernel, two
// gamma/beta-gradient kernels, two gradient-accumulation adds and the residual gradient add backward; here it is
// one kernel forward and two backward (the second a small column reduction), the normalised output is written
// directly in the consumer's dtype, and d(gamma), d(beta) land in the parameter arena's gradient by accumulation.
// HBM-bound: one wave per row, 16-byte accesses, every tensor touched once.
//
//   fwd:  s = a (+ b);  mean, rstd over C;  y = (s - mean) * rstd * gamma + beta
//   bwd:  g = dy * gamma;  dx = rstd * (g - mean_C(g) - xhat * mean_C(g * xhat)) (+ ds);  dgamma += Σ_rows dy * xhat;
//         dbeta += Σ_rows dy
#include "common.hpp"

namespace {

__device__ __forceinline__ float f16_to_f32(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
__device__ __forceinline__ unsigned f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }

// 4 consecutive channels of a row; `kind` = MBV_DT_F32 / MBV_DT_BF16 / MBV_DT_F16 storage (wave-uniform)
__device__ __forceinline__ float4 load4(const void* base, int kind, long elem) {
  if (kind == MBV_DT_F32) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
  const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
  if (kind == MBV_DT_BF16)
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  return make_float4(f16_to_f32(u.x & 0xffffu), f16_to_f32(u.x >> 16), f16_to_f32(u.y & 0xffffu), f16_to_f32(u.y >> 16));
}
__device__ __forceinline__ void store4(void* base, int kind, long elem, float4 v) {
  if (kind == MBV_DT_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem) = v;
    return;
  }
  uint2 u;
  if (kind == MBV_DT_BF16) {
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
  } else {
    u.x = f32_to_f16(v.x) | (f32_to_f16(v.y) << 16);
    u.y = f32_to_f16(v.z) | (f32_to_f16(v.w) << 16);
  }
  *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + elem) = u;
}

// Patch-merging rows (mmdet PatchMerging, mask_bev/models/networks/swin/swin.py:611-616: nn.Unfold(2, stride 2) of a
// (B, H, W, C) token map, channel order c * 4 + kh * 2 + kw): row (b, oh, ow) of 4 C elements is gathered from the
// four pixels (2 oh + kh, 2 ow + kw) of the f32 map — element 4 v + k of the row is channel v of pixel k — so the
// unfolded copy (and its scattered 4-byte writes) never exists; the backward scatters dx the same way.
struct MergeGeom {
  int on;                            // 0: plain rows
  int C, W, OW, OHW;                 // input channels, input width, output width, output pixels per sample
  long sample;                       // H * W * C
};
__device__ __forceinline__ long merge_base(const MergeGeom& m, long row, int v) {
  const long b = row / m.OHW;
  const int r = (int)(row - b * m.OHW), oh = r / m.OW, ow = r - oh * m.OW;
  return b * m.sample + ((long)(2 * oh) * m.W + 2 * ow) * m.C + v;
}
__device__ __forceinline__ float4 merge_load(const float* x, const MergeGeom& m, long row, int v) {
  const float* p = x + merge_base(m, row, v);
  return make_float4(p[0], p[m.C], p[(long)m.W * m.C], p[(long)m.W * m.C + m.C]);
}
__device__ __forceinline__ void merge_store(float* x, const MergeGeom& m, long row, int v, float4 d) {
  float* p = x + merge_base(m, row, v);
  p[0] = d.x; p[m.C] = d.y; p[(long)m.W * m.C] = d.z; p[(long)m.W * m.C + m.C] = d.w;
}

struct LnIo {
  const void* a; const void* b;      // inputs (b nullable)
  int a_bf16, b_bf16;
  const float* gamma; const float* beta;
  float* s;                          // the sum, f32 (saved for the backward; nullable only when b == nullptr && !a_bf16)
  void* y; int y_bf16;
  float* mean; float* rstd;
  MergeGeom mg;                      // mg.on: a is the f32 (B, H, W, C) map the rows are gathered from (b, s null)
  long b_mod;                        // > 0: b has b_mod rows and repeats (row % b_mod): a per-sample map added to every sample
  void* y2; int y2_kind;             // optional second copy of y in another storage type (post-LN: f32 for the next residual
                                     // add, 16 bits for the branch GEMM that follows): no cast launch in between
};

// ITERS float4 per lane: C <= 256 * ITERS, C % 4 == 0
template <int ITERS>
__global__ void __launch_bounds__(256) k_add_ln_fwd(LnIo io, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C >> 2;
  const long base = row * C;
  float4 x[ITERS];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int v = lane + 64 * i;
    x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < nvec) {
      x[i] = io.mg.on ? merge_load(reinterpret_cast<const float*>(io.a), io.mg, row, v)
                      : load4(io.a, io.a_bf16, base + 4 * v);
      if (io.b) {
        const float4 t = load4(io.b, io.b_bf16, (io.b_mod ? (row % io.b_mod) * C : base) + 4 * v);
        x[i].x += t.x; x[i].y += t.y; x[i].z += t.z; x[i].w += t.w;
      }
      if (io.s) *reinterpret_cast<float4*>(io.s + base + 4 * v) = x[i];
      sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      const float d0 = x[i].x - mean, d1 = x[i].y - mean, d2 = x[i].z - mean, d3 = x[i].w - mean;
      sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  if (lane == 0) {
    io.mean[row] = mean;
    io.rstd[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      const float4 g = *reinterpret_cast<const float4*>(io.gamma + 4 * v);
      const float4 be = *reinterpret_cast<const float4*>(io.beta + 4 * v);
      float4 o;
      o.x = (x[i].x - mean) * rstd * g.x + be.x;
      o.y = (x[i].y - mean) * rstd * g.y + be.y;
      o.z = (x[i].z - mean) * rstd * g.z + be.z;
      o.w = (x[i].w - mean) * rstd * g.w + be.w;
      store4(io.y, io.y_bf16, base + 4 * v, o);
      if (io.y2) store4(io.y2, io.y2_kind, base + 4 * v, o);
    }
  }
}

struct LnBwdIo {
  const void* dy; int dy_bf16;
  const void* dy2; int dy2_kind;     // optional second gradient of y (y fanned out to two consumers): dy + dy2 on load
  const void* ds; int ds_bf16;       // gradient arriving at the sum from the residual path (nullable)
  const float* s; const float* mean; const float* rstd; const float* gamma;
  float* dx;                         // f32 gradient of the sum (= of a and of b)
  void* dx_lo; int dx_lo_kind;       // optional 16-bit copy of dx for a 16-bit branch input (nullable), its storage kind
  float* partial;                    // (gridDim.x, NP, C) per-block Σ dy*xhat, Σ dy [, Σ dx]; NULL = few blocks:
                                     // add straight into dgamma / dbeta / dbranch (no reduction launch)
  float* dgamma; float* dbeta; float* dbranch;
  int np;                            // 2, or 3 when the column sums of dx are wanted too (the bias gradient of
                                     // the Linear that produced the residual branch: d(branch) = dx)
  MergeGeom mg;                      // mg.on: s is the f32 (B, H, W, C) map of the forward, dx has its layout
  unsigned* amax_dx;                 // optional absmax record (64 words) of dx: one max-combine per block (fp32 compute: the
                                     // Linear backward that takes dx as its output gradient runs on K20, csrc/gemm_f32s.hip)
};

// Wide rows (ITERS >= 4, C > 512) add their column sums to the block's LDS accumulators ROW BY ROW instead of keeping
// three register accumulators per channel across the rows of a wave: few rows reach such a wave (4 096 x 768 at Swin
// stage 3: one or two), the 36-72 f64 LDS adds per row cost ~0.2 us of a shared pipe, and the 48-96 VGPRs they free
// let two workgroups share a CU (ITERS = 4: 164 -> <= 128 VGPRs) — the kernel is latency-bound at these sizes.
template <int ITERS>
__global__ void __launch_bounds__(512, (ITERS == 4 ? 4 : 1)) k_add_ln_bwd(LnBwdIo io, long rows, int C) {
  constexpr bool ROW_LDS = ITERS >= 4;
  extern __shared__ double red[];    // [np][C]; f64: LDS ds_add_f32 is ≈ 20x slower than ds_add_f64 on gfx950
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 2;
  const int np = io.np;
  for (int i = threadIdx.x; i < np * C; i += 512) red[i] = 0.0;
  __syncthreads();
  float4 dg[ITERS], db[ITERS], dxs[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) dg[i] = db[i] = dxs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dmax = 0.f;
  float4 gam[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int v = lane + 64 * i;
    gam[i] = v < nvec ? *reinterpret_cast<const float4*>(io.gamma + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long row = (long)blockIdx.x * 8 + wave; row < rows; row += (long)gridDim.x * 8) {
    const long base = row * C;
    const float mean = io.mean[row], rstd = io.rstd[row];
    float4 g[ITERS], xh[ITERS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int v = lane + 64 * i;
      g[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < nvec) {
        float4 dy = load4(io.dy, io.dy_bf16, base + 4 * v);
        if (io.dy2) {
          const float4 t = load4(io.dy2, io.dy2_kind, base + 4 * v);
          dy.x += t.x; dy.y += t.y; dy.z += t.z; dy.w += t.w;
        }
        const float4 sv = io.mg.on ? merge_load(io.s, io.mg, row, v) : *reinterpret_cast<const float4*>(io.s + base + 4 * v);
        xh[i].x = (sv.x - mean) * rstd; xh[i].y = (sv.y - mean) * rstd;
        xh[i].z = (sv.z - mean) * rstd; xh[i].w = (sv.w - mean) * rstd;
        if (ROW_LDS) {
          atomicAdd(&red[4 * v + 0], (double)(dy.x * xh[i].x)); atomicAdd(&red[4 * v + 1], (double)(dy.y * xh[i].y));
          atomicAdd(&red[4 * v + 2], (double)(dy.z * xh[i].z)); atomicAdd(&red[4 * v + 3], (double)(dy.w * xh[i].w));
          atomicAdd(&red[C + 4 * v + 0], (double)dy.x); atomicAdd(&red[C + 4 * v + 1], (double)dy.y);
          atomicAdd(&red[C + 4 * v + 2], (double)dy.z); atomicAdd(&red[C + 4 * v + 3], (double)dy.w);
        } else {
          dg[i].x += dy.x * xh[i].x; dg[i].y += dy.y * xh[i].y; dg[i].z += dy.z * xh[i].z; dg[i].w += dy.w * xh[i].w;
          db[i].x += dy.x; db[i].y += dy.y; db[i].z += dy.z; db[i].w += dy.w;
        }
        g[i].x = dy.x * gam[i].x; g[i].y = dy.y * gam[i].y; g[i].z = dy.z * gam[i].z; g[i].w = dy.w * gam[i].w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
    }
    const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
        float4 d;
        d.x = rstd * (g[i].x - c1 - xh[i].x * c2);
        d.y = rstd * (g[i].y - c1 - xh[i].y * c2);
        d.z = rstd * (g[i].z - c1 - xh[i].z * c2);
        d.w = rstd * (g[i].w - c1 - xh[i].w * c2);
        if (io.ds) {
          const float4 t = load4(io.ds, io.ds_bf16, base + 4 * v);
          d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        }
        if (io.mg.on) merge_store(io.dx, io.mg, row, v, d);
        else *reinterpret_cast<float4*>(io.dx + base + 4 * v) = d;
        dmax = fmaxf(dmax, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
        if (io.dx_lo) store4(io.dx_lo, io.dx_lo_kind, base + 4 * v, d);
        if (ROW_LDS) {
          if (np == 3) {
            atomicAdd(&red[2 * C + 4 * v + 0], (double)d.x); atomicAdd(&red[2 * C + 4 * v + 1], (double)d.y);
            atomicAdd(&red[2 * C + 4 * v + 2], (double)d.z); atomicAdd(&red[2 * C + 4 * v + 3], (double)d.w);
          }
        } else {
          dxs[i].x += d.x; dxs[i].y += d.y; dxs[i].z += d.z; dxs[i].w += d.w;
        }
      }
    }
  }
  // the 8 waves' column sums meet in LDS, then one (2, C) row of partials per block
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int v = lane + 64 * i;
    if (!ROW_LDS && v < nvec) {
      atomicAdd(&red[4 * v + 0], (double)dg[i].x); atomicAdd(&red[4 * v + 1], (double)dg[i].y);
      atomicAdd(&red[4 * v + 2], (double)dg[i].z); atomicAdd(&red[4 * v + 3], (double)dg[i].w);
      atomicAdd(&red[C + 4 * v + 0], (double)db[i].x); atomicAdd(&red[C + 4 * v + 1], (double)db[i].y);
      atomicAdd(&red[C + 4 * v + 2], (double)db[i].z); atomicAdd(&red[C + 4 * v + 3], (double)db[i].w);
      if (np == 3) {
        atomicAdd(&red[2 * C + 4 * v + 0], (double)dxs[i].x); atomicAdd(&red[2 * C + 4 * v + 1], (double)dxs[i].y);
        atomicAdd(&red[2 * C + 4 * v + 2], (double)dxs[i].z); atomicAdd(&red[2 * C + 4 * v + 3], (double)dxs[i].w);
      }
    }
  }
  __syncthreads();
  if (io.amax_dx) {                   // the waves' maxima meet in LDS: ONE no-return atomic per block
    __shared__ float wmax[8];
    dmax = wave_max(dmax);
    if (lane == 0) wmax[wave] = dmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = wmax[0];
      for (int i = 1; i < 8; ++i) m = fmaxf(m, wmax[i]);
      const unsigned bits = __float_as_uint(m) & 0x7fffffffu;
      if (bits) atomicMax(io.amax_dx + (blockIdx.x & 63), bits);
    }
  }
  if (io.partial) {
    float* prow = io.partial + (long)blockIdx.x * np * C;
    for (int i = threadIdx.x; i < np * C; i += 512) prow[i] = (float)red[i];
  } else {
    for (int i = threadIdx.x; i < np * C; i += 512) {
      float* dst = i < C ? io.dgamma + i : (i < 2 * C ? io.dbeta + (i - C) : io.dbranch + (i - 2 * C));
      atomicAdd(dst, (float)red[i]);
    }
  }
}

// dgamma[c] += Σ_blocks partial[blk][0][c], dbeta likewise (and the branch-bias gradient when np == 3).  Block = 64 columns x 4 slices over one chunk of 64
// partial rows (blockIdx.y); chunks meet through f32 atomics (≤ 16 adds per address), the destination is cleared
// by the caller when it is not accumulated into.
__global__ void __launch_bounds__(256) k_ln_param_reduce(const float* __restrict__ partial, int nblk, int C, int np,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ dbranch) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);     // over np*C columns: gamma | beta | branch bias
  const int slice = threadIdx.x >> 6;
  const int b0 = blockIdx.y * 64, b1 = min(nblk, b0 + 64);
  const int ncol = np * C;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  if (col < ncol) {
    int b = b0 + slice;
    for (; b + 12 < b1; b += 16) {
      acc0 += partial[(long)b * ncol + col];
      acc1 += partial[(long)(b + 4) * ncol + col];
      acc2 += partial[(long)(b + 8) * ncol + col];
      acc3 += partial[(long)(b + 12) * ncol + col];
    }
    for (; b < b1; b += 4) acc0 += partial[(long)b * ncol + col];
  }
  red[slice][threadIdx.x & 63] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (slice == 0 && col < ncol) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    float* dst = col < C ? dgamma + col : (col < 2 * C ? dbeta + (col - C) : dbranch + (col - 2 * C));
    atomicAdd(dst, t);
  }
}

int iters_for(int C) {
  if (C <= 0 || (C & 3)) return 0;
  if (C <= 256) return 1;
  if (C <= 512) return 2;
  if (C <= 1024) return 4;
  if (C <= 2048) return 8;
  if (C <= 3072) return 12;        // patch merging in front of Swin stage 4: 4 x 768 channels (round 4; it fell back to ATen)
  return 0;
}

// acc[c][r] += sum_b g[b][r][c]: the gradient of a per-sample (R, C) token map that was added to every sample, accumulated
// into the parameter it is a transposed view of ((1, C, H, W) absolute position embedding: swin.py:579-586 / :750-760).
// 32 x 32 tiles through LDS: reads run along c, writes along r.
template <typename T>
__global__ void __launch_bounds__(256) k_transposed_batch_sum(const T* __restrict__ g, int batch, long R, int C,
                                                              float* __restrict__ acc) {
  __shared__ float tile[32][33];
  const long r0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < batch; ++b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long r = r0 + ty + 8 * k;
      const int c = c0 + tx;
      if (r < R && c < C) {
        const T v = g[((long)b * R + r) * C + c];
        s[k] += v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = s[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k;
    const long r = r0 + tx;
    if (r < R && c < C) acc[(long)c * R + r] += tile[tx][ty + 8 * k];
  }
}

}  // namespace

// g (batch, R, C) f32 → acc (C, R) f32 += the batch sum, transposed.
extern "C" int mbv_transposed_batch_sum_accum(const float* g, int32_t batch, int64_t R, int32_t C, float* acc, void* stream) {
  if (batch <= 0 || R <= 0 || C <= 0) return MBV_ERR_BAD_ARG;
  if (!g || !acc) return MBV_ERR_BAD_ARG;
  if ((R + 31) / 32 > 0x7fffffffL || (C + 31) / 32 > 65535) return MBV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_transposed_batch_sum<float>, dim3((unsigned)((R + 31) / 32), (unsigned)((C + 31) / 32)), dim3(256), 0,
                     (hipStream_t)stream, g, batch, (long)R, C, acc);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// (the add + LayerNorm entry points stop at 2048 channels: with three partial sums per channel — dgamma, dbeta, dbranch —
// 3072 channels would need 72 KB of LDS accumulators; the merging form has two)
extern "C" int mbv_add_layernorm_supported(int32_t C) { return (C <= 2048 && iters_for(C)) ? 1 : 0; }

// one 512-thread block per CU for the widest rows, two for C <= 1024 (per-row LDS accumulation), up to four for narrow ones
extern "C" int64_t mbv_add_layernorm_bwd_blocks(int64_t rows, int32_t C) {
  const int it = iters_for(C);
  const int64_t cap = it <= 2 ? 1024 : (it == 4 ? 512 : 256);
  int64_t b = (rows + 7) / 8;
  return b < 1 ? 1 : (b > cap ? cap : b);
}

// 1 = the backward adds dgamma / dbeta from inside its one kernel (few rows); 0 = it leaves per-block partial rows
extern "C" int mbv_add_layernorm_bwd_direct(int64_t rows, int32_t C) {
  return mbv_add_layernorm_bwd_blocks(rows, C) <= 64 ? 1 : 0;
}

static int add_ln_fwd_launch(const LnIo& io, int it, int64_t rows, int32_t C, float eps, hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  switch (it) {
    case 1: hipLaunchKernelGGL(k_add_ln_fwd<1>, grid, block, 0, st, io, (long)rows, C, eps); break;
    case 2: hipLaunchKernelGGL(k_add_ln_fwd<2>, grid, block, 0, st, io, (long)rows, C, eps); break;
    case 4: hipLaunchKernelGGL(k_add_ln_fwd<4>, grid, block, 0, st, io, (long)rows, C, eps); break;
    case 8: hipLaunchKernelGGL(k_add_ln_fwd<8>, grid, block, 0, st, io, (long)rows, C, eps); break;
    default: hipLaunchKernelGGL(k_add_ln_fwd<12>, grid, block, 0, st, io, (long)rows, C, eps); break;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_add_layernorm_fwd2(const void* a, int32_t a_bf16, const void* b, int32_t b_bf16, int64_t b_rows,
                                      const float* gamma, const float* beta, int64_t rows, int32_t C, float eps,
                                      float* sum_out, void* y, int32_t y_bf16, void* y2, int32_t y2_dtype, float* mean,
                                      float* rstd, void* stream);

extern "C" int mbv_add_layernorm_fwd(const void* a, int32_t a_bf16, const void* b, int32_t b_bf16, const float* gamma,
                                     const float* beta, int64_t rows, int32_t C, float eps, float* sum_out, void* y,
                                     int32_t y_bf16, float* mean, float* rstd, void* stream) {
  return mbv_add_layernorm_fwd2(a, a_bf16, b, b_bf16, 0, gamma, beta, rows, C, eps, sum_out, y, y_bf16, nullptr, 0, mean,
                                rstd, stream);
}

extern "C" int mbv_add_layernorm_fwd2(const void* a, int32_t a_bf16, const void* b, int32_t b_bf16, int64_t b_rows,
                                      const float* gamma, const float* beta, int64_t rows, int32_t C, float eps,
                                      float* sum_out, void* y, int32_t y_bf16, void* y2, int32_t y2_dtype, float* mean,
                                      float* rstd, void* stream) {
  const int it = C <= 2048 ? iters_for(C) : 0;
  if (b_rows < 0 || (b_rows > 0 && (!b || rows % b_rows))) return MBV_ERR_BAD_ARG;
  if (!it) return MBV_ERR_UNSUPPORTED;
  if (rows < 0) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!a || !gamma || !beta || !y || !mean || !rstd) return MBV_ERR_BAD_ARG;
  if (!sum_out && (b || a_bf16)) return MBV_ERR_BAD_ARG;       // the backward needs the f32 LN input
  LnIo io{a, b, a_bf16, b_bf16, gamma, beta, sum_out, y, y_bf16, mean, rstd, MergeGeom{}, b_rows == rows ? 0 : b_rows, y2,
          y2_dtype};
  return add_ln_fwd_launch(io, it, rows, C, eps, (hipStream_t)stream);
}

// The backward of `rows` rows of C elements: one kernel, then (many rows) the reduction of the per-block partial rows of
// the parameter gradients unless the caller defers it.
static int add_ln_bwd_launch(LnBwdIo io, int it, int64_t rows, int32_t C, int32_t accumulate, float* partial_ws,
                             int32_t defer_reduce, hipStream_t st) {
  const int np = io.np;
  const int nblk = (int)mbv_add_layernorm_bwd_blocks(rows, C);
  const bool direct = nblk <= 64;          // few rows (the decoder's B*Q tokens): ≤ 64 adds per address, one launch
  io.partial = direct ? nullptr : partial_ws;
  if (direct && !accumulate) {
    MBV_CHECK_HIP(mbv_fill_async(io.dgamma, 0, (size_t)C * 4, st));
    MBV_CHECK_HIP(mbv_fill_async(io.dbeta, 0, (size_t)C * 4, st));
  }
  const dim3 grid(nblk), block(512);
  const size_t lds = (size_t)np * C * sizeof(double);
  switch (it) {
    case 1: hipLaunchKernelGGL(k_add_ln_bwd<1>, grid, block, lds, st, io, (long)rows, C); break;
    case 2: hipLaunchKernelGGL(k_add_ln_bwd<2>, grid, block, lds, st, io, (long)rows, C); break;
    case 4: hipLaunchKernelGGL(k_add_ln_bwd<4>, grid, block, lds, st, io, (long)rows, C); break;
    case 8: hipLaunchKernelGGL(k_add_ln_bwd<8>, grid, block, lds, st, io, (long)rows, C); break;
    default: hipLaunchKernelGGL(k_add_ln_bwd<12>, grid, block, lds, st, io, (long)rows, C); break;
  }
  MBV_CHECK_LAUNCH();
  if (direct) return MBV_OK;
  // defer_reduce (accumulating callers only): the (nblk, np, C) partial rows stay in partial_ws and the caller adds
  // their column sums later — mbv_colsum_accum_group with rows = nblk, ld = np * C, one entry per parameter
  if (defer_reduce && accumulate) return MBV_OK;
  if (!accumulate) {
    MBV_CHECK_HIP(mbv_fill_async(io.dgamma, 0, (size_t)C * 4, st));
    MBV_CHECK_HIP(mbv_fill_async(io.dbeta, 0, (size_t)C * 4, st));
  }
  hipLaunchKernelGGL(k_ln_param_reduce, dim3((unsigned)((np * C + 63) / 64), (unsigned)((nblk + 63) / 64)), dim3(256), 0,
                     st, partial_ws, nblk, C, np, io.dgamma, io.dbeta, io.dbranch);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// dy2 (nullable, dy2_dtype): a second gradient of y, added to dy on load — a post-LN output that feeds both the next
// residual add and the next branch hands its two gradients over separately instead of through an autograd add launch
// (bwd3: the same with amax_dx — an optional absmax record, 64 zeroed words, that receives the bits of max|dx|)
extern "C" int mbv_add_layernorm_bwd3(const void* dy, int32_t dy_bf16, const void* dy2, int32_t dy2_dtype, const void* ds,
                                      int32_t ds_bf16, const float* s, const float* mean, const float* rstd,
                                      const float* gamma, int64_t rows, int32_t C, float* dx, void* dx_lo,
                                      int32_t dx_lo_dtype, float* dgamma, float* dbeta, int32_t accumulate,
                                      float* dbranch_bias, float* partial_ws, int32_t defer_reduce, uint32_t* amax_dx,
                                      void* stream) {
  const int it = C <= 2048 ? iters_for(C) : 0;
  if (!it) return MBV_ERR_UNSUPPORTED;
  if (rows < 0) return MBV_ERR_BAD_ARG;
  if (!dgamma || !dbeta) return MBV_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate) {
      MBV_CHECK_HIP(mbv_fill_async(dgamma, 0, (size_t)C * 4, st));
      MBV_CHECK_HIP(mbv_fill_async(dbeta, 0, (size_t)C * 4, st));
    }
    return MBV_OK;
  }
  if (!dy || !s || !mean || !rstd || !gamma || !dx || !partial_ws) return MBV_ERR_BAD_ARG;
  if (dx_lo && dx_lo_dtype != MBV_DT_BF16 && dx_lo_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  LnBwdIo io{dy, dy_bf16, dy2, dy2_dtype, ds, ds_bf16, s, mean, rstd, gamma, dx, dx_lo, dx_lo_dtype, nullptr,
             dgamma, dbeta, dbranch_bias, dbranch_bias ? 3 : 2, MergeGeom{}, amax_dx};
  return add_ln_bwd_launch(io, it, rows, C, accumulate, partial_ws, defer_reduce, st);
}

extern "C" int mbv_add_layernorm_bwd2(const void* dy, int32_t dy_bf16, const void* dy2, int32_t dy2_dtype, const void* ds,
                                      int32_t ds_bf16, const float* s, const float* mean, const float* rstd,
                                      const float* gamma, int64_t rows, int32_t C, float* dx, void* dx_lo,
                                      int32_t dx_lo_dtype, float* dgamma, float* dbeta, int32_t accumulate,
                                      float* dbranch_bias, float* partial_ws, int32_t defer_reduce, void* stream) {
  return mbv_add_layernorm_bwd3(dy, dy_bf16, dy2, dy2_dtype, ds, ds_bf16, s, mean, rstd, gamma, rows, C, dx, dx_lo, dx_lo_dtype,
                                dgamma, dbeta, accumulate, dbranch_bias, partial_ws, defer_reduce, nullptr, stream);
}

extern "C" int mbv_add_layernorm_bwd(const void* dy, int32_t dy_bf16, const void* ds, int32_t ds_bf16, const float* s,
                                     const float* mean, const float* rstd, const float* gamma, int64_t rows, int32_t C,
                                     float* dx, void* dx_lo, int32_t dx_lo_dtype, float* dgamma, float* dbeta,
                                     int32_t accumulate, float* dbranch_bias, float* partial_ws, int32_t defer_reduce,
                                     void* stream) {
  return mbv_add_layernorm_bwd2(dy, dy_bf16, nullptr, 0, ds, ds_bf16, s, mean, rstd, gamma, rows, C, dx, dx_lo,
                                dx_lo_dtype, dgamma, dbeta, accumulate, dbranch_bias, partial_ws, defer_reduce, stream);
}

// ---- patch merging: unfold(2 x 2, stride 2) + LayerNorm(4 C) without the unfolded copy (see MergeGeom) --------------
static bool merge_geom(int64_t batch, int32_t h, int32_t w, int32_t c, MergeGeom& m, int64_t& rows) {
  if (batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (h & 1) || (w & 1)) return false;
  m.on = 1; m.C = c; m.W = w; m.OW = w / 2; m.OHW = (h / 2) * (w / 2);
  m.sample = (long)h * w * c;
  rows = batch * m.OHW;
  return true;
}

extern "C" int mbv_merge_layernorm_supported(int32_t h, int32_t w, int32_t c) {
  return (h > 0 && w > 0 && !(h & 1) && !(w & 1) && c > 0 && iters_for(4 * c)) ? 1 : 0;
}

// y (batch, h/2, w/2, 4 c) = LayerNorm_{4c}(unfold_{2x2}(x (batch, h, w, c) f32)); y 16-bit or f32 (y_dtype);
// mean / rstd (batch * h/2 * w/2) saved for the backward
extern "C" int mbv_merge_layernorm_fwd(const float* x, int64_t batch, int32_t h, int32_t w, int32_t c, const float* gamma,
                                       const float* beta, float eps, void* y, int32_t y_dtype, float* mean, float* rstd,
                                       void* stream) {
  if (batch == 0) return MBV_OK;
  MergeGeom m{};
  int64_t rows = 0;
  if (!merge_geom(batch, h, w, c, m, rows)) return MBV_ERR_UNSUPPORTED;
  const int it = iters_for(4 * c);
  if (!it) return MBV_ERR_UNSUPPORTED;
  if (!x || !gamma || !beta || !y || !mean || !rstd) return MBV_ERR_BAD_ARG;
  LnIo io{x, nullptr, MBV_DT_F32, 0, gamma, beta, nullptr, y, y_dtype, mean, rstd, m};
  return add_ln_fwd_launch(io, it, rows, 4 * c, eps, (hipStream_t)stream);
}

// dx (batch, h, w, c) f32 = the input gradient (every pixel belongs to exactly one row: dx is fully written);
// dgamma / dbeta (4 c) as in mbv_add_layernorm_bwd (accumulate, partial_ws of mbv_add_layernorm_bwd_blocks(rows, 4 c)
// * 2 * 4 c floats, defer_reduce)
extern "C" int mbv_merge_layernorm_bwd(const void* dy, int32_t dy_dtype, const float* x, const float* mean,
                                       const float* rstd, const float* gamma, int64_t batch, int32_t h, int32_t w,
                                       int32_t c, float* dx, float* dgamma, float* dbeta, int32_t accumulate,
                                       float* partial_ws, int32_t defer_reduce, void* stream) {
  if (batch == 0) return MBV_OK;
  MergeGeom m{};
  int64_t rows = 0;
  if (!merge_geom(batch, h, w, c, m, rows)) return MBV_ERR_UNSUPPORTED;
  const int it = iters_for(4 * c);
  if (!it) return MBV_ERR_UNSUPPORTED;
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !partial_ws) return MBV_ERR_BAD_ARG;
  LnBwdIo io{dy, dy_dtype, nullptr, 0, nullptr, 0, x, mean, rstd, gamma, dx, nullptr, 0, nullptr, dgamma, dbeta, nullptr, 2, m};
  return add_ln_bwd_launch(io, it, rows, 4 * c, accumulate, partial_ws, defer_reduce, (hipStream_t)stream);
}
