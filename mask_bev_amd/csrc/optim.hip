// K11 — the parameter-arena kernels of the training step (HBM-bound, no MFMA).
//
//   mbv_adamw_step      one pass of Adam / AdamW over a flat f32 parameter arena: reads param, grad, exp_avg,
//                       exp_avg_sq (16 B/elem), writes param, exp_avg, exp_avg_sq (12 B/elem), optionally the
//                       bf16 shadow the GEMMs read (2 B/elem) and the zeroed gradient (4 B/elem).
//   mbv_colsum_accum    bias gradient: out[n] += sum_t g[t, n], accumulated straight into the f32 arena gradient.
//
// Reference: torch.optim.AdamW / Adam configured at /root/reference mask_bev/mask_bev_module.py:131-166
// (single-tensor update order of torch/optim/adamw.py is followed so that results agree to f32 rounding).
#include "common.hpp"
#include "adam.hpp"

#include <hip/hip_bf16.h>

namespace {

// 4 elements per thread per iteration (float4 loads: 16 B/lane, fully coalesced), grid-stride.
__global__ void __launch_bounds__(256) k_adamw(float* __restrict__ param, float* __restrict__ grad,
                                               float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                               unsigned short* __restrict__ shadow, long n, AdamArgs a) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  if (a.loss_scale) a.grad_scale /= *a.loss_scale;
  if (a.skip && *a.skip) {         // overflowed step: parameters, moments and the shadow stay; the gradient is dropped
    if (a.zero_grad) {
      for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      const long t = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;
      if (t < n) grad[t] = 0.f;
    }
    return;
  }
  if (a.applied) {                 // bias corrections from the device-side count of applied updates
    const double t = (double)(*a.applied + 1);
    a.bias_correction1 = (float)(1.0 - pow((double)a.beta1, t));
    a.bias_correction2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, t));
  }
  // Streaming accesses: every array is read once and written once per step and is far larger than the caches (6.8 GB at the
  // bench size), so loads and stores carry the non-temporal hint — 1.33 -> 1.25 ms measured; a second row per thread in
  // flight was slower (1.37 ms).
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f4 p = __builtin_nontemporal_load(reinterpret_cast<const f4*>(param) + i);
    const f4 g = __builtin_nontemporal_load(reinterpret_cast<const f4*>(grad) + i);
    f4 m = __builtin_nontemporal_load(reinterpret_cast<const f4*>(exp_avg) + i);
    f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(exp_avg_sq) + i);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float pc = p[c], mc = m[c], vc = v[c];
      adam_one(pc, g[c], mc, vc, a);
      p[c] = pc; m[c] = mc; v[c] = vc;
    }
    __builtin_nontemporal_store(p, reinterpret_cast<f4*>(param) + i);
    __builtin_nontemporal_store(m, reinterpret_cast<f4*>(exp_avg) + i);
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(exp_avg_sq) + i);
    if (shadow) {
      us4 s;
      s.x = shadow_bits(p.x, a.shadow_kind); s.y = shadow_bits(p.y, a.shadow_kind);
      s.z = shadow_bits(p.z, a.shadow_kind); s.w = shadow_bits(p.w, a.shadow_kind);
      reinterpret_cast<us4*>(shadow)[i] = s;              // (read again by the next forward: a normal store)
    }
    if (a.zero_grad) __builtin_nontemporal_store(f4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4*>(grad) + i);
  }
  // ragged tail (< 4 elements)
  const long t = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    float p = param[t], m = exp_avg[t], v = exp_avg_sq[t];
    adam_one(p, grad[t], m, v, a);
    param[t] = p; exp_avg[t] = m; exp_avg_sq[t] = v;
    if (shadow) shadow[t] = shadow_bits(p, a.shadow_kind);
    if (a.zero_grad) grad[t] = 0.f;
  }
}

// f32 -> 16-bit shadow refresh of an arena (after load_state_dict / parameter broadcast).
__global__ void __launch_bounds__(256) k_shadow(const float* __restrict__ param, unsigned short* __restrict__ shadow,
                                                long n, int kind) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) shadow[i] = shadow_bits(param[i], kind);
}

// ---- fp16 loss scaling, all on the device (no host synchronisation, replayable) ---------------------------------
// flag |= "some gradient element is inf or nan" (what torch.amp.GradScaler's unscale_ reports as found_inf)
__global__ void __launch_bounds__(256) k_grad_nonfinite(const float* __restrict__ grad, long n, int* __restrict__ flag) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const uint4 u = reinterpret_cast<const uint4*>(grad)[i];
    bad |= ((u.x & 0x7f800000u) == 0x7f800000u) | ((u.y & 0x7f800000u) == 0x7f800000u) |
           ((u.z & 0x7f800000u) == 0x7f800000u) | ((u.w & 0x7f800000u) == 0x7f800000u);
  }
  const long t = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) bad |= (__float_as_uint(grad[t]) & 0x7f800000u) == 0x7f800000u;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// GradScaler.update(): overflow -> scale *= backoff, streak = 0; else streak += 1 and every `interval` clean steps
// scale *= growth.  Clears the flag for the next step.
__global__ void k_loss_scale_update(float* __restrict__ scale, int* __restrict__ streak, int* __restrict__ flag,
                                    float growth, float backoff, int interval, int* __restrict__ applied) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*flag) {
    *scale = fmaxf(*scale * backoff, 1.f);
    *streak = 0;
  } else {
    if (applied) ++*applied;       // this step's update was applied: Adam's step count advances
    if (++*streak >= interval) {
      const float s = *scale * growth;
      if (s <= 3.0e38f && s == s) *scale = s;
      *streak = 0;
    }
  }
  *flag = 0;
}

// Column sums of a row-major (T, N) matrix accumulated into out (N,) f32.
// Block = 4 waves; each wave walks rows r = wave, wave + 4·gridDim.y·…; a lane owns 4 adjacent columns (8 B bf16 /
// 16 B f32 per lane, 256 columns per wave-row).  Partial sums of the 4 waves meet in LDS; one atomicAdd per column
// and block.
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 load4<unsigned short>(const unsigned short* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<unsigned short>(const unsigned short* p) { return __uint_as_float((unsigned)*p << 16); }

template <>
__device__ __forceinline__ float4 load4<_Float16>(const _Float16* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 q = *reinterpret_cast<const h4*>(p);
  return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
}
template <>
__device__ __forceinline__ float load1<_Float16>(const _Float16* p) { return (float)*p; }

// W = threads per row (power of two ≤ 256, each owning 4 adjacent columns); 256 / W rows are read per iteration,
// 4 iterations in flight.  Row slices (gridDim.y) are kept few (≤ 64 per column) — f32 atomics onto the same
// few hundred addresses serialise in L2, measured 58 us at 672 slices vs the 12 us of a two-pass reduction.
template <typename T>
__device__ __forceinline__ void colsum_block(const T* __restrict__ g, long rows, int n, long ld, float* __restrict__ out,
                                             const bool vec, const int W, int bx, int by, int ny, float4* part) {
  const int tid = threadIdx.x;
  const int cg = tid & (W - 1), ro = tid / W, rpi = 256 / W;
  const int c0 = (bx * W + cg) * 4;
  const long rows_per_block = (rows + ny - 1) / ny;
  const long r0 = (long)by * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < n) {
    if (vec) {
      float4 a1 = acc, a2 = acc, a3 = acc;
      long r = r0 + ro;
      for (; r + 3 * rpi < r1; r += 4 * rpi) {
        const float4 x0 = load4<T>(g + r * ld + c0);
        const float4 x1 = load4<T>(g + (r + rpi) * ld + c0);
        const float4 x2 = load4<T>(g + (r + 2 * rpi) * ld + c0);
        const float4 x3 = load4<T>(g + (r + 3 * rpi) * ld + c0);
        acc.x += x0.x; acc.y += x0.y; acc.z += x0.z; acc.w += x0.w;
        a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
        a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
        a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
      }
      for (; r < r1; r += rpi) {
        const float4 x0 = load4<T>(g + r * ld + c0);
        acc.x += x0.x; acc.y += x0.y; acc.z += x0.z; acc.w += x0.w;
      }
      acc.x += a1.x + a2.x + a3.x; acc.y += a1.y + a2.y + a3.y;
      acc.z += a1.z + a2.z + a3.z; acc.w += a1.w + a2.w + a3.w;
    } else {
      for (long r = r0 + ro; r < r1; r += rpi) {
        const T* row = g + r * ld + c0;
        acc.x += load1<T>(row);
        if (c0 + 1 < n) acc.y += load1<T>(row + 1);
        if (c0 + 2 < n) acc.z += load1<T>(row + 2);
        if (c0 + 3 < n) acc.w += load1<T>(row + 3);
      }
    }
  }
  part[tid] = acc;
  __syncthreads();
  if (ro == 0 && c0 < n) {
    float4 s = acc;
    for (int w = 1; w < rpi; ++w) {
      const float4 q = part[w * W + cg];
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    atomicAdd(out + c0, s.x);
    if (c0 + 1 < n) atomicAdd(out + c0 + 1, s.y);
    if (c0 + 2 < n) atomicAdd(out + c0 + 2, s.z);
    if (c0 + 3 < n) atomicAdd(out + c0 + 3, s.w);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_colsum(const T* __restrict__ g, long rows, int n, float* __restrict__ out,
                                                const bool vec, const int W) {
  __shared__ float4 part[256];
  colsum_block<T>(g, rows, n, (long)n, out, vec, W, blockIdx.x, blockIdx.y, gridDim.y, part);
}

// Up to kColsumGroup column sums in ONE launch: bias gradients and LayerNorm-parameter partials (rows of per-block
// partial sums, row stride ld > n) are nobody's input — collected during the backward, issued together at its end.
constexpr int kColsumGroup = 64;
struct ColsumEntry {
  const void* g; float* out;
  int rows, n, ld, kind, W, gx, gy, block_begin, vec;
};
struct ColsumGroupArgs {
  ColsumEntry e[kColsumGroup];
  int n, total_blocks;
};

__global__ void __launch_bounds__(256) k_colsum_group(const ColsumGroupArgs a) {
  __shared__ float4 part[256];
  const int blk = blockIdx.x;
  int i = 0;
  while (i + 1 < a.n && blk >= a.e[i + 1].block_begin) ++i;
  const ColsumEntry& e = a.e[i];
  const int local = blk - e.block_begin, bx = local % e.gx, by = local / e.gx;
  if (e.kind == MBV_DT_F16)
    colsum_block<_Float16>(reinterpret_cast<const _Float16*>(e.g), e.rows, e.n, e.ld, e.out, e.vec != 0, e.W, bx, by, e.gy, part);
  else if (e.kind == MBV_DT_BF16)
    colsum_block<unsigned short>(reinterpret_cast<const unsigned short*>(e.g), e.rows, e.n, e.ld, e.out, e.vec != 0, e.W, bx, by, e.gy, part);
  else
    colsum_block<float>(reinterpret_cast<const float*>(e.g), e.rows, e.n, e.ld, e.out, e.vec != 0, e.W, bx, by, e.gy, part);
}

// Activation backward fused with the bias gradient of the Linear in front of it:
//   gz = ga * act'(z)   (GELU, erf form, or ReLU)      bias_acc[c] += Σ_rows gz[r, c]
// — the column sums are taken from the values the kernel has just computed instead of by a second pass over gz
// (the fc1 layers of the FFNs have the widest outputs of the model: 4C columns).  Same tiling as k_colsum.
template <typename T>
__device__ __forceinline__ void store4t(T* p, float4 v);
template <>
__device__ __forceinline__ void store4t<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <>
__device__ __forceinline__ void store4t<unsigned short>(unsigned short* p, float4 v) {
  uint2 u;
  u.x = pack_bf16x2(v.x, v.y);
  u.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}

template <>
__device__ __forceinline__ void store4t<_Float16>(_Float16* p, float4 v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 q;
  q[0] = (_Float16)v.x; q[1] = (_Float16)v.y; q[2] = (_Float16)v.z; q[3] = (_Float16)v.w;
  *reinterpret_cast<h4*>(p) = q;
}

template <int KIND>   // 0 = ReLU, 1 = GELU (erf)
__device__ __forceinline__ float act_grad(float z, float g) {
  if (KIND == 0) return z > 0.f ? g : 0.f;
  // Phi(z) = 0.5 (1 + erf(z / sqrt 2)) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 in erf): the rational form needs
  // exp(-(z / sqrt 2)^2) = exp(-z^2 / 2) — the very exponential the density term uses — so the whole derivative
  // costs one exp, one rcp and a degree-5 Horner chain instead of libm's erff.
  const float e = __expf(-0.5f * z * z);
  const float az = fabsf(z) * 0.70710678118654752f;
  const float t = mbv_rcp(1.f + 0.3275911f * az);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float half_tail = 0.5f * poly * e;                       // 0.5 erfc(|z| / sqrt 2)
  const float cdf = z >= 0.f ? 1.f - half_tail : half_tail;
  return g * (cdf + z * 0.3989422804014327f * e);
}

template <typename T, int KIND>
__global__ void __launch_bounds__(256) k_act_bwd_colsum(const T* __restrict__ ga, const T* __restrict__ z, long rows,
                                                        int n, T* __restrict__ gz, float* __restrict__ bias_acc,
                                                        const int W) {
  __shared__ float4 part[256];
  const int tid = threadIdx.x;
  const int cg = tid & (W - 1), ro = tid / W, rpi = 256 / W;
  const int c0 = (blockIdx.x * W + cg) * 4;
  const long rows_per_block = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < n) {
    long r = r0 + ro;
    // 4 rows per iteration: 8 loads in flight per thread (one row at a time left the kernel on load latency —
    // 3.9 TB/s — whatever the arithmetic cost)
    for (; r + 3 * rpi < r1; r += 4 * rpi) {
      float4 g[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = load4<T>(ga + (r + u * rpi) * n + c0);
        x[u] = load4<T>(z + (r + u * rpi) * n + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 d;
        d.x = act_grad<KIND>(x[u].x, g[u].x); d.y = act_grad<KIND>(x[u].y, g[u].y);
        d.z = act_grad<KIND>(x[u].z, g[u].z); d.w = act_grad<KIND>(x[u].w, g[u].w);
        store4t<T>(gz + (r + u * rpi) * n + c0, d);
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
      }
    }
    for (; r < r1; r += rpi) {
      const float4 g = load4<T>(ga + r * n + c0);
      const float4 x = load4<T>(z + r * n + c0);
      float4 d;
      d.x = act_grad<KIND>(x.x, g.x); d.y = act_grad<KIND>(x.y, g.y);
      d.z = act_grad<KIND>(x.z, g.z); d.w = act_grad<KIND>(x.w, g.w);
      store4t<T>(gz + r * n + c0, d);
      acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
    }
  }
  part[tid] = acc;
  __syncthreads();
  if (ro == 0 && c0 < n && bias_acc) {
    float4 s = acc;
    for (int w = 1; w < rpi; ++w) {
      const float4 q = part[w * W + cg];
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    atomicAdd(bias_acc + c0, s.x); atomicAdd(bias_acc + c0 + 1, s.y);
    atomicAdd(bias_acc + c0 + 2, s.z); atomicAdd(bias_acc + c0 + 3, s.w);
  }
}

// Small-token weight gradient in exact f32: acc (O, I) += g^T x for g (T, O), x (T, I), T = the decoder's B*Q rows.
// The library's f32 GEMM heuristics pick one 256x256 macro-tile for this shape (measured 98 us per call); here one
// wave owns a 32 x 32 output tile and a slice of T, multiplies with v_mfma_f32_32x32x2_f32 straight from global
// memory (for a fixed t the 32 o's / i's of a tile are one contiguous 128 B line, and both operands stay in L2),
// and adds its tile into the arena gradient with f32 atomics.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void wgrad_small_wave(const float* __restrict__ g, const float* __restrict__ x, int T,
                                                 int O, int I, int tiles_i, int ntiles, int rows_per_slice,
                                                 float* __restrict__ acc, float* __restrict__ bias_acc, int wg) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int tile = wg % ntiles, slice = wg / ntiles;
  const int o0 = (tile / tiles_i) * 32, i0 = (tile % tiles_i) * 32;
  const int t0 = slice * rows_per_slice;
  const int t1 = t0 + rows_per_slice < T ? t0 + rows_per_slice : T;
  if (t0 >= T) return;
  const bool ov = o0 + r < O, iv = i0 + r < I;
  const float* gp = g + (o0 + r);
  const float* xp = x + (i0 + r);
  f32x16_t c;
#pragma unroll
  for (int k = 0; k < 16; ++k) c[k] = 0.f;
  float bsum = 0.f;                       // bias gradient: the waves of the first column of tiles also sum g over t
  int tb = t0;                            // wave-uniform row base; half h of the wave reads row tb + 2u + h
  for (; tb + 8 <= t1; tb += 8) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = ov ? gp[(long)(tb + 2 * u + h) * O] : 0.f;
      b[u] = iv ? xp[(long)(tb + 2 * u + h) * I] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], c, 0, 0, 0);
    bsum += (a[0] + a[1]) + (a[2] + a[3]);
  }
  for (; tb < t1; tb += 2) {              // out-of-range rows contribute 0
    const int t = tb + h;
    const bool tv = t < t1;
    const float a = (ov && tv) ? gp[(long)t * O] : 0.f;
    const float b = (iv && tv) ? xp[(long)t * I] : 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    bsum += a;
  }
  if (bias_acc && i0 == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0 && ov) atomicAdd(bias_acc + o0 + r, bsum);
  }
  if (iv) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int o = o0 + (k & 3) + 8 * (k >> 2) + 4 * h;
      if (o < O) atomicAdd(acc + (long)o * I + i0 + r, c[k]);
    }
  }
}


__global__ void __launch_bounds__(256) k_wgrad_small(const float* __restrict__ g, const float* __restrict__ x, int T,
                                                     int O, int I, int tiles_i, int ntiles, int rows_per_slice,
                                                     float* __restrict__ acc, float* __restrict__ bias_acc) {
  wgrad_small_wave(g, x, T, O, I, tiles_i, ntiles, rows_per_slice, acc, bias_acc, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// The same for up to kWgradGroup Linears in ONE launch.  The weight gradients of the decoder's 400-row layers are
// nobody's input: issued one by one between the data-gradient kernels they cost 63 launches of ≈ 10 µs with the chip
// mostly idle; collected during the backward and launched together at its end they fill it once.
constexpr int kWgradGroup = 48;
struct WgradEntry {
  const float* g; const float* x; float* acc; float* bias_acc;
  int T, O, I, tiles_i, ntiles, rows_per_slice, wave_begin;
};
struct WgradGroupArgs {
  WgradEntry e[kWgradGroup];
  int n, total_waves;
};

__global__ void __launch_bounds__(256) k_wgrad_small_group(const WgradGroupArgs a) {
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= a.total_waves) return;
  int i = 0;
  while (i + 1 < a.n && wg >= a.e[i + 1].wave_begin) ++i;       // wave-uniform: a scalar loop over <= 48 entries
  const WgradEntry& e = a.e[i];
  wgrad_small_wave(e.g, e.x, e.T, e.O, e.I, e.tiles_i, e.ntiles, e.rows_per_slice, e.acc, e.bias_acc, wg - e.wave_begin);
}

}  // namespace

extern "C" int mbv_adamw_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                              int32_t shadow_dtype, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int64_t step, float grad_scale, int32_t decoupled, int32_t zero_grad,
                              const float* loss_scale, const int32_t* skip_flag, const int32_t* applied_steps,
                              void* stream) {
  if (n < 0 || step < 1 || !param || !grad || !exp_avg || !exp_avg_sq) return MBV_ERR_BAD_ARG;
  if (n == 0) return MBV_OK;
  if ((reinterpret_cast<size_t>(param) | reinterpret_cast<size_t>(grad) | reinterpret_cast<size_t>(exp_avg) |
       reinterpret_cast<size_t>(exp_avg_sq)) & 15)
    return MBV_ERR_BAD_ARG;
  if (shadow_bf16 && (reinterpret_cast<size_t>(shadow_bf16) & 7)) return MBV_ERR_BAD_ARG;
  if (shadow_bf16 && shadow_dtype != MBV_DT_BF16 && shadow_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  AdamArgs a;
  a.shadow_kind = shadow_dtype; a.loss_scale = loss_scale; a.skip = skip_flag; a.applied = applied_steps;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bias_correction1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bias_correction2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = grad_scale; a.decoupled = decoupled; a.zero_grad = zero_grad;
  const long n4 = n >> 2;
  long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;      // 16 blocks per CU, grid-stride beyond
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_adamw, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, reinterpret_cast<unsigned short*>(shadow_bf16), (long)n, a);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_grad_nonfinite(const float* grad, int64_t n, int32_t* flag, void* stream) {
  if (n < 0 || !grad || !flag || (reinterpret_cast<size_t>(grad) & 15)) return MBV_ERR_BAD_ARG;
  if (n == 0) return MBV_OK;
  long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_grad_nonfinite, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, (long)n, flag);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_loss_scale_update(float* loss_scale, int32_t* clean_steps, int32_t* flag, float growth, float backoff,
                                     int32_t growth_interval, int32_t* applied_steps, void* stream) {
  if (!loss_scale || !clean_steps || !flag || growth < 1.f || backoff <= 0.f || backoff > 1.f || growth_interval < 1)
    return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_loss_scale_update, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_scale, clean_steps, flag,
                     growth, backoff, growth_interval, applied_steps);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_refresh_shadow(const float* param, void* shadow_bf16, int32_t shadow_dtype, int64_t n, void* stream) {
  if (n < 0 || !param || !shadow_bf16) return MBV_ERR_BAD_ARG;
  if (shadow_dtype != MBV_DT_BF16 && shadow_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  if (n == 0) return MBV_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_shadow, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param,
                     reinterpret_cast<unsigned short*>(shadow_bf16), (long)n, shadow_dtype);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// launch geometry of a column sum: W threads per row, gx column blocks, gy row slices
static void colsum_geometry(int64_t rows, int32_t n, int target_blocks, int& W, unsigned& gx, long& gy) {
  W = 1;
  while (W < 64 && W * 4 < n) W <<= 1;          // a wave per 256 columns: wide matrices get column blocks (gridDim.x), not idle lanes
  const int rpi = 256 / W;
  gx = (unsigned)((n + W * 4 - 1) / (W * 4));
  // ≈ 2 blocks per CU, at most 64 row slices per column, at least 8 iterations per block
  gy = (target_blocks + gx - 1) / gx;
  const long cap = (long)rows * n >= (16L << 20) ? 128 : 64;     // big inputs: bandwidth outweighs contention
  if (gy > cap) gy = cap;
  if (gy > rows / (8L * rpi)) gy = rows / (8L * rpi);
  if (gy < 1) gy = 1;
}

// count independent column sums out[i] (n[i]) += Σ_r g[i][r, :n[i]] (row stride ld[i] elements, dtype[i] an MBV_DT_*
// code) in one launch per 64.  Every array argument is a HOST array of length count.
extern "C" int mbv_colsum_accum_group(const void* const* g, const int32_t* dtype, const int64_t* rows, const int32_t* n,
                                      const int64_t* ld, float* const* out, int32_t count, void* stream) {
  if (count < 0 || (count > 0 && (!g || !dtype || !rows || !n || !ld || !out))) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < count; base += kColsumGroup) {
    const int cnt = count - base < kColsumGroup ? count - base : kColsumGroup;
    ColsumGroupArgs a;
    a.n = 0;
    int blocks = 0;
    const int target = 2048 / cnt > 16 ? 2048 / cnt : 16;          // ≈ 8 blocks per CU over the group
    for (int j = 0; j < cnt; ++j) {
      const int i = base + j;
      if (rows[i] < 0 || n[i] <= 0 || ld[i] < n[i] || !g[i] || !out[i] || dtype[i] < 0 || dtype[i] > 2) return MBV_ERR_BAD_ARG;
      if (rows[i] == 0) continue;
      ColsumEntry& e = a.e[a.n++];
      unsigned gx; long gy;
      colsum_geometry(rows[i], n[i], target, e.W, gx, gy);
      e.g = g[i]; e.out = out[i]; e.rows = (int)rows[i]; e.n = n[i]; e.ld = (int)ld[i]; e.kind = dtype[i];
      e.gx = (int)gx; e.gy = (int)gy; e.block_begin = blocks;
      e.vec = ((n[i] & 3) == 0 && (ld[i] & 3) == 0 && (reinterpret_cast<size_t>(g[i]) & (dtype[i] ? 7 : 15)) == 0) ? 1 : 0;
      blocks += e.gx * e.gy;
    }
    if (a.n == 0) continue;
    a.total_blocks = blocks;
    hipLaunchKernelGGL(k_colsum_group, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

extern "C" int mbv_colsum_accum(const void* g, int32_t is_bf16, int64_t rows, int32_t n, float* out, void* stream) {
  if (rows < 0 || n <= 0 || !g || !out) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  int W;
  unsigned gx;
  long gy;
  colsum_geometry(rows, n, 512, W, gx, gy);
  const bool vec = (n & 3) == 0 && (reinterpret_cast<size_t>(g) & (is_bf16 ? 7 : 15)) == 0;
  if (is_bf16 == MBV_DT_F16)
    hipLaunchKernelGGL(k_colsum<_Float16>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(g), (long)rows, n, out, vec, W);
  else if (is_bf16)
    hipLaunchKernelGGL(k_colsum<unsigned short>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(g), (long)rows, n, out, vec, W);
  else
    hipLaunchKernelGGL(k_colsum<float>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(g), (long)rows, n, out, vec, W);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_wgrad_small_f32(const float* g, const float* x, int32_t T, int32_t O, int32_t I, float* acc,
                                   float* bias_acc, void* stream) {
  if (T < 0 || O <= 0 || I <= 0 || !g || !x || !acc) return MBV_ERR_BAD_ARG;
  if (T == 0) return MBV_OK;
  const int tiles_o = (O + 31) / 32, tiles_i = (I + 31) / 32;
  const int ntiles = tiles_o * tiles_i;
  int slices = (2048 + ntiles - 1) / ntiles;            // ≈ 2048 waves = 8 per CU (1024: the 2048-wide FFN layers ran 200-row chains, 21 us)
  const int max_slices = (T + 31) / 32;                 // at least 32 rows per slice
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  int rows_per_slice = (T + slices - 1) / slices;
  rows_per_slice = (rows_per_slice + 1) & ~1;           // even: the two k-halves of the MFMA stay aligned
  slices = (T + rows_per_slice - 1) / rows_per_slice;
  const long waves = (long)ntiles * slices;
  hipLaunchKernelGGL(k_wgrad_small, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, x, T, O,
                     I, tiles_i, ntiles, rows_per_slice, acc, bias_acc);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// n independent products acc[i] (O[i], I[i]) += g[i]^T x[i] (and bias_acc[i] += column sums of g[i] where given) in as
// few launches as the kernel-argument size allows (48 per launch).  All arrays are HOST arrays of length n.
extern "C" int mbv_wgrad_small_f32_group(const float* const* g, const float* const* x, float* const* acc,
                                         float* const* bias_acc, const int32_t* T, const int32_t* O, const int32_t* I,
                                         int32_t n, void* stream) {
  if (n < 0 || (n > 0 && (!g || !x || !acc || !T || !O || !I))) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < n; base += kWgradGroup) {
    const int cnt = n - base < kWgradGroup ? n - base : kWgradGroup;
    WgradGroupArgs a;
    a.n = 0;
    int waves = 0;
    // ≈ 16 waves per CU over the whole group, at least 128 waves per product, at least 32 rows per slice
    const int per_entry = 4096 / cnt > 128 ? 4096 / cnt : 128;
    for (int j = 0; j < cnt; ++j) {
      const int i = base + j;
      if (T[i] < 0 || O[i] <= 0 || I[i] <= 0 || !g[i] || !x[i] || !acc[i]) return MBV_ERR_BAD_ARG;
      if (T[i] == 0) continue;
      WgradEntry& e = a.e[a.n++];
      e.g = g[i]; e.x = x[i]; e.acc = acc[i]; e.bias_acc = bias_acc ? bias_acc[i] : nullptr;
      e.T = T[i]; e.O = O[i]; e.I = I[i];
      const int tiles_o = (O[i] + 31) / 32;
      e.tiles_i = (I[i] + 31) / 32;
      e.ntiles = tiles_o * e.tiles_i;
      int slices = (per_entry + e.ntiles - 1) / e.ntiles;
      const int max_slices = (T[i] + 31) / 32;
      if (slices > max_slices) slices = max_slices;
      if (slices < 1) slices = 1;
      int rps = (T[i] + slices - 1) / slices;
      rps = (rps + 1) & ~1;
      slices = (T[i] + rps - 1) / rps;
      e.rows_per_slice = rps;
      e.wave_begin = waves;
      waves += e.ntiles * slices;
    }
    if (a.n == 0) continue;
    a.total_waves = waves;
    hipLaunchKernelGGL(k_wgrad_small_group, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

extern "C" int mbv_act_bwd_colsum(const void* grad_act, const void* pre_act, int32_t is_bf16, int32_t kind, int64_t rows,
                                  int32_t n, void* grad_pre, float* bias_acc, void* stream) {
  if (rows < 0 || n <= 0 || (n & 3) || kind < 0 || kind > 1) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!grad_act || !pre_act || !grad_pre) return MBV_ERR_BAD_ARG;
  const size_t al = is_bf16 ? 7 : 15;
  if ((reinterpret_cast<size_t>(grad_act) | reinterpret_cast<size_t>(pre_act) | reinterpret_cast<size_t>(grad_pre)) & al)
    return MBV_ERR_BAD_ARG;
  int W = 1;
  while (W < 256 && W * 4 < n) W <<= 1;
  const int rpi = 256 / W;
  const unsigned gx = (unsigned)((n + W * 4 - 1) / (W * 4));
  long gy = (2048 + gx - 1) / gx;                      // elementwise work: fill the chip; ≤ 256 atomics per column
  if (gy > 256) gy = 256;
  if (gy > rows / (4L * rpi)) gy = rows / (4L * rpi);
  if (gy < 1) gy = 1;
  const dim3 grid(gx, (unsigned)gy), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MBV_ACT(T, K)                                                                                              \
  hipLaunchKernelGGL((k_act_bwd_colsum<T, K>), grid, block, 0, st, reinterpret_cast<const T*>(grad_act),           \
                     reinterpret_cast<const T*>(pre_act), (long)rows, n, reinterpret_cast<T*>(grad_pre), bias_acc, W)
  if (is_bf16 == MBV_DT_F16) { if (kind) MBV_ACT(_Float16, 1); else MBV_ACT(_Float16, 0); }
  else if (is_bf16) { if (kind) MBV_ACT(unsigned short, 1); else MBV_ACT(unsigned short, 0); }
  else { if (kind) MBV_ACT(float, 1); else MBV_ACT(float, 0); }
#undef MBV_ACT
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
