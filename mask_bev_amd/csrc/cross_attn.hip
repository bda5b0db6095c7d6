// K6 — multi-head (masked) attention of the Mask2Former transformer decoder on MFMA, forward and backward.
//
// Replaces the `nn.MultiheadAttention` cores reached through mmcv's MultiheadAttention wrapper from the decoder
// loop of mask_bev/models/networks/mask2former_head/mask2former_head.py:535-560 (configured at
// mask_bev/models/head/mask_bev_panoptic_head.py:150-176): masked cross-attention of the Q = 100 queries over
// L in {256, 1024, 4096} memory tokens (8 heads x 32) and the 100 x 100 self-attention.
// The boolean mask is the (B, Q, L) byte map written by K7 — kept once per query, not once per head.
//
// Forward is split over L ("flash-decoding"): one workgroup owns (batch, head, 128-key split) and, like K4,
// holds the whole 128-query x 128-key score tile of its split in MFMA accumulators (S^T = K Q^T, exact
// softmax inside the split, O_s = P V with P as the next MFMA's operand).  It emits (m_s, l_s, O_s); a small
// combine kernel merges the splits and writes the output and the row log-sum-exp.  L = 4096 gives
// 32 splits x 32 (batch, head) = 1024 workgroups instead of 32.
// Backward per split: part 1 (lane = query) forms dQ (f32 atomics into (B, Q, E): splits x 400 x 256 floats),
// part 2 (wave = key block) forms dK, dV of the split's keys with plain stores.
#include "mfma_tiles.hpp"

namespace {

using namespace mbv_tiles;

struct AttnGeom {
  int B, Q, L, heads, E, nsplit, nsuper;
  int ldkv;      // row stride (elements) of the key / value inputs: E, or wider when several layers' projections of the
                 // same memory are written side by side by one GEMM (the pointers then carry the column offset)
  int ldg;       // row stride of grad_k / grad_v
  int gkv_bf16;  // grad_k / grad_v stored in the 16-bit type (single query super-block only: plain stores)
  int vec_ok;    // 16-bit tensors 16-byte aligned, E / ldkv / D multiples of 8: fused 16-byte staging
  int mask_vec;  // mask rows 16-byte aligned (L % 16 == 0): 16-byte mask loads
};

struct AttnBlock {
  int b, head, su, split, q0, k0, nq, nk, ws_row;
};

__device__ __forceinline__ AttnBlock decode(const AttnGeom& g) {
  int id = blockIdx.x;
  AttnBlock r;
  r.ws_row = id;
  r.split = id % g.nsplit;
  id /= g.nsplit;
  r.su = id % g.nsuper;
  id /= g.nsuper;
  r.head = id % g.heads;
  r.b = id / g.heads;
  r.q0 = r.su * NPAD;
  r.k0 = r.split * NPAD;
  r.nq = min(NPAD, g.Q - r.q0);
  r.nk = min(NPAD, g.L - r.k0);
  return r;
}

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// rows [r0, r0 + n) of src (B, R, E), columns [col, col + D) -> padded-row [row][d] image (the f32 path);
// `vec`: 16-byte loads (f32 tensors 16-byte aligned, E % 4 == 0)
template <int D, typename TIn>
__device__ __forceinline__ void stage_rows(const TIn* __restrict__ src, int64_t batch_off, int E, int r0, int n,
                                           int col, float* row_img, bool vec) {
  using L = Lay<false, D>;
  constexpr int CH = D / 4;
#pragma unroll 2
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c4 = (idx - t * CH) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < n) {
      const TIn* p = src + batch_off + (int64_t)(r0 + t) * E + col + c4;
      if constexpr (std::is_same_v<TIn, float>) {
        if (vec) {
          const float4 f = *reinterpret_cast<const float4*>(p);
          v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = p[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = to_f(p[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) row_img[t * L::RS + c4 + j] = v[j];
  }
}

// the same into a swizzled 16-bit image, element by element (tensors that miss the alignment of the fused path)
template <int D, typename TIn>
__device__ __forceinline__ void stage_rows_swz(const TIn* __restrict__ src, int64_t batch_off, int E, int r0, int n,
                                               int col, lo16_t* img) {
  constexpr int CH = D / 8;
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c = idx - t * CH;
    lo16_t* dst = img + Swz<D>::chunk_off(t, c);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dst[j] = (lo16_t)(t < n ? to_f(src[batch_off + (int64_t)(r0 + t) * E + col + 8 * c + j]) : 0.f);
  }
}

// 16-byte chunk c of row r0 + t (zeros beyond n)
__device__ __forceinline__ uint4 row_chunk(const lo16_t* __restrict__ src, int64_t batch_off, int E, int r0, int n,
                                           int col, int t, int c) {
  return t < n ? *reinterpret_cast<const uint4*>(src + batch_off + (int64_t)(r0 + t) * E + col + 8 * c)
               : make_uint4(0, 0, 0, 0);
}

// Blocked-pair bits of this block: word[q][kb] holds keys 32 kb .. 32 kb + 31 of query q (bit set = blocked: masked by
// the byte map, or a key / query beyond the range).  2 KB instead of a 16 KB byte tile; a lane that owns a query
// keeps its four words in registers, a lane that owns a key reads the query's word as an LDS broadcast.
__device__ __forceinline__ unsigned pack_bytes4(unsigned w) { return (((w & 0x01010101u) * 0x01020408u) >> 24) & 0xfu; }

__device__ __forceinline__ void stage_mask_bits(const uint8_t* __restrict__ mask, const AttnGeom& g, const AttnBlock& k,
                                                unsigned* __restrict__ words) {
  for (int idx = threadIdx.x; idx < NPAD * NBLK; idx += blockDim.x) {
    const int q = idx / NBLK, kb = idx - q * NBLK;
    unsigned w = 0xffffffffu;
    if (q < k.nq && 32 * kb < k.nk) {
      const int nvalid = min(32, k.nk - 32 * kb);
      w = 0;
      if (mask) {
        const uint8_t* row = mask + ((int64_t)k.b * g.Q + k.q0 + q) * g.L + k.k0 + 32 * kb;
        if (g.mask_vec && nvalid == 32) {
          const uint4 lo = *reinterpret_cast<const uint4*>(row), hi = *reinterpret_cast<const uint4*>(row + 16);
          w = pack_bytes4(lo.x) | (pack_bytes4(lo.y) << 4) | (pack_bytes4(lo.z) << 8) | (pack_bytes4(lo.w) << 12) |
              (pack_bytes4(hi.x) << 16) | (pack_bytes4(hi.y) << 20) | (pack_bytes4(hi.z) << 24) | (pack_bytes4(hi.w) << 28);
        } else {
          for (int j = 0; j < nvalid; ++j) w |= (row[j] ? 1u : 0u) << j;
        }
      }
      if (nvalid < 32) w |= 0xffffffffu << nvalid;
    }
    words[idx] = w;
  }
}

// ---------------------------------------------------------------------------------------------
// forward: per-split partials (m, l in log2 units)
// ---------------------------------------------------------------------------------------------
template <bool BF16, int D, typename TIn>
__global__ void __launch_bounds__(256) k_attn_fwd_split(const TIn* __restrict__ q, const TIn* __restrict__ k,
                                                        const TIn* __restrict__ v, const uint8_t* __restrict__ mask,
                                                        AttnGeom g, float scale, float* __restrict__ ws_m,
                                                        float* __restrict__ ws_l, float* __restrict__ ws_o,
                                                        TIn* __restrict__ out, float* __restrict__ lse) {
  using L = Lay<BF16, D>;
  using T = typename L::T;
  constexpr int IMG = BF16 ? Swz<D>::IMG : L::ROW_IMG;      // 16-bit: swizzled, unpadded (mfma_tiles.hpp); f32: padded rows
  __shared__ __attribute__((aligned(16))) T q_img[IMG];
  __shared__ __attribute__((aligned(16))) T k_img[IMG];
  __shared__ __attribute__((aligned(16))) T v_img[IMG];
  __shared__ unsigned mbits[NPAD * NBLK];
  const AttnBlock blk = decode(g);
  const int col = blk.head * D;
  const int64_t qoff = (int64_t)blk.b * g.Q * g.E, koff = (int64_t)blk.b * g.L * g.ldkv;
  if constexpr (BF16) {
    bool fused = false;
    if constexpr (std::is_same_v<TIn, lo16_t>) {
      if (g.vec_ok) {      // every 16-byte load of the block is issued before the first LDS store: one memory round trip
        fused = true;
        constexpr int CH = D / 8;
#pragma unroll 2
        for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
          const int t = idx / CH, c = idx - t * CH;
          const uint4 a = row_chunk(q, qoff, g.E, blk.q0, blk.nq, col, t, c);
          const uint4 b = row_chunk(k, koff, g.ldkv, blk.k0, blk.nk, col, t, c);
          const uint4 d = row_chunk(v, koff, g.ldkv, blk.k0, blk.nk, col, t, c);
          const int o = Swz<D>::chunk_off(t, c);
          *reinterpret_cast<uint4*>(q_img + o) = a;
          *reinterpret_cast<uint4*>(k_img + o) = b;
          *reinterpret_cast<uint4*>(v_img + o) = d;
        }
      }
    }
    if (!fused) {
      stage_rows_swz<D, TIn>(q, qoff, g.E, blk.q0, blk.nq, col, q_img);
      stage_rows_swz<D, TIn>(k, koff, g.ldkv, blk.k0, blk.nk, col, k_img);
      stage_rows_swz<D, TIn>(v, koff, g.ldkv, blk.k0, blk.nk, col, v_img);
    }
  } else {
    stage_rows<D, TIn>(q, qoff, g.E, blk.q0, blk.nq, col, q_img, g.vec_ok);
    stage_rows<D, TIn>(k, koff, g.ldkv, blk.k0, blk.nk, col, k_img, g.vec_ok);
    stage_rows<D, TIn>(v, koff, g.ldkv, blk.k0, blk.nk, col, v_img, g.vec_ok);
  }
  stage_mask_bits(mask, g, blk, mbits);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nqb = (blk.nq + 31) / 32, nkb = (blk.nk + 31) / 32;
  if (wave >= nqb) return;
  const int ql = 32 * wave + r;
  f32x16 s[NBLK];
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    s[kb] = zero16();
    if (kb < nkb) {
      if constexpr (BF16) mma_rows_swz<D>(k_img, 32 * kb, q_img, 32 * wave, s[kb]);
      else mma_rows<BF16, D>(k_img, 32 * kb, q_img, 32 * wave, s[kb]);
    }
  }
  // scores in log2 units; blocked pairs -> -inf.  This lane's query owns one mask word per key block; element i of
  // lane half h is key 32 kb + acc_row(i, h) = bit (i & 3) + 8 (i >> 2) of the word shifted by 4 h.
  const float sl2 = scale * kLog2e;
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    const unsigned w = mbits[ql * NBLK + kb] >> (4 * h);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool ok = kb < nkb && !((w >> ((i & 3) + 8 * (i >> 2))) & 1u);
      const float val = ok ? s[kb][i] * sl2 : -INFINITY;
      s[kb][i] = val;
      m = fmaxf(m, val);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float m_use = (m == -INFINITY) ? 0.f : m;
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(s[kb][i] - m_use);
      s[kb][i] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);
  const int64_t wrow = (int64_t)blk.ws_row * NPAD;
  const bool single = g.nsplit == 1;          // all keys in this block (the 100 x 100 self-attention): no combine pass
  if (h == 0) {
    if (!single) {
      ws_m[wrow + ql] = m;
      ws_l[wrow + ql] = sum;
    } else if (ql < blk.nq) {
      lse[((int64_t)blk.b * g.heads + blk.head) * g.Q + blk.q0 + ql] = (m + __log2f(sum)) * kLn2;
    }
  }
  // 1 / sum of each accumulator row's query, for the direct store (rows of O are queries, lanes are d)
  __shared__ float inv_s[NPAD];
  if (single) {
    if (h == 0) inv_s[ql] = 1.f / sum;
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave reads back only what it wrote
  }
  constexpr int NCB = (D + 31) / 32;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    f32x16 o = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb)
      if (kb < nkb) {
        if constexpr (BF16) mma_acc_tr<D>(s[kb], v_img, 32 * kb, cb, o);
        else mma_acc_operand<BF16, D>(s[kb], v_img, 32 * kb, cb, o);
      }
    const int dcol = r + 32 * cb;
    if (dcol < D) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * wave + acc_row(i, h);
        if (!single) ws_o[(wrow + qq) * D + dcol] = o[i];
        else if (qq < blk.nq)
          out[((int64_t)blk.b * g.Q + blk.q0 + qq) * g.E + blk.head * D + dcol] = (TIn)(o[i] * inv_s[qq]);
      }
    }
  }
}

// merge the splits: out[b][q][head*D + d], lse[b][head][q] (natural-log units; the partials are in log2 units)
template <typename TOut>
__global__ void __launch_bounds__(256) k_attn_combine(const float* __restrict__ ws_m, const float* __restrict__ ws_l,
                                                      const float* __restrict__ ws_o, AttnGeom g, int D,
                                                      TOut* __restrict__ out, float* __restrict__ lse) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)g.B * g.heads * g.Q * D;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  int64_t t = idx / D;
  const int qq = (int)(t % g.Q);
  t /= g.Q;
  const int head = (int)(t % g.heads);
  const int b = (int)(t / g.heads);
  const int su = qq / NPAD, ql = qq - su * NPAD;
  const int64_t base = (((int64_t)b * g.heads + head) * g.nsuper + su) * g.nsplit;
  float M = -INFINITY;
#pragma unroll 8
  for (int s = 0; s < g.nsplit; ++s) M = fmaxf(M, ws_m[(base + s) * NPAD + ql]);
  float den = 0.f, num = 0.f;
#pragma unroll 8
  for (int s = 0; s < g.nsplit; ++s) {      // independent loads, 8 splits in flight
    const float ms = ws_m[(base + s) * NPAD + ql];
    const float ls = ws_l[(base + s) * NPAD + ql];
    const float os = ws_o[((base + s) * NPAD + ql) * D + d];
    const float w = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - M);
    den += w * ls;
    num += w * os;
  }
  out[((int64_t)b * g.Q + qq) * g.E + head * D + d] = (TOut)(num / den);
  if (d == 0) lse[((int64_t)b * g.heads + head) * g.Q + qq] = (M + __log2f(den)) * kLn2;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// 512 threads: waves 0-3 own the query blocks (dQ partial of this key split), waves 4-7 the key blocks (dK, dV) —
// side by side on the same staged images, like K4's backward.
template <bool BF16, int D, typename TIn>
__global__ void __launch_bounds__(512, 2) k_attn_bwd(const TIn* __restrict__ q, const TIn* __restrict__ k,
                                                     const TIn* __restrict__ v, const uint8_t* __restrict__ mask,
                                                     const TIn* __restrict__ out, const TIn* __restrict__ grad_out,
                                                     const float* __restrict__ lse, AttnGeom g, float scale,
                                                     float* __restrict__ grad_q, float* __restrict__ grad_k,
                                                     float* __restrict__ grad_v) {
  using L = Lay<BF16, D>;
  using T = typename L::T;
  constexpr int IMG = BF16 ? Swz<D>::IMG : L::ROW_IMG;
  __shared__ __attribute__((aligned(16))) T q_img[IMG];
  __shared__ __attribute__((aligned(16))) T k_img[IMG];
  __shared__ __attribute__((aligned(16))) T v_img[IMG];
  __shared__ __attribute__((aligned(16))) T do_img[IMG];
  __shared__ unsigned mbits[NPAD * NBLK];
  __shared__ double delta_s[NPAD];     // f64: LDS ds_add_f32 is ≈ 20x slower than ds_add_f64 on gfx950
  __shared__ float2 ld_s[NPAD];        // (lse log2 e, delta) per query
  const AttnBlock blk = decode(g);
  const int col = blk.head * D;
  const int64_t qoff = (int64_t)blk.b * g.Q * g.E, koff = (int64_t)blk.b * g.L * g.ldkv;
  const int64_t goff = (int64_t)blk.b * g.L * g.ldg;
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) delta_s[t] = 0.0;
  __syncthreads();
  bool fused = false;
  if constexpr (BF16 && std::is_same_v<TIn, lo16_t>) {
    if (g.vec_ok) {
      fused = true;
      constexpr int CH = D / 8;
      for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
        const int t = idx / CH, c = idx - t * CH;
        union { uint4 u; lo16_t e[8]; } a, b, d, go, o;
        a.u = row_chunk(q, qoff, g.E, blk.q0, blk.nq, col, t, c);
        go.u = row_chunk(grad_out, qoff, g.E, blk.q0, blk.nq, col, t, c);
        o.u = row_chunk(out, qoff, g.E, blk.q0, blk.nq, col, t, c);
        b.u = row_chunk(k, koff, g.ldkv, blk.k0, blk.nk, col, t, c);
        d.u = row_chunk(v, koff, g.ldkv, blk.k0, blk.nk, col, t, c);
        const int off = Swz<D>::chunk_off(t, c);
        *reinterpret_cast<uint4*>(q_img + off) = a.u;
        *reinterpret_cast<uint4*>(do_img + off) = go.u;
        *reinterpret_cast<uint4*>(k_img + off) = b.u;
        *reinterpret_cast<uint4*>(v_img + off) = d.u;
        if (t < blk.nq) {                 // delta[q] = sum_d dO[q][d] * O[q][d]
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += (float)o.e[j] * (float)go.e[j];
          atomicAdd(&delta_s[t], (double)acc);
        }
      }
    }
  }
  if (!fused) {
    if constexpr (BF16) {
      stage_rows_swz<D, TIn>(q, qoff, g.E, blk.q0, blk.nq, col, q_img);
      stage_rows_swz<D, TIn>(k, koff, g.ldkv, blk.k0, blk.nk, col, k_img);
      stage_rows_swz<D, TIn>(v, koff, g.ldkv, blk.k0, blk.nk, col, v_img);
      stage_rows_swz<D, TIn>(grad_out, qoff, g.E, blk.q0, blk.nq, col, do_img);
    } else {
      stage_rows<D, TIn>(q, qoff, g.E, blk.q0, blk.nq, col, q_img, g.vec_ok);
      stage_rows<D, TIn>(k, koff, g.ldkv, blk.k0, blk.nk, col, k_img, g.vec_ok);
      stage_rows<D, TIn>(v, koff, g.ldkv, blk.k0, blk.nk, col, v_img, g.vec_ok);
      stage_rows<D, TIn>(grad_out, qoff, g.E, blk.q0, blk.nq, col, do_img, g.vec_ok);
    }
    constexpr int CH = D / 8;
    for (int idx = threadIdx.x; idx < blk.nq * CH; idx += blockDim.x) {
      const int t = idx / CH, c8 = (idx - t * CH) * 8;
      const int64_t o = qoff + (int64_t)(blk.q0 + t) * g.E + col + c8;
      float acc = 0.f;
      if constexpr (std::is_same_v<TIn, float>) {
        if (g.vec_ok) {
          const float4 a0 = *reinterpret_cast<const float4*>(out + o), a1 = *reinterpret_cast<const float4*>(out + o + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(grad_out + o), b1 = *reinterpret_cast<const float4*>(grad_out + o + 4);
          acc = a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
          atomicAdd(&delta_s[t], (double)acc);
          continue;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += to_f(out[o + j]) * to_f(grad_out[o + j]);
      atomicAdd(&delta_s[t], (double)acc);
    }
  }
  stage_mask_bits(mask, g, blk, mbits);
  __syncthreads();
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) {
    const float l = t < blk.nq ? lse[((int64_t)blk.b * g.heads + blk.head) * g.Q + blk.q0 + t] : 0.f;
    ld_s[t] = make_float2(l * kLog2e, (float)delta_s[t]);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, part = threadIdx.x >> 8;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (blk.nq + 31) / 32, nkb = (blk.nk + 31) / 32;
  constexpr int NCB = (D + 31) / 32;
  const float sl2 = scale * kLog2e;

  if (part == 0 && wave < nqb) {   // ---- part 1: lane = query → dQ partial of this key split
    const int ql = 32 * wave + r;
    const float my_lse2 = ld_s[ql].x, my_delta = ld_s[ql].y;
    f32x16 dq[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dq[cb] = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
      if (kb >= nkb) continue;
      f32x16 s = zero16(), dp = zero16();
      if constexpr (BF16) {
        mma_rows_swz<D>(k_img, 32 * kb, q_img, 32 * wave, s);
        mma_rows_swz<D>(v_img, 32 * kb, do_img, 32 * wave, dp);
      } else {
        mma_rows<BF16, D>(k_img, 32 * kb, q_img, 32 * wave, s);
        mma_rows<BF16, D>(v_img, 32 * kb, do_img, 32 * wave, dp);
      }
      const unsigned w = mbits[ql * NBLK + kb] >> (4 * h);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool blocked = (w >> ((i & 3) + 8 * (i >> 2))) & 1u;
        const float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, -my_lse2));
        s[i] = blocked ? 0.f : p * (dp[i] - my_delta) * scale;
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        if constexpr (BF16) mma_acc_tr<D>(s, k_img, 32 * kb, cb, dq[cb]);
        else mma_acc_operand<BF16, D>(s, k_img, 32 * kb, cb, dq[cb]);
      }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int dcol = r + 32 * cb;
      if (dcol >= D) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * wave + acc_row(i, h);
        if (qq < blk.nq) {
          float* dst = grad_q + qoff + (int64_t)(blk.q0 + qq) * g.E + col + dcol;
          if (g.nsplit > 1) atomicAdd(dst, dq[cb][i]);
          else *dst = dq[cb][i];
        }
      }
    }
  }
  if (part == 1 && wave < nkb) {   // ---- part 2: wave = key block → dK, dV of this split's keys
    const int kb = wave;
    f32x16 dk[NCB], dv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { dk[cb] = zero16(); dv[cb] = zero16(); }
#pragma unroll
    for (int qb = 0; qb < NBLK; ++qb) {
      if (qb >= nqb) continue;
      f32x16 s = zero16(), dp = zero16();
      if constexpr (BF16) {
        mma_rows_swz<D>(q_img, 32 * qb, k_img, 32 * kb, s);
        mma_rows_swz<D>(do_img, 32 * qb, v_img, 32 * kb, dp);
      } else {
        mma_rows<BF16, D>(q_img, 32 * qb, k_img, 32 * kb, s);
        mma_rows<BF16, D>(do_img, 32 * qb, v_img, 32 * kb, dp);
      }
      f32x16 ds;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * qb + acc_row(i, h);
        const float2 ld = ld_s[qq];                                        // LDS broadcasts: one address per lane half
        const bool blocked = (mbits[qq * NBLK + kb] >> r) & 1u;            // this lane's key inside the query's word
        float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, -ld.x));
        p = blocked ? 0.f : p;
        s[i] = p;
        ds[i] = p * (dp[i] - ld.y) * scale;
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        if constexpr (BF16) {
          mma_acc_tr<D>(s, do_img, 32 * qb, cb, dv[cb]);
          mma_acc_tr<D>(ds, q_img, 32 * qb, cb, dk[cb]);
        } else {
          mma_acc_operand<BF16, D>(s, do_img, 32 * qb, cb, dv[cb]);
          mma_acc_operand<BF16, D>(ds, q_img, 32 * qb, cb, dk[cb]);
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int dcol = r + 32 * cb;
      if (dcol >= D) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = 32 * kb + acc_row(i, h);
        if (kk < blk.nk) {
          const int64_t o = goff + (int64_t)(blk.k0 + kk) * g.ldg + col + dcol;
          if (g.gkv_bf16) {
            reinterpret_cast<unsigned short*>(grad_k)[o] = f32_to_lo16(dk[cb][i]);
            reinterpret_cast<unsigned short*>(grad_v)[o] = f32_to_lo16(dv[cb][i]);
          } else if (g.nsuper > 1) { atomicAdd(grad_k + o, dk[cb][i]); atomicAdd(grad_v + o, dv[cb][i]); }
          else { grad_k[o] = dk[cb][i]; grad_v[o] = dv[cb][i]; }
        }
      }
    }
  }
}

bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
               const void* e = nullptr) {
  return ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(c) |
           reinterpret_cast<size_t>(d) | reinterpret_cast<size_t>(e)) & 15) == 0;
}

bool make_attn_geom(int B, int Q, int L, int heads, int D, AttnGeom& g) {
  if (B <= 0 || Q <= 0 || L <= 0 || heads <= 0 || D <= 0) return false;
  g.B = B; g.Q = Q; g.L = L; g.heads = heads; g.E = heads * D;
  g.ldkv = g.ldg = g.E; g.gkv_bf16 = 0; g.vec_ok = 0; g.mask_vec = 0;
  g.nsplit = (L + NPAD - 1) / NPAD;
  g.nsuper = (Q + NPAD - 1) / NPAD;
  return true;
}

size_t attn_ws_floats(const AttnGeom& g, int D) {
  const size_t rows = (size_t)g.B * g.heads * g.nsuper * g.nsplit * NPAD;
  return rows * (2 + D);
}

template <bool BF16, typename TIn>
int attn_fwd_launch(const AttnGeom& g, int D, const void* q, const void* k, const void* v, const uint8_t* mask,
                    void* out, float* lse, float* ws, hipStream_t stream) {
  const float scale = 1.0f / sqrtf((float)D);
  const size_t rows = (size_t)g.B * g.heads * g.nsuper * g.nsplit * NPAD;
  float *ws_m = ws, *ws_l = ws + rows, *ws_o = ws + 2 * rows;
  const dim3 grid((unsigned)(g.B * g.heads * g.nsuper * g.nsplit)), block(256);
  const TIn *qq = (const TIn*)q, *kk = (const TIn*)k, *vv = (const TIn*)v;
  switch (D) {
    case 16: hipLaunchKernelGGL((k_attn_fwd_split<BF16, 16, TIn>), grid, block, 0, stream, qq, kk, vv, mask, g, scale, ws_m, ws_l, ws_o, (TIn*)out, lse); break;
    case 32: hipLaunchKernelGGL((k_attn_fwd_split<BF16, 32, TIn>), grid, block, 0, stream, qq, kk, vv, mask, g, scale, ws_m, ws_l, ws_o, (TIn*)out, lse); break;
    case 64: hipLaunchKernelGGL((k_attn_fwd_split<BF16, 64, TIn>), grid, block, 0, stream, qq, kk, vv, mask, g, scale, ws_m, ws_l, ws_o, (TIn*)out, lse); break;
    default: return MBV_ERR_UNSUPPORTED;
  }
  MBV_CHECK_LAUNCH();
  if (g.nsplit == 1) return MBV_OK;            // the split kernel normalised and stored the output itself
  const int64_t total = (int64_t)g.B * g.heads * g.Q * D;
  hipLaunchKernelGGL((k_attn_combine<TIn>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ws_m, ws_l,
                     ws_o, g, D, (TIn*)out, lse);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

template <bool BF16, typename TIn>
int attn_bwd_launch(const AttnGeom& g, int D, const void* q, const void* k, const void* v, const uint8_t* mask,
                    const void* out, const void* grad_out, const float* lse, float* gq, float* gk, float* gv,
                    hipStream_t stream) {
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 grid((unsigned)(g.B * g.heads * g.nsuper * g.nsplit)), block(512);
  const TIn *qq = (const TIn*)q, *kk = (const TIn*)k, *vv = (const TIn*)v, *oo = (const TIn*)out, *go = (const TIn*)grad_out;
  switch (D) {
    case 16: hipLaunchKernelGGL((k_attn_bwd<BF16, 16, TIn>), grid, block, 0, stream, qq, kk, vv, mask, oo, go, lse, g, scale, gq, gk, gv); break;
    case 32: hipLaunchKernelGGL((k_attn_bwd<BF16, 32, TIn>), grid, block, 0, stream, qq, kk, vv, mask, oo, go, lse, g, scale, gq, gk, gv); break;
    case 64: hipLaunchKernelGGL((k_attn_bwd<BF16, 64, TIn>), grid, block, 0, stream, qq, kk, vv, mask, oo, go, lse, g, scale, gq, gk, gv); break;
    default: return MBV_ERR_UNSUPPORTED;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

#ifdef MBV_H16
// ---------------------------------------------------------------------------------------------
// f32 tensors on the 16-bit matrix pipe ("split" mode; fp32 compute) — K4's split mode (csrc/window_attn.hip) for the decoder
// ---------------------------------------------------------------------------------------------
// Every f32 operand element of a block's tiles is split into an IEEE-half pair while the tile is staged (x 2^e = hi + lo: 22
// significant bits; two swizzled images per operand) and every product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16.
// The power-of-two scales are PER TILE here (the maximum of the 128 x D tile a block stages, found in registers between the
// loads and the LDS stores: a tile is small and each block's results are rescaled on their own); of the register operands the
// probabilities take 2^13 (<= 1) and every dS tile the scale of its own maximum (acc_tr_split_scaled).
union SPack8 {
  uint4 u;
  lo16_t h[8];
};
constexpr float kProbScale = 8192.f, kProbInv = 1.f / 8192.f;

__device__ __forceinline__ void sload8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ float smax8(const float (&v)[8]) {
  return fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
               fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
}
template <int D>
__device__ __forceinline__ void sput(lo16_t* img, int t, int c, const float (&v)[8], float s) {
  SPack8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float y = v[j] * s;
    hi.h[j] = (lo16_t)y;
    lo.h[j] = (lo16_t)(y - (float)hi.h[j]);
  }
  const int o = Swz<D>::chunk_off(t, c);
  *reinterpret_cast<uint4*>(img + o) = hi.u;
  *reinterpret_cast<uint4*>(img + Swz<D>::IMG + o) = lo.u;
}
// biased exponent e of the scale 2^(e - 127) that puts a maximum with these bits in [2^13, 2^14) (127 for an all-zero tile)
__device__ __forceinline__ int scale_exp(unsigned bits) {
  if (bits == 0u) return 127;
  const int e = 267 - (int)(bits >> 23);
  return e < 1 ? 1 : (e > 253 ? 253 : e);
}
__device__ __forceinline__ float pow2e(int e) {
  e = e < 1 ? 1 : (e > 254 ? 254 : e);
  return __uint_as_float((unsigned)e << 23);
}
// the maxima of up to four tiles of a block: per-thread values -> per-wave words in LDS -> (after the caller's barrier) maxima
template <int N>
__device__ __forceinline__ void publish_tile_max(const float (&m)[N], unsigned* wmax /* [waves][N] */) {
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float v = m[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[wave * N + i] = __float_as_uint(v) & 0x7fffffffu;
  }
}
template <int N>
__device__ __forceinline__ unsigned tile_max(const unsigned* wmax, int i) {
  unsigned m = 0u;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) m = m > wmax[w * N + i] ? m : wmax[w * N + i];
  return m;
}

// the power-of-two scale (biased exponent) of an accumulator tile that becomes an MFMA operand: its largest magnitude over the wave
__device__ __forceinline__ int acc_tile_scale_exp(const f32x16& x) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return scale_exp(__builtin_amdgcn_readfirstlane(__float_as_uint(m) & 0x7fffffffu));
}
// acc += (X^T . M) with X split at ITS OWN scale (a dS tile: with thousands of keys its elements sit 2^-11 and more below any
// static bound — measured 2e-5 in dK at L = 4 096 with one bound per block — so every tile takes the scale of its maximum and
// its product joins the f32 sum outside the matrix instruction)
template <int D>
__device__ __forceinline__ void acc_tr_split_scaled(const f32x16& x, const lo16_t* m_hi, const lo16_t* m_lo, int k0, int cb,
                                                    float m_inv, f32x16& acc) {
  const int e = acc_tile_scale_exp(x);
  f32x16 tmp = zero16();
  mma_acc_tr_split<D>(x, pow2e(e), m_hi, m_lo, k0, cb, tmp);
  const float inv = pow2e(254 - e) * m_inv;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = fmaf(tmp[i], inv, acc[i]);
}

template <int D>
__global__ void __launch_bounds__(256) k_attn_split_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const uint8_t* __restrict__ mask,
                                                        AttnGeom g, float scale, float* __restrict__ ws_m,
                                                        float* __restrict__ ws_l, float* __restrict__ ws_o,
                                                        float* __restrict__ out, float* __restrict__ lse) {
  constexpr int IMG = Swz<D>::IMG, CH = D / 8, ITEMS = (NPAD * CH + 255) / 256;
  __shared__ __attribute__((aligned(16))) lo16_t q_img[2 * IMG];      // hi image, lo image
  __shared__ __attribute__((aligned(16))) lo16_t k_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t v_img[2 * IMG];
  __shared__ unsigned mbits[NPAD * NBLK];
  __shared__ unsigned wmax[4 * 3];
  __shared__ float inv_s[NPAD];
  const AttnBlock blk = decode(g);
  const int col = blk.head * D;
  const int64_t qoff = (int64_t)blk.b * g.Q * g.E, koff = (int64_t)blk.b * g.L * g.ldkv;
  float qv[ITEMS][8], kv[ITEMS][8], vv[ITEMS][8];
  float mx[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = threadIdx.x + 256 * it, t = idx / CH, c = idx - t * CH;
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[it][j] = kv[it][j] = vv[it][j] = 0.f;
    if (idx < NPAD * CH) {
      if (t < blk.nq) sload8(q + qoff + (int64_t)(blk.q0 + t) * g.E + col + 8 * c, qv[it]);
      if (t < blk.nk) {
        sload8(k + koff + (int64_t)(blk.k0 + t) * g.ldkv + col + 8 * c, kv[it]);
        sload8(v + koff + (int64_t)(blk.k0 + t) * g.ldkv + col + 8 * c, vv[it]);
      }
    }
    mx[0] = fmaxf(mx[0], smax8(qv[it])); mx[1] = fmaxf(mx[1], smax8(kv[it])); mx[2] = fmaxf(mx[2], smax8(vv[it]));
  }
  publish_tile_max<3>(mx, wmax);
  stage_mask_bits(mask, g, blk, mbits);
  __syncthreads();
  const int eq = scale_exp(tile_max<3>(wmax, 0)), ek = scale_exp(tile_max<3>(wmax, 1)), ev = scale_exp(tile_max<3>(wmax, 2));
  const float s_q = pow2e(eq), s_k = pow2e(ek), s_v = pow2e(ev);
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = threadIdx.x + 256 * it, t = idx / CH, c = idx - t * CH;
    if (idx < NPAD * CH) {
      sput<D>(q_img, t, c, qv[it], s_q);
      sput<D>(k_img, t, c, kv[it], s_k);
      sput<D>(v_img, t, c, vv[it], s_v);
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nqb = (blk.nq + 31) / 32, nkb = (blk.nk + 31) / 32;
  if (wave >= nqb) return;
  const int ql = 32 * wave + r;
  f32x16 s[NBLK];
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    s[kb] = zero16();
    if (kb < nkb) mma_rows_split<D>(k_img, k_img + IMG, 32 * kb, q_img, q_img + IMG, 32 * wave, s[kb]);
  }
  const float sl2 = scale * kLog2e * pow2e(254 - eq) * pow2e(254 - ek);
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    const unsigned w = mbits[ql * NBLK + kb] >> (4 * h);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool ok = kb < nkb && !((w >> ((i & 3) + 8 * (i >> 2))) & 1u);
      const float val = ok ? s[kb][i] * sl2 : -INFINITY;
      s[kb][i] = val;
      m = fmaxf(m, val);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float m_use = (m == -INFINITY) ? 0.f : m;
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(s[kb][i] - m_use);
      s[kb][i] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 32, 64);
  const int64_t wrow = (int64_t)blk.ws_row * NPAD;
  const bool single = g.nsplit == 1;
  if (h == 0) {
    if (!single) {
      ws_m[wrow + ql] = m;
      ws_l[wrow + ql] = sum;
    } else if (ql < blk.nq) {
      lse[((int64_t)blk.b * g.heads + blk.head) * g.Q + blk.q0 + ql] = (m + __log2f(sum)) * kLn2;
    }
  }
  if (single) {
    if (h == 0) inv_s[ql] = 1.f / sum;
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave reads back only what it wrote
  }
  constexpr int NCB = (D + 31) / 32;
  const float o_inv = kProbInv * pow2e(254 - ev);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    f32x16 o = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb)
      if (kb < nkb) mma_acc_tr_split<D>(s[kb], kProbScale, v_img, v_img + IMG, 32 * kb, cb, o);
    const int dcol = r + 32 * cb;
    if (dcol < D) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * wave + acc_row(i, h);
        if (!single) ws_o[(wrow + qq) * D + dcol] = o[i] * o_inv;
        else if (qq < blk.nq)
          out[((int64_t)blk.b * g.Q + blk.q0 + qq) * g.E + blk.head * D + dcol] = o[i] * o_inv * inv_s[qq];
      }
    }
  }
}

template <int D>
__global__ void __launch_bounds__(512, 2) k_attn_split_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ out, const float* __restrict__ grad_out,
                                                           const float* __restrict__ lse, AttnGeom g, float scale,
                                                           float* __restrict__ grad_q, float* __restrict__ grad_k,
                                                           float* __restrict__ grad_v) {
  constexpr int IMG = Swz<D>::IMG, CH = D / 8, ITEMS = (NPAD * CH + 511) / 512;
  __shared__ __attribute__((aligned(16))) lo16_t q_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t k_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t v_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t do_img[2 * IMG];
  __shared__ unsigned mbits[NPAD * NBLK];
  __shared__ double delta_s[NPAD];
  __shared__ float2 ld_s[NPAD];
  __shared__ unsigned wmax[8 * 4];
  const AttnBlock blk = decode(g);
  const int col = blk.head * D;
  const int64_t qoff = (int64_t)blk.b * g.Q * g.E, koff = (int64_t)blk.b * g.L * g.ldkv;
  const int64_t goff = (int64_t)blk.b * g.L * g.ldg;
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) delta_s[t] = 0.0;
  __syncthreads();
  float qv[ITEMS][8], kv[ITEMS][8], vv[ITEMS][8], dv_[ITEMS][8];
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = threadIdx.x + 512 * it, t = idx / CH, c = idx - t * CH;
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[it][j] = kv[it][j] = vv[it][j] = dv_[it][j] = 0.f;
    if (idx < NPAD * CH) {
      if (t < blk.nq) {
        const int64_t o = qoff + (int64_t)(blk.q0 + t) * g.E + col + 8 * c;
        float ov[8];
        sload8(q + o, qv[it]);
        sload8(grad_out + o, dv_[it]);
        sload8(out + o, ov);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += ov[j] * dv_[it][j];
        atomicAdd(&delta_s[t], (double)acc);          // delta[q] = sum_d dO[q][d] * O[q][d]
      }
      if (t < blk.nk) {
        sload8(k + koff + (int64_t)(blk.k0 + t) * g.ldkv + col + 8 * c, kv[it]);
        sload8(v + koff + (int64_t)(blk.k0 + t) * g.ldkv + col + 8 * c, vv[it]);
      }
    }
    mx[0] = fmaxf(mx[0], smax8(qv[it])); mx[1] = fmaxf(mx[1], smax8(kv[it]));
    mx[2] = fmaxf(mx[2], smax8(vv[it])); mx[3] = fmaxf(mx[3], smax8(dv_[it]));
  }
  publish_tile_max<4>(mx, wmax);
  stage_mask_bits(mask, g, blk, mbits);
  __syncthreads();
  const int eq = scale_exp(tile_max<4>(wmax, 0)), ek = scale_exp(tile_max<4>(wmax, 1));
  const int ev = scale_exp(tile_max<4>(wmax, 2)), ed = scale_exp(tile_max<4>(wmax, 3));
  const float s_q = pow2e(eq), s_k = pow2e(ek), s_v = pow2e(ev), s_d = pow2e(ed);
  const float inv_q = pow2e(254 - eq), inv_k = pow2e(254 - ek), inv_v = pow2e(254 - ev), inv_d = pow2e(254 - ed);
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = threadIdx.x + 512 * it, t = idx / CH, c = idx - t * CH;
    if (idx < NPAD * CH) {
      sput<D>(q_img, t, c, qv[it], s_q);
      sput<D>(k_img, t, c, kv[it], s_k);
      sput<D>(v_img, t, c, vv[it], s_v);
      sput<D>(do_img, t, c, dv_[it], s_d);
    }
  }
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) {
    const float l = t < blk.nq ? lse[((int64_t)blk.b * g.heads + blk.head) * g.Q + blk.q0 + t] : 0.f;
    ld_s[t] = make_float2(l * kLog2e, (float)delta_s[t]);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, part = threadIdx.x >> 8;
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (blk.nq + 31) / 32, nkb = (blk.nk + 31) / 32;
  constexpr int NCB = (D + 31) / 32;
  const float sl2 = scale * kLog2e * inv_q * inv_k;
  const float dp_inv = inv_v * inv_d;

  if (part == 0 && wave < nqb) {   // ---- part 1: lane = query -> dQ partial of this key split
    const int ql = 32 * wave + r;
    const float my_lse2 = ld_s[ql].x, my_delta = ld_s[ql].y;
    f32x16 dq[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dq[cb] = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
      if (kb >= nkb) continue;
      f32x16 s = zero16(), dp = zero16();
      mma_rows_split<D>(k_img, k_img + IMG, 32 * kb, q_img, q_img + IMG, 32 * wave, s);
      mma_rows_split<D>(v_img, v_img + IMG, 32 * kb, do_img, do_img + IMG, 32 * wave, dp);
      const unsigned w = mbits[ql * NBLK + kb] >> (4 * h);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool blocked = (w >> ((i & 3) + 8 * (i >> 2))) & 1u;
        const float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, -my_lse2));
        s[i] = blocked ? 0.f : p * (dp[i] * dp_inv - my_delta) * scale;
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc_tr_split_scaled<D>(s, k_img, k_img + IMG, 32 * kb, cb, inv_k, dq[cb]);
    }
    const float dq_inv = 1.f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int dcol = r + 32 * cb;
      if (dcol >= D) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * wave + acc_row(i, h);
        if (qq < blk.nq) {
          float* dst = grad_q + qoff + (int64_t)(blk.q0 + qq) * g.E + col + dcol;
          if (g.nsplit > 1) atomicAdd(dst, dq[cb][i] * dq_inv);
          else *dst = dq[cb][i] * dq_inv;
        }
      }
    }
  }
  if (part == 1 && wave < nkb) {   // ---- part 2: wave = key block -> dK, dV of this split's keys
    const int kb = wave;
    f32x16 dk[NCB], dv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { dk[cb] = zero16(); dv[cb] = zero16(); }
#pragma unroll
    for (int qb = 0; qb < NBLK; ++qb) {
      if (qb >= nqb) continue;
      f32x16 s = zero16(), dp = zero16();
      mma_rows_split<D>(q_img, q_img + IMG, 32 * qb, k_img, k_img + IMG, 32 * kb, s);
      mma_rows_split<D>(do_img, do_img + IMG, 32 * qb, v_img, v_img + IMG, 32 * kb, dp);
      f32x16 ds;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * qb + acc_row(i, h);
        const float2 ld = ld_s[qq];
        const bool blocked = (mbits[qq * NBLK + kb] >> r) & 1u;
        float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, -ld.x));
        p = blocked ? 0.f : p;
        s[i] = p;
        ds[i] = p * (dp[i] * dp_inv - ld.y) * scale;
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        mma_acc_tr_split<D>(s, kProbScale, do_img, do_img + IMG, 32 * qb, cb, dv[cb]);
        acc_tr_split_scaled<D>(ds, q_img, q_img + IMG, 32 * qb, cb, inv_q, dk[cb]);
      }
    }
    const float dv_inv = kProbInv * inv_d, dk_inv = 1.f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int dcol = r + 32 * cb;
      if (dcol >= D) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = 32 * kb + acc_row(i, h);
        if (kk < blk.nk) {
          const int64_t o = goff + (int64_t)(blk.k0 + kk) * g.ldg + col + dcol;
          const float vk = dk[cb][i] * dk_inv, vv2 = dv[cb][i] * dv_inv;
          if (g.nsuper > 1) { atomicAdd(grad_k + o, vk); atomicAdd(grad_v + o, vv2); }
          else { grad_k[o] = vk; grad_v[o] = vv2; }
        }
      }
    }
  }
}
#endif

}  // namespace

#ifndef MBV_H16
extern "C" size_t mbv_attn_workspace_bytes(int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads,
                                           int32_t head_dim) {
  AttnGeom g;
  if (!make_attn_geom(batch, num_queries, num_keys, heads, head_dim, g)) return 0;
  return mbv_align_up(attn_ws_floats(g, head_dim) * sizeof(float), 256);
}

// the half build of this file (cross_attn_f16.hip); a dtype flag of MBV_DT_F16 forwards there
MBV_F16_TWIN int mbv_attn_fwd_ld_f16(const void*, const void*, const void*, int32_t, const uint8_t*, int32_t, int32_t,
                                     int32_t, int32_t, int32_t, int32_t, void*, float*, void*, size_t, void*);
MBV_F16_TWIN int mbv_attn_bwd_ld_f16(const void*, const void*, const void*, int32_t, const uint8_t*, const void*,
                                     const void*, const float*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                                     float*, void*, void*, int32_t, int32_t, void*);
#endif

MBV_ENTRY int MBV_SYM(mbv_attn_fwd_ld)(const void* q, const void* k, const void* v, int32_t ld_kv, const uint8_t* blocked,
                               int32_t is_bf16, int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads,
                               int32_t head_dim, void* out, float* lse, void* workspace, size_t workspace_bytes,
                               void* stream_) {
#ifndef MBV_H16
  if (is_bf16 == MBV_DT_F16)
    return mbv_attn_fwd_ld_f16(q, k, v, ld_kv, blocked, 1, batch, num_queries, num_keys, heads, head_dim, out, lse,
                               workspace, workspace_bytes, stream_);
#endif
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttnGeom g;
  if (!make_attn_geom(batch, num_queries, num_keys, heads, head_dim, g)) return MBV_ERR_BAD_ARG;
  if (ld_kv < g.E) return MBV_ERR_BAD_ARG;
  g.ldkv = ld_kv;
  if (!q || !k || !v || !out || !lse) return MBV_ERR_BAD_ARG;
  // fused 16-byte staging: 8-element chunks of the 16-bit types, 4-element chunks of f32
  g.vec_ok = is_bf16 ? ((g.E & 7) == 0 && (ld_kv & 7) == 0 && (head_dim & 7) == 0 && aligned16(q, k, v))
                     : ((g.E & 3) == 0 && (ld_kv & 3) == 0 && (head_dim & 3) == 0 && aligned16(q, k, v));
  g.mask_vec = blocked && (num_keys & 15) == 0 && aligned16(blocked);
  if (!workspace || workspace_bytes < mbv_attn_workspace_bytes(batch, num_queries, num_keys, heads, head_dim))
    return MBV_ERR_WORKSPACE;
  float* ws = reinterpret_cast<float*>(workspace);
  return is_bf16 ? attn_fwd_launch<true, lo16_t>(g, head_dim, q, k, v, blocked, out, lse, ws, stream)
                 : attn_fwd_launch<false, float>(g, head_dim, q, k, v, blocked, out, lse, ws, stream);
}

#ifndef MBV_H16
extern "C" int mbv_attn_fwd(const void* q, const void* k, const void* v, const uint8_t* blocked, int32_t is_bf16,
                            int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim,
                            void* out, float* lse, void* workspace, size_t workspace_bytes, void* stream_) {
  return mbv_attn_fwd_ld(q, k, v, heads * head_dim, blocked, is_bf16, batch, num_queries, num_keys, heads, head_dim,
                         out, lse, workspace, workspace_bytes, stream_);
}
#endif

MBV_ENTRY int MBV_SYM(mbv_attn_bwd_ld)(const void* q, const void* k, const void* v, int32_t ld_kv, const uint8_t* blocked,
                               const void* out, const void* grad_out, const float* lse, int32_t is_bf16, int32_t batch,
                               int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim, float* grad_q,
                               void* grad_k, void* grad_v, int32_t ld_grad_kv, int32_t grad_kv_bf16, void* stream_) {
#ifndef MBV_H16
  if (is_bf16 == MBV_DT_F16 || grad_kv_bf16 == MBV_DT_F16)
    return mbv_attn_bwd_ld_f16(q, k, v, ld_kv, blocked, out, grad_out, lse, MBV_LO_FLAG(is_bf16), batch, num_queries,
                               num_keys, heads, head_dim, grad_q, grad_k, grad_v, ld_grad_kv,
                               MBV_LO_FLAG(grad_kv_bf16), stream_);
#endif
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttnGeom g;
  if (!make_attn_geom(batch, num_queries, num_keys, heads, head_dim, g)) return MBV_ERR_BAD_ARG;
  if (!q || !k || !v || !out || !grad_out || !lse || !grad_q || !grad_k || !grad_v) return MBV_ERR_BAD_ARG;
  if (ld_kv < g.E || ld_grad_kv < g.E) return MBV_ERR_BAD_ARG;
  // strided or bf16 key / value gradients are plain stores of whole rows: one query super-block only
  if ((ld_grad_kv != g.E || grad_kv_bf16) && g.nsuper > 1) return MBV_ERR_UNSUPPORTED;
  g.ldkv = ld_kv; g.ldg = ld_grad_kv; g.gkv_bf16 = grad_kv_bf16 ? 1 : 0;
  g.vec_ok = is_bf16 ? ((g.E & 7) == 0 && (ld_kv & 7) == 0 && (head_dim & 7) == 0 && aligned16(q, k, v, out, grad_out))
                     : ((g.E & 3) == 0 && (ld_kv & 3) == 0 && (head_dim & 3) == 0 && aligned16(q, k, v, out, grad_out));
  g.mask_vec = blocked && (num_keys & 15) == 0 && aligned16(blocked);
  if (g.nsplit > 1)
    MBV_CHECK_HIP(mbv_fill_async(grad_q, 0, sizeof(float) * (size_t)batch * num_queries * g.E, stream));
  if (g.nsuper > 1) {
    MBV_CHECK_HIP(mbv_fill_async(grad_k, 0, sizeof(float) * (size_t)batch * num_keys * g.E, stream));
    MBV_CHECK_HIP(mbv_fill_async(grad_v, 0, sizeof(float) * (size_t)batch * num_keys * g.E, stream));
  }
  float* gk32 = reinterpret_cast<float*>(grad_k);
  float* gv32 = reinterpret_cast<float*>(grad_v);
  return is_bf16 ? attn_bwd_launch<true, lo16_t>(g, head_dim, q, k, v, blocked, out, grad_out, lse, grad_q, gk32,
                                                 gv32, stream)
                 : attn_bwd_launch<false, float>(g, head_dim, q, k, v, blocked, out, grad_out, lse, grad_q, gk32,
                                                 gv32, stream);
}

#ifndef MBV_H16
extern "C" int mbv_attn_bwd(const void* q, const void* k, const void* v, const uint8_t* blocked, const void* out,
                            const void* grad_out, const float* lse, int32_t is_bf16, int32_t batch,
                            int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim, float* grad_q,
                            float* grad_k, float* grad_v, void* stream_) {
  return mbv_attn_bwd_ld(q, k, v, heads * head_dim, blocked, out, grad_out, lse, is_bf16, batch, num_queries, num_keys,
                         heads, head_dim, grad_q, grad_k, grad_v, heads * head_dim, 0, stream_);
}
#endif

#ifdef MBV_H16
// ---- K6 on f32 tensors in the split mode (see k_attn_split_fwd): the arguments of mbv_attn_fwd_ld / _bwd_ld without the dtype flags
extern "C" int mbv_attn_split_supported(int32_t heads, int32_t head_dim, int32_t ld_kv) {
  return (heads > 0 && (head_dim == 16 || head_dim == 32 || head_dim == 64) && ld_kv >= heads * head_dim && (ld_kv & 3) == 0) ? 1 : 0;
}

extern "C" int mbv_attn_split_fwd_ld(const float* q, const float* k, const float* v, int32_t ld_kv, const uint8_t* blocked,
                                     int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim,
                                     float* out, float* lse, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttnGeom g;
  if (!make_attn_geom(batch, num_queries, num_keys, heads, head_dim, g)) return MBV_ERR_BAD_ARG;
  if (!q || !k || !v || !out || !lse) return MBV_ERR_BAD_ARG;
  if (!mbv_attn_split_supported(heads, head_dim, ld_kv) || !aligned16(q, k, v, out)) return MBV_ERR_UNSUPPORTED;
  g.ldkv = ld_kv;
  g.mask_vec = blocked && (num_keys & 15) == 0 && aligned16(blocked);
  if (!workspace || workspace_bytes < mbv_align_up(attn_ws_floats(g, head_dim) * sizeof(float), 256)) return MBV_ERR_WORKSPACE;
  float* ws = reinterpret_cast<float*>(workspace);
  const float scale = 1.0f / sqrtf((float)head_dim);
  const size_t rows = (size_t)g.B * g.heads * g.nsuper * g.nsplit * NPAD;
  float *ws_m = ws, *ws_l = ws + rows, *ws_o = ws + 2 * rows;
  const dim3 grid((unsigned)(g.B * g.heads * g.nsuper * g.nsplit)), block(256);
  switch (head_dim) {
    case 16: hipLaunchKernelGGL((k_attn_split_fwd<16>), grid, block, 0, stream, q, k, v, blocked, g, scale, ws_m, ws_l, ws_o, out, lse); break;
    case 32: hipLaunchKernelGGL((k_attn_split_fwd<32>), grid, block, 0, stream, q, k, v, blocked, g, scale, ws_m, ws_l, ws_o, out, lse); break;
    default: hipLaunchKernelGGL((k_attn_split_fwd<64>), grid, block, 0, stream, q, k, v, blocked, g, scale, ws_m, ws_l, ws_o, out, lse); break;
  }
  MBV_CHECK_LAUNCH();
  if (g.nsplit == 1) return MBV_OK;
  const int64_t total = (int64_t)g.B * g.heads * g.Q * head_dim;
  hipLaunchKernelGGL((k_attn_combine<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ws_m, ws_l, ws_o, g,
                     head_dim, out, lse);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_attn_split_bwd_ld(const float* q, const float* k, const float* v, int32_t ld_kv, const uint8_t* blocked,
                                     const float* out, const float* grad_out, const float* lse, int32_t batch,
                                     int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim, float* grad_q,
                                     float* grad_k, float* grad_v, int32_t ld_grad_kv, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttnGeom g;
  if (!make_attn_geom(batch, num_queries, num_keys, heads, head_dim, g)) return MBV_ERR_BAD_ARG;
  if (!q || !k || !v || !out || !grad_out || !lse || !grad_q || !grad_k || !grad_v) return MBV_ERR_BAD_ARG;
  if (ld_grad_kv < g.E) return MBV_ERR_BAD_ARG;
  if (!mbv_attn_split_supported(heads, head_dim, ld_kv) || !aligned16(q, k, v, out, grad_out)) return MBV_ERR_UNSUPPORTED;
  if (ld_grad_kv != g.E && g.nsuper > 1) return MBV_ERR_UNSUPPORTED;
  g.ldkv = ld_kv; g.ldg = ld_grad_kv;
  g.mask_vec = blocked && (num_keys & 15) == 0 && aligned16(blocked);
  if (g.nsplit > 1)
    MBV_CHECK_HIP(mbv_fill_async(grad_q, 0, sizeof(float) * (size_t)batch * num_queries * g.E, stream));
  if (g.nsuper > 1) {
    MBV_CHECK_HIP(mbv_fill_async(grad_k, 0, sizeof(float) * (size_t)batch * num_keys * g.E, stream));
    MBV_CHECK_HIP(mbv_fill_async(grad_v, 0, sizeof(float) * (size_t)batch * num_keys * g.E, stream));
  }
  const float scale = 1.0f / sqrtf((float)head_dim);
  const dim3 grid((unsigned)(g.B * g.heads * g.nsuper * g.nsplit)), block(512);
  switch (head_dim) {
    case 16: hipLaunchKernelGGL((k_attn_split_bwd<16>), grid, block, 0, stream, q, k, v, blocked, out, grad_out, lse, g, scale, grad_q, grad_k, grad_v); break;
    case 32: hipLaunchKernelGGL((k_attn_split_bwd<32>), grid, block, 0, stream, q, k, v, blocked, out, grad_out, lse, g, scale, grad_q, grad_k, grad_v); break;
    default: hipLaunchKernelGGL((k_attn_split_bwd<64>), grid, block, 0, stream, q, k, v, blocked, out, grad_out, lse, g, scale, grad_q, grad_k, grad_v); break;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
#endif
