// K4 — fused shifted-window multi-head attention (Swin W-MSA / SW-MSA), forward and backward, gfx950.
//
// Replaces ShiftWindowMSA.forward + WindowMSA.forward of the reference
// (mask_bev/models/networks/swin/swin.py:179-253 and :80-118; window_partition/reverse :255-284) between the
// qkv projection and the output projection.  The reference pads the map to a multiple of the window, rolls
// it, partitions it into windows, permutes qkv into heads, adds the gathered relative-position bias and the
// shift mask, soft-maxes, multiplies by v, and undoes all of that: five full-tensor layout copies per block
// plus unfused batched GEMMs.  Here ONE workgroup owns one (window, head): padding, cyclic shift, window
// partition/reverse and head split are pure addressing on the channels-last (B, H, W, 3C) qkv map, bias and
// mask are added in registers, and both contractions run on MFMA:
//     S^T = K Q^T            (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 for f32 inputs: exact f32)
//     O   = P V              (P stays in the accumulator registers: "accumulator tile as next operand",
//                             cdna_hip_programming.md §3 — no LDS round trip for the probabilities)
// A window has ws*ws <= 128 tokens, so the whole score matrix of a (window, head) lives in one workgroup
// (4 waves x 32 query rows) and the softmax is exact (no online rescaling).
//
// Backward is flash-attention style with the forward's log-sum-exp: part 1 (lane = query) recomputes S^T,
// P, dP, dS and forms dQ with the register trick; part 2 (wave = key block) recomputes S, P, dP, dS in the
// un-swapped orientation, where the tiles are directly the A operands of dV = P^T dO and dK = dS^T Q.
// Padded tokens (the reference pads AFTER LayerNorm, so their q, k, v equal the qkv bias) send their
// gradient to a (3C) bias-gradient buffer; the relative-position-bias gradient is reduced in LDS per
// workgroup and then added to the table gradient with one atomic per entry.
#include "mfma_tiles.hpp"
#ifdef MBV_H16
#include "amax.hpp"
#endif

namespace {

using namespace mbv_tiles;

// NPAD = 128 tokens per window at most (ws <= 11), 4 blocks of 32 (mfma_tiles.hpp)

struct WinGeom {
  int batch, H, W, C, heads, ws, shift, Hp, Wp, nWh, nWw, N;
  int vec_ok;   // bf16 tensors 16-byte aligned and C % 8 == 0: fused 16-byte staging
};

// token t of window (wy, wx) -> pixel index inside the (H, W) map, or -1 for a padded token
__device__ __forceinline__ int token_pixel(const WinGeom& g, int wy, int wx, int t) {
  const int ty = t / g.ws, tx = t - ty * g.ws;
  int y = wy * g.ws + ty + g.shift;
  int x = wx * g.ws + tx + g.shift;
  if (y >= g.Hp) y -= g.Hp;
  if (x >= g.Wp) x -= g.Wp;
  return (y < g.H && x < g.W) ? y * g.W + x : -1;
}

// region label of the shifted-window mask (swin.py:198-219)
__device__ __forceinline__ int region_label(const WinGeom& g, int wy, int wx, int t) {
  if (g.shift == 0) return 0;
  const int ty = t / g.ws, tx = t - ty * g.ws;
  const int ys = wy * g.ws + ty, xs = wx * g.ws + tx;
  const int ry = ys < g.Hp - g.ws ? 0 : (ys < g.Hp - g.shift ? 1 : 2);
  const int rx = xs < g.Wp - g.ws ? 0 : (xs < g.Wp - g.shift ? 1 : 2);
  return ry * 3 + rx;
}

struct BlockId {
  int b, wy, wx, head;
};
__device__ __forceinline__ BlockId decode_block(const WinGeom& g) {
  int id = blockIdx.x;
  BlockId r;
  r.head = id % g.heads;
  id /= g.heads;
  r.wx = id % g.nWw;
  id /= g.nWw;
  r.wy = id % g.nWh;
  r.b = id / g.nWh;
  return r;
}

// Stage one operand (q, k, v or dO) of this (window, head) into a padded-row [token][d] image (the f32 path).
// Padded tokens take the qkv bias (swin.py:185-188 pads zeros after LayerNorm => qkv == bias).
template <bool BF16, int D, typename TIn>
__device__ __forceinline__ void stage_part(const WinGeom& g, const BlockId& id, const int* __restrict__ pix_lds,
                                           const TIn* __restrict__ src, int src_row_stride, int src_col,
                                           const float* __restrict__ pad_vec /* may be null -> zeros */,
                                           typename Lay<BF16, D>::T* row_img) {
  using L = Lay<BF16, D>;
  using T = typename L::T;
  constexpr int CH = D / 8;   // 8-element chunks per token
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c8 = (idx - t * CH) * 8;
    float v[8];
    if (t < g.N) {
      const int pix = pix_lds[t];
      if (pix >= 0) {
        const TIn* p = src + ((int64_t)id.b * g.H * g.W + pix) * src_row_stride + src_col + c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = to_f(p[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pad_vec ? pad_vec[src_col + c8 + j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) row_img[t * L::RS + c8 + j] = (T)v[j];
  }
}

// scalar staging of one operand into a swizzled image (tensors that are not 16-byte aligned or C % 8 != 0)
template <int D, typename TIn>
__device__ __forceinline__ void stage_part_swz(const WinGeom& g, const BlockId& id, const int* __restrict__ pix_lds,
                                               const TIn* __restrict__ src, int src_row_stride, int src_col,
                                               const float* __restrict__ pad_vec, lo16_t* img) {
  constexpr int CH = D / 8;
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c = idx - t * CH;
    lo16_t* dst = img + Swz<D>::chunk_off(t, c);
    const int pix = t < g.N ? pix_lds[t] : -1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (t < g.N) {
        if (pix >= 0) v = to_f(src[((int64_t)id.b * g.H * g.W + pix) * src_row_stride + src_col + 8 * c + j]);
        else if (pad_vec) v = pad_vec[src_col + 8 * c + j];
      }
      dst[j] = (lo16_t)v;
    }
  }
}

// ---- fused staging for bf16 activations -------------------------------------------------------------------
// One pass over the (token, 8-channel chunk) items of a window: ALL the 16-byte global loads an item needs (q, k, v
// and, in the backward, dO and O) are issued together, then written to the swizzled LDS images with 16-byte stores.
// The per-part form pays one dependent HBM round trip per part and loop iteration (≈ 9 in the backward); this is
// one or two.  Requires C % 8 == 0 and 16-byte aligned tensors.
union Pack8 {
  uint4 u;
  lo16_t h[8];
};

__device__ __forceinline__ Pack8 pad_pack(const float* __restrict__ vec, int c) {
  Pack8 p;
#pragma unroll
  for (int j = 0; j < 8; ++j) p.h[j] = (lo16_t)(vec ? vec[c + j] : 0.f);
  return p;
}

template <int D>
__device__ __forceinline__ void put_row(lo16_t* img, int t, int c8, const Pack8& p) {
  *reinterpret_cast<uint4*>(img + Swz<D>::chunk_off(t, c8 >> 3)) = p.u;
}

template <int D>
__device__ __forceinline__ void stage_fwd_bf16(const WinGeom& g, const BlockId& id, const int* __restrict__ pix_lds,
                                               const lo16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                               int col, lo16_t* q_img, lo16_t* k_img, lo16_t* v_img) {
  constexpr int CH = D / 8;
  const int C3 = 3 * g.C;
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c8 = (idx - t * CH) * 8;
    Pack8 q, k, v;
    q.u = k.u = v.u = make_uint4(0, 0, 0, 0);
    if (t < g.N) {
      const int pix = pix_lds[t];
      if (pix >= 0) {
        const lo16_t* p = qkv + ((int64_t)id.b * g.H * g.W + pix) * C3 + col + c8;
        q.u = *reinterpret_cast<const uint4*>(p);
        k.u = *reinterpret_cast<const uint4*>(p + g.C);
        v.u = *reinterpret_cast<const uint4*>(p + 2 * g.C);
      } else {
        q = pad_pack(qkv_bias, col + c8);
        k = pad_pack(qkv_bias, g.C + col + c8);
        v = pad_pack(qkv_bias, 2 * g.C + col + c8);
      }
    }
    put_row<D>(q_img, t, c8, q);
    put_row<D>(k_img, t, c8, k);
    put_row<D>(v_img, t, c8, v);
  }
}

template <int D>
__device__ __forceinline__ void stage_bwd_bf16(const WinGeom& g, const BlockId& id, const int* __restrict__ pix_lds,
                                               const lo16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                               const lo16_t* __restrict__ out, const lo16_t* __restrict__ grad_out,
                                               int col, lo16_t* q_img, lo16_t* k_img, lo16_t* v_img, lo16_t* do_img,
                                               double* delta_s) {
  constexpr int CH = D / 8;
  const int C3 = 3 * g.C;
#pragma unroll 2
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c8 = (idx - t * CH) * 8;
    Pack8 q, k, v, d, o;
    q.u = k.u = v.u = d.u = o.u = make_uint4(0, 0, 0, 0);
    bool real = false;
    if (t < g.N) {
      const int pix = pix_lds[t];
      if (pix >= 0) {
        real = true;
        const int64_t row = (int64_t)id.b * g.H * g.W + pix;
        const lo16_t* p = qkv + row * C3 + col + c8;
        q.u = *reinterpret_cast<const uint4*>(p);
        k.u = *reinterpret_cast<const uint4*>(p + g.C);
        v.u = *reinterpret_cast<const uint4*>(p + 2 * g.C);
        d.u = *reinterpret_cast<const uint4*>(grad_out + row * g.C + col + c8);
        o.u = *reinterpret_cast<const uint4*>(out + row * g.C + col + c8);
      } else {                      // padded token: q, k, v are the qkv bias; dO is zero (swin.py:247-248 crops it)
        q = pad_pack(qkv_bias, col + c8);
        k = pad_pack(qkv_bias, g.C + col + c8);
        v = pad_pack(qkv_bias, 2 * g.C + col + c8);
      }
    }
    put_row<D>(q_img, t, c8, q);
    put_row<D>(k_img, t, c8, k);
    put_row<D>(v_img, t, c8, v);
    put_row<D>(do_img, t, c8, d);
    if (real) {                     // delta[q] = sum_d dO[q][d] * O[q][d]
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)o.h[j] * (float)d.h[j];
      atomicAdd(&delta_s[t], (double)acc);
    }
  }
}

// Additive relative-position bias + shift mask of (query, key), in log2 units.  Per token the LDS table `kinfo` holds
// lin = ty (2 ws - 1) + tx in its low half and the shifted-window region label in its high half: the bias-table index
// of a pair is lin(q) - lin(k) + (ws - 1) 2 ws — one subtraction per element instead of unpacking both coordinates —
// and the labels differ iff (qi ^ ki) has a high bit.  `tbl` is staged pre-multiplied by log2(e), so that a
// probability is ONE v_exp_f32 of an FMA: p = exp2(s * (scale log2 e) + bias2 - lse2).
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
constexpr float kMask2 = -100.0f * kLog2e;                     // swin.py:216: masked pairs get -100
__device__ __forceinline__ int token_info(const WinGeom& g, const BlockId& id, int t) {
  const int ty = t / g.ws, tx = t - ty * g.ws;
  return (ty * (2 * g.ws - 1) + tx) | (region_label(g, id.wy, id.wx, t) << 16);
}
__device__ __forceinline__ float bias_mask2(const float* __restrict__ tbl, int qi, int ki, int idx0, int* idx_out) {
  const int idx = (qi & 0xffff) - (ki & 0xffff) + idx0;
  if (idx_out) *idx_out = idx;
  float v = tbl[idx];
  if ((qi ^ ki) >> 16) v += kMask2;
  return v;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <bool BF16, int D, typename TIn>
__global__ void __launch_bounds__(256) k_window_attn_fwd(const TIn* __restrict__ qkv,
                                                         const float* __restrict__ qkv_bias,
                                                         const float* __restrict__ bias_table, WinGeom g, float scale,
                                                         TIn* __restrict__ out, float* __restrict__ lse) {
  using L = Lay<BF16, D>;
  using T = typename L::T;
  constexpr int IMG = BF16 ? Swz<D>::IMG : L::ROW_IMG;      // 16-bit: swizzled, unpadded; f32: padded rows
  __shared__ __attribute__((aligned(16))) T k_img[IMG];
  __shared__ __attribute__((aligned(16))) T q_img[IMG];
  __shared__ __attribute__((aligned(16))) T v_img[IMG];
  __shared__ float tbl[21 * 21];
  __shared__ int kinfo[NPAD];
  __shared__ int pix[NPAD];
  const BlockId id = decode_block(g);
  const int tsz = (2 * g.ws - 1) * (2 * g.ws - 1);
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) tbl[i] = bias_table[i * g.heads + id.head] * kLog2e;
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) {
    if (t < g.N) {
      kinfo[t] = token_info(g, id, t);
      pix[t] = token_pixel(g, id.wy, id.wx, t);
    } else {
      kinfo[t] = 0;
      pix[t] = -1;
    }
  }
  __syncthreads();
  const int C3 = 3 * g.C, col = id.head * D;
  const bool vec_ok = g.vec_ok != 0;
  bool fused = false;
  if constexpr (BF16 && std::is_same_v<TIn, lo16_t>) {
    if (vec_ok) {
      fused = true;
      stage_fwd_bf16<D>(g, id, pix, qkv, qkv_bias, col, q_img, k_img, v_img);
    }
  }
  if (fused) {
  } else if constexpr (BF16) {
    stage_part_swz<D, TIn>(g, id, pix, qkv, C3, col, qkv_bias, q_img);
    stage_part_swz<D, TIn>(g, id, pix, qkv, C3, g.C + col, qkv_bias, k_img);
    stage_part_swz<D, TIn>(g, id, pix, qkv, C3, 2 * g.C + col, qkv_bias, v_img);
  } else {
    stage_part<BF16, D, TIn>(g, id, pix, qkv, C3, col, qkv_bias, q_img);
    stage_part<BF16, D, TIn>(g, id, pix, qkv, C3, g.C + col, qkv_bias, k_img);
    stage_part<BF16, D, TIn>(g, id, pix, qkv, C3, 2 * g.C + col, qkv_bias, v_img);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nblk = (g.N + 31) / 32;
  if (wave >= nblk) return;
  const int q = 32 * wave + r;
  // S^T tiles: rows = keys (accumulator registers), cols = this wave's queries (lanes)
  f32x16 s[NBLK];
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    s[kb] = zero16();
    if (kb < nblk) {
      if constexpr (BF16) mma_rows_swz<D>(k_img, 32 * kb, q_img, 32 * wave, s[kb]);
      else mma_rows<BF16, D>(k_img, 32 * kb, q_img, 32 * wave, s[kb]);
    }
  }
  // scores in log2 units: v = s (scale log2 e) + bias2
  const float sl2 = scale * kLog2e;
  const int idx0 = (g.ws - 1) * 2 * g.ws;
  const int qi = kinfo[q];
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // branch-free: every lane evaluates its element (kinfo / tbl hold valid entries for all NPAD tokens), padded
      // keys are then set to -inf — the 16 elements' LDS reads of a tile are issued together, not one dependent
      // chain per exec-masked block
      const int k = 32 * kb + acc_row(i, h);
      float v = fmaf(s[kb][i], sl2, bias_mask2(tbl, qi, kinfo[k], idx0, nullptr));
      v = (q < g.N) ? v : 0.f;
      v = (kb < nblk && k < g.N) ? v : -INFINITY;
      s[kb][i] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(s[kb][i] - m);     // exp2(-inf) = 0 for padded keys
      s[kb][i] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
    for (int i = 0; i < 16; ++i) s[kb][i] *= inv;
  if (h == 0 && q < g.N) lse[(int64_t)blockIdx.x * NPAD + q] = (m + __log2f(sum)) * kLn2;     // natural-log units

  // O = P V : rows = queries (accumulator registers), cols = d (lanes)
  constexpr int NCB = (D + 31) / 32;
  f32x16 o[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    o[cb] = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb)
      if (kb < nblk) {
        if constexpr (BF16) mma_acc_tr<D>(s[kb], v_img, 32 * kb, cb, o[cb]);
        else mma_acc_operand<BF16, D>(s[kb], v_img, 32 * kb, cb, o[cb]);
      }
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int dcol = r + 32 * cb;
    if (dcol >= D) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = 32 * wave + acc_row(i, h);
      if (qq < g.N) {
        const int px = pix[qq];
        if (px >= 0) out[((int64_t)id.b * g.H * g.W + px) * g.C + col + dcol] = (TIn)o[cb][i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// 512 threads: waves 0-3 own the query blocks (dS^T tiles → bias-table gradient, dQ) while waves 4-7 own the key
// blocks (dK, dV) — the two halves of the backward share the staged LDS images and run side by side.  16-bit: four
// swizzled images (q, k, v, dO; 76 KB at D = 64) and ≤ 128 VGPRs, so two workgroups share a CU and one's staging
// overlaps the other's MFMA phase.
template <bool BF16, int D, typename TIn>
__global__ void __launch_bounds__(512, (BF16 ? 4 : 2)) k_window_attn_bwd(const TIn* __restrict__ qkv,
                                                         const float* __restrict__ qkv_bias,
                                                         const float* __restrict__ bias_table,
                                                         const TIn* __restrict__ out, const TIn* __restrict__ grad_out,
                                                         const float* __restrict__ lse, WinGeom g, float scale,
                                                         TIn* __restrict__ grad_qkv, float* __restrict__ grad_table,
                                                         float* __restrict__ grad_pad /* (3C) */, int full_bias) {
  using L = Lay<BF16, D>;
  using T = typename L::T;
  constexpr int IMG = BF16 ? Swz<D>::IMG : L::ROW_IMG;
  __shared__ __attribute__((aligned(16))) T q_img[IMG];
  __shared__ __attribute__((aligned(16))) T k_img[IMG];
  __shared__ __attribute__((aligned(16))) T v_img[IMG];
  __shared__ __attribute__((aligned(16))) T do_img[IMG];
  __shared__ float tbl[21 * 21];
  __shared__ double dtbl[21 * 21];     // f64: LDS ds_add_f32 is ≈ 20x slower than ds_add_f64 on gfx950
  __shared__ int kinfo[NPAD];
  __shared__ int pix[NPAD];
  __shared__ float lse_s[NPAD];
  __shared__ double delta_s[NPAD];
  __shared__ float2 ld_s[NPAD];        // (lse log2 e, delta) per query, f32: what the element loops read
  // column sums of dQ / dK / dV of this (window, head): the qkv-bias gradient.  Padded tokens always contribute (their
  // q, k, v ARE the bias); with full_bias the real tokens do too, which is the bias gradient of the qkv Linear itself
  // — that layer then skips its own pass over grad_qkv.
  __shared__ double colacc[3 * D];
  const BlockId id = decode_block(g);
  const int tsz = (2 * g.ws - 1) * (2 * g.ws - 1);
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) {
    tbl[i] = bias_table[i * g.heads + id.head] * kLog2e;
    dtbl[i] = 0.0;
  }
  for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) colacc[i] = 0.0;
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) {
    if (t < g.N) {
      kinfo[t] = token_info(g, id, t);
      pix[t] = token_pixel(g, id.wy, id.wx, t);
      lse_s[t] = lse[(int64_t)blockIdx.x * NPAD + t];
    } else {
      kinfo[t] = 0;
      pix[t] = -1;
      lse_s[t] = 0.f;
    }
    delta_s[t] = 0.0;
  }
  __syncthreads();
  const int C3 = 3 * g.C, col = id.head * D;
  bool fused = false;
  if constexpr (BF16 && std::is_same_v<TIn, lo16_t>) {
    if (g.vec_ok) {
      fused = true;
      stage_bwd_bf16<D>(g, id, pix, qkv, qkv_bias, out, grad_out, col, q_img, k_img, v_img, do_img, delta_s);
    }
  }
  if (!fused) {
    // dO is zero on padded tokens (the reference crops them away, swin.py:247-248)
    if constexpr (BF16) {
      stage_part_swz<D, TIn>(g, id, pix, qkv, C3, col, qkv_bias, q_img);
      stage_part_swz<D, TIn>(g, id, pix, qkv, C3, g.C + col, qkv_bias, k_img);
      stage_part_swz<D, TIn>(g, id, pix, qkv, C3, 2 * g.C + col, qkv_bias, v_img);
      stage_part_swz<D, TIn>(g, id, pix, grad_out, g.C, col, nullptr, do_img);
    } else {
      stage_part<BF16, D, TIn>(g, id, pix, qkv, C3, col, qkv_bias, q_img);
      stage_part<BF16, D, TIn>(g, id, pix, qkv, C3, g.C + col, qkv_bias, k_img);
      stage_part<BF16, D, TIn>(g, id, pix, qkv, C3, 2 * g.C + col, qkv_bias, v_img);
      stage_part<BF16, D, TIn>(g, id, pix, grad_out, g.C, col, nullptr, do_img);
    }
    // delta[q] = sum_d dO[q][d] * O[q][d]
    constexpr int CH = D / 8;
    for (int idx = threadIdx.x; idx < g.N * CH; idx += blockDim.x) {
      const int t = idx / CH, c8 = (idx - t * CH) * 8;
      const int px = pix[t];
      if (px < 0) continue;
      const int64_t o = ((int64_t)id.b * g.H * g.W + px) * g.C + col + c8;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += to_f(out[o + j]) * to_f(grad_out[o + j]);
      atomicAdd(&delta_s[t], (double)acc);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) ld_s[t] = make_float2(lse_s[t] * kLog2e, (float)delta_s[t]);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, part = threadIdx.x >> 8;
  const int r = lane & 31, h = lane >> 5;
  const int nblk = (g.N + 31) / 32;
  constexpr int NCB = (D + 31) / 32;
  const int64_t row0 = (int64_t)id.b * g.H * g.W;
  const float sl2 = scale * kLog2e;
  const int idx0 = (g.ws - 1) * 2 * g.ws;

  if (wave < nblk && part == 0) {
    // ---- part 1: lane = query.  dS^T tiles, relative-position-bias gradient, dQ
    const int q = 32 * wave + r;
    const float my_lse2 = ld_s[q].x, my_delta = ld_s[q].y;
    const int qi = kinfo[q];
    f32x16 dq[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dq[cb] = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
      if (kb >= nblk) continue;
      f32x16 s = zero16(), dp = zero16();
      if constexpr (BF16) {
        mma_rows_swz<D>(k_img, 32 * kb, q_img, 32 * wave, s);
        mma_rows_swz<D>(v_img, 32 * kb, do_img, 32 * wave, dp);
      } else {
        mma_rows<BF16, D>(k_img, 32 * kb, q_img, 32 * wave, s);
        mma_rows<BF16, D>(v_img, 32 * kb, do_img, 32 * wave, dp);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // branch-free up to the atomic (see the forward): padded pairs contribute ds = 0
        const int k = 32 * kb + acc_row(i, h);
        int idx;
        const float bm = bias_mask2(tbl, qi, kinfo[k], idx0, &idx);
        const float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, bm - my_lse2));
        const bool valid = q < g.N && k < g.N;
        const float ds = valid ? p * (dp[i] - my_delta) : 0.f;
        if (valid) atomicAdd(&dtbl[idx], (double)ds);      // 100 x 100 of the 128 x 128 pairs: the LDS-atomic pipe is the
                                                           // busiest unit of this kernel, padded pairs stay off it
        s[i] = ds * scale;                       // dQ = scale * dS K
        // four elements' LDS chains in flight are enough; without the fence the scheduler hoists all 16 and spills
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
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
      float csum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * wave + acc_row(i, h);
        if (qq >= g.N) continue;
        const int px = pix[qq];
        if (px >= 0) grad_qkv[(row0 + px) * C3 + col + dcol] = (TIn)dq[cb][i];
        if (px < 0 || full_bias) csum += dq[cb][i];
      }
      csum += __shfl_xor(csum, 32, 64);
      if (h == 0 && csum != 0.f) atomicAdd(&colacc[dcol], (double)csum);
    }
  }
  if (wave < nblk && part == 1) {
    // ---- part 2: wave = key block.  dK, dV
    const int kb = wave;
    f32x16 dk[NCB], dv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { dk[cb] = zero16(); dv[cb] = zero16(); }
    const int k = 32 * kb + r;                   // this lane's key (column of the un-swapped tiles)
    const int ki = kinfo[k];
#pragma unroll
    for (int qb = 0; qb < NBLK; ++qb) {
      if (qb >= nblk) continue;
      f32x16 s = zero16(), dp = zero16();
      if constexpr (BF16) {
        mma_rows_swz<D>(q_img, 32 * qb, k_img, 32 * kb, s);       // rows = queries, cols = keys
        mma_rows_swz<D>(do_img, 32 * qb, v_img, 32 * kb, dp);
      } else {
        mma_rows<BF16, D>(q_img, 32 * qb, k_img, 32 * kb, s);
        mma_rows<BF16, D>(do_img, 32 * qb, v_img, 32 * kb, dp);
      }
      f32x16 ds;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * qb + acc_row(i, h);
        const float2 ld = ld_s[qq];
        float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, bias_mask2(tbl, kinfo[qq], ki, idx0, nullptr) - ld.x));
        p = (qq < g.N && k < g.N) ? p : 0.f;
        s[i] = p;
        ds[i] = p * (dp[i] - ld.y) * scale;
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        if constexpr (BF16) {
          mma_acc_tr<D>(s, do_img, 32 * qb, cb, dv[cb]);                              // dV = P^T dO
          mma_acc_tr<D>(ds, q_img, 32 * qb, cb, dk[cb]);                              // dK = scale dS^T Q
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
      float ksum = 0.f, vsum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = 32 * kb + acc_row(i, h);
        if (kk >= g.N) continue;
        const int px = pix[kk];
        if (px >= 0) {
          grad_qkv[(row0 + px) * C3 + g.C + col + dcol] = (TIn)dk[cb][i];
          grad_qkv[(row0 + px) * C3 + 2 * g.C + col + dcol] = (TIn)dv[cb][i];
        }
        if (px < 0 || full_bias) { ksum += dk[cb][i]; vsum += dv[cb][i]; }
      }
      ksum += __shfl_xor(ksum, 32, 64);
      vsum += __shfl_xor(vsum, 32, 64);
      if (h == 0 && ksum != 0.f) atomicAdd(&colacc[D + dcol], (double)ksum);
      if (h == 0 && vsum != 0.f) atomicAdd(&colacc[2 * D + dcol], (double)vsum);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) atomicAdd(&grad_table[i * g.heads + id.head], (float)dtbl[i]);
  for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) {
    const double v = colacc[i];
    if (v != 0.0) atomicAdd(&grad_pad[(i / D) * g.C + col + (i % D)], (float)v);
  }
}

bool make_geom(int batch, int H, int W, int C, int heads, int ws, int shift, WinGeom& g) {
  if (batch <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || ws <= 0 || shift < 0 || shift >= ws) return false;
  if (C % heads != 0) return false;
  g.batch = batch; g.H = H; g.W = W; g.C = C; g.heads = heads; g.ws = ws; g.shift = shift;
  g.Hp = (H + ws - 1) / ws * ws;
  g.Wp = (W + ws - 1) / ws * ws;
  g.nWh = g.Hp / ws;
  g.nWw = g.Wp / ws;
  g.N = ws * ws;
  g.vec_ok = 0;
  return true;
}

template <bool BF16, typename TIn>
int launch_fwd(const WinGeom& g, int D, const void* qkv, const float* qkv_bias, const float* table, void* out,
               float* lse, hipStream_t stream) {
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 grid((unsigned)(g.batch * g.nWh * g.nWw * g.heads)), block(256);
  const TIn* q = reinterpret_cast<const TIn*>(qkv);
  TIn* o = reinterpret_cast<TIn*>(out);
  WinGeom gv = g;
  gv.vec_ok = BF16 && g.C % 8 == 0 && (reinterpret_cast<size_t>(qkv) & 15) == 0;
  switch (D) {
    case 16: hipLaunchKernelGGL((k_window_attn_fwd<BF16, 16, TIn>), grid, block, 0, stream, q, qkv_bias, table, gv, scale, o, lse); break;
    case 32: hipLaunchKernelGGL((k_window_attn_fwd<BF16, 32, TIn>), grid, block, 0, stream, q, qkv_bias, table, gv, scale, o, lse); break;
    case 64: hipLaunchKernelGGL((k_window_attn_fwd<BF16, 64, TIn>), grid, block, 0, stream, q, qkv_bias, table, gv, scale, o, lse); break;
    default: return MBV_ERR_UNSUPPORTED;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

template <bool BF16, typename TIn>
int launch_bwd(const WinGeom& g, int D, const void* qkv, const float* qkv_bias, const float* table, const void* out,
               const void* grad_out, const float* lse, void* grad_qkv, float* grad_table, float* grad_pad,
               int full_bias, hipStream_t stream) {
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 grid((unsigned)(g.batch * g.nWh * g.nWw * g.heads)), block(512);
  const TIn* q = reinterpret_cast<const TIn*>(qkv);
  const TIn* o = reinterpret_cast<const TIn*>(out);
  const TIn* go = reinterpret_cast<const TIn*>(grad_out);
  TIn* gq = reinterpret_cast<TIn*>(grad_qkv);
  WinGeom gv = g;
  gv.vec_ok = BF16 && g.C % 8 == 0 &&
              ((reinterpret_cast<size_t>(qkv) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(grad_out)) & 15) == 0;
  switch (D) {
    case 16: hipLaunchKernelGGL((k_window_attn_bwd<BF16, 16, TIn>), grid, block, 0, stream, q, qkv_bias, table, o, go, lse, gv, scale, gq, grad_table, grad_pad, full_bias); break;
    case 32: hipLaunchKernelGGL((k_window_attn_bwd<BF16, 32, TIn>), grid, block, 0, stream, q, qkv_bias, table, o, go, lse, gv, scale, gq, grad_table, grad_pad, full_bias); break;
    case 64: hipLaunchKernelGGL((k_window_attn_bwd<BF16, 64, TIn>), grid, block, 0, stream, q, qkv_bias, table, o, go, lse, gv, scale, gq, grad_table, grad_pad, full_bias); break;
    default: return MBV_ERR_UNSUPPORTED;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

#ifdef MBV_H16
// ---------------------------------------------------------------------------------------------
// f32 tensors on the 16-bit matrix pipe ("split" mode; fp32 compute)
// ---------------------------------------------------------------------------------------------
// The exact-f32 form above multiplies with v_mfma_f32_32x32x2_f32 (the f32 vector rate) out of padded-row f32 images that
// it fills and reads one word at a time.  Here every f32 operand element is split while its tile is staged — x 2^e = hi + lo,
// two IEEE halves, 22 significant bits (K20's arithmetic: csrc/gemm_f32s.hip) — into a PAIR of the 16-bit path's swizzled
// images, and every product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with f32 accumulation: three instructions at
// 16x the f32 rate, 16-byte staging and fragment reads, the transposing LDS read for the X^T . M products.  Ranges: one
// power-of-two scale per TENSOR from its absmax record (amax.hpp: qkv, dO); the register operands take static bounds —
// probabilities <= 1 (2^13), |dS| <= 2 D max|v| max|dO| / sqrt(D) — every scale is divided out exactly where a result is used.
struct SplitScales {
  float s_q, inv_q, s_d, inv_d, s_ds, inv_ds;
};
constexpr float kProbScale = 8192.f, kProbInv = 1.f / 8192.f;

__device__ __forceinline__ void split_pack(const float (&x)[8], float s, Pack8& hi, Pack8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float y = x[j] * s;
    hi.h[j] = (lo16_t)y;
    lo.h[j] = (lo16_t)(y - (float)hi.h[j]);
  }
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <int D>
__device__ __forceinline__ void put_split(lo16_t* hi_img, lo16_t* lo_img, int t, int c8, const float (&v)[8], float s) {
  Pack8 hi, lo;
  split_pack(v, s, hi, lo);
  put_row<D>(hi_img, t, c8, hi);
  put_row<D>(lo_img, t, c8, lo);
}

// q, k, v (and dO, with delta = sum_d dO O) of one (window, head): every 16-byte load of an item is issued before the first
// use; padded tokens take the qkv bias (dO: zero).  C % 4 == 0, 16-byte aligned tensors (head columns are multiples of 8).
template <int D, bool BWD>
__device__ __forceinline__ void stage_split(const WinGeom& g, const BlockId& id, const int* __restrict__ pix_lds,
                                            const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                            const float* __restrict__ out, const float* __restrict__ grad_out, int col,
                                            float s_q, float s_d, lo16_t* q_img, lo16_t* k_img, lo16_t* v_img,
                                            lo16_t* do_img, double* delta_s) {
  constexpr int CH = D / 8, IMG = Swz<D>::IMG;
  const int C3 = 3 * g.C;
  for (int idx = threadIdx.x; idx < NPAD * CH; idx += blockDim.x) {
    const int t = idx / CH, c8 = (idx - t * CH) * 8;
    float q[8], k[8], v[8], d[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = k[j] = v[j] = d[j] = o[j] = 0.f;
    bool real = false;
    if (t < g.N) {
      const int pix = pix_lds[t];
      if (pix >= 0) {
        real = true;
        const int64_t row = (int64_t)id.b * g.H * g.W + pix;
        const float* p = qkv + row * C3 + col + c8;
        load8(p, q);
        load8(p + g.C, k);
        load8(p + 2 * g.C, v);
        if (BWD) {
          load8(grad_out + row * g.C + col + c8, d);
          load8(out + row * g.C + col + c8, o);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          q[j] = qkv_bias[col + c8 + j];
          k[j] = qkv_bias[g.C + col + c8 + j];
          v[j] = qkv_bias[2 * g.C + col + c8 + j];
        }
      }
    }
    put_split<D>(q_img, q_img + IMG, t, c8, q, s_q);
    put_split<D>(k_img, k_img + IMG, t, c8, k, s_q);
    put_split<D>(v_img, v_img + IMG, t, c8, v, s_q);
    if (BWD) {
      put_split<D>(do_img, do_img + IMG, t, c8, d, s_d);
      if (real) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += o[j] * d[j];
        atomicAdd(&delta_s[t], (double)acc);
      }
    }
  }
}

// the maximum of the bits of |stored values| of a workgroup -> ONE no-return atomic into its slot of the record
__device__ __forceinline__ void publish_absmax(unsigned mine, unsigned* red /* >= 16 words of LDS */, unsigned* record) {
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const unsigned o2 = (unsigned)__shfl_xor((int)mine, sft, 64);
    mine = mine > o2 ? mine : o2;
  }
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned m = 0u;
    for (int i = 0; i < nw; ++i) m = m > red[i] ? m : red[i];
    if (m) atomicMax(record + (blockIdx.x & (kAmaxSlots - 1)), m);
  }
}

template <int D>
__global__ void __launch_bounds__(256) k_window_attn_split_fwd(const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                               const float* __restrict__ bias_table, WinGeom g, float scale,
                                                               const unsigned* __restrict__ amax_qkv,
                                                               float* __restrict__ out, float* __restrict__ lse) {
  constexpr int IMG = Swz<D>::IMG;
  __shared__ __attribute__((aligned(16))) lo16_t k_img[2 * IMG];      // hi image, lo image
  __shared__ __attribute__((aligned(16))) lo16_t q_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t v_img[2 * IMG];
  __shared__ float tbl[21 * 21];
  __shared__ int kinfo[NPAD];
  __shared__ int pix[NPAD];
  const BlockId id = decode_block(g);
  const int tsz = (2 * g.ws - 1) * (2 * g.ws - 1);
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) tbl[i] = bias_table[i * g.heads + id.head] * kLog2e;
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) {
    if (t < g.N) {
      kinfo[t] = token_info(g, id, t);
      pix[t] = token_pixel(g, id.wy, id.wx, t);
    } else {
      kinfo[t] = 0;
      pix[t] = -1;
    }
  }
  float inv_q;
  const float s_q = pow2_scale(amax_qkv, inv_q);
  __syncthreads();
  const int col = id.head * D;
  stage_split<D, false>(g, id, pix, qkv, qkv_bias, nullptr, nullptr, col, s_q, 1.f, q_img, k_img, v_img, nullptr, nullptr);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nblk = (g.N + 31) / 32;
  if (wave >= nblk) return;
  const int q = 32 * wave + r;
  f32x16 s[NBLK];
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    s[kb] = zero16();
    if (kb < nblk) mma_rows_split<D>(k_img, k_img + IMG, 32 * kb, q_img, q_img + IMG, 32 * wave, s[kb]);
  }
  const float sl2 = scale * kLog2e * inv_q * inv_q;            // the operands' scales leave with the softmax scale
  const int idx0 = (g.ws - 1) * 2 * g.ws;
  const int qi = kinfo[q];
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 32 * kb + acc_row(i, h);
      float v = fmaf(s[kb][i], sl2, bias_mask2(tbl, qi, kinfo[k], idx0, nullptr));
      v = (q < g.N) ? v : 0.f;
      v = (kb < nblk && k < g.N) ? v : -INFINITY;
      s[kb][i] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(s[kb][i] - m);
      s[kb][i] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
    for (int i = 0; i < 16; ++i) s[kb][i] *= inv;
  if (h == 0 && q < g.N) lse[(int64_t)blockIdx.x * NPAD + q] = (m + __log2f(sum)) * kLn2;

  constexpr int NCB = (D + 31) / 32;
  f32x16 o[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    o[cb] = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb)
      if (kb < nblk) mma_acc_tr_split<D>(s[kb], kProbScale, v_img, v_img + IMG, 32 * kb, cb, o[cb]);
  }
  const float o_inv = kProbInv * inv_q;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int dcol = r + 32 * cb;
    if (dcol >= D) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = 32 * wave + acc_row(i, h);
      if (qq < g.N) {
        const int px = pix[qq];
        if (px >= 0) out[((int64_t)id.b * g.H * g.W + px) * g.C + col + dcol] = o[cb][i] * o_inv;
      }
    }
  }
}

template <int D>
__global__ void __launch_bounds__(512, (D <= 32 ? 4 : 2)) k_window_attn_split_bwd(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
    const float* __restrict__ out, const float* __restrict__ grad_out, const float* __restrict__ lse, WinGeom g, float scale,
    const unsigned* __restrict__ amax_qkv, const unsigned* __restrict__ amax_do, float* __restrict__ grad_qkv,
    float* __restrict__ grad_table, float* __restrict__ grad_pad, int full_bias, unsigned* __restrict__ amax_out) {
  constexpr int IMG = Swz<D>::IMG;
  __shared__ __attribute__((aligned(16))) lo16_t q_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t k_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t v_img[2 * IMG];
  __shared__ __attribute__((aligned(16))) lo16_t do_img[2 * IMG];
  __shared__ float tbl[21 * 21];
  __shared__ double dtbl[21 * 21];
  __shared__ int kinfo[NPAD];
  __shared__ int pix[NPAD];
  __shared__ float lse_s[NPAD];
  __shared__ double delta_s[NPAD];
  __shared__ float2 ld_s[NPAD];
  __shared__ double colacc[3 * D];
  __shared__ unsigned red[16];
  const BlockId id = decode_block(g);
  const int tsz = (2 * g.ws - 1) * (2 * g.ws - 1);
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) {
    tbl[i] = bias_table[i * g.heads + id.head] * kLog2e;
    dtbl[i] = 0.0;
  }
  for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) colacc[i] = 0.0;
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) {
    if (t < g.N) {
      kinfo[t] = token_info(g, id, t);
      pix[t] = token_pixel(g, id.wy, id.wx, t);
      lse_s[t] = lse[(int64_t)blockIdx.x * NPAD + t];
    } else {
      kinfo[t] = 0;
      pix[t] = -1;
      lse_s[t] = 0.f;
    }
    delta_s[t] = 0.0;
  }
  // scales: the tensors' (their maxima in [2^13, 2^14)) and the bound of |dS| / sqrt(D): 2 D max|v| max|dO| / sqrt(D) <
  // 2^(3 + log2 D) 2^e(v) 2^e(dO) with max < 2^(e + 1), i.e. scale exponent 13 - (3 + log2 D + e_v + e_d)
  const int eq = pow2_scale_exp(amax_qkv), ed = pow2_scale_exp(amax_do);      // biased: scale = 2^(e - 127), e = 140 - e_max
  constexpr int LOG2D = D == 16 ? 4 : (D == 32 ? 5 : 6);
  const int eds = eq + ed - 127 - 13 - 3 - LOG2D;
  SplitScales sc;
  sc.s_q = pow2_from_exp(eq); sc.inv_q = pow2_from_exp(254 - eq);
  sc.s_d = pow2_from_exp(ed); sc.inv_d = pow2_from_exp(254 - ed);
  const int eds_c = eds < 1 ? 1 : (eds > 253 ? 253 : eds);
  sc.s_ds = pow2_from_exp(eds_c); sc.inv_ds = pow2_from_exp(254 - eds_c);
  __syncthreads();
  const int C3 = 3 * g.C, col = id.head * D;
  stage_split<D, true>(g, id, pix, qkv, qkv_bias, out, grad_out, col, sc.s_q, sc.s_d, q_img, k_img, v_img, do_img, delta_s);
  __syncthreads();
  for (int t = threadIdx.x; t < NPAD; t += blockDim.x) ld_s[t] = make_float2(lse_s[t] * kLog2e, (float)delta_s[t]);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, part = threadIdx.x >> 8;
  const int r = lane & 31, h = lane >> 5;
  const int nblk = (g.N + 31) / 32;
  constexpr int NCB = (D + 31) / 32;
  const int64_t row0 = (int64_t)id.b * g.H * g.W;
  const float sl2 = scale * kLog2e * sc.inv_q * sc.inv_q;
  const float dp_inv = sc.inv_q * sc.inv_d;
  const int idx0 = (g.ws - 1) * 2 * g.ws;
  unsigned out_max = 0u;

  if (wave < nblk && part == 0) {
    // ---- part 1: lane = query.  dS^T tiles, relative-position-bias gradient, dQ
    const int q = 32 * wave + r;
    const float my_lse2 = ld_s[q].x, my_delta = ld_s[q].y;
    const int qi = kinfo[q];
    f32x16 dq[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dq[cb] = zero16();
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
      if (kb >= nblk) continue;
      f32x16 s = zero16(), dp = zero16();
      mma_rows_split<D>(k_img, k_img + IMG, 32 * kb, q_img, q_img + IMG, 32 * wave, s);
      mma_rows_split<D>(v_img, v_img + IMG, 32 * kb, do_img, do_img + IMG, 32 * wave, dp);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = 32 * kb + acc_row(i, h);
        int idx;
        const float bm = bias_mask2(tbl, qi, kinfo[k], idx0, &idx);
        const float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, bm - my_lse2));
        const bool valid = q < g.N && k < g.N;
        const float ds = valid ? p * (dp[i] * dp_inv - my_delta) : 0.f;
        if (valid) atomicAdd(&dtbl[idx], (double)ds);
        s[i] = ds * scale;                       // dQ = scale * dS K
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) mma_acc_tr_split<D>(s, sc.s_ds, k_img, k_img + IMG, 32 * kb, cb, dq[cb]);
    }
    const float dq_inv = sc.inv_ds * sc.inv_q;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int dcol = r + 32 * cb;
      if (dcol >= D) continue;
      float csum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * wave + acc_row(i, h);
        if (qq >= g.N) continue;
        const int px = pix[qq];
        const float v = dq[cb][i] * dq_inv;
        if (px >= 0) {
          grad_qkv[(row0 + px) * C3 + col + dcol] = v;
          const unsigned b = __float_as_uint(v) & 0x7fffffffu;
          out_max = out_max > b ? out_max : b;
        }
        if (px < 0 || full_bias) csum += v;
      }
      csum += __shfl_xor(csum, 32, 64);
      if (h == 0 && csum != 0.f) atomicAdd(&colacc[dcol], (double)csum);
    }
  }
  if (wave < nblk && part == 1) {
    // ---- part 2: wave = key block.  dK, dV
    const int kb = wave;
    f32x16 dk[NCB], dv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { dk[cb] = zero16(); dv[cb] = zero16(); }
    const int k = 32 * kb + r;
    const int ki = kinfo[k];
#pragma unroll
    for (int qb = 0; qb < NBLK; ++qb) {
      if (qb >= nblk) continue;
      f32x16 s = zero16(), dp = zero16();
      mma_rows_split<D>(q_img, q_img + IMG, 32 * qb, k_img, k_img + IMG, 32 * kb, s);
      mma_rows_split<D>(do_img, do_img + IMG, 32 * qb, v_img, v_img + IMG, 32 * kb, dp);
      f32x16 ds;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = 32 * qb + acc_row(i, h);
        const float2 ld = ld_s[qq];
        float p = __builtin_amdgcn_exp2f(fmaf(s[i], sl2, bias_mask2(tbl, kinfo[qq], ki, idx0, nullptr) - ld.x));
        p = (qq < g.N && k < g.N) ? p : 0.f;
        s[i] = p;
        ds[i] = p * (dp[i] * dp_inv - ld.y) * scale;
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        mma_acc_tr_split<D>(s, kProbScale, do_img, do_img + IMG, 32 * qb, cb, dv[cb]);      // dV = P^T dO
        mma_acc_tr_split<D>(ds, sc.s_ds, q_img, q_img + IMG, 32 * qb, cb, dk[cb]);          // dK = scale dS^T Q
      }
    }
    const float dv_inv = kProbInv * sc.inv_d, dk_inv = sc.inv_ds * sc.inv_q;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int dcol = r + 32 * cb;
      if (dcol >= D) continue;
      float ksum = 0.f, vsum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = 32 * kb + acc_row(i, h);
        if (kk >= g.N) continue;
        const int px = pix[kk];
        const float vk = dk[cb][i] * dk_inv, vv = dv[cb][i] * dv_inv;
        if (px >= 0) {
          grad_qkv[(row0 + px) * C3 + g.C + col + dcol] = vk;
          grad_qkv[(row0 + px) * C3 + 2 * g.C + col + dcol] = vv;
          const unsigned b1 = __float_as_uint(vk) & 0x7fffffffu, b2 = __float_as_uint(vv) & 0x7fffffffu;
          const unsigned b = b1 > b2 ? b1 : b2;
          out_max = out_max > b ? out_max : b;
        }
        if (px < 0 || full_bias) { ksum += vk; vsum += vv; }
      }
      ksum += __shfl_xor(ksum, 32, 64);
      vsum += __shfl_xor(vsum, 32, 64);
      if (h == 0 && ksum != 0.f) atomicAdd(&colacc[D + dcol], (double)ksum);
      if (h == 0 && vsum != 0.f) atomicAdd(&colacc[2 * D + dcol], (double)vsum);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) atomicAdd(&grad_table[i * g.heads + id.head], (float)dtbl[i]);
  for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) {
    const double v = colacc[i];
    if (v != 0.0) atomicAdd(&grad_pad[(i / D) * g.C + col + (i % D)], (float)v);
  }
  if (amax_out) publish_absmax(out_max, red, amax_out);
}
#endif

}  // namespace

#ifndef MBV_H16
extern "C" int64_t mbv_window_attn_lse_elems(int32_t batch, int32_t H, int32_t W, int32_t heads, int32_t ws) {
  if (batch <= 0 || H <= 0 || W <= 0 || heads <= 0 || ws <= 0) return 0;
  const int64_t nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
  return (int64_t)batch * nWh * nWw * heads * NPAD;
}

// the half build of this file (window_attn_f16.hip); `is_bf16` = MBV_DT_F16 forwards there
MBV_F16_TWIN int mbv_window_attn_fwd_f16(const void*, const float*, const float*, int32_t, int32_t, int32_t, int32_t, int32_t,
                                         int32_t, int32_t, int32_t, void*, float*, void*);
MBV_F16_TWIN int mbv_window_attn_bwd_f16(const void*, const float*, const float*, const void*, const void*, const float*,
                                         int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, void*,
                                         float*, float*, int32_t, int32_t, void*);
#endif

MBV_ENTRY int MBV_SYM(mbv_window_attn_fwd)(const void* qkv, const float* qkv_bias, const float* bias_table, int32_t is_bf16,
                                   int32_t batch, int32_t H, int32_t W, int32_t C, int32_t heads, int32_t ws,
                                   int32_t shift, void* out, float* lse, void* stream_) {
#ifndef MBV_H16
  if (is_bf16 == MBV_DT_F16)
    return mbv_window_attn_fwd_f16(qkv, qkv_bias, bias_table, 1, batch, H, W, C, heads, ws, shift, out, lse, stream_);
#endif
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WinGeom g;
  if (!make_geom(batch, H, W, C, heads, ws, shift, g)) return MBV_ERR_BAD_ARG;
  if (g.N > NPAD || ws > 11) return MBV_ERR_UNSUPPORTED;
  if (!qkv || !qkv_bias || !bias_table || !out || !lse) return MBV_ERR_BAD_ARG;
  const int D = C / heads;
  return is_bf16 ? launch_fwd<true, lo16_t>(g, D, qkv, qkv_bias, bias_table, out, lse, stream)
                 : launch_fwd<false, float>(g, D, qkv, qkv_bias, bias_table, out, lse, stream);
}

MBV_ENTRY int MBV_SYM(mbv_window_attn_bwd)(const void* qkv, const float* qkv_bias, const float* bias_table, const void* out,
                                   const void* grad_out, const float* lse, int32_t is_bf16, int32_t batch, int32_t H,
                                   int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift, void* grad_qkv,
                                   float* grad_table, float* grad_qkv_bias, int32_t full_bias_grad,
                                   int32_t accumulate, void* stream_) {
#ifndef MBV_H16
  if (is_bf16 == MBV_DT_F16)
    return mbv_window_attn_bwd_f16(qkv, qkv_bias, bias_table, out, grad_out, lse, 1, batch, H, W, C, heads, ws, shift,
                                   grad_qkv, grad_table, grad_qkv_bias, full_bias_grad, accumulate, stream_);
#endif
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WinGeom g;
  if (!make_geom(batch, H, W, C, heads, ws, shift, g)) return MBV_ERR_BAD_ARG;
  if (g.N > NPAD || ws > 11) return MBV_ERR_UNSUPPORTED;
  if (!qkv || !qkv_bias || !bias_table || !out || !grad_out || !lse || !grad_qkv || !grad_table || !grad_qkv_bias)
    return MBV_ERR_BAD_ARG;
  const int D = C / heads;
  const int tsz = (2 * ws - 1) * (2 * ws - 1);
  if (accumulate) {
    // grad_table / grad_qkv_bias are (arena) gradients the kernel's atomics add into: no fill, no add afterwards
  } else if (grad_qkv_bias == grad_table + (size_t)tsz * heads) {      // one allocation (ops.py): one fill launch
    MBV_CHECK_HIP(mbv_fill_async(grad_table, 0, sizeof(float) * ((size_t)tsz * heads + 3 * (size_t)C), stream));
  } else {
    MBV_CHECK_HIP(mbv_fill_async(grad_table, 0, sizeof(float) * tsz * heads, stream));
    MBV_CHECK_HIP(mbv_fill_async(grad_qkv_bias, 0, sizeof(float) * 3 * C, stream));
  }
  return is_bf16 ? launch_bwd<true, lo16_t>(g, D, qkv, qkv_bias, bias_table, out, grad_out, lse, grad_qkv, grad_table,
                                            grad_qkv_bias, full_bias_grad, stream)
                 : launch_bwd<false, float>(g, D, qkv, qkv_bias, bias_table, out, grad_out, lse, grad_qkv, grad_table,
                                            grad_qkv_bias, full_bias_grad, stream);
}

#ifdef MBV_H16
// ---- K4 on f32 tensors in the split mode (see k_window_attn_split_fwd) ----------------------------------------------------
extern "C" int mbv_window_attn_split_supported(int32_t C, int32_t heads, int32_t ws) {
  if (C <= 0 || heads <= 0 || C % heads || ws <= 0 || ws > 11) return 0;
  const int D = C / heads;
  return (D == 16 || D == 32 || D == 64) ? 1 : 0;
}

extern "C" int mbv_window_attn_split_fwd(const float* qkv, const float* qkv_bias, const float* bias_table, int32_t batch,
                                         int32_t H, int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift,
                                         const uint32_t* amax_qkv, float* out, float* lse, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WinGeom g;
  if (!make_geom(batch, H, W, C, heads, ws, shift, g)) return MBV_ERR_BAD_ARG;
  if (!mbv_window_attn_split_supported(C, heads, ws) || g.N > NPAD) return MBV_ERR_UNSUPPORTED;
  if (!qkv || !qkv_bias || !bias_table || !out || !lse) return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(qkv) | reinterpret_cast<size_t>(out)) & 15) return MBV_ERR_UNSUPPORTED;
  const int D = C / heads;
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 grid((unsigned)(g.batch * g.nWh * g.nWw * g.heads)), block(256);
  switch (D) {
    case 16: hipLaunchKernelGGL((k_window_attn_split_fwd<16>), grid, block, 0, stream, qkv, qkv_bias, bias_table, g, scale, amax_qkv, out, lse); break;
    case 32: hipLaunchKernelGGL((k_window_attn_split_fwd<32>), grid, block, 0, stream, qkv, qkv_bias, bias_table, g, scale, amax_qkv, out, lse); break;
    default: hipLaunchKernelGGL((k_window_attn_split_fwd<64>), grid, block, 0, stream, qkv, qkv_bias, bias_table, g, scale, amax_qkv, out, lse); break;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_window_attn_split_bwd(const float* qkv, const float* qkv_bias, const float* bias_table, const float* out,
                                         const float* grad_out, const float* lse, int32_t batch, int32_t H, int32_t W,
                                         int32_t C, int32_t heads, int32_t ws, int32_t shift, const uint32_t* amax_qkv,
                                         const uint32_t* amax_grad_out, float* grad_qkv, float* grad_table,
                                         float* grad_qkv_bias, int32_t full_bias_grad, int32_t accumulate,
                                         uint32_t* amax_grad_qkv, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WinGeom g;
  if (!make_geom(batch, H, W, C, heads, ws, shift, g)) return MBV_ERR_BAD_ARG;
  if (!mbv_window_attn_split_supported(C, heads, ws) || g.N > NPAD) return MBV_ERR_UNSUPPORTED;
  if (!qkv || !qkv_bias || !bias_table || !out || !grad_out || !lse || !grad_qkv || !grad_table || !grad_qkv_bias)
    return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(qkv) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(grad_out)) & 15)
    return MBV_ERR_UNSUPPORTED;
  const int D = C / heads;
  const int tsz = (2 * ws - 1) * (2 * ws - 1);
  if (accumulate) {
  } else if (grad_qkv_bias == grad_table + (size_t)tsz * heads) {
    MBV_CHECK_HIP(mbv_fill_async(grad_table, 0, sizeof(float) * ((size_t)tsz * heads + 3 * (size_t)C), stream));
  } else {
    MBV_CHECK_HIP(mbv_fill_async(grad_table, 0, sizeof(float) * tsz * heads, stream));
    MBV_CHECK_HIP(mbv_fill_async(grad_qkv_bias, 0, sizeof(float) * 3 * C, stream));
  }
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 grid((unsigned)(g.batch * g.nWh * g.nWw * g.heads)), block(512);
#define MBV_K4S_BWD(DD)                                                                                                      \
  hipLaunchKernelGGL((k_window_attn_split_bwd<DD>), grid, block, 0, stream, qkv, qkv_bias, bias_table, out, grad_out, lse, g, \
                     scale, amax_qkv, amax_grad_out, grad_qkv, grad_table, grad_qkv_bias, full_bias_grad, amax_grad_qkv)
  switch (D) {
    case 16: MBV_K4S_BWD(16); break;
    case 32: MBV_K4S_BWD(32); break;
    default: MBV_K4S_BWD(64); break;
  }
#undef MBV_K4S_BWD
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
#endif
