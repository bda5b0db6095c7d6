// K13c — the matcher's mask-cost products straight from the sampled logits.
//
// HungarianAssigner's mask costs (mmdet CrossEntropyLossCost(use_sigmoid) 5.0 + DiceCost(pred_act, eps 1) 5.0 as evaluated
// at mask_bev/models/networks/mask2former_head/mask2former_head.py:199-205) need, per (decoder output, image) group and
// (query q, target column c) pair,
//     x[q] . t[c]      (BCE:  sum softplus(x) - x . t,  since softplus(-x) t + softplus(x) (1 - t) = softplus(x) - x t)
//     sig(x)[q] . t[c] (dice numerator)          and the row sums   sum softplus(x)[q],  sum sig(x)[q],  sum t[c]
// over the P sampled points.  Until round 3 that was a pass writing three (groups, Q, P) f32 term planes (604 MB at the
// bench batch) and an f32 library GEMM reading them (≈ 0.5 ms together).  Here the terms never leave the CU:
//   * a workgroup owns one group and one slice of the points; per chunk of 32 points it evaluates sigmoid / softplus of the
//     Q x 32 logits on the VALU, splits x, sig(x) and t into an IEEE-half pair (hi = half(v), lo = half(v - hi): 22
//     significant bits) and writes the pairs into LDS in MFMA fragment order;
//   * A = [x ; sig(x) ; ones] (2Q + 1 rows), B = [t ; ones] (G + 1 columns), C += A_hi B_hi + A_hi B_lo + A_lo B_hi on
//     v_mfma_f32_32x32x16_f16 with f32 accumulation — the dropped lo.lo term is 2^-22 relative, below the f32 rounding of the
//     12 544-term sums themselves.  The ones row / column return sum t, sum x and sum sig(x) from the same products;
//     sum softplus(x) is a VALU sum;
//   * every workgroup stores its partial (2Q + 1) x (G + 1) products; mbv_match_cost_split adds the slices in a fixed order
//     (no atomics: bit-reproducible) while it assembles the cost matrix.
// VALU-bound (three transcendentals per logit): ≈ 50 M logits per step.
#include "common.hpp"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kChunk = 32;          // points per chunk: two k-steps of 16
constexpr int kNTiles = 4;          // 32-column tiles of [t ; ones]: wave w owns tile w
constexpr int kMaxMTiles = 7;       // 32-row tiles of [x ; sig ; ones]: 2 Q + 1 <= 224

// fragment slot (16 bytes) of row-in-tile r, half h of k-step ks of `tile`, part 0 = hi / 1 = lo
__device__ __forceinline__ int slot(int tile, int ks, int part, int r, int h) {
  return (((tile * 2 + ks) * 2 + part) << 6) + r + 32 * h;
}

// v = hi + lo with hi = half(v) and lo = half(v - hi), two values per conversion (v_cvt_pkrtz_f16_f32: round toward zero —
// whatever hi leaves, lo carries; lo's own truncation is 2^-21 of v).  As instructions: pack, 2 unpacks, a packed subtract, pack.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float* v, h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const h2 a = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(v[j], v[j + 1]));
    const h2 b = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(v[j] - (float)a[0], v[j + 1] - (float)a[1]));
    hi[j] = a[0]; hi[j + 1] = a[1];
    lo[j] = b[0]; lo[j + 1] = b[1];
  }
}

struct Raw {
  float4 v[2];
};

__device__ __forceinline__ Raw load8(const float* p) {
  Raw r;
  r.v[0] = *reinterpret_cast<const float4*>(p);
  r.v[1] = *reinterpret_cast<const float4*>(p + 4);
  return r;
}

template <int MT>
__global__ void __launch_bounds__(kThreads, 2)
    k_match_products(const float* __restrict__ x, const float* __restrict__ t, int q, int g, int p, int splits,
                     float* __restrict__ prod, float* __restrict__ neg_sums) {
  __shared__ h8 lds_a[MT * 4 * 64];
  __shared__ h8 lds_t[kNTiles * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long grp = blockIdx.x / splits;
  const int s = blockIdx.x - (int)grp * splits;
  const int nchunks = (p + kChunk - 1) / kChunk;
  const int c0 = (int)((long)s * nchunks / splits), c1 = (int)((long)(s + 1) * nchunks / splits);
  const int rows_a = 2 * q + 1, cols_t = g + 1;

  // LDS: zeros, then the constant ones row of A (row 2q) and ones column of B (column g)
  {
    h8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
    for (int i = tid; i < MT * 4 * 64; i += kThreads) lds_a[i] = z;
    for (int i = tid; i < kNTiles * 4 * 64; i += kThreads) lds_t[i] = z;
    __syncthreads();
    h8 one;
#pragma unroll
    for (int j = 0; j < 8; ++j) one[j] = (_Float16)1.f;
    if (tid < 4) lds_a[slot((2 * q) >> 5, tid >> 1, 0, (2 * q) & 31, tid & 1)] = one;
    else if (tid < 8) lds_t[slot(g >> 5, (tid - 4) >> 1, 0, g & 31, tid & 1)] = one;
  }

  // conversion units: (row, 8-point block) — thread tid owns A units tid, tid + 256 and T units tid, tid + 256 of a chunk
  int a_row[2], t_row[2];
  const int blk = tid & 3;                              // 8-point block inside the chunk (256 % 4 == 0: same for both units)
  const float* a_ptr[2];
  const float* t_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = tid + kThreads * j;
    a_row[j] = (u >> 2) < q ? (u >> 2) : -1;
    t_row[j] = (u >> 2) < g ? (u >> 2) : -1;
    a_ptr[j] = x + (grp * q + (a_row[j] < 0 ? 0 : a_row[j])) * (long)p + blk * 8;
    t_ptr[j] = t + (grp * g + (t_row[j] < 0 ? 0 : t_row[j])) * (long)p + blk * 8;
  }
  float s_neg[2] = {0.f, 0.f};

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

  const int ks_u = blk >> 1, h_u = blk & 1;
  const int ntiles = (cols_t + 31) >> 5;

  // two register stages of raw chunks: while chunk c is converted and multiplied, the loads of chunks c + 1 and c + 2 are
  // in flight (one chunk of a workgroup is 25 KB; a single stage left the loop waiting on HBM latency every chunk)
  struct Stage {
    Raw a[2], t[2];
  };
  auto fetch = [&](int c, Stage& st) {
    const bool in = c < c1 && c * kChunk + blk * 8 < p;   // p % 8 == 0: a block is inside or outside as a whole
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (in && a_row[j] >= 0) st.a[j] = load8(a_ptr[j] + (long)c * kChunk);
      if (in && t_row[j] >= 0) st.t[j] = load8(t_ptr[j] + (long)c * kChunk);
    }
  };

  auto chunk = [&](int c, Stage& st) {
    const bool in = c * kChunk + blk * 8 < p;
    // ---- convert this chunk's logits and targets into fragment-ordered half pairs ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (a_row[j] >= 0) {
        h8 xh, xl, sh, sl;
        if (in) {
          float v[8] = {st.a[j].v[0].x, st.a[j].v[0].y, st.a[j].v[0].z, st.a[j].v[0].w,
                        st.a[j].v[1].x, st.a[j].v[1].y, st.a[j].v[1].z, st.a[j].v[1].w};
          float sg[8];
          float sn = 0.f, dprod = 1.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            // the raw v_exp_f32 / v_rcp_f32 / v_log_f32 (1 ulp): 1 + e lies in (1, 2], no denormal or range handling is
            // needed, and libm's forms of the three cost more VALU issue slots than everything else in this loop
            v[i] = __builtin_amdgcn_fmed3f(v[i], -60000.f, 60000.f);   // half's range; the costs saturate long before
            const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(v[i]));
            const float d = 1.0f + e;
            const float inv = __builtin_amdgcn_rcpf(d);
            sg[i] = v[i] >= 0.f ? inv : e * inv;
            sn += fmaxf(v[i], 0.f);
            dprod *= d;                                       // in (1, 256]: one logarithm for the eight points
          }
          s_neg[j] += fmaf(__builtin_amdgcn_logf(dprod), 0.6931471805599453f, sn);
          split8(v, xh, xl);
          split8(sg, sh, sl);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) xh[i] = xl[i] = sh[i] = sl[i] = (_Float16)0.f;
        }
        const int m0 = a_row[j], m1 = q + a_row[j];
        lds_a[slot(m0 >> 5, ks_u, 0, m0 & 31, h_u)] = xh;
        lds_a[slot(m0 >> 5, ks_u, 1, m0 & 31, h_u)] = xl;
        lds_a[slot(m1 >> 5, ks_u, 0, m1 & 31, h_u)] = sh;
        lds_a[slot(m1 >> 5, ks_u, 1, m1 & 31, h_u)] = sl;
      }
      if (t_row[j] >= 0) {
        h8 th, tl;
        if (in) {
          const float v[8] = {st.t[j].v[0].x, st.t[j].v[0].y, st.t[j].v[0].z, st.t[j].v[0].w,
                              st.t[j].v[1].x, st.t[j].v[1].y, st.t[j].v[1].z, st.t[j].v[1].w};
          split8(v, th, tl);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) th[i] = tl[i] = (_Float16)0.f;
        }
        const int n = t_row[j];
        lds_t[slot(n >> 5, ks_u, 0, n & 31, h_u)] = th;
        lds_t[slot(n >> 5, ks_u, 1, n & 31, h_u)] = tl;
      }
    }
    // the ones row / column stand for points that exist: a trailing partial chunk clears the blocks beyond p
    if (c * kChunk + kChunk > p && tid < 8) {
      h8 val;
      const int b4 = tid & 3;
      const _Float16 o = (c * kChunk + b4 * 8 < p) ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) val[i] = o;
      if (tid < 4) lds_a[slot((2 * q) >> 5, b4 >> 1, 0, (2 * q) & 31, b4 & 1)] = val;
      else lds_t[slot(g >> 5, b4 >> 1, 0, g & 31, b4 & 1)] = val;
    }
    fetch(c + 2, st);                                   // this stage is free again
    __syncthreads();
    // ---- C += A_hi B_hi + A_hi B_lo + A_lo B_hi; wave = column tile.  The three products of a row tile go back to back
    //      into the same accumulator (same-opcode accumulate chains issue without a stall) so that only one pair of A
    //      fragments is live: the registers that buys hold the second load stage ----
    if (wave < ntiles) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const h8 bh = lds_t[slot(wave, ks, 0, lane, 0)];
        const h8 bl = lds_t[slot(wave, ks, 1, lane, 0)];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const h8 ah = lds_a[slot(m, ks, 0, lane, 0)];
          const h8 al = lds_a[slot(m, ks, 1, lane, 0)];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  };

  Stage s0, s1;
  fetch(c0, s0);
  fetch(c0 + 1, s1);
  for (int c = c0; c < c1; c += 2) {
    chunk(c, s0);
    if (c + 1 < c1) chunk(c + 1, s1);
  }

  // ---- this slice's partial products and softplus sums ----
  float* out = prod + (grp * splits + s) * (long)rows_a * cols_t;
  if (wave < ntiles) {
    const int col = wave * 32 + (lane & 31), hh = lane >> 5;
    if (col < cols_t) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (row < rows_a) out[(long)row * cols_t + col] = acc[m][i];
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v = s_neg[j];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    if (blk == 0 && a_row[j] >= 0) neg_sums[(grp * splits + s) * (long)q + a_row[j]] = v;
  }
}

// cost = -2 softmax(cls)[label] + 5 (sum softplus(x) - x.t) / P + 5 (1 - (2 sig.t + 1) / (sum sig + sum t + 1)) from the
// sliced products of k_match_products (cf. k_match_cost in mask_loss.hip, which takes the library GEMM's products).
__global__ void __launch_bounds__(256) k_match_cost_split(const float* __restrict__ cls, const int64_t* __restrict__ labels,
                                                          const float* __restrict__ prod,
                                                          const float* __restrict__ neg_sums, long total, int q, int g,
                                                          int k1, int batch, int splits, float inv_points,
                                                          float* __restrict__ cost) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % g);
  const long qi = (i / g) % q, grp = i / ((long)g * q);
  const int b = (int)(grp % batch);
  const float* c = cls + (grp * q + qi) * k1;
  float mx = c[0];
  for (int k = 1; k < k1; ++k) mx = fmaxf(mx, c[k]);
  float den = 0.f;
  for (int k = 0; k < k1; ++k) den += __expf(c[k] - mx);
  const int lab = (int)labels[(long)b * g + col];
  const float prob = __expf(c[lab] - mx) / den;
  const int ct = g + 1;
  const long slice = (long)(2 * q + 1) * ct;
  float xt = 0.f, st = 0.f, s_sig = 0.f, tsum = 0.f, s_neg = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float* pg = prod + (grp * splits + s) * slice;
    xt += pg[qi * ct + col];
    st += pg[(q + qi) * ct + col];
    s_sig += pg[(q + qi) * ct + g];
    tsum += pg[2L * q * ct + col];
    s_neg += neg_sums[(grp * splits + s) * (long)q + qi];
  }
  const float bce = (s_neg - xt) * inv_points;
  const float dice = 1.0f - (2.0f * st + 1.0f) / (s_sig + tsum + 1.0f);
  cost[i] = -2.0f * prob + 5.0f * bce + 5.0f * dice;
}

template <int MT>
void launch_products(const float* x, const float* t, long groups, int q, int g, int p, int splits, float* prod,
                     float* neg_sums, hipStream_t stream) {
  hipLaunchKernelGGL(k_match_products<MT>, dim3((unsigned)(groups * splits)), dim3(kThreads), 0, stream, x, t, q, g, p,
                     splits, prod, neg_sums);
}

}  // namespace

extern "C" int mbv_match_products_supported(int32_t queries, int32_t targets, int32_t points) {
  return queries > 0 && targets > 0 && points > 0 && 2 * queries + 1 <= kMaxMTiles * 32 && targets + 1 <= kNTiles * 32 &&
         points % 8 == 0;
}

extern "C" int mbv_match_products(const float* logits, const float* targets, int64_t groups, int32_t queries,
                                  int32_t targets_n, int32_t points, int32_t splits, float* prod, float* neg_sums,
                                  void* stream) {
  if (groups < 0 || splits <= 0) return MBV_ERR_BAD_ARG;
  if (!mbv_match_products_supported(queries, targets_n, points)) return MBV_ERR_UNSUPPORTED;
  if (groups == 0) return MBV_OK;
  if (!logits || !targets || !prod || !neg_sums) return MBV_ERR_BAD_ARG;
  if (((reinterpret_cast<size_t>(logits) | reinterpret_cast<size_t>(targets)) & 15) != 0) return MBV_ERR_BAD_ARG;
  if (groups * splits > 0x7fffffffL || splits > (points + kChunk - 1) / kChunk) return MBV_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch ((2 * queries + 1 + 31) / 32) {
    case 1: launch_products<1>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
    case 2: launch_products<2>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
    case 3: launch_products<3>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
    case 4: launch_products<4>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
    case 5: launch_products<5>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
    case 6: launch_products<6>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
    default: launch_products<7>(logits, targets, groups, queries, targets_n, points, splits, prod, neg_sums, st); break;
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_match_cost_split(const float* cls, const int64_t* labels, const float* prod, const float* neg_sums,
                                    int64_t groups, int32_t queries, int32_t targets, int32_t classes_plus_one,
                                    int32_t batch, int32_t points, int32_t splits, float* cost, void* stream) {
  if (groups < 0 || queries <= 0 || targets <= 0 || classes_plus_one <= 0 || batch <= 0 || points <= 0 || splits <= 0)
    return MBV_ERR_BAD_ARG;
  if (groups == 0) return MBV_OK;
  if (!cls || !labels || !prod || !neg_sums || !cost) return MBV_ERR_BAD_ARG;
  const long total = groups * (long)queries * targets;
  hipLaunchKernelGGL(k_match_cost_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cls,
                     labels, prod, neg_sums, total, queries, targets, classes_plus_one, batch, splits,
                     1.0f / (float)points, cost);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
