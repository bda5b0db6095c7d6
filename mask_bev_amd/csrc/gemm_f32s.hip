// K20 — f32 GEMMs on the 16-bit matrix pipe: every f32 operand element is split, while its tile is staged, into an
// IEEE-half PAIR  x * 2^e = hi + lo  (hi = half(x 2^e), lo = half(x 2^e - hi): 22 significant bits) and the product is
// formed as  hi.hi + hi.lo + lo.hi  on v_mfma_f32_32x32x16_f16 with f32 accumulation — three matrix instructions at 16x the
// rate of v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md § Matrix cores: the exact-f32 MFMA runs at the f32 VECTOR rate,
// 157 TFLOP/s, and the library's f32 GEMMs already sit at 0.72 of it — VERDICT r04 #3).  The dropped lo.lo term is 2^-22
// relative.  It serves the reference-precision (fp32) step's token-major Linears — the Swin qkv / proj / fc1 / fc2 / patch
// merging projections (/root/reference: mask_bev/models/networks/swin/swin.py:89-116, 347-355, 611-616), the pixel
// decoder's Linears (mask_bev/models/head/mask_bev_panoptic_head.py:119-146) — in the three layouts of K17 (gemm.hip):
//
//   NT  C[m][n] = sum_k X[m][k] W[n][k]          forward Linear
//   NN  C[m][k] = sum_n G[m][n] W[n][k]          data gradient
//   TN  C[n][k] = sum_m G[m][n] X[m][k]          weight gradient (split over m: parts stored, owner adds — no atomics)
//
// Range: half has 5 exponent bits, so each operand is scaled by a power of two 2^e that puts its largest magnitude in
// [2^13, 2^14) — PER TENSOR, from one device word holding the bits of max|x| (mbv_f32_absmax: one streaming pass, an
// integer atomic max per workgroup; weights: once per optimizer step).  An element 2^-j of the maximum keeps 11 + 11 bits
// for j <= 18 and loses one bit per binade below that (lo becomes subnormal; 2^-24 absolute on the scaled values, i.e.
// 2^-38 of the operand's maximum): the error of a product row is <= 2^-22 sum|a||b| + K 2^-37 max|a| max|b|.  The scales
// are exact (powers of two) and are divided out of the f32 accumulators in the epilogue.
//
// Structure: K17's 128 x 128 x 32 block (4 waves as 2 x 2, wave tile 64 x 64 = 2 x 2 MFMA tiles), but the operands arrive
// through REGISTERS (buffer_load_dwordx4 with the descriptor's range check: ragged edges read as zeros) because they have
// to be split on the way: one register set, step kt + 1 is split and written to the other LDS stage behind step kt's
// matrix instructions, step kt + 2 is requested right after (cdna_hip_programming.md §5, T14 as G15 writes it); the LDS
// images are K17's swizzled 16-bit images — one for the hi halves, one for the lo halves of each operand — so the
// fragment reads (ds_read_b128 rows / ds_read_b64_tr_b16 transposed) are K17's.  Each element is split ONCE per workgroup
// (6 VALU instructions per pair: v_pk_mul, v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_fma, v_cvt_pk_f16_f32).

#include "common.hpp"
#include "gemm_tiles.hpp"
#include "amax.hpp"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int KB32 = 32;                          // K-step
constexpr int IMG32 = 128 * KB32 * 2;             // one 16-bit image of a 128-wide operand tile: 8 KB
constexpr int STAGE32 = 4 * IMG32;                // A hi, A lo, B hi, B lo
constexpr int LDS32 = 2 * STAGE32;                // two stages: 64 KB -> two workgroups per CU
constexpr int NLD = 4;                            // 16-byte loads per thread and operand tile (128 x 32 f32 / 256 threads)

enum { EPI32_NONE = 0, EPI32_RELU = 1, EPI32_GELU = 2, EPI32_DRELU = 3, EPI32_DGELU = 4 };

struct Gemm32Args {
  const float* a;
  const float* b;
  float* c;             // (gm, gn) row-major f32
  float* c2;            // EPI32_RELU / EPI32_GELU: optional pre-activation output
  const float* bias;    // optional, (gn)
  const unsigned* amax_a;   // kAmaxSlots device words whose maximum is the bits of (a bound of) max|a| (NULL: scale 1)
  const unsigned* amax_b;
  unsigned* amax_out;       // optional: the bits of max|stored c| are max-combined into these kAmaxSlots words
  const float* aux;         // EPI32_DRELU / EPI32_DGELU: the saved pre-activation (gm, gn) f32, row stride ldx
  float* colsum_rows;       // EPI32_D*: (2 ntm, gn) f32 — the column sums of the stored values per 64-row wave row (the bias
                            // gradient of the layer in front, reduced by the caller: no atomics)
  int ldx;
  int gm, gn, gk;
  int lda, ldb, ldc;
  long long sa, sb, sc;     // batch strides (elements)
  long long ssplit;         // splits > 1: element stride between the partial results of the K parts
  unsigned a_bytes, b_bytes;
  int ntm, ntn, splits, ksteps;
  int acc_out;              // splits == 1: the product is ADDED to c (read-modify-write by the owning workgroup)
  int pc, ph, pw;           // GATHER modes: the (B, pc, ph, pw) NCHW image behind the 4 x 4 patch rows
  int tap[9], tap_steps;    // GATHER 4: byte offset added to A's rows for the K-steps of tap t (tap_steps K-steps each)
};

// The rows of a non-overlapping 4 x 4 patch projection (mmdet PatchEmbed: Conv2d(C, E, 4, stride 4) on an NCHW image,
// /root/reference: mask_bev/models/networks/swin/swin.py:579-586) are never materialised: element (token, k') of the
// (B * H/4 * W/4, 16 C) row matrix, k' = c * 16 + dy * 4 + dx — the flattening of the (E, C, 4, 4) weight — IS image element
// (b, c, 4 oy + dy, 4 ox + dx).  Four consecutive k' (dx = 0 .. 3) are 16 contiguous bytes, so a 16-byte operand piece of
// the GEMM loaders is one aligned load from the image, and 16-byte pieces of consecutive tokens of an image row are
// contiguous (the thread maps below give consecutive lanes consecutive tokens).
//   GATHER 1: NT  out (tokens, E) = rows . W^T          — A is the gathered row matrix
//   GATHER 2: NN  d image = (d out (tokens, E) . W) scattered back to NCHW — C is the scattered row matrix
//   GATHER 3: TN  dW (E, 16 C) += d out^T . rows        — B is the gathered row matrix (contraction over tokens)
//   GATHER 4: NT  a 3 x 3 convolution on a zero-bordered channels-last image as ONE product over k = (tap, channel): the A
//             row of output position m for tap t is row m + shift_t of the same matrix — a per-K-step scalar offset
__device__ __forceinline__ unsigned patch_elem(const Gemm32Args& p, int token, int kp) {
  const int tw = p.pw >> 2, th = p.ph >> 2;
  const int b = token / (th * tw), rem = token - b * th * tw;
  const int oy = rem / tw, ox = rem - oy * tw;
  const int c = kp >> 4, dy = (kp >> 2) & 3;
  return (unsigned)(((b * p.pc + c) * p.ph + 4 * oy + dy) * p.pw + 4 * ox + (kp & 3));
}

template <bool KS>
__device__ __forceinline__ unsigned load_offset(int i, int tid, int x0, int x_total, int ld_bytes, int k0, int k_end) {
  const int q = tid + 256 * i;
  if (KS) {                                       // [k rows][128 cols] f32: 32 float4 per k row, 8 rows per pass
    const int gk = k0 + (q >> 5), gx = x0 + 4 * (q & 31);
    return (gk < k_end && gx < x_total) ? (unsigned)gk * (unsigned)ld_bytes + (unsigned)gx * 4u : OOB;
  }
  const int gx = x0 + (q >> 3), gk = k0 + 4 * (q & 7);      // [128 rows][32 k] f32: 8 float4 per row, 32 rows per pass
  return (gx < x_total && gk < k_end) ? (unsigned)gx * (unsigned)ld_bytes + (unsigned)gk * 4u : OOB;
}

// where this thread's 4 halves of load i go in K17's 16-bit image (gemm_tiles.hpp: kc_lane<32> / ks_lane swizzles)
template <bool KS>
__device__ __forceinline__ unsigned image_offset(int i, int tid) {
  const int q = tid + 256 * i;
  if (KS) {
    const int r = q >> 5, cq = q & 31;
    return (unsigned)(r * 256 + ((((cq >> 1) ^ ks_swz(r))) << 4) + (cq & 1) * 8);
  }
  const int r = q >> 3, kq = q & 7;
  return (unsigned)(r * 64 + ((((kq >> 1) ^ ((r >> 2) & 3))) << 4) + (kq & 1) * 8);
}

__device__ __forceinline__ void split4(const u32x4 raw, const float s, uint2& hi, uint2& lo) {
  const f32x4 f = __builtin_bit_cast(f32x4, raw) * s;
  const f32x2 f01 = {f[0], f[1]}, f23 = {f[2], f[3]};
  const f16x2 h01 = __builtin_convertvector(f01, f16x2), h23 = __builtin_convertvector(f23, f16x2);
  const f32x2 r01 = f01 - __builtin_convertvector(h01, f32x2), r23 = f23 - __builtin_convertvector(h23, f32x2);
  const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

__device__ __forceinline__ f32x16 mma16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// One workgroup's share of a product: work item `bid` of its (batch, split, tile_m, tile_n) index.
template <bool A_KS, bool B_KS, int EPI, int GATHER = 0>
__device__ __forceinline__ void gemm32s_body(const Gemm32Args& p, const int bid, char* smem) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles = p.ntm * p.ntn;
  const int z = bid / tiles, t = bid - z * tiles;
  const int tile_m = t / p.ntn, tile_n = t - tile_m * p.ntn;
  const int bz = z / p.splits, sp = z - bz * p.splits;
  const int k_begin = sp * p.ksteps * KB32;
  int k_end = k_begin + p.ksteps * KB32;
  if (k_end > p.gk) k_end = p.gk;
  const int nk = (k_end - k_begin + KB32 - 1) / KB32;
  if (nk <= 0) return;

  float inv_a, inv_b;
  const float sca = pow2_scale(p.amax_a, inv_a), scb = pow2_scale(p.amax_b, inv_b);

  const float* abase = p.a + (size_t)bz * (size_t)p.sa;
  const float* bbase = p.b + (size_t)bz * (size_t)p.sb;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bbase), 0, (int)p.b_bytes, 0x00020000);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bool ragged = ((k_end - k_begin) & (KB32 - 1)) != 0;
  // (GATHER 1: a K-step of 32 k' is two channels of the image; GATHER 3's token steps are not uniform in the image: its
  // offsets are formed per step)
  const unsigned stepa = GATHER == 1 ? (unsigned)(2 * 4) * (unsigned)p.ph * (unsigned)p.pw
                                     : (A_KS ? (unsigned)(KB32 * 4) * (unsigned)p.lda : (unsigned)(KB32 * 4));
  const unsigned stepb = GATHER == 3 ? 0u : (B_KS ? (unsigned)(KB32 * 4) * (unsigned)p.ldb : (unsigned)(KB32 * 4));
  // operand-piece offsets of K-step (k0 .. kend) for this thread's load i
  auto a_offset = [&](int i, int k0, int kend) -> unsigned {
    if (GATHER == 1) {                             // lane = token (contiguous 16-byte pieces), 8 k' quads per tile row
      const int q = tid + 256 * i, token = m0 + (q & 127), kp = k0 + 4 * (q >> 7);
      return (token < p.gm && kp < kend) ? patch_elem(p, token, kp) * 4u : OOB;
    }
    return load_offset<A_KS>(i, tid, m0, p.gm, p.lda * 4, k0, kend);
  };
  auto b_offset = [&](int i, int k0, int kend) -> unsigned {
    if (GATHER == 3) {                             // [32 tokens][128 k'] tile: lane = token, 32 k' quads
      const int q = tid + 256 * i, token = k0 + (q & 31), kp = n0 + 4 * (q >> 5);
      return (token < kend && kp < p.gn) ? patch_elem(p, token, kp) * 4u : OOB;
    }
    return load_offset<B_KS>(i, tid, n0, p.gn, p.ldb * 4, k0, kend);
  };
  unsigned offa[NLD], offb[NLD], imga[NLD], imgb[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    offa[i] = a_offset(i, k_begin, k_begin + KB32);
    offb[i] = b_offset(i, k_begin, k_begin + KB32);
    const int q = tid + 256 * i;
    // (row, 16-byte piece) of this load in the 16-bit images: the gather maps give consecutive lanes consecutive rows
    imga[i] = GATHER == 1 ? (unsigned)((q & 127) * 64 + (((((q >> 7) >> 1) ^ (((q & 127) >> 2) & 3))) << 4) + ((q >> 7) & 1) * 8)
                          : image_offset<A_KS>(i, tid);
    imgb[i] = GATHER == 3 ? (unsigned)((q & 31) * 256 + (((((q >> 5) >> 1) ^ ks_swz(q & 31))) << 4) + ((q >> 5) & 1) * 8)
                          : image_offset<B_KS>(i, tid);
  }

  u32x4 rawa[NLD], rawb[NLD];
  auto request = [&](int kt, bool tail) {          // the f32 tiles of K-step kt -> registers
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      unsigned va = offa[i], vb = offb[i], sa = (unsigned)kt * stepa, sb = (unsigned)kt * stepb;
      if (GATHER == 4) sa += (unsigned)p.tap[(k_begin / KB32 + kt) / p.tap_steps];
      if (tail) {
        va = a_offset(i, k_begin + kt * KB32, k_end);
        vb = b_offset(i, k_begin + kt * KB32, k_end);
        sa = GATHER == 4 ? (unsigned)p.tap[(k_begin / KB32 + kt) / p.tap_steps] : 0u;
        sb = 0u;
      } else if (GATHER == 3) {
        vb = b_offset(i, k_begin + kt * KB32, k_end);
      }
      rawa[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)va, (int)sa, 0);
      rawb[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)vb, (int)sb, 0);
    }
  };
  auto split_store = [&](int slot) {               // registers -> hi / lo images of stage `slot`
    char* st = smem + slot * STAGE32;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      uint2 hi, lo;
      split4(rawa[i], sca, hi, lo);
      *reinterpret_cast<uint2*>(st + imga[i]) = hi;
      *reinterpret_cast<uint2*>(st + IMG32 + imga[i]) = lo;
      split4(rawb[i], scb, hi, lo);
      *reinterpret_cast<uint2*>(st + 2 * IMG32 + imgb[i]) = hi;
      *reinterpret_cast<uint2*>(st + 3 * IMG32 + imgb[i]) = lo;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragments of K-step kt (both 16-k halves): [ks][hi / lo][tile]
  uint4 fa[2][2][TM], fb[2][2][TN];
  auto read_frags = [&](int kt) {
    const char* st = smem + (kt & 1) * STAGE32;
#pragma unroll
    for (int ks = 0; ks < KB32 / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int x = 32 * (wm * TM + i);
        fa[ks][0][i] = A_KS ? frag_ks(st, x, ks, lane) : frag_kc<KB32>(st, x, ks, lane);
        fa[ks][1][i] = A_KS ? frag_ks(st + IMG32, x, ks, lane) : frag_kc<KB32>(st + IMG32, x, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int x = 32 * (wn * TN + j);
        fb[ks][0][j] = B_KS ? frag_ks(st + 2 * IMG32, x, ks, lane) : frag_kc<KB32>(st + 2 * IMG32, x, ks, lane);
        fb[ks][1][j] = B_KS ? frag_ks(st + 3 * IMG32, x, ks, lane) : frag_kc<KB32>(st + 3 * IMG32, x, ks, lane);
      }
    }
  };
  auto multiply = [&]() {                          // the 24 matrix instructions of the K-step whose fragments are loaded
#pragma unroll
    for (int ks = 0; ks < KB32 / 16; ++ks)
      // C^T accumulators (a lane holds 4 consecutive columns of C): the small cross terms first, then hi . hi
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = mma16(fb[ks][1][j], fa[ks][0][i], acc[i][j]);
          acc[i][j] = mma16(fb[ks][0][j], fa[ks][1][i], acc[i][j]);
          acc[i][j] = mma16(fb[ks][0][j], fa[ks][0][i], acc[i][j]);
        }
  };

  // Order of a K-step: step kt + 1's f32 tiles — requested a whole step ago — are split into the other stage (last read in
  // step kt - 1, behind the barrier) and step kt + 2 is requested into the freed registers WHILE this step's matrix
  // instructions run: the steady-state steps are one basic block (no tail / last-step branches) in which the scheduler is told
  // to put five vector instructions of the split behind every matrix instruction (an MFMA holds the vector issue for 8 of its
  // 32 cycles: ~ 5 single-issue instructions fit a gap, MI355X_MICROARCH.md § cycle constants; ~ 125 VALU per 24 MFMA here).
  const int n_main = nk - 2 - (ragged ? 1 : 0);    // steps whose successor's successor is a full step
  request(0, ragged && nk == 1);
  split_store(0);
  if (nk > 1) request(1, ragged && nk == 2);
  __syncthreads();
  int kt = 0;
  for (; kt < n_main; ++kt) {
    read_frags(kt);                                // (in front of the split's LDS stores: the compiler orders LDS accesses)
    char* nst = smem + ((kt + 1) & 1) * STAGE32;
    unsigned sa2 = (unsigned)(kt + 2) * stepa;
    const unsigned sb2 = (unsigned)(kt + 2) * stepb;
    if (GATHER == 4) sa2 += (unsigned)p.tap[(k_begin / KB32 + kt + 2) / p.tap_steps];
    // eight groups of { three matrix instructions (one accumulator's cross terms and hi . hi), the split of ONE 16-byte
    // piece of step kt + 1 and its two LDS stores, the request of that piece of step kt + 2 }, pinned in this order
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int ks = g >> 2, i = (g >> 1) & 1, j = g & 1;
      acc[i][j] = mma16(fb[ks][1][j], fa[ks][0][i], acc[i][j]);
      acc[i][j] = mma16(fb[ks][0][j], fa[ks][1][i], acc[i][j]);
      acc[i][j] = mma16(fb[ks][0][j], fa[ks][0][i], acc[i][j]);
      uint2 hi, lo;
      if (g < NLD) {
        split4(rawa[g], sca, hi, lo);
        *reinterpret_cast<uint2*>(nst + imga[g]) = hi;
        *reinterpret_cast<uint2*>(nst + IMG32 + imga[g]) = lo;
        rawa[g] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)offa[g], (int)sa2, 0);
      } else {
        split4(rawb[g - NLD], scb, hi, lo);
        *reinterpret_cast<uint2*>(nst + 2 * IMG32 + imgb[g - NLD]) = hi;
        *reinterpret_cast<uint2*>(nst + 3 * IMG32 + imgb[g - NLD]) = lo;
        const unsigned vb2 = GATHER == 3 ? b_offset(g - NLD, k_begin + (kt + 2) * KB32, k_end) : offb[g - NLD];
        rawb[g - NLD] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)vb2, (int)sb2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  for (; kt < nk; ++kt) {                          // the last (up to three) steps: nothing / a ragged step left to fetch
    if (kt + 1 < nk) {
      split_store((kt + 1) & 1);
      if (kt + 2 < nk) request(kt + 2, ragged && kt + 2 == nk - 1);
    }
    read_frags(kt);
    multiply();
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][e] = C[m = 64 wm + 32 i + r][n = 64 wn + 32 j + acc_row(e, h)] (K17's STORE form, f32) ----
  const int r = lane & 31, h = lane >> 5;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * STG_LD);
  const int prow = lane >> 3, cg = lane & 7;
  const int gn = n0 + 64 * wn + 8 * cg;
  const bool col_ok = gn < p.gn;                    // gn % 8 == 0 is required: a column group is all in or all out
  const size_t cbase = (size_t)bz * (size_t)p.sc + (size_t)sp * (size_t)p.ssplit;
  const int mrow0 = m0 + 32 * TM * wm + prow;
  const float inv = inv_a * inv_b;                  // each factor is a normal power of two; the product may be subnormal
  const bool two_step = !(inv >= 1.1754944e-38f);   // then scale in two exact steps
  unsigned out_max = 0u;                            // bits of the largest magnitude this thread stores (amax_out)
  float bias8[8], csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; csum[e] = 0.f; }
  if (p.bias && col_ok) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + gn);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + gn + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
    bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = acc[i][j][4 * g]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
        *reinterpret_cast<f32x4*>(stg + r * STG_LD + 32 * j + 8 * g + 4 * h) = v;
      }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = 8 * ps + prow;
      const int gm = mrow0 + 32 * i + 8 * ps;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * STG_LD + 8 * cg);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * STG_LD + 8 * cg + 4);
      if (gm >= p.gm || !col_ok) continue;
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (two_step ? (v[e] * inv_a) * inv_b : v[e] * inv) + bias8[e];
      if (GATHER == 2) {                            // the row matrix element (token gm, k' gn ..) scattered to the NCHW image
        float* d0 = p.c + patch_elem(p, gm, gn);
        float* d1 = p.c + patch_elem(p, gm, gn + 4);
        *reinterpret_cast<float4*>(d0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(d1) = make_float4(v[4], v[5], v[6], v[7]);
        continue;
      }
      const size_t o = cbase + (size_t)gm * p.ldc + gn;
      if (EPI == EPI32_DRELU || EPI == EPI32_DGELU) {     // d(pre-activation) = product * act'(pre), and its column sums
        const float* ax = p.aux + cbase + (size_t)gm * p.ldx + gn;
        const float4 z0 = *reinterpret_cast<const float4*>(ax), z1 = *reinterpret_cast<const float4*>(ax + 4);
        const float z[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (EPI == EPI32_DRELU) {
            v[e] = z[e] > 0.f ? v[e] : 0.f;
          } else {
            float dens;
            const float cdf = gelu_cdf_parts(z[e], dens);
            v[e] *= cdf + z[e] * dens;
          }
          csum[e] += v[e];
        }
      } else if (EPI != EPI32_NONE) {
        if (p.c2) {
          *reinterpret_cast<float4*>(p.c2 + o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(p.c2 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (EPI == EPI32_RELU) {
            v[e] = v[e] > 0.f ? v[e] : 0.f;
          } else {
            float dens;
            v[e] *= gelu_cdf_parts(v[e], dens);
          }
        }
      }
      if (p.acc_out) {
        const float4 d0 = *reinterpret_cast<const float4*>(p.c + o), d1 = *reinterpret_cast<const float4*>(p.c + o + 4);
        v[0] += d0.x; v[1] += d0.y; v[2] += d0.z; v[3] += d0.w;
        v[4] += d1.x; v[5] += d1.y; v[6] += d1.z; v[7] += d1.w;
      }
      if (p.amax_out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned w = __float_as_uint(v[e]) & 0x7fffffffu;
          out_max = out_max > w ? out_max : w;
        }
      }
      *reinterpret_cast<float4*>(p.c + o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(p.c + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  if ((EPI == EPI32_DRELU || EPI == EPI32_DGELU) && p.colsum_rows) {
    // one partial row per (tile_m, wave row): the lanes that own a column group meet by shuffles (lane = 8 prow + cg)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float sm = csum[e];
      sm += __shfl_xor(sm, 8, 64);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      csum[e] = sm;
    }
    if (prow == 0 && col_ok) {
      float* d = p.colsum_rows + ((size_t)bz * 2 * p.ntm + 2 * tile_m + wm) * (size_t)p.gn + gn;
      *reinterpret_cast<float4*>(d) = make_float4(csum[0], csum[1], csum[2], csum[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
    }
  }
  if (p.amax_out) {
    // ONE fire-and-forget max-combine per workgroup (the waves' maxima meet in LDS).  A read of the slot in front of the
    // atomic ("skip when it already holds more") made every wave wait for its own stores to be acknowledged — up to + 29 us
    // per launch (65 536 x 768 -> 192); a no-return atomic costs the workgroup nothing and <= 3 072 of them per launch
    // trickle in as the workgroups finish.
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      const unsigned o2 = (unsigned)__shfl_xor((int)out_max, sft, 64);
      out_max = out_max > o2 ? out_max : o2;
    }
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    __syncthreads();                                // the staging tiles above are done with
    if (lane == 0) red[wave] = out_max;
    __syncthreads();
    if (tid == 0) {
      const unsigned m01 = red[0] > red[1] ? red[0] : red[1], m23 = red[2] > red[3] ? red[2] : red[3];
      const unsigned m = m01 > m23 ? m01 : m23;
      if (m) atomicMax(p.amax_out + (bid & (kAmaxSlots - 1)), m);
    }
  }
}

template <bool A_KS, bool B_KS, int EPI, int GATHER = 0>
__global__ void __launch_bounds__(256, 2) k_gemm32s(const Gemm32Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm32s_body<A_KS, B_KS, EPI, GATHER>(p, xcd_contiguous(blockIdx.x, gridDim.x), smem);
}

// ---- grouped weight gradients: up to kTn32Group products dw_i (n_i, k_i) += g_i (m_i, n_i)^T x_i (m_i, k_i) in ONE launch ------
// A weight gradient is nobody's input, so the fp32 step's Linears hand theirs over during the backward pass and the pass
// issues them together at its end (as K17's group does for the 16-bit modes): the few-row ones (the decoder's 400 query rows
// x 9 layers, /root/reference: mask_bev/models/networks/mask2former_head/mask2former_head.py:535-560) are a handful of
// 128 x 128 tiles each, the token-major ones (Swin / pixel decoder: 1 024 - 65 536 tokens) alone fill the chip only by cutting
// their token sum into ~ 40 slivers with a round trip of partial tiles each.  Work item = (entry, token range, tile); every
// entry is cut into ranges of about kTn32Depth tokens, so the items are about equally long.  One range: the owner adds its
// tile to dw in place; several: partial tiles go to the workspace and k_add_parts32_group folds them into dw (no atomics).
constexpr int kTn32Group = 48;
constexpr int kTn32Depth = 4096;
struct Tn32Entry {
  const float* g; const float* x; float* out;     // out: dw (splits == 1) or this entry's parts in the workspace
  const unsigned* amax_g; const unsigned* amax_x;
  int m, n, k, ldg, ldx, ntn, splits, ksteps, item_begin;
};
struct Tn32GroupArgs {
  Tn32Entry e[kTn32Group];
  int n;
};

__global__ void __launch_bounds__(256, 2) k_gemm32s_tn_group(const Tn32GroupArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int item = xcd_contiguous(blockIdx.x, gridDim.x);
  int i = 0;
  while (i + 1 < a.n && item >= a.e[i + 1].item_begin) ++i;     // block-uniform: a scalar loop over <= 48 entries
  const Tn32Entry& e = a.e[i];
  Gemm32Args p;
  p.a = e.g; p.b = e.x; p.c = e.out; p.c2 = nullptr; p.bias = nullptr;
  p.amax_a = e.amax_g; p.amax_b = e.amax_x; p.amax_out = nullptr;
  p.gm = e.n; p.gn = e.k; p.gk = e.m;
  p.lda = e.ldg; p.ldb = e.ldx; p.ldc = e.k;
  p.sa = 0; p.sb = 0; p.sc = 0; p.ssplit = (long long)e.n * e.k;
  p.a_bytes = (unsigned)(((long long)(e.m - 1) * e.ldg + e.n) * 4);
  p.b_bytes = (unsigned)(((long long)(e.m - 1) * e.ldx + e.k) * 4);
  p.ntm = (e.n + 127) / 128; p.ntn = e.ntn; p.splits = e.splits; p.ksteps = e.ksteps;
  p.acc_out = e.splits == 1;
  p.pc = 0; p.ph = 0; p.pw = 0;
  gemm32s_body<true, true, EPI32_NONE, 0>(p, item - e.item_begin, smem);
}

// out (n) += sum of the `parts` partial results part (parts, n): a thread owns 4 consecutive elements (owner adds: no
// atomics, bit-reproducible)
__global__ void __launch_bounds__(256) k_add_parts32(const float* __restrict__ part, int parts, long long n,
                                                     float* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = 0; p < parts; p += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = p + u < parts ? *reinterpret_cast<const float4*>(part + (long long)(p + u) * n + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  float4 o = *reinterpret_cast<const float4*>(out + i);
  o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
  *reinterpret_cast<float4*>(out + i) = o;
}

constexpr int kParts32Group = 96;
struct Parts32Entry {
  const float* part; float* out;
  long long n;                 // elements of out (a multiple of 4)
  int parts, block_begin;
};
struct Parts32GroupArgs {
  Parts32Entry e[kParts32Group];
  int n;
};

// out_i (n_i) += sum of its parts (parts_i, n_i), for up to kParts32Group outputs; a thread owns 4 consecutive elements
__global__ void __launch_bounds__(256) k_add_parts32_group(const Parts32GroupArgs a) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.e[i + 1].block_begin) ++i;
  const Parts32Entry& e = a.e[i];
  const long long j = ((long long)(blockIdx.x - e.block_begin) * 256 + threadIdx.x) * 4;
  if (j >= e.n) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = 0; p < e.parts; p += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = p + u < e.parts ? *reinterpret_cast<const float4*>(e.part + (long long)(p + u) * e.n + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  float4 o = *reinterpret_cast<const float4*>(e.out + j);
  o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
  *reinterpret_cast<float4*>(e.out + j) = o;
}

// ---- max|x| of f32 tensors as the BITS of the maximum (non-negative floats order like integers) -------------------------
constexpr int kAbsmaxGroup = 64;
struct AbsmaxEntry {
  const float* x;
  unsigned* out;
  long long rows, cols, ld;       // cols % 4 == 0, 16-byte aligned rows
  int block_begin, blocks;
};
struct AbsmaxArgs {
  AbsmaxEntry e[kAbsmaxGroup];
  int n;
};

__global__ void __launch_bounds__(256) k_absmax_group(const AbsmaxArgs a) {
  __shared__ unsigned red[4];
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.e[i + 1].block_begin) ++i;
  const AbsmaxEntry& e = a.e[i];
  const long long c4 = e.cols >> 2, total = e.rows * c4;
  const long long stride = (long long)e.blocks * 256;
  unsigned m = 0u;
  for (long long q = (long long)(blockIdx.x - e.block_begin) * 256 + threadIdx.x; q < total; q += 4 * stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long qq = q + u * stride;
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (qq < total) {
        const long long row = qq / c4, col = qq - row * c4;
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(e.x + row * e.ld + 4 * col));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const u32x4 w = v[u] & 0x7fffffffu;
      const unsigned m0 = w[0] > w[1] ? w[0] : w[1], m1 = w[2] > w[3] ? w[2] : w[3];
      const unsigned mm = m0 > m1 ? m0 : m1;
      m = m > mm ? m : mm;
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, s, 64);
    m = m > o ? m : o;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a01 = red[0] > red[1] ? red[0] : red[1], a23 = red[2] > red[3] ? red[2] : red[3];
    const unsigned mm = a01 > a23 ? a01 : a23;
    if (mm) atomicMax(e.out + (blockIdx.x & (kAmaxSlots - 1)), mm);
  }
}

}  // namespace

static bool fits32(long long rows, long long ld) { return rows * ld * 4 < 0x7fff0000LL; }

template <bool AKS, bool BKS, int G = 0>
static int gemm32s_launch(int epi, const Gemm32Args& a, int batch, hipStream_t st) {
  const long long nblk = (long long)a.ntm * a.ntn * a.splits * batch;
  if (nblk <= 0) return MBV_OK;
  if (nblk > 0x7fffffffLL) return MBV_ERR_UNSUPPORTED;
#define MBV_G32_LAUNCH(E)                                                                                                \
  do {                                                                                                                   \
    static bool done = false;       /* idempotent attribute of the code object, not library state */                     \
    if (!done) {                                                                                                         \
      MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm32s<AKS, BKS, E, G>),                       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS32));                             \
      done = true;                                                                                                       \
    }                                                                                                                    \
    hipLaunchKernelGGL((k_gemm32s<AKS, BKS, E, G>), dim3((unsigned)nblk), dim3(256), LDS32, st, a);                       \
  } while (0)
  if constexpr (G == 0 && !AKS && !BKS) {          // the activation epilogues exist for the forward form only
    if (epi == EPI32_RELU) MBV_G32_LAUNCH(EPI32_RELU);
    else if (epi == EPI32_GELU) MBV_G32_LAUNCH(EPI32_GELU);
    else MBV_G32_LAUNCH(EPI32_NONE);
  } else if constexpr (G == 0 && !AKS && BKS) {    // the data-gradient form: optionally times the activation's derivative
    if (epi == EPI32_DRELU) MBV_G32_LAUNCH(EPI32_DRELU);
    else if (epi == EPI32_DGELU) MBV_G32_LAUNCH(EPI32_DGELU);
    else if (epi != EPI32_NONE) return MBV_ERR_UNSUPPORTED;
    else MBV_G32_LAUNCH(EPI32_NONE);
  } else {
    if (epi != EPI32_NONE) return MBV_ERR_UNSUPPORTED;
    MBV_G32_LAUNCH(EPI32_NONE);
  }
#undef MBV_G32_LAUNCH
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_gemm32s_supported(int32_t layout, int64_t m, int64_t n, int64_t k) {
  if (layout < 0 || layout > 2 || m <= 0 || n <= 0 || k <= 0) return 0;
  if ((n & 7) || (k & 7)) return 0;
  return fits32(m, n > k ? n : k) ? 1 : 0;
}

// ---- absmax BOUNDS of LayerNorm outputs ------------------------------------------------------------------------------------
// y = xhat * gamma + beta with |xhat| <= sqrt(C - 1): the record of y holds the bits of sqrt(C) max|gamma| + max|beta| — a bound
// ~ 3x above the true maximum of a 65 536 x 192 map (1.6 of the 18 binades a K20 operand keeps at full precision) that costs
// two reductions over C elements instead of a pass over the map.  One workgroup per LayerNorm, up to kLnBoundGroup per launch;
// only word 0 of the record is written (the others stay as the caller zeroed them).
constexpr int kLnBoundGroup = 96;
struct LnBoundEntry {
  const float* gamma; const float* beta; unsigned* out;
  int c;
};
struct LnBoundArgs {
  LnBoundEntry e[kLnBoundGroup];
};

namespace {
__global__ void __launch_bounds__(256) k_ln_bound_group(const LnBoundArgs a) {
  __shared__ float red[8];
  const LnBoundEntry& e = a.e[blockIdx.x];
  float mg = 0.f, mb = 0.f;
  for (int i = threadIdx.x; i < e.c; i += 256) {
    mg = fmaxf(mg, fabsf(e.gamma[i]));
    if (e.beta) mb = fmaxf(mb, fabsf(e.beta[i]));
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    mg = fmaxf(mg, __shfl_xor(mg, sft, 64));
    mb = fmaxf(mb, __shfl_xor(mb, sft, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave] = mg; red[4 + wave] = mb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mg = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    mb = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    const float bound = sqrtf((float)e.c) * mg + mb;
    e.out[0] = __float_as_uint(bound) & 0x7fffffffu;      // (NaN / inf parameters give a NaN / inf bound: the consumer clamps)
  }
}
}  // namespace

// records[i][0] = bits of sqrt(c[i]) max|gamma[i]| + max|beta[i]| (beta[i] may be NULL) for i < count; HOST arrays
extern "C" int mbv_ln_bound_group(const float* const* gamma, const float* const* beta, const int32_t* c, uint32_t* const* records,
                                  int32_t count, void* stream) {
  if (count < 0 || (count > 0 && (!gamma || !c || !records))) return MBV_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i)
    if (!gamma[i] || !records[i] || c[i] <= 0) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < count; base += kLnBoundGroup) {
    const int cnt = count - base < kLnBoundGroup ? count - base : kLnBoundGroup;
    LnBoundArgs a;
    for (int j = 0; j < cnt; ++j) {
      a.e[j].gamma = gamma[base + j];
      a.e[j].beta = beta ? beta[base + j] : nullptr;
      a.e[j].out = records[base + j];
      a.e[j].c = c[base + j];
    }
    hipLaunchKernelGGL(k_ln_bound_group, dim3((unsigned)cnt), dim3(256), 0, (hipStream_t)stream, a);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

// max|x| of `count` f32 tensors (rows[i], cols[i]) with row stride ld[i] (cols % 4 == 0, 16-byte aligned rows), as the BITS
// of the maximum, max-combined into the kAmaxSlots (64) words at out[i]: they must hold 0 (or earlier partial maxima) when
// the launch starts; the maximum over the 64 words is the result.  Every array argument is a HOST array of length count; one launch per 64 tensors.
extern "C" int mbv_f32_absmax_group(const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ld,
                                    uint32_t* const* out, int32_t count, void* stream) {
  if (count < 0 || (count > 0 && (!x || !rows || !cols || !ld || !out))) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < count; base += kAbsmaxGroup) {
    const int cnt = count - base < kAbsmaxGroup ? count - base : kAbsmaxGroup;
    AbsmaxArgs a;
    a.n = 0;
    long long blocks = 0;
    for (int j = 0; j < cnt; ++j) {
      const int i = base + j;
      if (rows[i] < 0 || cols[i] < 0 || !out[i]) return MBV_ERR_BAD_ARG;
      if (rows[i] == 0 || cols[i] == 0) continue;
      if (!x[i] || (cols[i] & 3) || (ld[i] & 3) || ld[i] < cols[i] || (reinterpret_cast<size_t>(x[i]) & 15)) return MBV_ERR_UNSUPPORTED;
      AbsmaxEntry& e = a.e[a.n++];
      e.x = x[i]; e.out = out[i]; e.rows = rows[i]; e.cols = cols[i]; e.ld = ld[i];
      const long long q = rows[i] * (cols[i] >> 2);
      long long nb = (q + 256 * 16 - 1) / (256 * 16);       // 16 float4 per thread
      if (nb > 2048) nb = 2048;
      if (nb < 1) nb = 1;
      e.block_begin = (int)blocks; e.blocks = (int)nb;
      blocks += nb;
    }
    if (a.n == 0) continue;
    hipLaunchKernelGGL(k_absmax_group, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

static int gemm32s_common(int layout, const float* a_op, const float* b_op, float* out, float* out_pre, const float* bias,
                          const uint32_t* amax_a, const uint32_t* amax_b, uint32_t* amax_out, int64_t gm, int64_t gn,
                          int64_t gk, int64_t lda,
                          int64_t ldb, int64_t ldc, int64_t a_rows, int64_t a_cols, int64_t b_rows, int64_t b_cols,
                          int32_t act, int32_t batch, int64_t sa, int64_t sb, int64_t sc, hipStream_t st) {
  Gemm32Args a = {};
  a.a = a_op; a.b = b_op; a.c = out; a.c2 = out_pre; a.bias = bias; a.amax_a = amax_a; a.amax_b = amax_b;
  a.amax_out = amax_out;
  a.gm = (int)gm; a.gn = (int)gn; a.gk = (int)gk;
  a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = (int)ldc;
  a.sa = sa; a.sb = sb; a.sc = sc;
  a.a_bytes = (unsigned)(((a_rows - 1) * lda + a_cols) * 4); a.b_bytes = (unsigned)(((b_rows - 1) * ldb + b_cols) * 4);
  a.ntm = (int)((gm + 127) / 128); a.ntn = (int)((gn + 127) / 128); a.splits = 1; a.ksteps = (int)((gk + KB32 - 1) / KB32);
  if (layout == 0) return gemm32s_launch<false, false>(act, a, batch, st);
  return gemm32s_launch<false, true>(act, a, batch, st);
}

// out (m, n) f32 = act(x (m, k) . w (n, k)^T + bias); out_pre (optional, act != 0) = the pre-activation.
// amax_x / amax_w: device words holding the bits of max|x| / max|w| (mbv_f32_absmax_group), NULL = no scaling (operand
// magnitudes within [2^-14, 2^15) keep full accuracy unscaled).
extern "C" int mbv_gemm32s_nt(const float* x, const float* w, const float* bias, float* out, float* out_pre, int64_t m,
                              int64_t n, int64_t k, int64_t ldx, int64_t ldw, int64_t ldo, const uint32_t* amax_x,
                              const uint32_t* amax_w, uint32_t* amax_out, int32_t act, int32_t batch, int64_t stride_x,
                              int64_t stride_w, int64_t stride_o, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || batch < 0 || !x || !w || !out) return MBV_ERR_BAD_ARG;
  if (act < 0 || act > 2) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldx & 3) || (ldw & 3) || (ldo & 3) || ldx < k || ldw < k || ldo < n) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(out) |
       reinterpret_cast<size_t>(out_pre) | reinterpret_cast<size_t>(bias)) & 15)
    return MBV_ERR_UNSUPPORTED;
  if (!fits32(m, ldx) || !fits32(n, ldw)) return MBV_ERR_UNSUPPORTED;
  if (m == 0 || batch == 0) return MBV_OK;
  return gemm32s_common(0, x, w, out, out_pre, bias, amax_x, amax_w, amax_out, m, n, k, ldx, ldw, ldo, m, k, n, k, act, batch,
                        stride_x, stride_w, stride_o, (hipStream_t)stream);
}

// out (m, k) f32 = g (m, n) . w (n, k)
extern "C" int mbv_gemm32s_nn(const float* g, const float* w, float* out, int64_t m, int64_t n, int64_t k, int64_t ldg,
                              int64_t ldw, int64_t ldo, const uint32_t* amax_g, const uint32_t* amax_w, uint32_t* amax_out,
                              int32_t batch, int64_t stride_g, int64_t stride_w, int64_t stride_o, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || batch < 0 || !g || !w || !out) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldg & 3) || (ldw & 3) || (ldo & 3) || ldg < n || ldw < k || ldo < k) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(out)) & 15) return MBV_ERR_UNSUPPORTED;
  if (!fits32(m, ldg) || !fits32(n, ldw)) return MBV_ERR_UNSUPPORTED;
  if (m == 0 || batch == 0) return MBV_OK;
  return gemm32s_common(1, g, w, out, nullptr, nullptr, amax_g, amax_w, amax_out, m, k, n, ldg, ldw, ldo, m, n, n, k, 0,
                        batch, stride_g, stride_w, stride_o, (hipStream_t)stream);
}

// Token ranges of a weight gradient: the chip holds 512 of these workgroups at a time (two per CU), and a launch of 516 takes
// as long as one of 1024 — so the split is the one that minimises (rounds of 512) / (ranges), at least 1024 tokens deep
// (43 ranges x 12 tiles = 516 workgroups measured 115 us at 65 536 x 768 x 192; 42 x 12 = 504 fit one round: 92 us).
static void tn32_split(int64_t m, int64_t n, int64_t k, int& splits, int& ksteps) {
  const int total_steps = (int)((m + KB32 - 1) / KB32);
  const long long tiles = ((n + 127) / 128) * ((k + 127) / 128);
  long long cap = (m + 1023) / 1024;
  if (cap > 256) cap = 256;
  if (cap < 1) cap = 1;
  long long best = 1;
  double best_cost = 1e30;
  for (long long s = 1; s <= cap; ++s) {
    const long long rounds = (tiles * s + 511) / 512;
    const double cost = (double)rounds / (double)s;
    if (cost < best_cost * 0.999) { best_cost = cost; best = s; }
  }
  ksteps = (int)((total_steps + best - 1) / best);
  splits = (total_steps + ksteps - 1) / ksteps;
}

extern "C" size_t mbv_gemm32s_tn_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  int splits, ksteps;
  tn32_split(m, n, k, splits, ksteps);
  return splits > 1 ? (size_t)splits * (size_t)n * (size_t)k * 4 : 0;
}

// dw (n, k) f32, contiguous  +=  g (m, n)^T . x (m, k): the token sum is cut into parts that are STORED to the workspace and
// added into dw by their owner (no atomics; bit-reproducible); a single part adds its tile to dw in place.
extern "C" int mbv_gemm32s_tn_acc(const float* g, const float* x, float* dw, int64_t m, int64_t n, int64_t k, int64_t ldg,
                                  int64_t ldx, const uint32_t* amax_g, const uint32_t* amax_x, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || !g || !x || !dw) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldg & 3) || (ldx & 3) || ldg < n || ldx < k) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dw) |
       reinterpret_cast<size_t>(workspace)) & 15)
    return MBV_ERR_UNSUPPORTED;
  if (!fits32(m, ldg) || !fits32(m, ldx) || n * k >= 0x7fffffffLL) return MBV_ERR_UNSUPPORTED;
  if (m == 0) return MBV_OK;
  Gemm32Args a = {};
  a.a = g; a.b = x; a.amax_a = amax_g; a.amax_b = amax_x;
  a.gm = (int)n; a.gn = (int)k; a.gk = (int)m;
  a.lda = (int)ldg; a.ldb = (int)ldx; a.ldc = (int)k;
  a.a_bytes = (unsigned)(((m - 1) * ldg + n) * 4); a.b_bytes = (unsigned)(((m - 1) * ldx + k) * 4);
  a.ntm = (int)((n + 127) / 128); a.ntn = (int)((k + 127) / 128);
  tn32_split(m, n, k, a.splits, a.ksteps);
  if (a.splits == 1) {
    a.c = dw; a.acc_out = 1;
    return gemm32s_launch<true, true>(EPI32_NONE, a, 1, (hipStream_t)stream);
  }
  if (!workspace || workspace_bytes < (size_t)a.splits * (size_t)n * (size_t)k * 4) return MBV_ERR_WORKSPACE;
  a.c = reinterpret_cast<float*>(workspace);
  a.ssplit = n * k;
  const int rc = gemm32s_launch<true, true>(EPI32_NONE, a, 1, (hipStream_t)stream);
  if (rc != MBV_OK) return rc;
  const long long total = n * k;
  hipLaunchKernelGGL(k_add_parts32, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float*>(workspace), a.splits, total, dw);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// ---- the 4 x 4 non-overlapping patch projection on an NCHW f32 image, without materialising its rows -------------------
static bool patch_ok(int64_t batch, int64_t c, int64_t h, int64_t w, int64_t e) {
  return batch > 0 && c > 0 && h > 0 && w > 0 && e > 0 && (h & 3) == 0 && (w & 127) == 0 && (e & 7) == 0 && (c & 1) == 0 &&
         batch * c * h * w * 4 < 0x7fff0000LL && (w >> 2) % 32 == 0;
}

extern "C" int mbv_patch_embed32_supported(int64_t batch, int64_t channels, int64_t h, int64_t w, int64_t embed) {
  return patch_ok(batch, channels, h, w, embed) ? 1 : 0;
}

// out (B * h/4 * w/4, E) = rows(image) . weight (E, 16 C)^T + bias
extern "C" int mbv_patch_embed32_fwd(const float* image, const float* weight, const float* bias, float* out, int64_t batch,
                                     int64_t channels, int64_t h, int64_t w, int64_t embed, const uint32_t* amax_image,
                                     const uint32_t* amax_w, void* stream) {
  if (!image || !weight || !out) return MBV_ERR_BAD_ARG;
  if (!patch_ok(batch, channels, h, w, embed)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(image) | reinterpret_cast<size_t>(weight) | reinterpret_cast<size_t>(out) |
       reinterpret_cast<size_t>(bias)) & 15)
    return MBV_ERR_UNSUPPORTED;
  const int64_t m = batch * (h / 4) * (w / 4), k = 16 * channels;
  Gemm32Args a = {};
  a.a = image; a.b = weight; a.c = out; a.bias = bias; a.amax_a = amax_image; a.amax_b = amax_w;
  a.gm = (int)m; a.gn = (int)embed; a.gk = (int)k;
  a.lda = (int)k; a.ldb = (int)k; a.ldc = (int)embed;
  a.a_bytes = (unsigned)(batch * channels * h * w * 4); a.b_bytes = (unsigned)(embed * k * 4);
  a.ntm = (int)((m + 127) / 128); a.ntn = (int)((embed + 127) / 128); a.splits = 1; a.ksteps = (int)(k / KB32);
  a.pc = (int)channels; a.ph = (int)h; a.pw = (int)w;
  return gemm32s_launch<false, false, 1>(EPI32_NONE, a, 1, (hipStream_t)stream);
}

// d image (B, C, h, w) = (d out (tokens, E) . weight (E, 16 C)) scattered back (every image element is written)
extern "C" int mbv_patch_embed32_bwd_image(const float* d_out, const float* weight, float* d_image, int64_t batch,
                                           int64_t channels, int64_t h, int64_t w, int64_t embed, const uint32_t* amax_g,
                                           const uint32_t* amax_w, void* stream) {
  if (!d_out || !weight || !d_image) return MBV_ERR_BAD_ARG;
  if (!patch_ok(batch, channels, h, w, embed)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(d_out) | reinterpret_cast<size_t>(weight) | reinterpret_cast<size_t>(d_image)) & 15)
    return MBV_ERR_UNSUPPORTED;
  const int64_t m = batch * (h / 4) * (w / 4), k = 16 * channels;
  Gemm32Args a = {};
  a.a = d_out; a.b = weight; a.c = d_image; a.amax_a = amax_g; a.amax_b = amax_w;
  a.gm = (int)m; a.gn = (int)k; a.gk = (int)embed;
  a.lda = (int)embed; a.ldb = (int)k; a.ldc = (int)k;
  a.a_bytes = (unsigned)(m * embed * 4); a.b_bytes = (unsigned)(embed * k * 4);
  a.ntm = (int)((m + 127) / 128); a.ntn = (int)((k + 127) / 128); a.splits = 1; a.ksteps = (int)((embed + KB32 - 1) / KB32);
  a.pc = (int)channels; a.ph = (int)h; a.pw = (int)w;
  return gemm32s_launch<false, true, 2>(EPI32_NONE, a, 1, (hipStream_t)stream);
}

extern "C" size_t mbv_patch_embed32_bwd_weight_workspace_bytes(int64_t batch, int64_t channels, int64_t h, int64_t w,
                                                               int64_t embed) {
  if (!patch_ok(batch, channels, h, w, embed)) return 0;
  return mbv_gemm32s_tn_workspace_bytes(batch * (h / 4) * (w / 4), embed, 16 * channels);
}

// d weight (E, 16 C) f32, contiguous += d out (tokens, E)^T . rows(image)
extern "C" int mbv_patch_embed32_bwd_weight(const float* d_out, const float* image, float* d_weight, int64_t batch,
                                            int64_t channels, int64_t h, int64_t w, int64_t embed, const uint32_t* amax_g,
                                            const uint32_t* amax_image, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  if (!d_out || !image || !d_weight) return MBV_ERR_BAD_ARG;
  if (!patch_ok(batch, channels, h, w, embed)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(d_out) | reinterpret_cast<size_t>(image) | reinterpret_cast<size_t>(d_weight) |
       reinterpret_cast<size_t>(workspace)) & 15)
    return MBV_ERR_UNSUPPORTED;
  const int64_t m = batch * (h / 4) * (w / 4), k = 16 * channels;
  Gemm32Args a = {};
  a.a = d_out; a.b = image; a.amax_a = amax_g; a.amax_b = amax_image;
  a.gm = (int)embed; a.gn = (int)k; a.gk = (int)m;
  a.lda = (int)embed; a.ldb = (int)k; a.ldc = (int)k;
  a.a_bytes = (unsigned)(m * embed * 4); a.b_bytes = (unsigned)(batch * channels * h * w * 4);
  a.ntm = (int)((embed + 127) / 128); a.ntn = (int)((k + 127) / 128);
  tn32_split(m, embed, k, a.splits, a.ksteps);
  a.pc = (int)channels; a.ph = (int)h; a.pw = (int)w;
  if (a.splits == 1) {
    a.c = d_weight; a.acc_out = 1;
    return gemm32s_launch<true, true, 3>(EPI32_NONE, a, 1, (hipStream_t)stream);
  }
  if (!workspace || workspace_bytes < (size_t)a.splits * (size_t)embed * (size_t)k * 4) return MBV_ERR_WORKSPACE;
  a.c = reinterpret_cast<float*>(workspace);
  a.ssplit = embed * k;
  const int rc = gemm32s_launch<true, true, 3>(EPI32_NONE, a, 1, (hipStream_t)stream);
  if (rc != MBV_OK) return rc;
  const long long total = embed * k;
  hipLaunchKernelGGL(k_add_parts32, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float*>(workspace), a.splits, total, d_weight);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

static void tn32_group_split(int64_t m, int& splits, int& ksteps) {
  const int total_steps = (int)((m + KB32 - 1) / KB32);
  int s = (int)((m + kTn32Depth - 1) / kTn32Depth);
  if (s < 1) s = 1;
  ksteps = (total_steps + s - 1) / s;
  splits = (total_steps + ksteps - 1) / ksteps;
}

extern "C" size_t mbv_gemm32s_tn_group_workspace_bytes(const int64_t* m, const int64_t* n, const int64_t* k, int32_t count) {
  size_t total = 0;
  for (int i = 0; i < count; ++i) {
    if (m[i] <= 0 || n[i] <= 0 || k[i] <= 0) continue;
    int splits, ksteps;
    tn32_group_split(m[i], splits, ksteps);
    if (splits > 1) total += (size_t)splits * (size_t)n[i] * (size_t)k[i] * 4;
  }
  return total;
}

// dw[i] (n[i], k[i]) f32, contiguous  +=  g[i] (m[i], n[i])^T . x[i] (m[i], k[i])  for i < count, in one launch (+ one parts-add
// launch) per 48 products; the dw[i] of one call must not overlap.  amax_g[i] / amax_x[i]: absmax records (NULL entries =
// unscaled).  Array arguments are HOST arrays of length count; workspace: mbv_gemm32s_tn_group_workspace_bytes, 16-byte aligned.
extern "C" int mbv_gemm32s_tn_group(const float* const* g, const float* const* x, float* const* dw, const int64_t* m,
                                    const int64_t* n, const int64_t* k, const int64_t* ldg, const int64_t* ldx,
                                    const uint32_t* const* amax_g, const uint32_t* const* amax_x, int32_t count,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (count < 0 || (count > 0 && (!g || !x || !dw || !m || !n || !k || !ldg || !ldx))) return MBV_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    if (m[i] < 0) return MBV_ERR_BAD_ARG;
    if (m[i] == 0) continue;
    if (n[i] <= 0 || k[i] <= 0 || !g[i] || !x[i] || !dw[i]) return MBV_ERR_BAD_ARG;
    if ((n[i] & 7) || (k[i] & 7) || (ldg[i] & 3) || (ldx[i] & 3) || ldg[i] < n[i] || ldx[i] < k[i]) return MBV_ERR_UNSUPPORTED;
    if ((reinterpret_cast<size_t>(g[i]) | reinterpret_cast<size_t>(x[i]) | reinterpret_cast<size_t>(dw[i])) & 15)
      return MBV_ERR_UNSUPPORTED;
    if (!fits32(m[i], ldg[i]) || !fits32(m[i], ldx[i]) || n[i] * k[i] >= 0x7fffffffLL) return MBV_ERR_UNSUPPORTED;
  }
  if (mbv_gemm32s_tn_group_workspace_bytes(m, n, k, count) > workspace_bytes) return MBV_ERR_WORKSPACE;
  if (workspace_bytes && (!workspace || (reinterpret_cast<size_t>(workspace) & 15))) return MBV_ERR_WORKSPACE;
  static bool attr = false;
  if (!attr) {
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm32s_tn_group),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS32));
    attr = true;
  }
  float* ws = reinterpret_cast<float*>(workspace);
  for (int base = 0; base < count; base += kTn32Group) {
    const int cnt = count - base < kTn32Group ? count - base : kTn32Group;
    Tn32GroupArgs a;
    Parts32GroupArgs pa;
    a.n = 0; pa.n = 0;
    long long items = 0, pblocks = 0;
    for (int j = 0; j < cnt; ++j) {
      const int i = base + j;
      if (m[i] == 0) continue;
      Tn32Entry& e = a.e[a.n++];
      e.g = g[i]; e.x = x[i];
      e.amax_g = amax_g ? amax_g[i] : nullptr; e.amax_x = amax_x ? amax_x[i] : nullptr;
      e.m = (int)m[i]; e.n = (int)n[i]; e.k = (int)k[i]; e.ldg = (int)ldg[i]; e.ldx = (int)ldx[i];
      e.ntn = (int)((k[i] + 127) / 128);
      tn32_group_split(m[i], e.splits, e.ksteps);
      e.item_begin = (int)items;
      items += (long long)((n[i] + 127) / 128) * e.ntn * e.splits;
      if (e.splits == 1) {
        e.out = dw[i];
      } else {
        e.out = ws;
        Parts32Entry& q = pa.e[pa.n++];
        q.part = ws; q.out = dw[i]; q.n = n[i] * k[i]; q.parts = e.splits; q.block_begin = (int)pblocks;
        pblocks += (q.n / 4 + 255) / 256;
        ws += (size_t)e.splits * (size_t)q.n;
      }
    }
    if (a.n == 0) continue;
    if (items > 0x7fffffffLL || pblocks > 0x7fffffffLL) return MBV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_gemm32s_tn_group, dim3((unsigned)items), dim3(256), LDS32, (hipStream_t)stream, a);
    MBV_CHECK_LAUNCH();
    if (pa.n) {
      hipLaunchKernelGGL(k_add_parts32_group, dim3((unsigned)pblocks), dim3(256), 0, (hipStream_t)stream, pa);
      MBV_CHECK_LAUNCH();
    }
  }
  return MBV_OK;
}

// ---- a 3 x 3 convolution (stride 1, padding 1) as one K20 product over k = (tap, channel) ---------------------------------
// rows / out_rows: zero-bordered channels-last buffers of csrc/conv_pad.hip ((2 G + B (H + 2)(W + 2), C) with G = W + 3 guard
// rows; mbv_conv_rows).  out_rows[m][co] = sum_t sum_ci rows[m + shift_t][ci] wm[co][t C + ci] for every padded position m
// (t = 3 dy + dx, shift_t = (dy - 1)(W + 2) + (dx - 1)): the interior positions hold the convolution, the border positions
// partial sums nobody reads.  The data gradient is the same call on the output gradient's rows with the taps flipped in wm.
extern "C" int mbv_conv3x3_gemm32s(const float* rows, const float* wm, float* out_rows, int64_t batch, int64_t H, int64_t W,
                                   int64_t C, int64_t cout, const uint32_t* amax_rows, const uint32_t* amax_w,
                                   uint32_t* amax_out, void* stream) {
  if (!rows || !wm || !out_rows || batch <= 0 || H <= 0 || W <= 0 || C <= 0 || cout <= 0) return MBV_ERR_BAD_ARG;
  if ((C % KB32) || (cout & 7)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(rows) | reinterpret_cast<size_t>(wm) | reinterpret_cast<size_t>(out_rows)) & 15)
    return MBV_ERR_UNSUPPORTED;
  const int64_t G = W + 3, mp = batch * (H + 2) * (W + 2), total = mp + 2 * G;
  if (total * C * 4 >= 0x7fff0000LL || cout * 9 * C * 4 >= 0x7fff0000LL) return MBV_ERR_UNSUPPORTED;
  Gemm32Args a = {};
  a.a = rows; a.b = wm; a.c = out_rows + G * cout; a.amax_a = amax_rows; a.amax_b = amax_w; a.amax_out = amax_out;
  a.gm = (int)mp; a.gn = (int)cout; a.gk = (int)(9 * C);
  a.lda = (int)C; a.ldb = (int)(9 * C); a.ldc = (int)cout;
  a.a_bytes = (unsigned)(total * C * 4); a.b_bytes = (unsigned)(cout * 9 * C * 4);
  a.ntm = (int)((mp + 127) / 128); a.ntn = (int)((cout + 127) / 128); a.splits = 1; a.ksteps = (int)(9 * C / KB32);
  a.tap_steps = (int)(C / KB32);
  for (int t = 0; t < 9; ++t) {
    const int64_t shift = (t / 3 - 1) * (W + 2) + (t % 3 - 1);
    a.tap[t] = (int)((G + shift - t) * C * 4);          // >= 0: the row of tap t, minus the t C columns k has advanced
  }
  return gemm32s_launch<false, false, 4>(EPI32_NONE, a, 1, (hipStream_t)stream);
}

// Partial column-sum rows an mbv_gemm32s_nn_act of this shape leaves: one per 64 output rows of a 128-row tile.
extern "C" int64_t mbv_gemm32s_nn_part_rows(int64_t m, int32_t batch) {
  if (m <= 0 || batch <= 0) return 0;
  return 2 * ((m + 127) / 128) * batch;
}

// out (m, k) f32 = act'(pre (m, k)) * (g (m, n) . w (n, k));  parts (mbv_gemm32s_nn_part_rows(m, 1), k) f32 (every element
// written): partial column sums of out — their sum over the rows is the bias gradient of the layer that produced `pre`.
// act: 1 ReLU, 2 GELU (erf).  The fp32 FFN's backward: the data gradient of fc2 times the activation's derivative in one launch.
extern "C" int mbv_gemm32s_nn_act(const float* g, const float* w, float* out, const float* pre, float* parts,
                                  size_t parts_bytes, int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldw, int64_t ldo,
                                  int64_t ldpre, const uint32_t* amax_g, const uint32_t* amax_w, uint32_t* amax_out,
                                  int32_t act, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || !g || !w || !out || !pre) return MBV_ERR_BAD_ARG;
  if (act < 1 || act > 2) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldg & 3) || (ldw & 3) || (ldo & 3) || (ldpre & 3) || ldg < n || ldw < k || ldo < k || ldpre < k)
    return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(out) |
       reinterpret_cast<size_t>(pre) | reinterpret_cast<size_t>(parts)) & 15)
    return MBV_ERR_UNSUPPORTED;
  if (!fits32(m, ldg) || !fits32(n, ldw) || !fits32(m, ldpre)) return MBV_ERR_UNSUPPORTED;
  if (parts && parts_bytes < (size_t)mbv_gemm32s_nn_part_rows(m, 1) * (size_t)k * 4) return MBV_ERR_WORKSPACE;
  if (m == 0) return MBV_OK;
  Gemm32Args a = {};
  a.a = g; a.b = w; a.c = out; a.amax_a = amax_g; a.amax_b = amax_w; a.amax_out = amax_out;
  a.aux = pre; a.ldx = (int)ldpre; a.colsum_rows = parts;
  a.gm = (int)m; a.gn = (int)k; a.gk = (int)n;
  a.lda = (int)ldg; a.ldb = (int)ldw; a.ldc = (int)ldo;
  a.a_bytes = (unsigned)(((m - 1) * ldg + n) * 4); a.b_bytes = (unsigned)(((n - 1) * ldw + k) * 4);
  a.ntm = (int)((m + 127) / 128); a.ntn = (int)((k + 127) / 128); a.splits = 1; a.ksteps = (int)((n + KB32 - 1) / KB32);
  return gemm32s_launch<false, true>(act == 1 ? EPI32_DRELU : EPI32_DGELU, a, 1, (hipStream_t)stream);
}
