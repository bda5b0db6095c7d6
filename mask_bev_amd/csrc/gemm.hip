// K17 — 16-bit (bf16 / fp16) MFMA GEMM family for the token-major Linear layers of the path: the Swin qkv / proj /
// fc1 / fc2 projections, patch projection and patch merging (/root/reference: mask_bev/models/networks/swin/swin.py:
// 89-116, 357-377, 579-586, 611; mmcv FFN), the pixel decoder's and the decoder's Linears
// (mask_bev/models/head/mask_bev_panoptic_head.py:119-175) and the backward of the mask-logit contraction
// (mask_bev/models/networks/mask2former_head/mask2former_head.py:459).  One kernel template, three operand layouts:
//
//   NT  C[m][n] = sum_k X[m][k] W[n][k]          forward Linear              (both operands contraction-contiguous)
//   NN  C[m][k] = sum_n G[m][n] W[n][k]          data gradient               (W read with transposing LDS reads)
//   TN  C[n][k] = sum_m G[m][n] X[m][k]          weight gradient, split over m (both operands transposed reads)
//
// Structure (cdna_hip_programming.md §5, "minimum 2-phase" recipe): 256 threads = 4 waves as 2 x 2, block tile
// 128 x 128 x KB (KB = 32: 35 KB of LDS, four workgroups per CU hide each other's load latency; KB = 64: two), wave
// tile 64 x 64 = 2 x 2 v_mfma_f32_32x32x16; both operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds`
// (no staging registers; out-of-range rows / columns read as zero through the buffer range check, so every edge is
// handled by the descriptor), double buffered, one barrier per K-step; LDS images are XOR-swizzled on the SOURCE
// address (the LDS side of an LDS-DMA is lane-linear) so that both the ds_read_b128 row reads and the
// ds_read_b64_tr_b16 transposed reads are bank-conflict free.
// Epilogue: STORE — the wave's accumulators (computed as C^T so that a lane holds 4 consecutive columns) are turned
// through a private LDS tile and leave as whole 128-byte (16-bit) / 256-byte (f32) row segments, with bias,
// ReLU / GELU (optionally also storing the pre-activation), the activation-backward products ReLU' / GELU' against
// a saved tensor and the column sums of the result (the bias gradient of the layer in front; per-wave partial rows
// in the workspace, then one small reduction launch) applied in f32 on the way; ATOMIC — f32 atomic adds of the
// accumulators straight into the destination (weight gradients accumulate in the parameter arena; each
// wave-instruction adds two 128-byte row segments, the full-rate shape of MI355X_MICROARCH.md §Global float atomics).

#include "common.hpp"
#include "gemm_tiles.hpp"

namespace {

// Block shape: WM x 2 waves, each wave TM x TN MFMA tiles of 32 x 32  =>  BM = 32 TM WM rows, BN = 64 TN columns.
//   128 x 128 (WM 2, 2 x 2 tiles, 256 threads)  — the default
//   256 x 128 (WM 4, 2 x 2 tiles, 512 threads)  — twice the rows per B-tile fetch: 85 instead of 64 flop per staged byte
//   384 x 192 (WM 4, 3 x 3 tiles, 512 threads)  — the weight-gradient shape: 128 flop per byte, one workgroup per CU
// An operand tile wider than 128 rows / columns is staged as 128-wide sub-images, so the swizzles below never change.
template <int KB, int NS, int WM = 2, int TM = 2, int TN = 2>
struct Geo {
  static constexpr int NW = 2 * WM;                                   // waves
  static constexpr int BM = 32 * TM * WM, BN = 64 * TN;
  static constexpr int SA = (BM + 127) / 128, SB = (BN + 127) / 128;  // 128-wide sub-images per operand tile
  static constexpr int IMG = 128 * KB * 2;                            // one sub-image
  static constexpr int STAGE = (SA + SB) * IMG;
  static constexpr int PPS = IMG / 1024;                              // 1-KiB LDS-DMA pieces per sub-image
  static constexpr int PIECES = (SA + SB) * PPS;
  static constexpr int NP = (PIECES + NW - 1) / NW;                   // pieces per wave and K-step
  static constexpr int STG = NW * 32 * STG_LD * 4;                    // epilogue staging (STORE, TN == 2)
  static constexpr int LDS = NS * STAGE > STG ? NS * STAGE : STG;
  static constexpr int BLOCKS = LDS <= 40960 ? 4 : (LDS <= 53248 ? 3 : (LDS <= 81920 ? 2 : 1));   // per CU, by LDS
  static constexpr int WPS = (BLOCKS * NW + 3) / 4 > 8 ? 8 : (BLOCKS * NW + 3) / 4;               // waves per SIMD
};

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_GELU = 2, EPI_DRELU = 3, EPI_DGELU = 4,
       EPI_CONV = 5 };       // EPI_NONE's epilogue; the A rows of a K-step are shifted by its tap (GemmArgs::tap)

struct GemmArgs {
  const void* a;
  const void* b;
  void* c;              // STORE: (gm, gn) row-major, 16-bit or f32; ATOMIC: f32, accumulated into
  void* c2;             // STORE + EPI_GELU / EPI_RELU: optional pre-activation output (same type as c)
  const void* aux;      // EPI_DRELU / EPI_DGELU: saved activation output / pre-activation, (gm, gn) 16-bit, ld = ldx
  const float* bias;    // STORE: optional, (gn)
  float* colsum_rows;   // STORE: optional, (WM ntm, gn) f32: column sums of the stored values per wave row
  int gm, gn, gk;       // GEMM dims: C (gm x gn) = A (gm x gk) . B (gk x gn)
  int lda, ldb, ldc, ldx;
  long long sa, sb, sc; // batch strides (elements)
  long long ssplit;     // STORE with splits > 1: element stride between the partial results of the K parts
  unsigned a_bytes, b_bytes;   // extent of one batch item of A / B for the buffer descriptors
  int ntm, ntn, splits, ksteps;  // tiles, split-K parts, K-steps of KB per part
  int out_f32;
  int acc_out;          // STORE, f32 output, splits == 1: the product is ADDED to c (read-modify-write, no atomics)
  int tap_steps;        // != 0: a 3 x 3 convolution on zero-bordered channels-last rows (csrc/conv_pad.hip) as ONE product over
  int tap[9];           //   k = (tap, channel): the A rows of the K-steps of tap t (tap_steps each) are tap[t] bytes further on
};

// One workgroup's share of a product: work item `bid` of the (batch, split, tile_m, tile_n) index, tile_n fastest, so
// that the items that share an A panel are neighbours (and, after xcd_contiguous, share an L2).
template <int KB, int NS, int WM, int TM, int TN, bool A_KS, bool B_KS, int OUT, int EPI, typename T>
__device__ __forceinline__ void gemm16_body(const GemmArgs& p, const int bid, char* smem) {
  using G = Geo<KB, NS, WM, TM, TN>;
  static_assert(OUT == 1 || TN == 2, "the STORE epilogue turns 64-column wave tiles");
  constexpr int BM = G::BM, BN = G::BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles = p.ntm * p.ntn;
  const int z = bid / tiles, t = bid - z * tiles;
  const int tile_m = t / p.ntn, tile_n = t - tile_m * p.ntn;
  const int bz = z / p.splits, sp = z - bz * p.splits;
  const int k_begin = sp * p.ksteps * KB;
  int k_end = k_begin + p.ksteps * KB;
  if (k_end > p.gk) k_end = p.gk;
  const int nk = (k_end - k_begin + KB - 1) / KB;
  if (nk <= 0) return;

  const char* abase = reinterpret_cast<const char*>(p.a) + (size_t)bz * (size_t)p.sa * 2;
  const char* bbase = reinterpret_cast<const char*>(p.b) + (size_t)bz * (size_t)p.sb * 2;
  const i32x4 ra = make_rsrc(abase, p.a_bytes), rb = make_rsrc(bbase, p.b_bytes);
  const unsigned smem_addr = lds_addr_of(smem);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // This wave's LDS-DMA pieces: piece pi = wave + NW j of the stage (sub-images of A first, then of B).  Offsets at
  // the first K-step are loop-invariant; a full K-step s adds the scalar s * step of its operand.
  const bool ragged = ((k_end - k_begin) & (KB - 1)) != 0;
  const unsigned stepa = A_KS ? (unsigned)(KB * 2) * (unsigned)p.lda : (unsigned)(KB * 2);
  const unsigned stepb = B_KS ? (unsigned)(KB * 2) * (unsigned)p.ldb : (unsigned)(KB * 2);
  auto piece_off = [&](int j, int k0, int kend) -> unsigned {
    const int pi = wave + G::NW * j;
    const int sub = pi / G::PPS, local = pi - sub * G::PPS;
    if (pi >= G::PIECES) return OOB;
    if (sub < G::SA) return piece_offset<KB, A_KS>(sub, local, m0, p.gm, p.lda * 2, k0, kend, lane);
    return piece_offset<KB, B_KS>(sub - G::SA, local, n0, p.gn, p.ldb * 2, k0, kend, lane);
  };
  unsigned off[G::NP];
#pragma unroll
  for (int j = 0; j < G::NP; ++j) off[j] = piece_off(j, k_begin, k_begin + KB);

  // K-step kt of this workgroup goes into ring slot kt % NS; a step beyond the range is issued as zero fills (no
  // memory traffic) so that every step leaves the same number of loads on the wave's counter
  auto stage = [&](int kt) {
    const unsigned slot = smem_addr + (kt % NS) * G::STAGE;
    const bool tail = ragged && kt == nk - 1;            // the only K-step with k's beyond the range
#pragma unroll
    for (int j = 0; j < G::NP; ++j) {
      const int pi = wave + G::NW * j;
      if (G::PIECES % G::NW != 0 && pi >= G::PIECES) continue;       // (wave-uniform) this wave has one piece fewer
      const bool is_a = pi < G::SA * G::PPS;
      unsigned v = off[j], sof = (unsigned)kt * (is_a ? stepa : stepb);
      if (kt >= nk) { v = OOB; sof = 0u; }
      else if (tail) { v = piece_off(j, k_begin + kt * KB, k_end); sof = 0u; }
      if constexpr (EPI == EPI_CONV) {
        if (is_a && kt < nk) sof += (unsigned)p.tap[(k_begin / KB + kt) / p.tap_steps];
      }
      glds16(is_a ? ra : rb, slot + pi * 1024, v, __builtin_amdgcn_readfirstlane(sof));
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // NS - 1 steps in flight.  Per step: wait for the oldest (the wave's own pieces: counted vmcnt, the younger
  // NS - 2 steps stay in flight), barrier (everybody's pieces of that step are in LDS, and everybody is done reading
  // the slot of the step before), refill that slot with step kt + NS - 1, multiply.
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) stage(s);
  for (int kt = 0; kt < nk; ++kt) {
    if (G::PIECES % G::NW == 0 || wave < G::PIECES % G::NW)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G::NP) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (G::NP - 1)) : "memory");
    __syncthreads();
    stage(kt + NS - 1);
    const char* aimg = smem + (kt % NS) * G::STAGE;
    const char* bimg = aimg + G::SA * G::IMG;
    uint4 af[KB / 16][TM], bf[KB / 16][TN];
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int x = 32 * (wm * TM + i);                 // 32-row tiles never straddle a 128-wide sub-image
        const char* im = aimg + (x >> 7) * G::IMG;
        af[ks][i] = A_KS ? frag_ks(im, x & 127, ks, lane) : frag_kc<KB>(im, x & 127, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int x = 32 * (wn * TN + j);
        const char* im = bimg + (x >> 7) * G::IMG;
        bf[ks][j] = B_KS ? frag_ks(im, x & 127, ks, lane) : frag_kc<KB>(im, x & 127, ks, lane);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = OUT == 0 ? Mma<T>::run(bf[ks][j], af[ks][i], acc[i][j]) : Mma<T>::run(af[ks][i], bf[ks][j], acc[i][j]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero fills of the tail: nothing may land in LDS later
  __syncthreads();

  const int r = lane & 31, h = lane >> 5;
  if (OUT == 1) {
    // acc[i][j][e]: row m = acc_row(e, h) of tile i, column n = r of tile j
    float* c = reinterpret_cast<float*>(p.c) + (size_t)bz * (size_t)p.sc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int gn = n0 + 32 * (wn * TN + j) + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int gm = m0 + 32 * (wm * TM + i) + acc_row(e, h);
          if (gm < p.gm && gn < p.gn) atomicAdd(c + (size_t)gm * p.ldc + gn, acc[i][j][e]);
        }
      }
    return;
  }

  // ---- STORE: acc[i][j][e] = C[m = 64 wm + 32 i + r][n = 64 wn + 32 j + acc_row(e, h)] -------------------------------
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * STG_LD);
  const int prow = lane >> 3, cg = lane & 7;
  const int gn = n0 + 64 * wn + 8 * cg;
  const bool col_ok = gn < p.gn;                      // gn % 8 == 0 is required: a column group is all in or all out
  const size_t cbase = (size_t)bz * (size_t)p.sc + (size_t)sp * (size_t)p.ssplit;
  const int mrow0 = m0 + 32 * TM * wm + prow;         // this lane's rows: mrow0 + 32 i + 8 ps
  constexpr bool HAS_AUX = EPI == EPI_DRELU || EPI == EPI_DGELU;
  uint4 auxv[TM][4];
  if (HAS_AUX) {                                      // every row requested before the accumulators are turned
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int gm = mrow0 + 32 * i + 8 * ps;
        auxv[i][ps] = make_uint4(0, 0, 0, 0);
        if (gm < p.gm && col_ok)
          auxv[i][ps] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.aux) + cbase +
                                                        (size_t)gm * p.ldx + gn);
      }
  }
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
      for (int e = 0; e < 8; ++e) v[e] += bias8[e];
      const size_t o = cbase + (size_t)gm * p.ldc + gn;
      if (EPI == EPI_RELU || EPI == EPI_GELU) {
        if (p.c2) store8<T>(p.c2, o, v, p.out_f32);        // the pre-activation as the backward wants it
        if (!p.out_f32) round8<T>(v);                     // the activation sees what the unfused path reads back
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (EPI == EPI_RELU) {
            v[e] = v[e] > 0.f ? v[e] : 0.f;
          } else {
            float dens;
            v[e] *= gelu_cdf_parts(v[e], dens);
          }
        }
      } else if (HAS_AUX) {
        const unsigned w[4] = {auxv[i][ps].x, auxv[i][ps].y, auxv[i][ps].z, auxv[i][ps].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float zz = Mma<T>::to_f((unsigned short)(w[e >> 1] >> (16 * (e & 1))));
          if (EPI == EPI_DRELU) {
            v[e] = zz > 0.f ? v[e] : 0.f;
          } else {
            float dens;
            const float cdf = gelu_cdf_parts(zz, dens);
            v[e] *= cdf + zz * dens;
          }
        }
      }
      if (p.acc_out) {                                    // f32 destination owned by this workgroup: out += product
        const float* d = reinterpret_cast<const float*>(p.c) + o;
        const float4 d0 = *reinterpret_cast<const float4*>(d), d1 = *reinterpret_cast<const float4*>(d + 4);
        v[0] += d0.x; v[1] += d0.y; v[2] += d0.z; v[3] += d0.w;
        v[4] += d1.x; v[5] += d1.y; v[6] += d1.z; v[7] += d1.w;
      }
      store8<T>(p.c, o, v, p.out_f32);
      if (p.colsum_rows) {
        if (!p.out_f32) round8<T>(v);                     // sum what was stored
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += v[e];
      }
    }
  }
  if (p.colsum_rows) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = csum[e];
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      csum[e] = s;
    }
    if (prow == 0 && col_ok) {                        // one partial row per (tile_m, wave row): no atomics, no contention
      float* d = p.colsum_rows + ((size_t)bz * WM * p.ntm + WM * tile_m + wm) * (size_t)p.gn + gn;
      *reinterpret_cast<float4*>(d) = make_float4(csum[0], csum[1], csum[2], csum[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
    }
  }
}

template <int KB, int NS, int WM, int TM, int TN, bool A_KS, bool B_KS, int OUT, int EPI, typename T>
__global__ void __launch_bounds__(64 * 2 * WM, (Geo<KB, NS, WM, TM, TN>::WPS)) k_gemm16(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm16_body<KB, NS, WM, TM, TN, A_KS, B_KS, OUT, EPI, T>(p, xcd_contiguous(blockIdx.x, gridDim.x), smem);
}

// ---- grouped weight gradients ------------------------------------------------------------------------------------
// Up to kTnGroup products dw_i (n_i, k_i) f32 += g_i (m_i, n_i)^T x_i (m_i, k_i) in ONE launch.  A weight gradient is
// nobody's input, so the Linears of a backward pass hand theirs over and the pass issues them together at its end: a
// 192 x 192 ... 2304 x 768 output is 2 ... 108 tiles of 128 x 128 — alone it fills the chip only by cutting the token
// sum into slivers (and pays a pipeline fill per sliver and a round trip of partial tiles per split), together the
// tiles of ~40 layers do.  Work item = (entry, split, tile_m, tile_n); every entry is cut into token ranges of about
// the same depth, so the items are about equally long.  splits == 1: the owner adds its tile to dw in place;
// splits > 1: partial tiles go to the workspace and k_add_parts_group folds them into dw (no atomics either way).
constexpr int kTnGroup = 48;
struct TnEntry {
  const void* g; const void* x; float* out;     // out: dw (splits == 1) or this entry's parts in the workspace
  int m, n, k, ldg, ldx, ldo;
  int ntn, splits, ksteps, item_begin;
};
struct TnGroupArgs {
  TnEntry e[kTnGroup];
  int n, total_items;
};

template <int KB, int NS, typename T>
__global__ void __launch_bounds__(256, (Geo<KB, NS, 2, 2, 2>::WPS)) k_gemm16_tn_group(const TnGroupArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int item = xcd_contiguous(blockIdx.x, gridDim.x);
  int i = 0;
  while (i + 1 < a.n && item >= a.e[i + 1].item_begin) ++i;     // block-uniform: a scalar loop over <= 48 entries
  const TnEntry& e = a.e[i];
  GemmArgs p;
  p.a = e.g; p.b = e.x; p.c = e.out; p.c2 = nullptr; p.aux = nullptr; p.bias = nullptr; p.colsum_rows = nullptr;
  p.gm = e.n; p.gn = e.k; p.gk = e.m;
  p.lda = e.ldg; p.ldb = e.ldx; p.ldc = e.ldo; p.ldx = 0;
  p.sa = 0; p.sb = 0; p.sc = 0;
  p.ssplit = (long long)e.n * e.k;
  p.a_bytes = (unsigned)(((long long)(e.m - 1) * e.ldg + e.n) * 2);
  p.b_bytes = (unsigned)(((long long)(e.m - 1) * e.ldx + e.k) * 2);
  p.ntm = (e.n + 127) / 128; p.ntn = e.ntn; p.splits = e.splits; p.ksteps = e.ksteps;
  p.out_f32 = 1;
  p.acc_out = e.splits == 1;
  p.tap_steps = 0;
  gemm16_body<KB, NS, 2, 2, 2, true, true, 0, EPI_NONE, T>(p, item - e.item_begin, smem);
}

constexpr int kPartsGroup = 96;
struct PartsEntry {
  const float* part; float* out;
  long long n;                 // elements of out (a multiple of 4)
  int parts, block_begin;
};
struct PartsGroupArgs {
  PartsEntry e[kPartsGroup];
  int n;
};

// out_i (n_i) += sum of its parts (parts_i, n_i), for up to kPartsGroup outputs; a thread owns 4 consecutive elements
__global__ void __launch_bounds__(256) k_add_parts_group(const PartsGroupArgs a) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.e[i + 1].block_begin) ++i;
  const PartsEntry& e = a.e[i];
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

// out (n) += sum over rows of part (rows, n)
__global__ void __launch_bounds__(256) k_sum_rows(const float* __restrict__ part, int rows, int n, float* __restrict__ out) {
  __shared__ float red[256];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float s = 0.f;
  if (col < n)
    for (int r = r0 + sub; r < r1; r += 4) s += part[(size_t)r * n + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0 && col < n)
    atomicAdd(out + col, (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]));
}

// out (n) += sum over the `parts` partial results part (parts, n).  A thread owns 4 consecutive elements and walks its
// share of the parts 8 at a time (8 independent 16-byte loads in flight).  gridDim.y == 1: the owner adds (no atomics,
// bit-reproducible); small outputs with many parts are cut into gridDim.y part ranges that finish with f32 atomics.
__global__ void __launch_bounds__(256) k_add_parts(const float* __restrict__ part, int parts, long long n,
                                                   float* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const int per = (parts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = p0 + per < parts ? p0 + per : parts;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = p0; p < p1; p += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = p + u < p1 ? *reinterpret_cast<const float4*>(part + (long long)(p + u) * n + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  if (gridDim.y == 1) {
    float4 o = *reinterpret_cast<const float4*>(out + i);
    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
    *reinterpret_cast<float4*>(out + i) = o;
  } else {
    atomicAdd(out + i, s.x); atomicAdd(out + i + 1, s.y); atomicAdd(out + i + 2, s.z); atomicAdd(out + i + 3, s.w);
  }
}

}  // namespace

// dtype: 0 = bf16, 1 = fp16.  layout: 0 = NT, 1 = NN, 2 = TN (see the header).
constexpr int KB = 32, NS = 3;      // K-step depth x ring slots (measured best of 32x2, 32x3, 32x4, 64x2, 64x3)

template <int WM, int TM, int TN, bool AKS, bool BKS, int OUT, int EPI>
static int gemm16_launch_t(int dtype, const GemmArgs& a, unsigned nblk, hipStream_t st) {
  using G = Geo<KB, NS, WM, TM, TN>;
  constexpr int lds = G::LDS;
  const dim3 block(64 * G::NW);
  if (dtype == 0) {
    if (lds > 65536) {
      static bool done = false;       // idempotent attribute of the code object, not library state
      if (!done) {
        MBV_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&k_gemm16<KB, NS, WM, TM, TN, AKS, BKS, OUT, EPI, __bf16>),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done = true;
      }
    }
    hipLaunchKernelGGL((k_gemm16<KB, NS, WM, TM, TN, AKS, BKS, OUT, EPI, __bf16>), dim3(nblk), block, lds, st, a);
  } else {
    if (lds > 65536) {
      static bool done = false;
      if (!done) {
        MBV_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&k_gemm16<KB, NS, WM, TM, TN, AKS, BKS, OUT, EPI, _Float16>),
            hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done = true;
      }
    }
    hipLaunchKernelGGL((k_gemm16<KB, NS, WM, TM, TN, AKS, BKS, OUT, EPI, _Float16>), dim3(nblk), block, lds, st, a);
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

template <int WM, int TM, int TN>
static int gemm16_launch_s(int layout, int atomic, int epi, int dtype, const GemmArgs& a, unsigned n, hipStream_t st) {
  if (layout == 0) {
    if (atomic) return gemm16_launch_t<WM, TM, TN, false, false, 1, EPI_NONE>(dtype, a, n, st);
    if constexpr (TN == 2) {
      if (epi == EPI_RELU) return gemm16_launch_t<WM, TM, TN, false, false, 0, EPI_RELU>(dtype, a, n, st);
      if (epi == EPI_GELU) return gemm16_launch_t<WM, TM, TN, false, false, 0, EPI_GELU>(dtype, a, n, st);
      if (epi == EPI_CONV) return gemm16_launch_t<WM, TM, TN, false, false, 0, EPI_CONV>(dtype, a, n, st);
      return gemm16_launch_t<WM, TM, TN, false, false, 0, EPI_NONE>(dtype, a, n, st);
    }
    return MBV_ERR_UNSUPPORTED;
  }
  if (layout == 1) {
    if constexpr (TN == 2) {
      if (epi == EPI_DRELU) return gemm16_launch_t<WM, TM, TN, false, true, 0, EPI_DRELU>(dtype, a, n, st);
      if (epi == EPI_DGELU) return gemm16_launch_t<WM, TM, TN, false, true, 0, EPI_DGELU>(dtype, a, n, st);
      return gemm16_launch_t<WM, TM, TN, false, true, 0, EPI_NONE>(dtype, a, n, st);
    }
    return MBV_ERR_UNSUPPORTED;
  }
  if (atomic) return gemm16_launch_t<WM, TM, TN, true, true, 1, EPI_NONE>(dtype, a, n, st);
  if constexpr (TN == 2) return gemm16_launch_t<WM, TM, TN, true, true, 0, EPI_NONE>(dtype, a, n, st);
  return MBV_ERR_UNSUPPORTED;
}

// Block shapes (Geo): 0 = 128 x 128, 1 = 256 x 128.  (A 384 x 192 weight-gradient shape — 3 x 3 tiles per wave, one
// workgroup per CU — was built and measured 2-2.6x SLOWER: with ~512 workgroups in flight the atomic traffic is
// workgroups x tile bytes, 151 MB instead of 33 MB per launch.)
static const int SHAPE_BM[2] = {128, 256}, SHAPE_BN[2] = {128, 128};

// Which block shape a problem gets: the largest shape that still (a) wastes little of its tiles on the edges of the
// (gm x gn) output and (b) leaves at least ~2 workgroup slots per CU busy (the weight gradients make up for small
// outputs with their split over the tokens).
static int gemm16_pick_shape(int atomic, long long gm, long long gn, long long work_units) {
  auto eff = [&](int s) {
    const long long tm = (gm + SHAPE_BM[s] - 1) / SHAPE_BM[s], tn = (gn + SHAPE_BN[s] - 1) / SHAPE_BN[s];
    return (double)(gm * gn) / (double)(tm * SHAPE_BM[s] * tn * SHAPE_BN[s]);
  };
  auto blocks = [&](int s) {
    return ((gm + SHAPE_BM[s] - 1) / SHAPE_BM[s]) * ((gn + SHAPE_BN[s] - 1) / SHAPE_BN[s]) * work_units;
  };
  // measured (scratch/bench_gemm.py, profiles/r02): 256 x 128 wins 5-13 % on STORE products with >= 128 such tiles;
  // products that end in atomics pay (workgroups x tile bytes) of atomic traffic, so they keep the smallest tile
  if (atomic) return 0;
  if (eff(1) >= 0.9 * eff(0) && blocks(1) >= 128) return 1;
  return 0;
}

static int gemm16_launch(int shape, int layout, int atomic, int epi, int dtype, const GemmArgs& a, int batch,
                         hipStream_t st) {
  const long long nblk = (long long)a.ntm * a.ntn * a.splits * batch;
  if (nblk <= 0) return MBV_OK;
  if (nblk > 0x7fffffffLL) return MBV_ERR_UNSUPPORTED;
  const unsigned n = (unsigned)nblk;
  if (shape == 1) return gemm16_launch_s<4, 2, 2>(layout, atomic, epi, dtype, a, n, st);
  return gemm16_launch_s<2, 2, 2>(layout, atomic, epi, dtype, a, n, st);
}

// Split of the contraction of an ATOMIC product over workgroups: about two workgroups per CU, at least 256 deep each.
static void gemm16_split(GemmArgs& a, int splits, long long contraction, int batch) {
  const int total_steps = (int)((contraction + KB - 1) / KB);
  int s = splits;
  if (s <= 0) {
    const long long tiles = (long long)a.ntm * a.ntn * batch;
    constexpr int target = 512;         // about two workgroups per CU (256 and 1024 measured slower on the step's shapes)
    s = (int)((target + tiles - 1) / tiles);
    if (s > total_steps * KB / 256) s = total_steps * KB / 256;
  }
  if (s < 1) s = 1;
  if (s > total_steps) s = total_steps;
  a.ksteps = (total_steps + s - 1) / s;
  a.splits = (total_steps + a.ksteps - 1) / a.ksteps;
}

static bool fits_2g(long long rows, long long ld) { return rows * ld * 2 < 0x7fff0000LL; }

extern "C" int mbv_gemm16_supported(int32_t layout, int64_t m, int64_t n, int64_t k) {
  if (layout < 0 || layout > 2 || m <= 0 || n <= 0 || k <= 0) return 0;
  if ((n & 7) || (k & 7)) return 0;          // 16-byte chunks; every other edge is handled by the range check
  return m * (n > k ? n : k) * 2 < 0x7fff0000LL ? 1 : 0;
}

// out (m, n) = epi(x (m, k) . w (n, k)^T + bias)
extern "C" int mbv_gemm16_nt(const void* x, const void* w, const float* bias, void* out, void* out_pre, int64_t m,
                             int64_t n, int64_t k, int64_t ldx, int64_t ldw, int64_t ldo, int32_t dtype,
                             int32_t out_f32, int32_t act, int32_t batch, int64_t stride_x, int64_t stride_w,
                             int64_t stride_o, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || batch < 0 || !x || !w || !out) return MBV_ERR_BAD_ARG;
  if (dtype < 0 || dtype > 1 || act < 0 || act > 2) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldx & 7) || (ldw & 7) || (ldo & 7) || ldx < k || ldw < k || ldo < n) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(out) |
       reinterpret_cast<size_t>(out_pre) | reinterpret_cast<size_t>(bias)) & 15)
    return MBV_ERR_UNSUPPORTED;
  if (!fits_2g(m, ldx) || !fits_2g(n, ldw)) return MBV_ERR_UNSUPPORTED;
  if (m == 0 || batch == 0) return MBV_OK;
  const int shape = gemm16_pick_shape(0, m, n, batch);
  const int BM = SHAPE_BM[shape], BN = SHAPE_BN[shape];
  GemmArgs a = {};
  a.a = x; a.b = w; a.c = out; a.c2 = out_pre; a.bias = bias;
  a.gm = (int)m; a.gn = (int)n; a.gk = (int)k;
  a.lda = (int)ldx; a.ldb = (int)ldw; a.ldc = (int)ldo;
  a.sa = stride_x; a.sb = stride_w; a.sc = stride_o;
  a.a_bytes = (unsigned)(((m - 1) * ldx + k) * 2); a.b_bytes = (unsigned)(((n - 1) * ldw + k) * 2);
  a.ntm = (int)((m + BM - 1) / BM); a.ntn = (int)((n + BN - 1) / BN); a.splits = 1; a.ksteps = (int)((k + KB - 1) / KB);
  a.out_f32 = out_f32;
  return gemm16_launch(shape, 0, 0, act, dtype, a, batch, (hipStream_t)stream);
}

// acc (m, n) f32 += x (m, k) . w (n, k)^T, the sum over k split over workgroups (f32 atomic adds): for few-row
// products with a long contraction (the mask-logit backward: 1000 x 256 outputs over 16 384 pixels)
extern "C" int mbv_gemm16_nt_acc(const void* x, const void* w, float* acc, int64_t m, int64_t n, int64_t k, int64_t ldx,
                                 int64_t ldw, int64_t ldacc, int32_t dtype, int32_t splits, int32_t batch,
                                 int64_t stride_x, int64_t stride_w, int64_t stride_acc, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || batch < 0 || !x || !w || !acc) return MBV_ERR_BAD_ARG;
  if (dtype < 0 || dtype > 1) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldx & 7) || (ldw & 7) || ldx < k || ldw < k || ldacc < n) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(acc)) & 15) return MBV_ERR_UNSUPPORTED;
  if (!fits_2g(m, ldx) || !fits_2g(n, ldw)) return MBV_ERR_UNSUPPORTED;
  if (m == 0 || batch == 0) return MBV_OK;
  const int shape = gemm16_pick_shape(1, m, n, (long long)batch * ((k + 511) / 512));
  const int BM = SHAPE_BM[shape], BN = SHAPE_BN[shape];
  GemmArgs a = {};
  a.a = x; a.b = w; a.c = acc;
  a.gm = (int)m; a.gn = (int)n; a.gk = (int)k;
  a.lda = (int)ldx; a.ldb = (int)ldw; a.ldc = (int)ldacc;
  a.sa = stride_x; a.sb = stride_w; a.sc = stride_acc;
  a.a_bytes = (unsigned)(((m - 1) * ldx + k) * 2); a.b_bytes = (unsigned)(((n - 1) * ldw + k) * 2);
  a.ntm = (int)((m + BM - 1) / BM); a.ntn = (int)((n + BN - 1) / BN);
  gemm16_split(a, splits, k, batch);
  a.out_f32 = 1;
  return gemm16_launch(shape, 0, 1, EPI_NONE, dtype, a, batch, (hipStream_t)stream);
}

// A 3 x 3 convolution (stride 1, padding 1) of 16-bit maps as one K17 product over k = (tap, channel) on the zero-bordered
// channels-last rows of csrc/conv_pad.hip (see mbv_conv3x3_gemm32s, csrc/gemm_f32s.hip, for the layout and the data-gradient
// form): out_rows[m][co] = sum_t sum_ci rows[m + shift_t][ci] wm[co][t C + ci].
extern "C" int mbv_conv3x3_gemm16(const void* rows, const void* wm, void* out_rows, int64_t batch, int64_t H, int64_t W, int64_t C,
                                  int64_t cout, int32_t dtype, int32_t out_f32, void* stream) {
  if (!rows || !wm || !out_rows || batch <= 0 || H <= 0 || W <= 0 || C <= 0 || cout <= 0 || dtype < 0 || dtype > 1)
    return MBV_ERR_BAD_ARG;
  if ((C % KB) || (cout & 7)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(rows) | reinterpret_cast<size_t>(wm) | reinterpret_cast<size_t>(out_rows)) & 15)
    return MBV_ERR_UNSUPPORTED;
  const int64_t G = W + 3, mp = batch * (H + 2) * (W + 2), total = mp + 2 * G;
  if (total * C * 2 >= 0x7fff0000LL || cout * 9 * C * 2 >= 0x7fff0000LL) return MBV_ERR_UNSUPPORTED;
  const int shape = gemm16_pick_shape(0, mp, cout, 1);
  const int BM = SHAPE_BM[shape], BN = SHAPE_BN[shape];
  GemmArgs a = {};
  a.a = rows; a.b = wm;
  a.c = reinterpret_cast<char*>(out_rows) + (size_t)G * (size_t)cout * (out_f32 ? 4 : 2);
  a.gm = (int)mp; a.gn = (int)cout; a.gk = (int)(9 * C);
  a.lda = (int)C; a.ldb = (int)(9 * C); a.ldc = (int)cout;
  a.a_bytes = (unsigned)(total * C * 2); a.b_bytes = (unsigned)(cout * 9 * C * 2);
  a.ntm = (int)((mp + BM - 1) / BM); a.ntn = (int)((cout + BN - 1) / BN); a.splits = 1; a.ksteps = (int)(9 * C / KB);
  a.out_f32 = out_f32;
  a.tap_steps = (int)(C / KB);
  for (int t = 0; t < 9; ++t) {
    const int64_t shift = (t / 3 - 1) * (W + 2) + (t % 3 - 1);
    a.tap[t] = (int)((G + shift - t) * C * 2);          // >= 0: the row of tap t, minus the t C columns k has advanced
  }
  return gemm16_launch(shape, 0, 0, EPI_CONV, dtype, a, 1, (hipStream_t)stream);
}

extern "C" size_t mbv_gemm16_nn_workspace_bytes(int64_t m, int64_t k, int32_t batch) {
  // one partial row per 64 output rows (a wave row of any block shape)
  return (size_t)((m + 63) / 64 + 4) * (size_t)k * 4 * (size_t)(batch > 0 ? batch : 1);
}

// Partial column-sum rows an mbv_gemm16_nn of this shape leaves in its workspace: one per 64 output rows of a tile row.
extern "C" int64_t mbv_gemm16_nn_part_rows(int64_t m, int64_t k, int32_t batch) {
  if (m <= 0 || k <= 0 || batch <= 0) return 0;
  const int BM = SHAPE_BM[gemm16_pick_shape(0, m, k, batch)];
  return (int64_t)(BM / 64) * ((m + BM - 1) / BM) * batch;
}

// out (m, k) = act'(aux) * (g (m, n) . w (n, k));  colsum (k) += column sums of out.  parts_only: the per-wave-row partial
// sums stay in `workspace` ((mbv_gemm16_nn_part_rows, k) f32, every element written) and the caller reduces them later.
static int gemm16_nn_impl(const void* g, const void* w, void* out, const void* aux, float* colsum, bool parts_only, int64_t m,
                          int64_t n, int64_t k, int64_t ldg, int64_t ldw, int64_t ldo, int64_t ldaux, int32_t dtype,
                          int32_t out_f32, int32_t act, int32_t batch, int64_t stride_g, int64_t stride_w,
                          int64_t stride_o, void* workspace, size_t workspace_bytes, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || batch < 0 || !g || !w || !out) return MBV_ERR_BAD_ARG;
  if (dtype < 0 || dtype > 1 || act < 0 || act > 2 || (act && !aux)) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldg & 7) || (ldw & 7) || (ldo & 7) || (ldaux & 7) || ldg < n || ldw < k || ldo < k)
    return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(out) |
       reinterpret_cast<size_t>(aux) | reinterpret_cast<size_t>(workspace)) & 15)
    return MBV_ERR_UNSUPPORTED;
  if (!fits_2g(m, ldg) || !fits_2g(n, ldw)) return MBV_ERR_UNSUPPORTED;
  const bool sums = colsum != nullptr || parts_only;
  if (sums && (!workspace || workspace_bytes < mbv_gemm16_nn_workspace_bytes(m, k, batch))) return MBV_ERR_WORKSPACE;
  if (m == 0 || batch == 0) return MBV_OK;
  const int shape = gemm16_pick_shape(0, m, k, batch);
  const int BM = SHAPE_BM[shape], BN = SHAPE_BN[shape];
  GemmArgs a = {};
  a.a = g; a.b = w; a.c = out; a.aux = aux; a.colsum_rows = sums ? reinterpret_cast<float*>(workspace) : nullptr;
  a.gm = (int)m; a.gn = (int)k; a.gk = (int)n;
  a.lda = (int)ldg; a.ldb = (int)ldw; a.ldc = (int)ldo; a.ldx = (int)ldaux;
  a.sa = stride_g; a.sb = stride_w; a.sc = stride_o;
  a.a_bytes = (unsigned)(((m - 1) * ldg + n) * 2); a.b_bytes = (unsigned)(((n - 1) * ldw + k) * 2);
  a.ntm = (int)((m + BM - 1) / BM); a.ntn = (int)((k + BN - 1) / BN); a.splits = 1; a.ksteps = (int)((n + KB - 1) / KB);
  a.out_f32 = out_f32;
  const int rc = gemm16_launch(shape, 1, 0, act == 0 ? EPI_NONE : (act == 1 ? EPI_DRELU : EPI_DGELU), dtype, a, batch,
                               (hipStream_t)stream);
  if (rc != MBV_OK || !colsum || parts_only) return rc;
  const int rows = (BM / 64) * a.ntm * batch;
  int gy = rows / 16;
  if (gy < 1) gy = 1;
  if (gy > 32) gy = 32;
  hipLaunchKernelGGL(k_sum_rows, dim3((unsigned)((k + 63) / 64), (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float*>(workspace), rows, (int)k, colsum);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_gemm16_nn(const void* g, const void* w, void* out, const void* aux, float* colsum, int64_t m,
                             int64_t n, int64_t k, int64_t ldg, int64_t ldw, int64_t ldo, int64_t ldaux, int32_t dtype,
                             int32_t out_f32, int32_t act, int32_t batch, int64_t stride_g, int64_t stride_w,
                             int64_t stride_o, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm16_nn_impl(g, w, out, aux, colsum, false, m, n, k, ldg, ldw, ldo, ldaux, dtype, out_f32, act, batch, stride_g,
                        stride_w, stride_o, workspace, workspace_bytes, stream);
}

// mbv_gemm16_nn whose column sums are left as partial rows: parts (mbv_gemm16_nn_part_rows(m, k, batch), k) f32, contiguous,
// 16-byte aligned, at least mbv_gemm16_nn_workspace_bytes(m, k, batch) long; their sum over the rows is the column sum.
extern "C" int mbv_gemm16_nn_parts(const void* g, const void* w, void* out, const void* aux, float* parts, size_t parts_bytes,
                                   int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldw, int64_t ldo, int64_t ldaux,
                                   int32_t dtype, int32_t out_f32, int32_t act, int32_t batch, int64_t stride_g,
                                   int64_t stride_w, int64_t stride_o, void* stream) {
  if (!parts) return MBV_ERR_BAD_ARG;
  return gemm16_nn_impl(g, w, out, aux, nullptr, true, m, n, k, ldg, ldw, ldo, ldaux, dtype, out_f32, act, batch, stride_g,
                        stride_w, stride_o, parts, parts_bytes, stream);
}

// accumulate != 0:  dw (n, k) f32 += g (m, n)^T . x (m, k), the sum over m split over workgroups (f32 atomic adds)
// accumulate == 0:  dw (n, k) = g^T . x stored (16-bit or f32), one workgroup per tile
// Bytes of workspace with which an accumulating mbv_gemm16_tn replaces its f32 atomics by partial results + one
// owner-adds pass (dw contiguous, batch 1): room for the largest split it may choose.
extern "C" size_t mbv_gemm16_tn_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  const long long tiles = ((n + 127) / 128) * ((k + 127) / 128);
  long long s = (512 + tiles - 1) / tiles;
  const long long cap = (m + 255) / 256;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (size_t)(s + 1) * (size_t)n * (size_t)k * 4;
}

extern "C" int mbv_gemm16_tn(const void* g, const void* x, void* dw, int64_t m, int64_t n, int64_t k, int64_t ldg,
                             int64_t ldx, int64_t lddw, int32_t dtype, int32_t accumulate, int32_t out_f32,
                             int32_t splits, int32_t batch, int64_t stride_g, int64_t stride_x, int64_t stride_dw,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || batch < 0 || !g || !x || !dw) return MBV_ERR_BAD_ARG;
  if (dtype < 0 || dtype > 1) return MBV_ERR_BAD_ARG;
  if ((n & 7) || (k & 7) || (ldg & 7) || (ldx & 7) || ldg < n || ldx < k || lddw < k) return MBV_ERR_UNSUPPORTED;
  if (!accumulate && (lddw & 7)) return MBV_ERR_UNSUPPORTED;
  if ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dw)) & 15) return MBV_ERR_UNSUPPORTED;
  if (!fits_2g(m, ldg) || !fits_2g(m, ldx)) return MBV_ERR_UNSUPPORTED;
  if (m == 0 || batch == 0) return MBV_OK;
  const int shape = gemm16_pick_shape(accumulate ? 1 : 0, n, k, accumulate ? (long long)batch * ((m + 511) / 512) : batch);
  const int BM = SHAPE_BM[shape], BN = SHAPE_BN[shape];
  GemmArgs a = {};
  a.a = g; a.b = x; a.c = dw;
  a.gm = (int)n; a.gn = (int)k; a.gk = (int)m;
  a.lda = (int)ldg; a.ldb = (int)ldx; a.ldc = (int)lddw;
  a.sa = stride_g; a.sb = stride_x; a.sc = stride_dw;
  a.a_bytes = (unsigned)(((m - 1) * ldg + n) * 2); a.b_bytes = (unsigned)(((m - 1) * ldx + k) * 2);
  a.ntm = (int)((n + BM - 1) / BM); a.ntn = (int)((k + BN - 1) / BN);
  if (accumulate) {
    gemm16_split(a, splits, m, batch);
  } else {
    a.splits = 1;
    a.ksteps = (int)((m + KB - 1) / KB);
  }
  a.out_f32 = accumulate ? 1 : out_f32;
  // partial results + owner-adds instead of atomics: plain full-line stores (~6 TB/s against ~1.3 for atomics) and
  // a bit-reproducible sum; needs a contiguous dw, 16-byte alignment and a workspace for the parts
  if (accumulate && a.splits > 1 && batch == 1 && workspace && lddw == k && (n * k) % 4 == 0 &&
      (reinterpret_cast<size_t>(workspace) & 15) == 0 &&
      workspace_bytes >= (size_t)a.splits * (size_t)n * (size_t)k * 4) {
    GemmArgs b = a;
    b.c = workspace;
    b.ssplit = n * k;
    b.ldc = (int)k;
    const int rc = gemm16_launch(shape, 2, 0, EPI_NONE, dtype, b, batch, (hipStream_t)stream);
    if (rc != MBV_OK) return rc;
    const long long total = n * k;
    // outputs of fewer than 64 k elements with many parts: a few part ranges per element keep the chip busy
    const unsigned gy = (total < 65536 && a.splits >= 32) ? (unsigned)((a.splits + 15) / 16) : 1u;
    hipLaunchKernelGGL(k_add_parts, dim3((unsigned)((total / 4 + 255) / 256), gy), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(workspace), a.splits, total, reinterpret_cast<float*>(dw));
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
  return gemm16_launch(shape, 2, accumulate ? 1 : 0, EPI_NONE, dtype, a, batch, (hipStream_t)stream);
}

// ---- grouped weight gradients (see k_gemm16_tn_group) -------------------------------------------------------------
// Token depth of one work item (2048 / 8192 measured slower over the step's 62 products).
static constexpr int tn_group_depth() { return 4096; }

static void tn_group_split(int64_t m, int& splits, int& ksteps) {
  const int total_steps = (int)((m + KB - 1) / KB);
  int s = (int)((m + tn_group_depth() - 1) / tn_group_depth());
  if (s < 1) s = 1;
  ksteps = (total_steps + s - 1) / s;
  splits = (total_steps + ksteps - 1) / ksteps;
}

static bool tn_group_entry_ok(const void* g, const void* x, const float* dw, int64_t m, int64_t n, int64_t k,
                              int64_t ldg, int64_t ldx) {
  if (m <= 0 || n <= 0 || k <= 0 || !g || !x || !dw) return false;
  if ((n & 7) || (k & 7) || (ldg & 7) || (ldx & 7) || ldg < n || ldx < k) return false;
  if ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dw)) & 15) return false;
  return fits_2g(m, ldg) && fits_2g(m, ldx) && n * k < 0x7fffffffLL;
}

extern "C" size_t mbv_gemm16_tn_group_workspace_bytes(const int64_t* m, const int64_t* n, const int64_t* k,
                                                      int32_t count) {
  size_t total = 0;
  for (int i = 0; i < count; ++i) {
    if (m[i] <= 0 || n[i] <= 0 || k[i] <= 0) continue;
    int splits, ksteps;
    tn_group_split(m[i], splits, ksteps);
    if (splits > 1) total += (size_t)splits * (size_t)n[i] * (size_t)k[i] * 4;
  }
  return total;
}

// dw[i] (n[i], k[i]) f32, contiguous  +=  g[i] (m[i], n[i])^T . x[i] (m[i], k[i])   for i < count, in one GEMM launch
// (+ one parts-add launch) per 48 products.  Every array argument is a HOST array of length count.
extern "C" int mbv_gemm16_tn_group(const void* const* g, const void* const* x, float* const* dw, const int64_t* m,
                                   const int64_t* n, const int64_t* k, const int64_t* ldg, const int64_t* ldx,
                                   int32_t count, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (count < 0 || dtype < 0 || dtype > 1) return MBV_ERR_BAD_ARG;
  if (count > 0 && (!g || !x || !dw || !m || !n || !k || !ldg || !ldx)) return MBV_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    if (m[i] < 0) return MBV_ERR_BAD_ARG;
    if (m[i] > 0 && !tn_group_entry_ok(g[i], x[i], dw[i], m[i], n[i], k[i], ldg[i], ldx[i])) return MBV_ERR_UNSUPPORTED;
  }
  if (mbv_gemm16_tn_group_workspace_bytes(m, n, k, count) > workspace_bytes) return MBV_ERR_WORKSPACE;
  if (workspace_bytes && (!workspace || (reinterpret_cast<size_t>(workspace) & 15))) return MBV_ERR_WORKSPACE;
  using G = Geo<KB, NS, 2, 2, 2>;
  float* ws = reinterpret_cast<float*>(workspace);
  for (int base = 0; base < count; base += kTnGroup) {
    const int cnt = count - base < kTnGroup ? count - base : kTnGroup;
    TnGroupArgs a;
    PartsGroupArgs pa;
    a.n = 0; pa.n = 0;
    long long items = 0, pblocks = 0;
    for (int j = 0; j < cnt; ++j) {
      const int i = base + j;
      if (m[i] == 0) continue;
      TnEntry& e = a.e[a.n++];
      e.g = g[i]; e.x = x[i];
      e.m = (int)m[i]; e.n = (int)n[i]; e.k = (int)k[i]; e.ldg = (int)ldg[i]; e.ldx = (int)ldx[i]; e.ldo = (int)k[i];
      e.ntn = (int)((k[i] + 127) / 128);
      tn_group_split(m[i], e.splits, e.ksteps);
      e.item_begin = (int)items;
      items += (long long)((n[i] + 127) / 128) * e.ntn * e.splits;
      if (e.splits == 1) {
        e.out = dw[i];
      } else {
        e.out = ws;
        PartsEntry& q = pa.e[pa.n++];
        q.part = ws; q.out = dw[i]; q.n = n[i] * k[i]; q.parts = e.splits; q.block_begin = (int)pblocks;
        pblocks += (q.n / 4 + 255) / 256;
        ws += (size_t)e.splits * (size_t)q.n;
      }
    }
    if (a.n == 0) continue;
    if (items > 0x7fffffffLL || pblocks > 0x7fffffffLL) return MBV_ERR_UNSUPPORTED;
    a.total_items = (int)items;
    if (dtype == 0)
      hipLaunchKernelGGL((k_gemm16_tn_group<KB, NS, __bf16>), dim3((unsigned)items), dim3(256), G::LDS,
                         (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((k_gemm16_tn_group<KB, NS, _Float16>), dim3((unsigned)items), dim3(256), G::LDS,
                         (hipStream_t)stream, a);
    MBV_CHECK_LAUNCH();
    if (pa.n) {
      hipLaunchKernelGGL(k_add_parts_group, dim3((unsigned)pblocks), dim3(256), 0, (hipStream_t)stream, pa);
      MBV_CHECK_LAUNCH();
    }
  }
  return MBV_OK;
}
