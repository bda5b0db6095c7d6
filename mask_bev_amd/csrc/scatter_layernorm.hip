// K3 — PointPillarsScatter fused with the (C, ny, nx) LayerNorm, forward and backward, gfx950.
//
// Replaces mask_bev/models/encoders/mask_bev_encoders.py:122-123 (mmdet3d PointPillarsScatter) and
// :75,:92 (nn.LayerNorm([C, ny, nx], eps=1e-3)).
//
// HBM-bound.  The reference writes a dense zero canvas (134 MB / scan at 128x512x512), reads it back for
// the statistics and again for the affine.  Here the canvas never exists:
//   * mean / variance come from the V x C pillar features alone (every empty cell is exactly 0);
//   * the apply kernel streams weight / bias once per BATCH (registers, reused for every scan) and
//     writes the output once; pillar rows are gathered through an LDS tile that transposes
//     (pillar, channel) rows into the x-contiguous NCHW layout so every global access is a full
//     16 B / lane coalesced access.
// Algorithmic bytes per batch: 2*C*G*4 (weight + bias) + B*C*G*4 (out) + V*C*4 (+ B*G*4 cell map).
#include "common.hpp"
#include "adam.hpp"

namespace {

constexpr int kCT = 32;  // channels per block tile (4 waves x 8 channels)

// ---------------------------------------------------------------------------------------------
// statistics over the sparse features
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ln_stats(const float* __restrict__ feats,
                                                  const int32_t* __restrict__ pillar_batch_start, int channels,
                                                  double* __restrict__ sums /* [batch][2] */) {
  const int b = blockIdx.y;
  const int64_t begin = (int64_t)pillar_batch_start[b] * channels;
  const int64_t end = (int64_t)pillar_batch_start[b + 1] * channels;
  double s = 0.0, q = 0.0;
  // channels % 4 == 0 is enforced by the launcher, so the range is float4 aligned
  for (int64_t i = begin + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < end;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    const float4 v = *reinterpret_cast<const float4*>(feats + i);
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  __shared__ double red[2][4];
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[b * 2 + 0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[b * 2 + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

__global__ void k_ln_finalize(const double* __restrict__ sums, int batch, double inv_count, float eps,
                              float* __restrict__ stats) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double mean = sums[b * 2] * inv_count;
  double var = sums[b * 2 + 1] * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[b * 2 + 0] = (float)mean;
  stats[b * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---------------------------------------------------------------------------------------------
// tile helpers.  A block owns one BEV row y, XT = 64*VEC consecutive cells and kCT channels.
// ---------------------------------------------------------------------------------------------
template <int VEC, int THREADS = 256>
struct Tile {
  static constexpr int XT = 64 * VEC;          // cells per tile
  static constexpr int LD = XT + 4;            // padded row (keeps float4 reads 16 B aligned)
  static constexpr int GROUPS = THREADS / XT;  // threads sharing one cell in the gather phase
  static constexpr int CH_PER_THREAD = kCT / GROUPS;
};

// gather feats[pid][c0 .. c0+kCT) of the tile's cells into lds[c][x] (zeros for empty cells)
template <int VEC, int THREADS = 256>
__device__ __forceinline__ void gather_tile(const float* __restrict__ feats, int channels, int c0, int32_t pid,
                                            float* __restrict__ lds) {
  using T = Tile<VEC, THREADS>;
  const int cell = threadIdx.x % T::XT;
  const int grp = threadIdx.x / T::XT;
  const int cbeg = grp * T::CH_PER_THREAD;
#pragma unroll
  for (int k = 0; k < T::CH_PER_THREAD; k += 4) {
    const int c = c0 + cbeg + k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pid >= 0 && c < channels) v = *reinterpret_cast<const float4*>(feats + (int64_t)pid * channels + c);
    lds[(cbeg + k + 0) * T::LD + cell] = v.x;
    lds[(cbeg + k + 1) * T::LD + cell] = v.y;
    lds[(cbeg + k + 2) * T::LD + cell] = v.z;
    lds[(cbeg + k + 3) * T::LD + cell] = v.w;
  }
}

// the same gather in two halves: the loads of the NEXT scan's rows are issued before the current scan's outputs are
// computed and stored, so that the (cell -> pillar id -> pillar row) chain of scan b+1 overlaps the store phase of b
template <int VEC, int THREADS = 256>
__device__ __forceinline__ void gather_load(const float* __restrict__ feats, int channels, int c0, int32_t pid,
                                            float4 (&r)[Tile<VEC, THREADS>::CH_PER_THREAD / 4]) {
  using T = Tile<VEC, THREADS>;
  const int grp = threadIdx.x / T::XT;
  const int cbeg = grp * T::CH_PER_THREAD;
#pragma unroll
  for (int k = 0; k < T::CH_PER_THREAD; k += 4) {
    const int c = c0 + cbeg + k;
    r[k / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pid >= 0 && c < channels) r[k / 4] = *reinterpret_cast<const float4*>(feats + (int64_t)pid * channels + c);
  }
}
template <int VEC, int THREADS = 256>
__device__ __forceinline__ void gather_store(const float4 (&r)[Tile<VEC, THREADS>::CH_PER_THREAD / 4],
                                             float* __restrict__ lds) {
  using T = Tile<VEC, THREADS>;
  const int cell = threadIdx.x % T::XT;
  const int grp = threadIdx.x / T::XT;
  const int cbeg = grp * T::CH_PER_THREAD;
#pragma unroll
  for (int k = 0; k < T::CH_PER_THREAD; k += 4) {
    lds[(cbeg + k + 0) * T::LD + cell] = r[k / 4].x;
    lds[(cbeg + k + 1) * T::LD + cell] = r[k / 4].y;
    lds[(cbeg + k + 2) * T::LD + cell] = r[k / 4].z;
    lds[(cbeg + k + 3) * T::LD + cell] = r[k / 4].w;
  }
}

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<1> {
  using type = float;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else {
    r[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    *p = r[0];
  }
}

// ---------------------------------------------------------------------------------------------
// Patch-token layout (PATCH = true, VEC = 4, patch = 4): the consumer of the BEV map is the backbone's
// non-overlapping 4 x 4 patch projection (mmdet PatchEmbed, swin.py:579-586), a GEMM over 16*C values per token.
// Writing the map as bf16 rows  tokens[b][y/4][x/4][(y%4)*4C + c*4 + x%4]  makes that projection (and its data
// gradient) one plain GEMM with no layout transform, no cast pass and half the bytes; the gradient comes back in
// the same layout.  A thread then owns NCH consecutive channels of one token column (4 cells): its NCH*4 values
// are contiguous in the row, and the lanes of a wave that share a token cover a full 256 B run of it.
// ---------------------------------------------------------------------------------------------
template <bool PATCH, int NCH>
struct Map {
  // first channel (within the tile) and first cell (within the tile) of this thread
  static __device__ __forceinline__ int chan0(int wave, int lane) {
    return PATCH ? (lane & (kCT / NCH - 1)) * NCH : wave * NCH;
  }
  static __device__ __forceinline__ int cell0(int wave, int lane, int vec) {
    constexpr int LPT = kCT / NCH;                  // lanes per token
    return PATCH ? (wave * (64 / LPT) + lane / LPT) * 4 : lane * vec;
  }
};

__device__ __forceinline__ int64_t patch_row_offset(int b, int channels, int ny, int nx, int y, int xv, int c) {
  const int64_t token = ((int64_t)b * (ny >> 2) + (y >> 2)) * (nx >> 2) + (xv >> 2);
  return token * (16 * (int64_t)channels) + (int64_t)(y & 3) * 4 * channels + (int64_t)c * 4;
}

// ---------------------------------------------------------------------------------------------
// forward apply
// ---------------------------------------------------------------------------------------------
// two values <-> one dword of 16-bit storage (KIND = MBV_DT_BF16 or MBV_DT_F16)
template <int KIND>
__device__ __forceinline__ unsigned pack_lo2(float a, float b) {
  if constexpr (KIND == MBV_DT_F16)
    return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) |
           ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
  else
    return pack_bf16x2(a, b);
}
template <int KIND>
__device__ __forceinline__ void unpack_lo2(unsigned u, float& a, float& b) {
  if constexpr (KIND == MBV_DT_F16) {
    a = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
    b = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
  } else {
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xffff0000u);
  }
}

template <int VEC, int PATCH>     // PATCH: 0 = (B, C, ny, nx) f32 map, MBV_DT_BF16 / MBV_DT_F16 = 4 x 4 patch rows
__global__ void __launch_bounds__(256) k_ln_apply(const float* __restrict__ feats,
                                                  const int32_t* __restrict__ cell_to_pillar,
                                                  const float* __restrict__ weight, const float* __restrict__ bias,
                                                  const float* __restrict__ stats, int batch, int channels, int ny,
                                                  int nx, int xtiles, void* __restrict__ out_,
                                                  unsigned* __restrict__ amax_out /* f32 map only; may be null */) {
  using T = Tile<VEC>;
  float omax = 0.f;
  __shared__ __attribute__((aligned(16))) float lds[kCT * T::LD];
  constexpr int kTokDw = kCT * 2 + 2;                         // dwords per token in the turn tile (64 data + 2 pad)
  __shared__ __attribute__((aligned(8))) uint32_t olds[PATCH ? 64 * kTokDw : 2];
  const int y = blockIdx.x / xtiles;
  const int x0 = (blockIdx.x % xtiles) * T::XT;
  const int c0 = blockIdx.y * kCT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t cells = (int64_t)ny * nx;
  // apply-phase ownership: a wave owns 8 channels, a lane VEC consecutive cells of each, so the affine parameters are
  // read as whole KiB rows.  For patch rows the results are turned through a second LDS tile (token-major, padded to
  // 66 dwords per token: conflict-free 8-byte writes and reads) and leave as whole 256-byte runs of the token rows.
  const int ch0 = wave * 8;
  const int cv = lane * VEC;
  auto chan_of = [&](int k) { return ch0 + k; };
  auto cell_of = [&](int) { return cv; };
  // affine parameters: read once, reused for every scan of the batch
  float w[8][VEC], bz[8][VEC];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + chan_of(k);
    const int xk = x0 + cell_of(k);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { w[k][e] = 0.f; bz[k][e] = 0.f; }
    if (xk < nx && c < channels) {             // nx % VEC == 0 → the whole vector is in range
      const int64_t o = ((int64_t)c * ny + y) * nx + xk;
      load_vec<VEC>(weight + o, w[k]);
      load_vec<VEC>(bias + o, bz[k]);
    }
  }
  const int gcell = threadIdx.x % T::XT;
  const bool g_ok = x0 + gcell < nx;
  const int64_t cell_off = (int64_t)y * nx + x0 + gcell;
  float4 rows[T::CH_PER_THREAD / 4];
  {
    const int32_t pid0 = g_ok ? cell_to_pillar[cell_off] : -1;
    gather_load<VEC>(feats, channels, c0, pid0, rows);
  }
  int32_t pid_next = (g_ok && batch > 1) ? cell_to_pillar[cells + cell_off] : -1;
  for (int b = 0; b < batch; ++b) {
    const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
    gather_store<VEC>(rows, lds);
    __syncthreads();
    if (b + 1 < batch) {                       // scan b+1: rows requested now, its successor's pillar id too
      gather_load<VEC>(feats, channels, c0, pid_next, rows);
      pid_next = (g_ok && b + 2 < batch) ? cell_to_pillar[(int64_t)(b + 2) * cells + cell_off] : -1;
    }
    {
      if constexpr (PATCH) {
        // channels % kCT == 0 in this mode.  token = lane (4 cells), channel cl = ch0 + k: 4 bf16 = 2 dwords
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f[VEC];
          load_vec<VEC>(&lds[(ch0 + k) * T::LD + cv], f);
          const float a0 = (f[0] - mean) * rstd * w[k][0] + bz[k][0];
          const float a1 = (f[1] - mean) * rstd * w[k][1] + bz[k][1];
          const float a2 = (f[2] - mean) * rstd * w[k][2] + bz[k][2];
          const float a3 = (f[3] - mean) * rstd * w[k][3] + bz[k][3];
          uint2 u;
          u.x = pack_lo2<PATCH>(a0, a1);
          u.y = pack_lo2<PATCH>(a2, a3);
          *reinterpret_cast<uint2*>(&olds[lane * kTokDw + (ch0 + k) * 2]) = u;
        }
        __syncthreads();
        // 64 tokens x 256 B: a wave instruction stores two whole token runs (32 lanes x 8 B each)
        unsigned short* obase = reinterpret_cast<unsigned short*>(out_);
#pragma unroll
        for (int it = 0; it < (64 * kCT * 2) / (256 * 2); ++it) {
          const int idx = it * 256 + threadIdx.x;
          const int tok = idx >> 5, piece = idx & 31;             // 32 pieces of 8 bytes per token
          const int xt = x0 + tok * 4;
          if (xt < nx) {
            const uint2 u = *reinterpret_cast<const uint2*>(&olds[tok * kTokDw + piece * 2]);
            unsigned short* orow = obase + patch_row_offset(b, channels, ny, nx, y, xt, c0) + piece * 4;
            *reinterpret_cast<uint2*>(orow) = u;
          }
        }
      } else {
        float* out = reinterpret_cast<float*>(out_);
        const int xv = x0 + cv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int cl = ch0 + k;
          const int c = c0 + cl;
          if (xv < nx && c < channels) {
            float f[VEC], r[VEC];
            load_vec<VEC>(&lds[cl * T::LD + cv], f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              r[e] = (f[e] - mean) * rstd * w[k][e] + bz[k][e];
              omax = fmaxf(omax, fabsf(r[e]));
            }
            store_vec<VEC>(out + (((int64_t)b * channels + c) * ny + y) * nx + xv, r);
          }
        }
      }
    }
    // NCHW: the f32 tile is about to be overwritten by the next scan's rows.  Patch rows: every read of the f32 tile
    // precedes the barrier above, and the turn tile is rewritten only after the next scan's first barrier.
    if constexpr (!PATCH) __syncthreads();
  }
  if constexpr (!PATCH) {
    // fp32 compute: the absmax record of the map for the K20 patch projection behind it — one max-combine per workgroup
    if (amax_out) {
      omax = wave_max(omax);
      if (lane == 0) olds[wave & 1] = 0u;
      __syncthreads();
      if (lane == 0) atomicMax(&olds[0], __float_as_uint(omax) & 0x7fffffffu);
      __syncthreads();
      if (threadIdx.x == 0 && olds[0]) atomicMax(amax_out + ((blockIdx.x + blockIdx.y * gridDim.x) & 63), olds[0]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, pass 1: grad_weight / grad_bias, per-scan sums of (g*w) and (g*w*xhat), and g*w at the
// occupied cells scattered back into (pillar, channel) rows.
// ---------------------------------------------------------------------------------------------
// ADAM (round 6, one GPU): the AdamW update of the two (C, ny, nx) affine parameters — a third of the model's parameters —
// happens HERE, where their gradients are complete in registers (the batch sum is per thread): the gradient never goes to
// memory, so its accumulate read + write, the optimizer pass's read of it and its zero fill (16 B per parameter, 1.07 GB
// at 128 x 512 x 512) are gone.  `weight` is then read-modify-written; grad_weight / grad_bias are not touched.
struct LnAdam {
  float* bias;                      // the LayerNorm bias parameter (the weight is the kernel's `weight`)
  float *m_w, *v_w, *m_b, *v_b;     // exp_avg / exp_avg_sq of weight and bias
  unsigned short *sh_w, *sh_b;      // 16-bit shadows (nullable)
  AdamArgs a;
};

template <int VEC, int PATCH, bool ADAM = false>
__global__ void __launch_bounds__(512) k_ln_bwd_dense(const void* __restrict__ grad_out_,
                                                      const float* __restrict__ feats,
                                                      const int32_t* __restrict__ cell_to_pillar,
                                                      const float* __restrict__ weight,
                                                      const float* __restrict__ stats, int batch, int channels,
                                                      int ny, int nx, int xtiles, float* __restrict__ grad_feats,
                                                      float* __restrict__ grad_weight, float* __restrict__ grad_bias,
                                                      int accumulate, double* __restrict__ sums /* [batch][2] */,
                                                      const LnAdam ad = LnAdam{}) {
  // 8 waves x 4 channels (126 VGPRs: two workgroups per CU).  Per scan the 4 grad_out vectors of a thread are
  // requested FIRST, so that they are in flight underneath the two dependent round trips of the gather (cell →
  // pillar id → pillar row), and the pillar id of the next scan is prefetched.  The per-scan sums meet in LDS and
  // leave as one pair of f64 atomics per block and scan after the loop.  (Measured at B=4, 128 x 512 x 512:
  // 463 us with 4 waves x 8 channels and the reduction + atomics inside the loop, 244 us in this form; a
  // register-only variant without the LDS tile — occupied cells as per-lane float4 loads — ran at 300-370 us:
  // its dependent, divergent loads sit on every wave's critical path.)
  constexpr int THREADS = 512, WAVES = THREADS / 64, CPW = kCT / WAVES;
  using T = Tile<VEC, THREADS>;
  using M = Map<false, CPW>;           // a wave owns CPW channels, a lane VEC cells: parameters move as whole KiB rows
  __shared__ __attribute__((aligned(16))) float lds[kCT * T::LD];
  // patch rows: the token-major gradient is read in whole 256-byte runs and turned through this tile (66 dwords per
  // token, as in the forward) into the (channel, cell) ownership above
  constexpr int kTokDw = kCT * 2 + 2;
  __shared__ __attribute__((aligned(8))) uint32_t gtile[PATCH ? 64 * kTokDw : 2];
  constexpr int MAXB = 16;
  __shared__ double acc[MAXB][2][WAVES];
  const int y = blockIdx.x / xtiles;
  const int x0 = (blockIdx.x % xtiles) * T::XT;
  const int c0 = blockIdx.y * kCT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t cells = (int64_t)ny * nx;
  const int ch0 = M::chan0(wave, lane), cv = M::cell0(wave, lane, VEC);
  const int xv = x0 + cv;
  const bool x_ok = xv < nx;
  float w[CPW][VEC], dw[CPW][VEC], db[CPW][VEC];
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int c = c0 + ch0 + k;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { w[k][e] = 0.f; dw[k][e] = 0.f; db[k][e] = 0.f; }
    if (x_ok && c < channels) load_vec<VEC>(weight + ((int64_t)c * ny + y) * nx + xv, w[k]);
  }
  const int gcell = threadIdx.x % T::XT;
  const int grp = threadIdx.x / T::XT;
  for (int bb0 = 0; bb0 < batch; bb0 += MAXB) {
    int32_t pid_next = -1;
    if (x0 + gcell < nx) pid_next = cell_to_pillar[(int64_t)bb0 * cells + (int64_t)y * nx + x0 + gcell];
    const int bend = min(batch, bb0 + MAXB);
#pragma unroll 1
    for (int b = bb0; b < bend; ++b) {
      const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
      const int32_t pid = pid_next;
      float g[CPW][VEC];
      uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
      const int g_tok = wave * 8 + (lane >> 3), g_piece = lane & 7;     // 8 lanes x 32 B = one token's 256-byte run
      if constexpr (PATCH) {
        const int xt = x0 + g_tok * 4;
        if (xt < nx) {
          const unsigned short* grow = reinterpret_cast<const unsigned short*>(grad_out_) +
                                       patch_row_offset(b, channels, ny, nx, y, xt, c0 + g_piece * 4);
          q0 = *reinterpret_cast<const uint4*>(grow);
          q1 = *reinterpret_cast<const uint4*>(grow + 8);
        }
      } else {
        const float* grad_out = reinterpret_cast<const float*>(grad_out_);
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int c = c0 + ch0 + k;
#pragma unroll
          for (int e = 0; e < VEC; ++e) g[k][e] = 0.f;
          if (x_ok && c < channels) load_vec<VEC>(grad_out + (((int64_t)b * channels + c) * ny + y) * nx + xv, g[k]);
        }
      }
      pid_next = -1;
      if (b + 1 < bend && x0 + gcell < nx)
        pid_next = cell_to_pillar[(int64_t)(b + 1) * cells + (int64_t)y * nx + x0 + gcell];
      gather_tile<VEC, THREADS>(feats, channels, c0, pid, lds);
      if constexpr (PATCH) {
        uint32_t* gt = &gtile[g_tok * kTokDw + g_piece * 8];
        *reinterpret_cast<uint2*>(gt) = make_uint2(q0.x, q0.y);
        *reinterpret_cast<uint2*>(gt + 2) = make_uint2(q0.z, q0.w);
        *reinterpret_cast<uint2*>(gt + 4) = make_uint2(q1.x, q1.y);
        *reinterpret_cast<uint2*>(gt + 6) = make_uint2(q1.z, q1.w);
      }
      __syncthreads();
      if constexpr (PATCH) {                     // token = lane, channels ch0 .. ch0 + CPW - 1
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const uint2 u = *reinterpret_cast<const uint2*>(&gtile[lane * kTokDw + (ch0 + k) * 2]);
          unpack_lo2<PATCH>(u.x, g[k][0], g[k][1]);
          unpack_lo2<PATCH>(u.y, g[k][2], g[k][3]);
        }
      }
      float s1f = 0.f, s2f = 0.f;   // <= 16 terms per thread and scan: f32 partials, f64 across threads
      if (x_ok) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int cl = ch0 + k;
          const int c = c0 + cl;
          if (c < channels) {
            float f[VEC], gw[VEC];
            load_vec<VEC>(&lds[cl * T::LD + cv], f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float xhat = (f[e] - mean) * rstd;
              dw[k][e] += g[k][e] * xhat;
              db[k][e] += g[k][e];
              gw[e] = g[k][e] * w[k][e];
              s1f += gw[e];
              s2f += gw[e] * xhat;
            }
            store_vec<VEC>(&lds[cl * T::LD + cv], gw);
          }
        }
      }
      const double s1 = wave_sum_d((double)s1f);
      const double s2 = wave_sum_d((double)s2f);
      if (lane == 0) { acc[b - bb0][0][wave] = s1; acc[b - bb0][1][wave] = s2; }
      __syncthreads();
      // scatter g*w of the occupied cells back to (pillar, channel) rows
      if (pid >= 0) {
        const int cbeg = grp * T::CH_PER_THREAD;
#pragma unroll
        for (int k = 0; k < T::CH_PER_THREAD; k += 4) {
          const int c = c0 + cbeg + k;
          if (c < channels) {
            const float4 v = make_float4(lds[(cbeg + k + 0) * T::LD + gcell], lds[(cbeg + k + 1) * T::LD + gcell],
                                         lds[(cbeg + k + 2) * T::LD + gcell], lds[(cbeg + k + 3) * T::LD + gcell]);
            *reinterpret_cast<float4*>(grad_feats + (int64_t)pid * channels + c) = v;
          }
        }
      }
      __syncthreads();
    }
    // one pair of f64 atomics per block and scan (the waves' sums met in LDS, ordered by the barriers above)
    if ((int)threadIdx.x < 2 * (bend - bb0)) {
      const int bi = threadIdx.x >> 1, j = threadIdx.x & 1;
      atomicAdd(&sums[(bb0 + bi) * 2 + j], ((acc[bi][j][0] + acc[bi][j][1]) + (acc[bi][j][2] + acc[bi][j][3])) +
                                               ((acc[bi][j][4] + acc[bi][j][5]) + (acc[bi][j][6] + acc[bi][j][7])));
    }
    __syncthreads();
  }
  if (x_ok) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = c0 + ch0 + k;
      if (c < channels) {
        const int64_t o = ((int64_t)c * ny + y) * nx + xv;
        if constexpr (ADAM) {
          float pb[VEC], mw[VEC], vw[VEC], mb[VEC], vb[VEC];
          load_vec<VEC>(ad.bias + o, pb);
          load_vec<VEC>(ad.m_w + o, mw);
          load_vec<VEC>(ad.v_w + o, vw);
          load_vec<VEC>(ad.m_b + o, mb);
          load_vec<VEC>(ad.v_b + o, vb);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            adam_one(w[k][e], dw[k][e], mw[e], vw[e], ad.a);
            adam_one(pb[e], db[k][e], mb[e], vb[e], ad.a);
          }
          store_vec<VEC>(const_cast<float*>(weight) + o, w[k]);
          store_vec<VEC>(ad.bias + o, pb);
          store_vec<VEC>(ad.m_w + o, mw);
          store_vec<VEC>(ad.v_w + o, vw);
          store_vec<VEC>(ad.m_b + o, mb);
          store_vec<VEC>(ad.v_b + o, vb);
          if (ad.sh_w) {
            if constexpr (VEC == 4) {                     // o % 4 == 0: one 8-byte store per parameter run
              const int kd = ad.a.shadow_kind;
              *reinterpret_cast<uint2*>(ad.sh_w + o) =
                  make_uint2((unsigned)shadow_bits(w[k][0], kd) | ((unsigned)shadow_bits(w[k][1], kd) << 16),
                             (unsigned)shadow_bits(w[k][2], kd) | ((unsigned)shadow_bits(w[k][3], kd) << 16));
              *reinterpret_cast<uint2*>(ad.sh_b + o) =
                  make_uint2((unsigned)shadow_bits(pb[0], kd) | ((unsigned)shadow_bits(pb[1], kd) << 16),
                             (unsigned)shadow_bits(pb[2], kd) | ((unsigned)shadow_bits(pb[3], kd) << 16));
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) {
                ad.sh_w[o + e] = shadow_bits(w[k][e], ad.a.shadow_kind);
                ad.sh_b[o + e] = shadow_bits(pb[e], ad.a.shadow_kind);
              }
            }
          }
          continue;
        }
        if (accumulate) {
          float a[VEC], bb[VEC];
          load_vec<VEC>(grad_weight + o, a);
          load_vec<VEC>(grad_bias + o, bb);
#pragma unroll
          for (int e = 0; e < VEC; ++e) { dw[k][e] += a[e]; db[k][e] += bb[e]; }
        }
        store_vec<VEC>(grad_weight + o, dw[k]);
        store_vec<VEC>(grad_bias + o, db[k]);
      }
    }
  }
}

// backward, pass 2: dfeat = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat)) on the (V, C) rows
__global__ void __launch_bounds__(256) k_ln_bwd_rows(const float* __restrict__ feats,
                                                     const int32_t* __restrict__ pillar_batch_start, int batch,
                                                     int channels, int64_t num_pillars,
                                                     const float* __restrict__ stats,
                                                     const double* __restrict__ sums, double inv_count,
                                                     float* __restrict__ grad_feats) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= num_pillars * channels) return;
  const int64_t v = i4 / channels;
  int b = 0;
  while (b + 1 < batch && v >= pillar_batch_start[b + 1]) ++b;
  const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
  const float m1 = (float)(sums[b * 2] * inv_count), m2 = (float)(sums[b * 2 + 1] * inv_count);
  const float4 f = *reinterpret_cast<const float4*>(feats + i4);
  float4 g = *reinterpret_cast<const float4*>(grad_feats + i4);
  g.x = rstd * (g.x - m1 - (f.x - mean) * rstd * m2);
  g.y = rstd * (g.y - m1 - (f.y - mean) * rstd * m2);
  g.z = rstd * (g.z - m1 - (f.z - mean) * rstd * m2);
  g.w = rstd * (g.w - m1 - (f.w - mean) * rstd * m2);
  *reinterpret_cast<float4*>(grad_feats + i4) = g;
}

}  // namespace

extern "C" size_t mbv_scatter_layernorm_workspace_bytes(int32_t batch) {
  if (batch <= 0) return 0;
  return mbv_align_up(sizeof(double) * 2 * (size_t)batch, 256);
}

// patch-token output (bf16 rows of 16*C per 4 x 4 patch): whole patches and whole channel tiles only
extern "C" int mbv_scatter_layernorm_patch_supported(int32_t channels, int32_t ny, int32_t nx, int32_t patch) {
  return patch == 4 && channels > 0 && channels % kCT == 0 && ny > 0 && nx > 0 && ny % 4 == 0 && nx % 4 == 0;
}

extern "C" int mbv_scatter_layernorm_fwd2(const float* feats, const int32_t* pillar_batch_start,
                                          const int32_t* cell_to_pillar, const float* weight, const float* bias,
                                          int32_t batch, int32_t channels, int32_t ny, int32_t nx, float eps,
                                          int32_t patch, int32_t patch_dtype, void* out, float* stats,
                                          void* workspace, size_t workspace_bytes, uint32_t* amax_out, void* stream_,
                                          void* ev_start, void* ev_stop);

extern "C" int mbv_scatter_layernorm_fwd(const float* feats, const int32_t* pillar_batch_start,
                                         const int32_t* cell_to_pillar, const float* weight, const float* bias,
                                         int32_t batch, int32_t channels, int32_t ny, int32_t nx, float eps,
                                         int32_t patch, int32_t patch_dtype, void* out, float* stats,
                                         void* workspace, size_t workspace_bytes, void* stream_, void* ev_start,
                                         void* ev_stop) {
  return mbv_scatter_layernorm_fwd2(feats, pillar_batch_start, cell_to_pillar, weight, bias, batch, channels, ny, nx, eps, patch,
                                    patch_dtype, out, stats, workspace, workspace_bytes, nullptr, stream_, ev_start, ev_stop);
}

// ... with amax_out: an optional absmax record (64 zeroed words) of the f32 (B, C, ny, nx) map (patch == 0), max-combined by one
// atomic per workgroup — the K20 patch projection behind it (fp32 compute) then needs no pass over the 0.5 GB map
extern "C" int mbv_scatter_layernorm_fwd2(const float* feats, const int32_t* pillar_batch_start,
                                          const int32_t* cell_to_pillar, const float* weight, const float* bias,
                                          int32_t batch, int32_t channels, int32_t ny, int32_t nx, float eps,
                                          int32_t patch, int32_t patch_dtype, void* out, float* stats,
                                          void* workspace, size_t workspace_bytes, uint32_t* amax_out, void* stream_,
                                          void* ev_start, void* ev_stop) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || channels <= 0 || ny <= 0 || nx <= 0) return MBV_ERR_BAD_ARG;
  if (channels % 4 != 0) return MBV_ERR_UNSUPPORTED;
  if (patch != 0 && !mbv_scatter_layernorm_patch_supported(channels, ny, nx, patch)) return MBV_ERR_UNSUPPORTED;
  if (patch != 0 && patch_dtype != MBV_DT_BF16 && patch_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  if (!feats || !pillar_batch_start || !cell_to_pillar || !weight || !bias || !out || !stats) return MBV_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mbv_scatter_layernorm_workspace_bytes(batch)) return MBV_ERR_WORKSPACE;
  double* sums = reinterpret_cast<double*>(workspace);
  MBV_CHECK_HIP(mbv_fill_async(sums, 0, sizeof(double) * 2 * batch, stream));
  hipLaunchKernelGGL(k_ln_stats, dim3(256, batch), dim3(256), 0, stream, feats, pillar_batch_start, channels, sums);
  MBV_CHECK_LAUNCH();
  const double inv_count = 1.0 / ((double)channels * ny * nx);
  hipLaunchKernelGGL(k_ln_finalize, dim3((batch + 63) / 64), dim3(64), 0, stream, sums, batch, inv_count, eps, stats);
  MBV_CHECK_LAUNCH();
  const int ctiles = (channels + kCT - 1) / kCT;
  if (ev_start) MBV_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_start), stream));
  if (patch) {
    const int xtiles = (nx + Tile<4>::XT - 1) / Tile<4>::XT;
    if (patch_dtype == MBV_DT_F16)
      hipLaunchKernelGGL((k_ln_apply<4, MBV_DT_F16>), dim3(xtiles * ny, ctiles), dim3(256), 0, stream, feats,
                         cell_to_pillar, weight, bias, stats, batch, channels, ny, nx, xtiles, out, amax_out);
    else
      hipLaunchKernelGGL((k_ln_apply<4, MBV_DT_BF16>), dim3(xtiles * ny, ctiles), dim3(256), 0, stream, feats,
                         cell_to_pillar, weight, bias, stats, batch, channels, ny, nx, xtiles, out, amax_out);
  } else if (nx % 4 == 0) {
    const int xtiles = (nx + Tile<4>::XT - 1) / Tile<4>::XT;
    hipLaunchKernelGGL((k_ln_apply<4, 0>), dim3(xtiles * ny, ctiles), dim3(256), 0, stream, feats, cell_to_pillar,
                       weight, bias, stats, batch, channels, ny, nx, xtiles, out, amax_out);
  } else {
    const int xtiles = (nx + Tile<1>::XT - 1) / Tile<1>::XT;
    hipLaunchKernelGGL((k_ln_apply<1, 0>), dim3(xtiles * ny, ctiles), dim3(256), 0, stream, feats, cell_to_pillar,
                       weight, bias, stats, batch, channels, ny, nx, xtiles, out, amax_out);
  }
  MBV_CHECK_LAUNCH();
  if (ev_stop) MBV_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_stop), stream));
  return MBV_OK;
}

static int ln_bwd_launch(const void* grad_out, int32_t patch, int32_t patch_dtype, const float* feats,
                         const int32_t* pillar_batch_start, const int32_t* cell_to_pillar, const float* weight,
                         const float* stats, int32_t batch, int32_t channels, int32_t ny, int32_t nx, int64_t num_pillars,
                         float* grad_feats, float* grad_weight, float* grad_bias, int32_t accumulate, void* workspace,
                         size_t workspace_bytes, void* stream_, void* ev_start, void* ev_stop, const LnAdam* ad) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || channels <= 0 || ny <= 0 || nx <= 0 || num_pillars < 0) return MBV_ERR_BAD_ARG;
  if (channels % 4 != 0) return MBV_ERR_UNSUPPORTED;
  if (patch != 0 && !mbv_scatter_layernorm_patch_supported(channels, ny, nx, patch)) return MBV_ERR_UNSUPPORTED;
  if (patch != 0 && patch_dtype != MBV_DT_BF16 && patch_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  if (!grad_out || !feats || !pillar_batch_start || !cell_to_pillar || !weight || !stats || !grad_feats) return MBV_ERR_BAD_ARG;
  if (!ad && (!grad_weight || !grad_bias)) return MBV_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mbv_scatter_layernorm_workspace_bytes(batch)) return MBV_ERR_WORKSPACE;
  double* sums = reinterpret_cast<double*>(workspace);
  MBV_CHECK_HIP(mbv_fill_async(sums, 0, sizeof(double) * 2 * batch, stream));
  const int ctiles = (channels + kCT - 1) / kCT;
  if (ev_start) MBV_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_start), stream));
#define MBV_LN_BWD(VEC_, PATCH_)                                                                                          \
  do {                                                                                                                   \
    const int xtiles = (nx + Tile<VEC_>::XT - 1) / Tile<VEC_>::XT;                                                        \
    if (ad)                                                                                                              \
      hipLaunchKernelGGL((k_ln_bwd_dense<VEC_, PATCH_, true>), dim3(xtiles * ny, ctiles), dim3(512), 0, stream, grad_out,  \
                         feats, cell_to_pillar, weight, stats, batch, channels, ny, nx, xtiles, grad_feats, grad_weight,   \
                         grad_bias, accumulate, sums, *ad);                                                              \
    else                                                                                                                 \
      hipLaunchKernelGGL((k_ln_bwd_dense<VEC_, PATCH_, false>), dim3(xtiles * ny, ctiles), dim3(512), 0, stream, grad_out, \
                         feats, cell_to_pillar, weight, stats, batch, channels, ny, nx, xtiles, grad_feats, grad_weight,   \
                         grad_bias, accumulate, sums, LnAdam{});                                                         \
  } while (0)
  if (patch) {
    if (patch_dtype == MBV_DT_F16) MBV_LN_BWD(4, MBV_DT_F16);
    else MBV_LN_BWD(4, MBV_DT_BF16);
  } else if (nx % 4 == 0) {
    MBV_LN_BWD(4, 0);
  } else {
    MBV_LN_BWD(1, 0);
  }
#undef MBV_LN_BWD
  MBV_CHECK_LAUNCH();
  if (ev_stop) MBV_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_stop), stream));
  if (num_pillars > 0) {
    const double inv_count = 1.0 / ((double)channels * ny * nx);
    const int64_t n4 = num_pillars * channels / 4;
    hipLaunchKernelGGL(k_ln_bwd_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, feats,
                       pillar_batch_start, batch, channels, num_pillars, stats, sums, inv_count, grad_feats);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

extern "C" int mbv_scatter_layernorm_bwd(const void* grad_out, int32_t patch, int32_t patch_dtype, const float* feats,
                                         const int32_t* pillar_batch_start, const int32_t* cell_to_pillar,
                                         const float* weight, const float* stats, int32_t batch, int32_t channels,
                                         int32_t ny, int32_t nx, int64_t num_pillars, float* grad_feats, float* grad_weight, float* grad_bias, int32_t accumulate,
                                         void* workspace, size_t workspace_bytes, void* stream_, void* ev_start,
                                         void* ev_stop) {
  return ln_bwd_launch(grad_out, patch, patch_dtype, feats, pillar_batch_start, cell_to_pillar, weight, stats, batch, channels,
                       ny, nx, num_pillars, grad_feats, grad_weight, grad_bias, accumulate, workspace, workspace_bytes, stream_,
                       ev_start, ev_stop, nullptr);
}

extern "C" int mbv_scatter_layernorm_bwd_adamw(const void* grad_out, int32_t patch, int32_t patch_dtype, const float* feats,
                                               const int32_t* pillar_batch_start, const int32_t* cell_to_pillar,
                                               float* weight, float* bias, const float* stats, int32_t batch,
                                               int32_t channels, int32_t ny, int32_t nx, int64_t num_pillars,
                                               float* grad_feats, float* exp_avg_w, float* exp_avg_sq_w, float* exp_avg_b,
                                               float* exp_avg_sq_b, void* shadow_w, void* shadow_b, int32_t shadow_dtype,
                                               float lr, float beta1, float beta2, float eps, float weight_decay,
                                               int64_t step, int32_t decoupled, void* workspace, size_t workspace_bytes,
                                               void* stream_, void* ev_start, void* ev_stop) {
  if (!weight || !bias || !exp_avg_w || !exp_avg_sq_w || !exp_avg_b || !exp_avg_sq_b || step < 1) return MBV_ERR_BAD_ARG;
  if ((shadow_w == nullptr) != (shadow_b == nullptr)) return MBV_ERR_BAD_ARG;
  if (shadow_w && shadow_dtype != MBV_DT_BF16 && shadow_dtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  if ((reinterpret_cast<size_t>(weight) | reinterpret_cast<size_t>(bias) | reinterpret_cast<size_t>(exp_avg_w) |
       reinterpret_cast<size_t>(exp_avg_sq_w) | reinterpret_cast<size_t>(exp_avg_b) | reinterpret_cast<size_t>(exp_avg_sq_b)) & 15)
    return MBV_ERR_BAD_ARG;
  if (shadow_w && ((reinterpret_cast<size_t>(shadow_w) | reinterpret_cast<size_t>(shadow_b)) & 7)) return MBV_ERR_BAD_ARG;
  LnAdam ad;
  ad.bias = bias; ad.m_w = exp_avg_w; ad.v_w = exp_avg_sq_w; ad.m_b = exp_avg_b; ad.v_b = exp_avg_sq_b;
  ad.sh_w = reinterpret_cast<unsigned short*>(shadow_w); ad.sh_b = reinterpret_cast<unsigned short*>(shadow_b);
  AdamArgs& a = ad.a;
  a.shadow_kind = shadow_dtype; a.loss_scale = nullptr; a.skip = nullptr; a.applied = nullptr;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bias_correction1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bias_correction2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.grad_scale = 1.0f; a.decoupled = decoupled; a.zero_grad = 0;
  return ln_bwd_launch(grad_out, patch, patch_dtype, feats, pillar_batch_start, cell_to_pillar, weight, stats, batch, channels,
                       ny, nx, num_pillars, grad_feats, nullptr, nullptr, 0, workspace, workspace_bytes, stream_, ev_start,
                       ev_stop, &ad);
}
