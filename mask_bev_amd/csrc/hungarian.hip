// K9 — batched linear-sum assignment (Hungarian / Kuhn-Munkres with potentials) on the device.
//
// Replaces the host round trip of mmdet's HungarianAssigner → scipy.optimize.linear_sum_assignment, reached
// from Mask2FormerHead._get_targets_single (mask_bev/models/networks/mask2former_head/mask2former_head.py:
// 207-210) once per image per decoder output (10 x B device→host syncs per training step).
// One wavefront solves one (rows x cols) cost matrix with the O(n^3) shortest-augmenting-path algorithm; the
// per-column work of every step (reduced-cost update, arg-min, potential update) is spread over the 64 lanes
// (two columns per lane for up to 128 columns).  All matrices of a step (decoder outputs x images) are solved
// by one launch.  Potentials are kept in f64 like scipy's solver.  The optimum it returns is the same
// assignment as scipy's whenever the optimum is unique; among exactly tied optima (e.g. identical zero-padded
// ground-truth columns) any of them gives identical targets and loss.
#include "common.hpp"

namespace {

constexpr int kMaxDim = 128;

// wave64 unsigned-min on the DPP network (no LDS round trips): 4 row shifts, then the two cross-row
// broadcasts; the total lands in lane 63 and is read back through a scalar register.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define MBV_DPP_MIN(ctrl, rmask)                                                                  \
  {                                                                                               \
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); \
    v = t < v ? t : v;                                                                            \
  }
  MBV_DPP_MIN(0x111, 0xf)   // row_shr:1
  MBV_DPP_MIN(0x112, 0xf)   // row_shr:2
  MBV_DPP_MIN(0x114, 0xf)   // row_shr:4
  MBV_DPP_MIN(0x118, 0xf)   // row_shr:8
  MBV_DPP_MIN(0x142, 0xa)   // row_bcast:15
  MBV_DPP_MIN(0x143, 0xc)   // row_bcast:31
#undef MBV_DPP_MIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// order-preserving map double -> u64 (total order of finite values)
__device__ __forceinline__ unsigned long long orderable(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// rows <= cols.  cost: (rows, cols) row-major f32.  row_to_col: (rows) i32.
__global__ void __launch_bounds__(64) k_hungarian(const float* __restrict__ cost_all, int rows, int cols,
                                                  int transposed, int32_t* __restrict__ out_all, int out_len) {
  __shared__ double u[kMaxDim + 1];
  __shared__ int p[kMaxDim + 1];      // p[j] = row matched to column j (1-based), 0 = free
  __shared__ int way[kMaxDim + 1];
  __shared__ float a_lds[kMaxDim * (kMaxDim + 1)];   // the whole cost matrix: every search step reads one row
  const int lane = threadIdx.x;
  const float* cost = cost_all + (int64_t)blockIdx.x * rows * cols;
  int32_t* out = out_all + (int64_t)blockIdx.x * out_len;
  // element (i, j) of the rows<=cols problem; `transposed` means the caller's matrix is (cols x rows)
  constexpr int LD = kMaxDim + 1;
  for (int e = lane; e < rows * cols; e += 64) {
    int i, j;
    if (transposed) { j = e / rows; i = e - j * rows; } else { i = e / cols; j = e - i * cols; }
    a_lds[i * LD + j] = cost[e];
  }
  auto a = [&](int i, int j) -> double { return (double)a_lds[i * LD + j]; };
  // this lane owns columns j = lane + 1 and lane + 65 (1-based)
  const int jA = lane + 1, jB = lane + 65;
  const bool hasA = jA <= cols, hasB = jB <= cols;
  double vA = 0.0, vB = 0.0;
  for (int i = lane; i <= kMaxDim; i += 64) {
    u[i] = 0.0;
    p[i] = 0;
    way[i] = 0;
  }
  __syncthreads();
  const double INF = 1e300;
  for (int i = 1; i <= rows; ++i) {
    if (lane == 0) p[0] = i;
    double minA = INF, minB = INF;
    bool usedA = false, usedB = false;
    int j0 = 0;
    __syncthreads();
    while (true) {
      // mark j0 used
      if (j0 == jA) usedA = true;
      if (j0 == jB) usedB = true;
      const int i0 = p[j0];
      const double ui0 = u[i0];
      double best = INF;
      int bestj = 0x7fffffff;
      if (hasA && !usedA) {
        const double cur = a(i0 - 1, jA - 1) - ui0 - vA;
        if (cur < minA) { minA = cur; way[jA] = j0; }
        if (minA < best) { best = minA; bestj = jA; }
      }
      if (hasB && !usedB) {
        const double cur = a(i0 - 1, jB - 1) - ui0 - vB;
        if (cur < minB) { minB = cur; way[jB] = j0; }
        if (minB < best) { best = minB; bestj = jB; }
      }
      // wave arg-min: 64-bit key = (orderable reduced cost with its 8 lowest mantissa bits cleared | column),
      // minimised in two 32-bit DPP passes (high word, then low word among the lanes holding the minimum)
      const unsigned long long key =
          bestj == 0x7fffffff ? ~0ull : ((orderable(best) & ~0xffull) | (unsigned long long)bestj);
      const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
      const unsigned mhi = wave_min_u32(hi);
      const unsigned mlo = wave_min_u32(hi == mhi ? lo : 0xffffffffu);
      const int j1 = (int)(mlo & 0xffu);
      // exact delta = minv[j1], read from the lane that owns column j1
      const int owner = (j1 - 1) & 63;
      const double cand = (j1 == jA) ? minA : minB;
      const double delta = __shfl(cand, owner, 64);
      // potentials: used columns (incl. the virtual column 0) move their rows; free columns tighten
      if (lane == 0) u[p[0]] += delta;
      if (hasA) { if (usedA) { u[p[jA]] += delta; vA -= delta; } else minA -= delta; }
      if (hasB) { if (usedB) { u[p[jB]] += delta; vB -= delta; } else minB -= delta; }
      j0 = j1;
      __syncthreads();
      if (p[j0] == 0) break;
    }
    // augment along the alternating path (sequential, <= rows steps)
    if (lane == 0) {
      int j = j0;
      while (j != 0) {
        const int jp = way[j];
        p[j] = p[jp];
        j = jp;
      }
    }
    __syncthreads();
  }
  // outputs
  if (!transposed) {
    // out[row] = col
    for (int j = lane + 1; j <= cols; j += 64)
      if (p[j] > 0) out[p[j] - 1] = j - 1;
  } else {
    // the caller's rows are this problem's columns: out[caller_row] = caller_col or -1
    for (int j = lane + 1; j <= cols; j += 64) out[j - 1] = p[j] > 0 ? p[j] - 1 : -1;
  }
}

}  // namespace

extern "C" int mbv_hungarian(const float* cost, int32_t batch, int32_t num_rows, int32_t num_cols,
                             int32_t* row_to_col, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch < 0 || num_rows <= 0 || num_cols <= 0) return MBV_ERR_BAD_ARG;
  if (num_rows > kMaxDim || num_cols > kMaxDim) return MBV_ERR_UNSUPPORTED;
  if (batch == 0) return MBV_OK;
  if (!cost || !row_to_col) return MBV_ERR_BAD_ARG;
  if (num_rows <= num_cols) {
    hipLaunchKernelGGL(k_hungarian, dim3(batch), dim3(64), 0, stream, cost, num_rows, num_cols, 0, row_to_col,
                       num_rows);
  } else {
    // more rows than columns: solve the transposed problem; unmatched rows get -1
    hipLaunchKernelGGL(k_hungarian, dim3(batch), dim3(64), 0, stream, cost, num_cols, num_rows, 1, row_to_col,
                       num_rows);
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
