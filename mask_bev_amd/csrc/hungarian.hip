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

__device__ __forceinline__ double readlane_f64(double x, int src_lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)b, src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// rows <= cols.  cost: (rows, cols) row-major f32.  row_to_col: (rows) i32.
//
// State lives in registers of the lane that owns a column (columns lane + 1 and lane + 65, 1-based): its dual v,
// the row matched to it (p), THAT ROW's dual u (a row is matched to at most one column, so its u can travel with
// the column), the search's minv / way / used.  The search column j0 is wave-uniform, so everything the classic
// formulation reads through arrays (p[j0], u[p[j0]], way[j]) is one v_readlane from the owner lane; LDS holds
// only the read-only cost matrix and there is no barrier inside the search.
//
// Start: the duals of the textbook row / column reduction (u_i = min_j a_ij, v_j = min_i (a_ij - u_i)) and the
// greedy matching on their zero reduced costs — a feasible dual with complementary slackness, so the
// shortest-augmenting-path search only has to run for the rows the greedy pass left unmatched and still ends in
// an optimal assignment (the same one as scipy's when the optimum is unique).
//
// Padded mode (`real_cols` != NULL, one entry per problem): the caller's matrix is SQUARE, (rows = predictions) x (cols
// = ground-truth slots), and its columns real_cols[p] .. cols - 1 are IDENTICAL (the dataset pads the instance list
// with all-zero masks of label 0, so those columns hold the same cost for a given prediction).  Assigning prediction i
// to any of them costs d_i = a[i][cols - 1], so the square problem equals  sum_i d_i + min over injections of the K =
// real_cols[p] real columns into the predictions of  sum_j (a[s(j)][j] - d_s(j)):  a RECTANGULAR problem with K rows
// (the real instances) and `rows` columns (the predictions) — K augmenting paths instead of `rows`, each at most K
// long.  It is solved on the original f32 costs in f64: the opt-out cost d enters as the START VALUE of the
// predictions' duals (v_q = d_q instead of 0: every reduced cost a - u - v is then the reduced cost of the transformed
// matrix), which is what a free column's dual has to be at the optimum.  The predictions left over take the padded
// columns in ascending order (identical targets: any order gives the same loss).  Same optimum as the square solve —
// the same real pairs whenever that optimum is unique — at ≈ (K / rows)^2 of its search steps.
__global__ void __launch_bounds__(64) k_hungarian(const float* __restrict__ cost_all, int rows, int cols,
                                                  int transposed, int32_t* __restrict__ out_all, int out_len,
                                                  const int32_t* __restrict__ real_cols) {
  __shared__ double u_row[kMaxDim];
  __shared__ double d_col[kMaxDim];
  __shared__ int row_matched[kMaxDim];
  __shared__ float a_lds[kMaxDim * (kMaxDim + 1)];   // the whole cost matrix: every search step reads one row
  const int lane = threadIdx.x;
  const float* cost = cost_all + (int64_t)blockIdx.x * rows * cols;
  int32_t* out = out_all + (int64_t)blockIdx.x * out_len;
  constexpr int LD = kMaxDim + 1;
  const int n_pred = rows, n_gt = cols;          // the caller's shape (padded mode)
  bool padded = false;
  if (real_cols) {
    int k = real_cols[blockIdx.x];
    k = k < 0 ? 0 : (k > n_gt ? n_gt : k);
    // square problems only: every column — hence every REAL column — is matched there, which is what the rectangular
    // form assumes (with more slots than predictions a real column may stay unmatched: solved in the plain form)
    if (k < n_gt && n_gt == n_pred) {
      padded = true;
      // internal rows = real instances, internal columns = predictions: a_lds[g][q] = a[q][g]
      for (int e = lane; e < n_pred * k; e += 64) {
        const int q = e / k, gidx = e - q * k;
        a_lds[gidx * LD + q] = cost[q * n_gt + gidx];
      }
      for (int q = lane; q < n_pred; q += 64) d_col[q] = (double)cost[q * n_gt + n_gt - 1];
      rows = k;
      cols = n_pred;
      transposed = 1;
    }
  }
  if (!padded) {
    for (int e = lane; e < rows * cols; e += 64) {
      int i, j;
      if (transposed) { j = e / rows; i = e - j * rows; } else { i = e / cols; j = e - i * cols; }
      a_lds[i * LD + j] = cost[e];
    }
    for (int q = lane; q < kMaxDim; q += 64) d_col[q] = 0.0;
  }
  __syncthreads();
  const int jA = lane + 1, jB = lane + 65;
  const bool hasA = jA <= cols, hasB = jB <= cols;
  const double INF = 1e300;
  // -- row reduction: lane r owns rows r and r + 64 (row stride LD = 129 words: conflict-free)
  for (int r = lane; r < rows; r += 64) {
    double m = (double)a_lds[r * LD] - d_col[0];
    for (int j = 1; j < cols; ++j) m = fmin(m, (double)a_lds[r * LD + j] - d_col[j]);
    u_row[r] = m;
    row_matched[r] = 0;
  }
  __syncthreads();
  // -- column reduction.  Only for square problems: with rows < cols a column may stay unmatched, and the dual of
  // an unmatched column has to be 0 at the optimum (v = 0 is the start the search itself keeps for free columns;
  // d in padded mode).
  double vA = hasA ? d_col[jA - 1] : 0.0, vB = hasB ? d_col[jB - 1] : 0.0;
  if (rows == cols && !padded) {
    vA = vB = INF;
    for (int i = 0; i < rows; ++i) {
      const double ui = u_row[i];
      if (hasA) vA = fmin(vA, (double)a_lds[i * LD + jA - 1] - ui);
      if (hasB) vB = fmin(vB, (double)a_lds[i * LD + jB - 1] - ui);
    }
  }
  // -- greedy matching on zero reduced costs (lowest free column first)
  int pA = 0, pB = 0;                 // row matched to my columns (1-based, 0 = free)
  double uA = 0.0, uB = 0.0;          // dual of that row
  for (int i = 0; i < rows; ++i) {
    const double ui = u_row[i];
    unsigned cand = 0xffffffffu;
    if (hasA && pA == 0 && ((double)a_lds[i * LD + jA - 1] - ui) - vA == 0.0) cand = (unsigned)jA;
    else if (hasB && pB == 0 && ((double)a_lds[i * LD + jB - 1] - ui) - vB == 0.0) cand = (unsigned)jB;
    const unsigned j = wave_min_u32(cand);
    if (j != 0xffffffffu) {
      if ((int)j == jA) { pA = i + 1; uA = ui; }
      if ((int)j == jB) { pB = i + 1; uB = ui; }
      if (lane == 0) row_matched[i] = 1;
    }
  }
  __syncthreads();
  // -- shortest augmenting paths for the remaining rows
  for (int i = 1; i <= rows; ++i) {
    if (row_matched[i - 1]) continue;
    double ui = u_row[i - 1];          // dual of the root row (wave-uniform)
    double minA = INF, minB = INF;
    bool usedA = false, usedB = false;
    int wayA = 0, wayB = 0;
    int j0 = 0;
    while (true) {
      int i0;
      double ui0;
      if (j0 == 0) {
        i0 = i;
        ui0 = ui;
      } else {
        const int owner = (j0 - 1) & 63;
        const bool second = j0 > 64;
        if (j0 == jA) usedA = true;
        if (j0 == jB) usedB = true;
        i0 = __builtin_amdgcn_readlane(second ? pB : pA, owner);
        ui0 = readlane_f64(second ? uB : uA, owner);
      }
      const float* arow = a_lds + (i0 - 1) * LD;
      double best = INF;
      int bestj = 0x7fffffff;
      if (hasA && !usedA) {
        const double cur = (double)arow[jA - 1] - ui0 - vA;
        if (cur < minA) { minA = cur; wayA = j0; }
        if (minA < best) { best = minA; bestj = jA; }
      }
      if (hasB && !usedB) {
        const double cur = (double)arow[jB - 1] - ui0 - vB;
        if (cur < minB) { minB = cur; wayB = j0; }
        if (minB < best || (minB == best && pA != 0 && pB == 0)) { best = minB; bestj = jB; }
      }
      // wave arg-min: 64-bit key = (orderable reduced cost with its 9 lowest mantissa bits cleared | matched? |
      // column), minimised in two 32-bit DPP passes (high word, then low word among the lanes holding the
      // minimum).  Among equal reduced costs a FREE column wins: the path ends there instead of wandering through
      // every matched column of a tie — with the dataset's ≈ 70 identical zero-padded GT columns that alone
      // is a 7x shorter search (2.2 ms → 0.3 ms per launch).
      const unsigned long long taken = bestj == jA ? (pA != 0) : (pB != 0);
      const unsigned long long key =
          bestj == 0x7fffffff ? ~0ull : ((orderable(best) & ~0x1ffull) | (taken << 8) | (unsigned long long)bestj);
      const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
      const unsigned mhi = wave_min_u32(hi);
      const unsigned mlo = wave_min_u32(hi == mhi ? lo : 0xffffffffu);
      const int j1 = (int)(mlo & 0xffu);          // 1..128 (bit 8 is the matched flag)
      const int owner1 = (j1 - 1) & 63;
      const bool second1 = j1 > 64;
      // exact delta = minv[j1] from the lane that owns column j1
      const double delta = readlane_f64(second1 ? minB : minA, owner1);
      // potentials: the rows of the tree (root + rows matched to used columns) go up, used columns go down,
      // free columns tighten
      ui += delta;
      if (hasA) { if (usedA) { uA += delta; vA -= delta; } else minA -= delta; }
      if (hasB) { if (usedB) { uB += delta; vB -= delta; } else minB -= delta; }
      j0 = j1;
      if (__builtin_amdgcn_readlane(second1 ? pB : pA, owner1) == 0) break;
    }
    // augment along the alternating path: p[j] = p[way[j]] (and that row's dual with it) back to the root
    int j = j0;
    while (j != 0) {
      const int owner = (j - 1) & 63;
      const bool second = j > 64;
      const int jp = __builtin_amdgcn_readlane(second ? wayB : wayA, owner);
      int row;
      double urow;
      if (jp == 0) {
        row = i;
        urow = ui;
      } else {
        const int ownp = (jp - 1) & 63;
        const bool secp = jp > 64;
        row = __builtin_amdgcn_readlane(secp ? pB : pA, ownp);
        urow = readlane_f64(secp ? uB : uA, ownp);
      }
      if (j == jA) { pA = row; uA = urow; }
      if (j == jB) { pB = row; uB = urow; }
      j = jp;
    }
  }
  // outputs
  if (padded) {
    // out[prediction] = its real instance, or — for the predictions left over — the padded columns in ascending order
    const bool freeA = hasA && pA == 0, freeB = hasB && pB == 0;
    const unsigned long long mA = __ballot(freeA), mB = __ballot(freeB);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (hasA) out[jA - 1] = pA > 0 ? pA - 1 : rows + __popcll(mA & below);
    if (hasB) out[jB - 1] = pB > 0 ? pB - 1 : rows + __popcll(mA) + __popcll(mB & below);
  } else if (!transposed) {
    if (hasA && pA > 0) out[pA - 1] = jA - 1;       // out[row] = col
    if (hasB && pB > 0) out[pB - 1] = jB - 1;
  } else {
    // the caller's rows are this problem's columns: out[caller_row] = caller_col or -1
    if (hasA) out[jA - 1] = pA > 0 ? pA - 1 : -1;
    if (hasB) out[jB - 1] = pB > 0 ? pB - 1 : -1;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Wide problems (128 < columns <= 320: the 200-query KITTI and 300-query Waymo configurations).  Same algorithm and
// register-resident state with CPL columns per lane; the cost matrix (up to 360 KB) does not fit LDS and is read
// from global memory / L2, one coalesced row per search step — latency-bound, a supported path rather than a fast
// one (the 100-query configuration never takes it).  rows <= cols, row-major (rows, cols) input.
template <int CPL>
__device__ __forceinline__ int pick_i(const int (&x)[CPL], int c) {
  int r = x[0];
#pragma unroll
  for (int k = 1; k < CPL; ++k) r = c == k ? x[k] : r;
  return r;
}
template <int CPL>
__device__ __forceinline__ double pick_d(const double (&x)[CPL], int c) {
  double r = x[0];
#pragma unroll
  for (int k = 1; k < CPL; ++k) r = c == k ? x[k] : r;
  return r;
}

//
// Padded mode (round 6; `real_cols` != NULL, square caller matrix): the rectangular problem of the K real ground-truth columns
// against the predictions, exactly as in k_hungarian — K augmenting paths of length <= K instead of `rows` paths through ties
// (the 200 / 300-query configurations carry 5-40 real instances: 5.6 ms -> 0.2 ms per launch at 200 queries).  Its K x rows
// matrix (the transposed real columns) is staged in dynamic LDS (`lds_rows` rows of it fit); a problem with more real columns
// than that is solved in the plain form from global memory.
template <int CPL>
__global__ void __launch_bounds__(64) k_hungarian_wide(const float* __restrict__ cost_all, int rows, int cols,
                                                       int transposed_out, int32_t* __restrict__ out_all, int out_len,
                                                       const int32_t* __restrict__ real_cols, int lds_rows) {
  constexpr int MAXD = 64 * CPL;
  extern __shared__ float a_pad[];        // padded mode: (K, predictions) f32, row stride `cols`
  __shared__ double u_row[MAXD];
  __shared__ int row_matched[MAXD];
  const int lane = threadIdx.x;
  const float* cost = cost_all + (int64_t)blockIdx.x * rows * cols;
  int32_t* out = out_all + (int64_t)blockIdx.x * out_len;
  const double INF = 1e300;
  const int n_pred = rows, n_gt = cols;
  bool padded = false;
  if (real_cols) {
    int k = real_cols[blockIdx.x];
    k = k < 0 ? 0 : (k > n_gt ? n_gt : k);
    if (k < n_gt && n_gt == n_pred && k <= lds_rows) {
      padded = true;
      // a_pad[g][q] = cost[q][g]: consecutive lanes read the consecutive real columns of a prediction's row
#pragma unroll 8
      for (int e = lane; e < n_pred * k; e += 64) {
        const int q = e / k, g = e - q * k;
        a_pad[g * n_pred + q] = cost[(int64_t)q * n_gt + g];
      }
      rows = k;
      cols = n_pred;
      transposed_out = 1;
    }
  }
  const float* amat = padded ? a_pad : cost;
  int jc[CPL];
  bool has[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) { jc[c] = lane + 1 + 64 * c; has[c] = jc[c] <= cols; }
  double v[CPL], uu[CPL], minv[CPL];
  int p[CPL], way[CPL];
  bool used[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {         // padded: the opt-out cost of prediction q is the start value of its dual
    v[c] = (padded && has[c]) ? (double)cost[(int64_t)(jc[c] - 1) * n_gt + n_gt - 1] : 0.0;
    uu[c] = 0.0;
    p[c] = 0;
  }
  __syncthreads();                        // a_pad staged
  // row reduction (coalesced: the lanes of the wave read one row)
  for (int i = 0; i < rows; ++i) {
    double m = INF;
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      if (has[c]) m = fmin(m, (double)amat[(int64_t)i * cols + jc[c] - 1] - v[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
    if (lane == 0) { u_row[i] = m; row_matched[i] = 0; }
  }
  __syncthreads();
  if (rows == cols && !padded) {          // column reduction (square problems only, see k_hungarian)
#pragma unroll
    for (int c = 0; c < CPL; ++c) v[c] = INF;
    for (int i = 0; i < rows; ++i) {
      const double ui = u_row[i];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        if (has[c]) v[c] = fmin(v[c], (double)amat[(int64_t)i * cols + jc[c] - 1] - ui);
    }
  }
  for (int i = 0; i < rows; ++i) {        // greedy matching on zero reduced costs
    const double ui = u_row[i];
    unsigned cand = 0xffffffffu;
#pragma unroll
    for (int c = CPL - 1; c >= 0; --c)
      if (has[c] && p[c] == 0 && ((double)amat[(int64_t)i * cols + jc[c] - 1] - ui) - v[c] == 0.0) cand = (unsigned)jc[c];
    const unsigned j = wave_min_u32(cand);
    if (j != 0xffffffffu) {
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        if ((int)j == jc[c]) { p[c] = i + 1; uu[c] = ui; }
      if (lane == 0) row_matched[i] = 1;
    }
  }
  __syncthreads();
  for (int i = 1; i <= rows; ++i) {
    if (row_matched[i - 1]) continue;
    double ui = u_row[i - 1];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { minv[c] = INF; used[c] = false; way[c] = 0; }
    int j0 = 0;
    while (true) {
      int i0;
      double ui0;
      if (j0 == 0) {
        i0 = i;
        ui0 = ui;
      } else {
        const int owner = (j0 - 1) & 63, cc = (j0 - 1) >> 6;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          if (j0 == jc[c]) used[c] = true;
        i0 = __builtin_amdgcn_readlane(pick_i<CPL>(p, cc), owner);
        ui0 = readlane_f64(pick_d<CPL>(uu, cc), owner);
      }
      const float* arow = amat + (int64_t)(i0 - 1) * cols;
      double best = INF;
      int bestj = 0x7fffffff;
      bool best_taken = true;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if (has[c] && !used[c]) {
          const double cur = (double)arow[jc[c] - 1] - ui0 - v[c];
          if (cur < minv[c]) { minv[c] = cur; way[c] = j0; }
          const bool taken = p[c] != 0;
          if (minv[c] < best || (minv[c] == best && best_taken && !taken)) { best = minv[c]; bestj = jc[c]; best_taken = taken; }
        }
      }
      // key = orderable reduced cost (10 low mantissa bits cleared) | matched? | column (9 bits); free columns win ties
      const unsigned long long key =
          bestj == 0x7fffffff ? ~0ull
                              : ((orderable(best) & ~0x3ffull) | ((unsigned long long)(best_taken ? 1 : 0) << 9) |
                                 (unsigned long long)bestj);
      const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
      const unsigned mhi = wave_min_u32(hi);
      const unsigned mlo = wave_min_u32(hi == mhi ? lo : 0xffffffffu);
      const int j1 = (int)(mlo & 0x1ffu);
      const int owner1 = (j1 - 1) & 63, c1 = (j1 - 1) >> 6;
      const double delta = readlane_f64(pick_d<CPL>(minv, c1), owner1);
      ui += delta;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if (has[c]) { if (used[c]) { uu[c] += delta; v[c] -= delta; } else minv[c] -= delta; }
      }
      j0 = j1;
      if (__builtin_amdgcn_readlane(pick_i<CPL>(p, c1), owner1) == 0) break;
    }
    int j = j0;
    while (j != 0) {                      // augment
      const int owner = (j - 1) & 63, cc = (j - 1) >> 6;
      const int jp = __builtin_amdgcn_readlane(pick_i<CPL>(way, cc), owner);
      int row;
      double urow;
      if (jp == 0) {
        row = i;
        urow = ui;
      } else {
        const int ownp = (jp - 1) & 63, cp = (jp - 1) >> 6;
        row = __builtin_amdgcn_readlane(pick_i<CPL>(p, cp), ownp);
        urow = readlane_f64(pick_d<CPL>(uu, cp), ownp);
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        if (j == jc[c]) { p[c] = row; uu[c] = urow; }
      j = jp;
    }
  }
  if (padded) {
    // out[prediction] = its real instance, or — for the predictions left over — the padded columns in ascending order
    int before = 0;                       // free predictions in the column sets below c
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const bool fr = has[c] && p[c] == 0;
      const unsigned long long m = __ballot(fr);
      if (has[c]) out[jc[c] - 1] = p[c] > 0 ? p[c] - 1 : rows + before + __popcll(m & ((1ull << lane) - 1ull));
      before += __popcll(m);
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    if (!has[c]) continue;
    if (!transposed_out) {
      if (p[c] > 0) out[p[c] - 1] = jc[c] - 1;       // out[row] = col
    } else {
      out[jc[c] - 1] = p[c] > 0 ? p[c] - 1 : -1;     // the caller's rows are this problem's columns
    }
  }
}

}  // namespace

extern "C" int mbv_hungarian(const float* cost, int32_t batch, int32_t num_rows, int32_t num_cols,
                             int32_t* row_to_col, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch < 0 || num_rows <= 0 || num_cols <= 0) return MBV_ERR_BAD_ARG;
  if (num_rows > 320 || num_cols > 320) return MBV_ERR_UNSUPPORTED;
  if (batch == 0) return MBV_OK;
  if (!cost || !row_to_col) return MBV_ERR_BAD_ARG;
  if (num_rows <= kMaxDim && num_cols <= kMaxDim) {
    if (num_rows <= num_cols) {
      hipLaunchKernelGGL(k_hungarian, dim3(batch), dim3(64), 0, stream, cost, num_rows, num_cols, 0, row_to_col,
                         num_rows, (const int32_t*)nullptr);
    } else {
      // more rows than columns: solve the transposed problem; unmatched rows get -1
      hipLaunchKernelGGL(k_hungarian, dim3(batch), dim3(64), 0, stream, cost, num_cols, num_rows, 1, row_to_col,
                         num_rows, (const int32_t*)nullptr);
    }
  } else {
    // wide problems read the matrix from global memory and need it in (rows <= cols) orientation:
    // mbv_hungarian_wide_t takes the materialised transpose for num_rows > num_cols
    if (num_rows > num_cols) return MBV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_hungarian_wide<5>, dim3(batch), dim3(64), 0, stream, cost, num_rows, num_cols, 0, row_to_col,
                       num_rows, (const int32_t*)nullptr, 0);
  }
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

// Square-or-wider problems whose trailing columns are identical padding (see k_hungarian): real_cols (batch) i32 on the
// device = number of leading real columns of each problem.  num_rows <= num_cols <= 320 (above 128 columns: the wide kernel).
extern "C" int mbv_hungarian_padded(const float* cost, int32_t batch, int32_t num_rows, int32_t num_cols,
                                    const int32_t* real_cols, int32_t* row_to_col, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch < 0 || num_rows <= 0 || num_cols <= 0 || num_rows > num_cols) return MBV_ERR_BAD_ARG;
  if (num_cols > 320) return MBV_ERR_UNSUPPORTED;
  if (batch == 0) return MBV_OK;
  if (!cost || !row_to_col || !real_cols) return MBV_ERR_BAD_ARG;
  if (num_cols > kMaxDim) {
    // wide problems: the K real columns' transposed block in dynamic LDS (as many rows of it as 128 KB hold)
    constexpr int kPadBytes = 128 * 1024;
    static bool attr_done = false;       // idempotent attribute of the code object, not library state
    if (!attr_done) {
      MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hungarian_wide<5>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kPadBytes));
      attr_done = true;
    }
    int lds_rows = kPadBytes / (4 * num_rows);
    lds_rows = lds_rows > num_cols ? num_cols : lds_rows;
    hipLaunchKernelGGL(k_hungarian_wide<5>, dim3(batch), dim3(64), (size_t)lds_rows * num_rows * 4, stream, cost, num_rows,
                       num_cols, 0, row_to_col, num_rows, real_cols, num_rows == num_cols ? lds_rows : 0);
    MBV_CHECK_LAUNCH();
    return MBV_OK;
  }
  hipLaunchKernelGGL(k_hungarian, dim3(batch), dim3(64), 0, stream, cost, num_rows, num_cols, 0, row_to_col, num_rows,
                     real_cols);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_hungarian_wide_t(const float* cost_t, int32_t batch, int32_t num_rows, int32_t num_cols,
                                    int32_t* row_to_col, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch < 0 || num_rows <= 0 || num_cols <= 0 || num_rows <= num_cols) return MBV_ERR_BAD_ARG;
  if (num_rows > 320) return MBV_ERR_UNSUPPORTED;
  if (batch == 0) return MBV_OK;
  if (!cost_t || !row_to_col) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_hungarian_wide<5>, dim3(batch), dim3(64), 0, stream, cost_t, num_cols, num_rows, 1, row_to_col,
                     num_rows, (const int32_t*)nullptr, 0);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
