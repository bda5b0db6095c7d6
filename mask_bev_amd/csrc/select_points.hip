// K10 — importance sampling of the mask loss: pick the k most uncertain of n sampled points per mask.
//
// Replaces `torch.topk(point_uncertainties[:, 0, :], k=num_uncertain_points, dim=1)` + the coordinate gather of
// mmdet's get_uncertain_point_coords_with_randomness, called from Mask2FormerHead._loss_by_feat_single
// (mask_bev/models/networks/mask2former_head/mask2former_head.py:401-404): uncertainty = -|logit|, k = 9408 of
// n = 37 632 points for each of the 10 x B x Q masks.  torch implements it with a full segmented sort.
// Here one workgroup owns one row: a 3-pass radix SELECT (11 + 11 + 10 bits of the order-preserving key of
// |logit|, histogram in LDS) finds the k-th smallest key exactly, then one ordered compaction pass writes the
// selected points' (x, y) coordinates in ascending index order (ties at the threshold: lowest indices first).
// The SET of selected points equals top-k's; their order differs, which no consumer depends on (the loss sums).
#include "common.hpp"

namespace {

constexpr int kBins = 2048;
constexpr int kThreads = 512;

__device__ __forceinline__ uint32_t abs_key(float x) { return __float_as_uint(fabsf(x)); }   // monotone for |x|

// block-wide exclusive scan of one int per thread (kThreads threads); returns exclusive prefix, total in *total
__device__ __forceinline__ int block_scan_excl(int v, int* total, int* lds /* >= 8 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < kThreads / 64; ++w) {
    const int s = lds[w];
    if (w < wave) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

__global__ void __launch_bounds__(kThreads) k_select_smallest_abs(const float* __restrict__ x,
                                                                  const float* __restrict__ coords, int n, int k,
                                                                  float* __restrict__ out_coords) {
  __shared__ int hist[kBins];
  __shared__ int scan_lds[8];
  __shared__ uint32_t s_prefix;
  __shared__ int s_krem;
  const int64_t row = blockIdx.x;
  const float* xr = x + row * n;
  uint32_t prefix = 0, prefix_mask = 0;
  int krem = k;                       // still to be taken among the elements matching the prefix
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass], nb = 1 << widths[pass];
    for (int i = threadIdx.x; i < kBins; i += kThreads) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kThreads) {
      const uint32_t key = abs_key(xr[i]);
      if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1);
    }
    __syncthreads();
    // find the bin where the cumulative count reaches krem: each thread owns kBins / kThreads = 4 bins
    constexpr int PER = kBins / kThreads;
    int local[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int b = threadIdx.x * PER + j;
      local[j] = b < nb ? hist[b] : 0;
      sum += local[j];
    }
    int total;
    const int excl = block_scan_excl(sum, &total, scan_lds);
    if (excl < krem && krem <= excl + sum) {          // exactly one thread
      int acc = excl;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (acc < krem && krem <= acc + local[j]) {
          s_prefix = prefix | ((uint32_t)(threadIdx.x * PER + j) << shift);
          s_krem = krem - acc;
        }
        acc += local[j];
      }
    }
    __syncthreads();
    prefix = s_prefix;
    krem = s_krem;
    prefix_mask |= (uint32_t)(nb - 1) << shift;
    __syncthreads();
  }
  // prefix is now the exact k-th smallest key T; take every key < T and the first `krem` keys == T
  const uint32_t T = prefix;
  const float* cr = coords + row * n * 2;
  float* orow = out_coords + row * (int64_t)k * 2;
  int base_lt = 0, base_eq = 0;       // running counts before the current chunk (in index order)
  for (int c0 = 0; c0 < n; c0 += kThreads) {
    const int i = c0 + threadIdx.x;
    uint32_t key = 0xffffffffu;
    if (i < n) key = abs_key(xr[i]);
    const int is_lt = (i < n && key < T) ? 1 : 0, is_eq = (i < n && key == T) ? 1 : 0;
    int tot_lt, tot_eq;
    const int ex_lt = block_scan_excl(is_lt, &tot_lt, scan_lds);
    const int ex_eq = block_scan_excl(is_eq, &tot_eq, scan_lds);
    // output position = selected elements before i = (all < T before i) + min(krem, == T before i)
    if (is_lt || (is_eq && base_eq + ex_eq < krem)) {
      const int eq_before = min(krem, base_eq + ex_eq);
      const int pos = base_lt + ex_lt + eq_before;
      const float2 xy = *reinterpret_cast<const float2*>(cr + (int64_t)i * 2);
      *reinterpret_cast<float2*>(orow + (int64_t)pos * 2) = xy;
    }
    base_lt += tot_lt;
    base_eq += tot_eq;
  }
}

}  // namespace

extern "C" int mbv_select_uncertain_points(const float* logits, const float* coords, int64_t rows, int32_t n,
                                           int32_t k, float* out_coords, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (rows < 0 || n <= 0 || k <= 0 || k > n) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!logits || !coords || !out_coords) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_select_smallest_abs, dim3((unsigned)rows), dim3(kThreads), 0, stream, logits, coords, n, k,
                     out_coords);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
