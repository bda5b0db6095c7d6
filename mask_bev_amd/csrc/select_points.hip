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
#include "bilinear.hpp"

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

// ---------------------------------------------------------------------------------------------------------
// Fused form: sample the n candidate points of a row from its (H, W) logit map AND select the k most uncertain, in
// one workgroup per row — the (rows, n) over-sampled logits (600 MB per step at 10 x B x Q rows, n = 37 632)
// never exist in HBM.  The map sits in a 64 KB LDS tile (K8's layout), every thread keeps the |logit| keys of its
// n / 1024 points in registers, a 3-pass radix select (11 + 10 + 10 bits) histograms them from there, and the
// compaction (same order and tie rule as above) writes each selected candidate's INDEX to its output position in
// LDS from per-(chunk, wave) ballot counts; a dense last pass turns the k indices into coordinates.  The uniform
// tail of the reference's sampling (rand_coords) is copied behind the selected points, which removes the torch.cat
// of the two coordinate sets as well.  The kernel is VALU-bound (5 400 VALU issue slots per thread, 55 % of them the
// generator + bilinear set-up of the sampling loop); 64 VGPRs so that two workgroups share a CU.
constexpr int kFusedThreads = 1024;
constexpr int kKeysPerThread = 40;                       // n <= 40 960
constexpr int kFusedWaves = kFusedThreads / 64;

// Counter-based uniform points for the training path (PCG output hash of a per-row offset + point counter): with
// RNG = true the candidate coordinates are generated where they are consumed — once for sampling, once more in
// the compaction — instead of being drawn by torch.rand into a 1.2 GB tensor, read, and read again.  The same
// generator filled into a tensor by mbv_uniform_points gives the tests an exact two-kernel reference.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  const uint32_t s = v * 747796405u + 2891336453u;
  const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
__device__ __forceinline__ uint32_t row_stream(int64_t seed, int64_t row) {
  return pcg_hash((uint32_t)seed ^ pcg_hash((uint32_t)(seed >> 32) + (uint32_t)row * 0x9E3779B9u + (uint32_t)(row >> 32)));
}
__device__ __forceinline__ float2 uniform_point(uint32_t stream, int p) {
  const uint32_t a = pcg_hash(stream + 2u * (uint32_t)p), b = pcg_hash(stream + 2u * (uint32_t)p + 1u);
  return make_float2((float)(a >> 8) * 5.9604644775390625e-08f, (float)(b >> 8) * 5.9604644775390625e-08f);   // [0, 1)
}

// (round 6: four pairs of points per thread, 16-byte stores — one point per thread was 1.76 M workgroups of 2 KB each for the
// 12 000 x 37 632 candidates of the 300-query configuration: 6.7 ms for 3.6 GB)
__global__ void __launch_bounds__(256) k_uniform_points(const int64_t* __restrict__ seed, int64_t rows, int n,
                                                        float* __restrict__ out) {
  const int64_t row = blockIdx.y;
  const uint32_t stream = row_stream(seed[0], row);
  float* orow = out + row * (int64_t)n * 2;
  const bool vec = ((reinterpret_cast<size_t>(orow) & 15) == 0);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = ((blockIdx.x * 4 + u) * (int)blockDim.x + threadIdx.x) * 2;       // this thread's pair of points
    if (p + 1 < n && vec) {
      const float2 a = uniform_point(stream, p), b = uniform_point(stream, p + 1);
      *reinterpret_cast<float4*>(orow + (int64_t)p * 2) = make_float4(a.x, a.y, b.x, b.y);
    } else {
      if (p < n) *reinterpret_cast<float2*>(orow + (int64_t)p * 2) = uniform_point(stream, p);
      if (p + 1 < n) *reinterpret_cast<float2*>(orow + (int64_t)(p + 1) * 2) = uniform_point(stream, p + 1);
    }
  }
}

template <bool RNG>
__global__ void __launch_bounds__(kFusedThreads, RNG ? 8 : 4) k_sample_select(const float* __restrict__ src,
                                                                 const int32_t* __restrict__ src_index,
                                                                 const float* __restrict__ coords,
                                                                 const int64_t* __restrict__ seed, int n, int k, int H,
                                                                 int W, const float* __restrict__ rand_coords,
                                                                 int n_rand, float* __restrict__ out_coords) {
  __shared__ __attribute__((aligned(16))) float tile[16384];
  __shared__ int cnt_lt[kKeysPerThread * kFusedWaves + 1], cnt_eq[kKeysPerThread * kFusedWaves + 1];
  __shared__ uint32_t s_prefix;
  __shared__ int s_krem;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hw = H * W;
  // [stamps begin]
  const float* s = src + (int64_t)src_index[row] * hw;
  if ((hw & 3) == 0) {
    // every 16-byte load of the map is issued before the first LDS store (one memory round trip, not four)
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* t4 = reinterpret_cast<float4*>(tile);
    const int n4 = hw >> 2;
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * kFusedThreads;
      v[u] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * kFusedThreads;
      if (i < n4) t4[i] = v[u];
    }
  } else {
    for (int i = tid; i < hw; i += kFusedThreads) tile[i] = s[i];
  }
  __syncthreads();
  // [phase 0 map staged]
  const float* cr = RNG ? nullptr : coords + row * (int64_t)n * 2;
  const uint32_t stream = RNG ? row_stream(seed[0], row) : 0u;
  uint32_t keys[kKeysPerThread];
  // batches of BATCH points: their coordinate loads are issued together (one HBM round trip per batch, not per
  // point); a scheduling barrier after every batch keeps the compiler from hoisting later batches' loads over the
  // 40 live key registers (it spilled 197 VGPRs without it)
  constexpr int BATCH = RNG ? 2 : 4;
#pragma unroll
  for (int jb = 0; jb < kKeysPerThread; jb += BATCH) {
    float2 xy[BATCH];
    const float* crb = cr;
    if constexpr (!RNG) asm volatile("" : "+s"(crb));      // (an opaque copy per batch: the 40 loads were hoisted to the top — 283 spills)
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int p = (jb + u) * kFusedThreads + tid;
      if constexpr (RNG) xy[u] = uniform_point(stream, p);
      else xy[u] = *reinterpret_cast<const float2*>(crb + (int64_t)min(p, n - 1) * 2);   // clamped, not predicated (slots >= n get the sentinel key)
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int p = (jb + u) * kFusedThreads + tid;
      Bil b;
      bil_setup_clamped(xy[u].x, xy[u].y, H, W, b);
      // same association as K8's gather loop ((((0 + t0) + t1) + t2) + t3): the fused form stays bit-identical to it
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) v += b.w[c] * tile[b.o[c]];
      keys[jb + u] = p < n ? abs_key(v) : 0xffffffffu;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // [phase 1 sampled]
  // [wavephase 16 sampled]
  // -- radix select of the k-th smallest key (11 + 10 + 10 bits below the sign bit, which |x| never sets), keys in registers.
  // The histogram lives in the map's tile, dead by now, as FOUR replicas (lane % 4) of 2048 64-BIT counters: on gfx950
  // a 32-bit LDS atomic costs ~200 clocks per wave instruction whatever the addresses (measured: 120 of them per
  // thread were 169 us of a row's 203, the same on clustered and on spread keys), ds_add_u64 ~10.  A replica's bins
  // are XOR-permuted by its number so that equal bins of different replicas sit in different banks.
  __syncthreads();                                        // every wave is done sampling from the tile
  // (opaque redefinition: keeps the histogram addresses of the first pass — 40 more live values — from being
  // computed up in the sampling loop, where they spilled 229 VGPRs)
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) asm volatile("" : "+v"(keys[j]));
  unsigned long long* rep = reinterpret_cast<unsigned long long*>(tile);
  const uint32_t rep_key = (uint32_t)(lane & 3) * (2048u * 8u) + (((uint32_t)(lane & 3) * 8u) << 3);   // byte offset ^ rotation
  char* rep_bytes = reinterpret_cast<char*>(tile);
  uint32_t prefix = 0, prefix_mask = 0;
  int krem = k;
  const int shifts[3] = {20, 10, 0};
  const int widths[3] = {11, 10, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass], nb = 1 << widths[pass];
    {
      float4* z = reinterpret_cast<float4*>(tile);
#pragma unroll
      for (int u = 0; u < 4; ++u) z[tid + u * kFusedThreads] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // [phase 10+pass zeroed]
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
      // slots beyond n hold the largest key (0xffffffff): counted in the top bin, they never reach the k-th smallest
      // (k <= n), so no validity test — and no 40 live lane masks — is needed from here on
      const uint32_t bin = (keys[j] >> shift) & (uint32_t)(nb - 1);
      if ((keys[j] & prefix_mask) == prefix)
        atomicAdd(reinterpret_cast<unsigned long long*>(rep_bytes + ((bin << 3) ^ rep_key)), 1ull);
    }
    __syncthreads();
    // [phase 7+pass histogram]
    // cumulative search: thread t owns bins 2t and 2t + 1 (adjacent in every replica: the permutation keeps bit 0);
    // wave totals through LDS
    constexpr int PERB = 2;
    int hb[PERB] = {0, 0}, hsum = 0;
    if (tid * PERB < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint4 v = *reinterpret_cast<const uint4*>(rep + r * 2048 + ((tid * PERB) ^ (r * 8)));
        hb[0] += (int)v.x;                                // (counts < 2^31: the low words)
        hb[1] += (int)v.z;
      }
    }
    hsum = hb[0] + hb[1];
    int inc = hsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) cnt_lt[wave] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += cnt_lt[w];
    int acc = off + inc - hsum;
    if (acc < krem && krem <= acc + hsum) {               // exactly one thread
#pragma unroll
      for (int q = 0; q < PERB; ++q) {
        if (acc < krem && krem <= acc + hb[q]) {
          s_prefix = prefix | ((uint32_t)(tid * PERB + q) << shift);
          s_krem = krem - acc;
        }
        acc += hb[q];
      }
    }
    __syncthreads();
    prefix = s_prefix;
    krem = s_krem;
    prefix_mask |= (uint32_t)(nb - 1) << shift;
    // [phase 3 bin search]
  }
  // -- compaction in index order: element (chunk j, thread t) has index j * kFusedThreads + t
  const uint32_t T = prefix;
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const unsigned long long m_lt = __ballot(keys[j] < T), m_eq = __ballot(keys[j] == T && T != 0xffffffffu);
    if (lane == 0) {
      cnt_lt[j * kFusedWaves + wave] = __popcll(m_lt);
      cnt_eq[j * kFusedWaves + wave] = __popcll(m_eq);
    }
  }
  __syncthreads();
  if (wave < 2) {                                         // wave 0 scans cnt_lt, wave 1 cnt_eq (exclusive, in place)
    int* c = wave == 0 ? cnt_lt : cnt_eq;
    constexpr int PER = (kKeysPerThread * kFusedWaves + 63) / 64;      // 10 consecutive entries per lane
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = lane * PER + q;
      loc[q] = e < kKeysPerThread * kFusedWaves ? c[e] : 0;
      sum += loc[q];
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    int run = inc - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = lane * PER + q;
      if (e < kKeysPerThread * kFusedWaves) c[e] = run;
      run += loc[q];
    }
  }
  __syncthreads();
  float* orow = out_coords + row * (int64_t)(k + n_rand) * 2;
  // [phase 4 positions]
  // Two steps.  (1) every candidate that is selected writes its INDEX to its output position, in LDS (the tile again:
  // the histogram is dead).  (2) the k selected indices are turned into coordinates — regenerated from the counter, or
  // read from the candidate list — by threads that are all busy, and stored with consecutive addresses.  Producing
  // the coordinates inside step 1 (the first version) ran the generator for all 40 slots of a thread to keep the one
  // in four that is selected: 1 700 VALU instructions per thread of a kernel that is VALU-bound.
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) asm volatile("" : "+v"(keys[j]));   // (no reuse of the count loop's 80 masks)
  uint32_t* sel = reinterpret_cast<uint32_t*>(tile);      // k <= 16 384 positions (checked by the host entry)
  // selected elements in front of chunk (j, wave): all smaller keys + the threshold-equal ones still taken; lane j of a
  // wave keeps its wave's entry of chunk j (one LDS read per lane, one v_readlane per chunk below)
  int my_base = 0, my_eq = 0;
  if (lane < kKeysPerThread) {
    my_eq = cnt_eq[lane * kFusedWaves + wave];
    my_base = cnt_lt[lane * kFusedWaves + wave] + min(krem, my_eq);
  }
#pragma unroll
  for (int j = 0; j < kKeysPerThread; ++j) {
    const bool is_lt = keys[j] < T;
    const unsigned long long m_lt = __ballot(is_lt);
    const unsigned long long m_eq = __ballot(keys[j] == T && T != 0xffffffffu);
    const int base = __builtin_amdgcn_readlane(my_base, j);
    int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_lt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_lt, 0u));
    bool take = is_lt;
    if (m_eq != 0ull) {                                   // (wave-uniform, rare: a key equal to the threshold in this chunk)
      const int eq_here = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_eq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_eq, 0u));
      const int room = max(krem - __builtin_amdgcn_readlane(my_eq, j), 0);      // threshold-equal keys this chunk may still take
      pos += min(room, eq_here);
      take = is_lt || (((m_eq >> lane) & 1ull) && eq_here < room);
    }
    if (take) sel[pos] = (uint32_t)(j * kFusedThreads + tid);
  }
  __syncthreads();
  {
    float2* od = reinterpret_cast<float2*>(orow);
#pragma unroll 2
    for (int i = tid; i < k; i += kFusedThreads) {
      const int p = (int)sel[i];
      if constexpr (RNG) od[i] = uniform_point(stream, p);
      else od[i] = *reinterpret_cast<const float2*>(cr + (int64_t)p * 2);
    }
  }
  // [phase 5 compaction]
  if (n_rand > 0) {
    const float2* rr = reinterpret_cast<const float2*>(rand_coords + row * (int64_t)n_rand * 2);
    float2* ot = reinterpret_cast<float2*>(orow + (int64_t)k * 2);
    for (int i = tid; i < n_rand; i += kFusedThreads) ot[i] = rr[i];
  }
  // [phase 6 tail]
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

extern "C" int mbv_sample_select_uncertain(const float* src, const int32_t* src_index, const float* coords,
                                           const int64_t* seed, int64_t rows, int32_t n, int32_t k, int32_t H,
                                           int32_t W, const float* rand_coords, int32_t n_rand, float* out_coords,
                                           void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (rows < 0 || n <= 0 || k <= 0 || k > n || H <= 0 || W <= 0 || n_rand < 0) return MBV_ERR_BAD_ARG;
  if ((int64_t)H * W > 16384 || n > kKeysPerThread * kFusedThreads || k > 16384) return MBV_ERR_UNSUPPORTED;
  if (rows == 0) return MBV_OK;
  if (!src || !src_index || !out_coords || (n_rand > 0 && !rand_coords)) return MBV_ERR_BAD_ARG;
  if ((coords == nullptr) == (seed == nullptr)) return MBV_ERR_BAD_ARG;      // exactly one source of points
  if (coords)
    hipLaunchKernelGGL(k_sample_select<false>, dim3((unsigned)rows), dim3(kFusedThreads), 0, stream, src, src_index,
                       coords, seed, n, k, H, W, rand_coords, n_rand, out_coords);
  else
    hipLaunchKernelGGL(k_sample_select<true>, dim3((unsigned)rows), dim3(kFusedThreads), 0, stream, src, src_index,
                       coords, seed, n, k, H, W, rand_coords, n_rand, out_coords);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_uniform_points(const int64_t* seed, int64_t rows, int32_t n, float* out_coords, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (rows < 0 || n <= 0 || rows > 65535) return MBV_ERR_BAD_ARG;
  if (rows == 0) return MBV_OK;
  if (!seed || !out_coords) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_uniform_points, dim3((unsigned)((n + 2047) / 2048), (unsigned)rows), dim3(256), 0, stream, seed,
                     rows, n, out_coords);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
