// K1 — range filter + deterministic hard voxelisation on gfx950, and the pillar gather/decoration
// kernels that sit directly on its output.
//
// Replaces mask_bev/models/encoders/mask_bev_encoders.py:95-117 (→ mmcv.ops.Voxelization, :69).
//
// The CPU reference hands out pillar ids in order of first appearance and keeps the first
// `max_points` points of a pillar in input order.  mmcv's CUDA path gets that order with an O(N^2)
// "scan all earlier points" kernel plus a single-thread kernel; here it is a sort/rank formulation:
//   1. key[i]   = scan * cells + cell(i)        (f32 floor-div, IEEE division; invalid → sentinel)
//   2. stable LSD radix sort of (key, i)        (8-bit digits; stable ⇒ runs are in input order)
//   3. run starts give each pillar's first point f; flag[f] = 1; exclusive scan of flag over the
//      ORIGINAL point order ranks the pillars by first appearance
//   4. one thread per run writes coors / num_points / the ≤ max_points kept point indices.
// Everything is integer/byte work bound by HBM/L2 latency; no MFMA.
#include "common.hpp"

namespace {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 16;
constexpr int kSortTile = kSortThreads * kSortItems;  // keys per block and pass
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

// ---------------------------------------------------------------------------------------------
// keys
// ---------------------------------------------------------------------------------------------
struct VoxelGeom {
  float x_min, y_min, z_min, x_max, y_max, z_max;
  float vx, vy, vz;
  int gx, gy, gz;
};

__global__ void __launch_bounds__(256) k_point_keys(const float* __restrict__ points, int dim,
                                                    const int32_t* __restrict__ scan_offsets, VoxelGeom g,
                                                    int prefilter, uint32_t cells, uint32_t invalid_key,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int b = blockIdx.y;
  const int64_t begin = scan_offsets[b], end = scan_offsets[b + 1];
  const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= end) return;
  const float* p = points + i * dim;
  const float x = p[0], y = p[1], z = p[2];
  bool ok = true;
  if (prefilter) {  // mask_bev_encoders.py:113-117, strict <, f32 compares
    ok = (g.x_min < x) && (x < g.x_max) && (g.y_min < y) && (y < g.y_max) && (g.z_min < z) && (z < g.z_max);
  }
  // mmcv dynamic_voxelize: c = floor((p - min) / vs), reject c < 0 || c >= grid.  The subtraction and
  // the division must stay separate IEEE f32 operations (compiled with -ffp-contract=off).
  const float qx = floorf((x - g.x_min) / g.vx);
  const float qy = floorf((y - g.y_min) / g.vy);
  const float qz = floorf((z - g.z_min) / g.vz);
  ok = ok && (qx >= 0.f) && (qx < (float)g.gx) && (qy >= 0.f) && (qy < (float)g.gy) && (qz >= 0.f) &&
       (qz < (float)g.gz);  // NaN fails every compare
  uint32_t key = invalid_key;
  if (ok) {
    const uint32_t cx = (uint32_t)qx, cy = (uint32_t)qy, cz = (uint32_t)qz;
    key = (uint32_t)b * cells + (cz * (uint32_t)g.gy + cy) * (uint32_t)g.gx + cx;
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan (u32), three small kernels: tile sums → scan of sums → tile scan + offset
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total, uint32_t* lds /*>=4*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t wave_off = 0, tot = 0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) {
    const uint32_t s = lds[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wave_off + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_reduce(const uint32_t* __restrict__ in, int64_t n,
                                                              uint32_t* __restrict__ partials) {
  __shared__ uint32_t lds[4];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) s += in[base + j];
  uint32_t tot;
  block_exclusive_scan(s, &tot, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) k_scan_partials(uint32_t* __restrict__ partials, int64_t nb) {
  __shared__ uint32_t lds[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? partials[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, &tot, lds);
    const uint32_t carry = carry_s;
    if (i < nb) partials[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
}

// `in` may alias `out` (in-place scan of the radix histogram): no __restrict__ on them.
__global__ void __launch_bounds__(kScanThreads) k_scan_apply(const uint32_t* in, int64_t n_in, int64_t n,
                                                             const uint32_t* __restrict__ partials, uint32_t* out) {
  __shared__ uint32_t lds[4];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    v[j] = (base + j < n_in) ? in[base + j] : 0u;
    s += v[j];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot, lds) + partials[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// out[i] = sum of in[0 .. i) for i < n.  `in` holds n_in <= n elements (elements from n_in on count as zero and are never
// read): the row-start scan has one more output than inputs (row_start[V] = K) and must not read past the caller's array.
int launch_exclusive_scan(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* partials, hipStream_t s,
                          int64_t n_in = -1) {
  if (n <= 0) return 0;
  if (n_in < 0 || n_in > n) n_in = n;
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, n_in, partials);
  MBV_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, partials, nb);
  MBV_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, n_in, n, partials, out);
  MBV_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort, 8-bit digits
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kSortThreads) k_radix_hist(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                             uint32_t* __restrict__ hist, int nblocks) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll 4
  for (int j = 0; j < kSortItems; ++j) {
    const int64_t i = base + j * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];  // digit-major
}

__global__ void __launch_bounds__(kSortThreads) k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                                const uint32_t* __restrict__ hist_scanned,
                                                                int nblocks) {
  __shared__ uint32_t base[256];     // next free global slot of each digit for this block
  __shared__ uint32_t wcnt[4][256];  // per-wave digit counts of the current 256-key slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  base[tid] = hist_scanned[(int64_t)tid * nblocks + blockIdx.x];
  const int64_t tile = (int64_t)blockIdx.x * kSortTile;
  for (int j = 0; j < kSortItems; ++j) {
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const int64_t i = tile + j * kSortThreads + tid;
    const bool valid = i < n;
    const uint32_t key = valid ? keys_in[i] : 0u;
    const uint32_t val = valid ? vals_in[i] : 0u;
    const uint32_t d = (key >> shift) & 255u;
    // lanes of this wave holding the same digit: 8 ballots instead of a serial match loop
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcnt[wave][d] = __popcll(same);
    __syncthreads();
    if (valid) {
      uint32_t pos = base[d] + rank;
      for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __syncthreads();
    base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// pillar ranking and output
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mark_first(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                    int64_t n, uint32_t invalid_key, uint32_t* __restrict__ flags) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t k = keys[p];
  if (k == invalid_key) return;
  if (p == 0 || keys[p - 1] != k) flags[vals[p]] = 1u;
}

// One block, one thread per scan: pillars per scan (capped), bases, total.
__global__ void k_scan_bases(const uint32_t* __restrict__ fscan, const int32_t* __restrict__ scan_offsets, int batch,
                             int max_voxels, int64_t capacity, int32_t* __restrict__ pillar_base,
                             int32_t* __restrict__ counts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int32_t acc = 0;
  for (int b = 0; b < batch; ++b) {
    int32_t c = (int32_t)(fscan[scan_offsets[b + 1]] - fscan[scan_offsets[b]]);
    if (max_voxels >= 0 && c > max_voxels) c = max_voxels;
    if ((int64_t)acc + c > capacity) c = (int32_t)(capacity - acc);  // never write past the caller's rows
    pillar_base[b] = acc;
    counts[b] = c;
    acc += c;
  }
  pillar_base[batch] = acc;
  counts[batch] = acc;
}

__global__ void __launch_bounds__(256) k_emit_pillars(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                      int64_t n, uint32_t invalid_key, uint32_t cells, int gx, int gy,
                                                      const uint32_t* __restrict__ fscan,
                                                      const int32_t* __restrict__ scan_offsets,
                                                      const int32_t* __restrict__ pillar_base,
                                                      const int32_t* __restrict__ counts, int max_points,
                                                      int max_voxels, int32_t* __restrict__ coors,
                                                      int32_t* __restrict__ num_points,
                                                      int32_t* __restrict__ pillar_points,
                                                      int32_t* __restrict__ cell_to_pillar) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t k = keys[p];
  if (k == invalid_key) return;
  if (p != 0 && keys[p - 1] == k) return;  // not a run start
  const uint32_t b = k / cells, cell = k - b * cells;
  const uint32_t first = vals[p];
  const int32_t local = (int32_t)(fscan[first] - fscan[scan_offsets[b]]);
  if (max_voxels >= 0 && local >= max_voxels) return;  // mmcv: voxel_num >= max_voxels → dropped
  if (local >= counts[b]) return;                      // capacity clamp of k_scan_bases
  const int64_t v = (int64_t)pillar_base[b] + local;
  const uint32_t x = cell % (uint32_t)gx, yz = cell / (uint32_t)gx;
  const uint32_t y = yz % (uint32_t)gy, z = yz / (uint32_t)gy;
  coors[v * 4 + 0] = (int32_t)b;
  coors[v * 4 + 1] = (int32_t)z;
  coors[v * 4 + 2] = (int32_t)y;
  coors[v * 4 + 3] = (int32_t)x;
  cell_to_pillar[k] = (int32_t)v;
  int cnt = 0;
  for (; cnt < max_points; ++cnt) {
    const int64_t q = p + cnt;
    if (q >= n || keys[q] != k) break;
    pillar_points[v * max_points + cnt] = (int32_t)vals[q];
  }
  num_points[v] = cnt;
  for (int j = cnt; j < max_points; ++j) pillar_points[v * max_points + j] = -1;
}

__global__ void k_total_rows(const int32_t* __restrict__ row_start, int32_t* __restrict__ counts, int batch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) counts[batch + 1] = row_start[counts[batch]];
}

// ---------------------------------------------------------------------------------------------
// dense voxel tensor (API only) and decoration of real points
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_voxels(const float* __restrict__ points, int dim,
                                                       const int32_t* __restrict__ pillar_points, int64_t total_slots,
                                                       float* __restrict__ voxels) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total_slots) return;
  const int32_t idx = pillar_points[s];
  float* o = voxels + s * dim;
  if (idx < 0) {
    for (int k = 0; k < dim; ++k) o[k] = 0.f;
  } else {
    const float* p = points + (int64_t)idx * dim;
    for (int k = 0; k < dim; ++k) o[k] = p[k];
  }
}

__global__ void __launch_bounds__(256) k_pfn_decorate(const float* __restrict__ points, int dim,
                                                      const int32_t* __restrict__ pillar_points,
                                                      const int32_t* __restrict__ num_points,
                                                      const int32_t* __restrict__ row_start,
                                                      const int32_t* __restrict__ coors, int64_t num_pillars,
                                                      int max_points, float vx, float vy, float vz, float x_off,
                                                      float y_off, float z_off, float* __restrict__ rows,
                                                      int64_t* __restrict__ row_pillar) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= num_pillars) return;
  const int n = num_points[v];
  const int32_t* pp = pillar_points + v * max_points;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = 0; j < n; ++j) {
    const float* p = points + (int64_t)pp[j] * dim;
    sx += p[0];
    sy += p[1];
    sz += p[2];
  }
  const float fn = (float)n;
  const float mx = sx / fn, my = sy / fn, mz = sz / fn;
  // mmdet3d PillarFeatureNet: centre = coor * v + (v / 2 + range_min), f32 arithmetic
  const float cx = (float)coors[v * 4 + 3] * vx + x_off;
  const float cy = (float)coors[v * 4 + 2] * vy + y_off;
  const float cz = (float)coors[v * 4 + 1] * vz + z_off;
  const int width = dim + 7;
  const int64_t r0 = row_start[v];
  for (int j = 0; j < n; ++j) {
    const float* p = points + (int64_t)pp[j] * dim;
    float* o = rows + (r0 + j) * width;
    const float fx = p[0] - cx, fy = p[1] - cy, fz = p[2] - cz;
    o[0] = fx;
    o[1] = fy;
    o[2] = fz;
    for (int k = 3; k < dim; ++k) o[k] = p[k];
    o[dim + 0] = p[0] - mx;
    o[dim + 1] = p[1] - my;
    o[dim + 2] = p[2] - mz;
    o[dim + 3] = fx;
    o[dim + 4] = fy;
    o[dim + 5] = fz;
    o[dim + 6] = sqrtf(fx * fx + fy * fy + fz * fz);
    row_pillar[r0 + j] = v;
  }
}

// The same decoration with EIGHT lanes per pillar (k_pfn_decorate gives a pillar to one thread, which walks its points as a
// chain of dependent loads — slot index, then the point — twice over: ~4 n memory round trips per pillar and 4-byte stores
// scattered over 64 pillars per wave instruction; 69 us for 17 MB of rows at the bench batch).  Every lane of a pillar's group
// forms the pillar mean itself, in slot order (the sum the one-thread form and the oracle compute, bit for bit): the slot
// indices and then the points of 8 slots are requested together (the group's lanes ask for the same addresses), and lane j
// writes rows j, j + 8, ...: a group's stores of one column fall into 352 consecutive bytes.
__global__ void __launch_bounds__(256) k_pfn_decorate8(const float* __restrict__ points, int dim,
                                                       const int32_t* __restrict__ pillar_points,
                                                       const int32_t* __restrict__ num_points,
                                                       const int32_t* __restrict__ row_start,
                                                       const int32_t* __restrict__ coors, int64_t num_pillars,
                                                       int max_points, float vx, float vy, float vz, float x_off,
                                                       float y_off, float z_off, float* __restrict__ rows,
                                                       int64_t* __restrict__ row_pillar) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int j8 = threadIdx.x & 7;
  if (v >= num_pillars) return;
  const int n = num_points[v];
  const int32_t* pp = pillar_points + v * max_points;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j0 = 0; j0 < n; j0 += 8) {
    int32_t idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) idx[u] = pp[j0 + u < n ? j0 + u : n - 1];
    float px[8], py[8], pz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* p = points + (int64_t)idx[u] * dim;
      px[u] = p[0]; py[u] = p[1]; pz[u] = p[2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u < n) { sx += px[u]; sy += py[u]; sz += pz[u]; }
  }
  const float fn = (float)n;
  const float mx = sx / fn, my = sy / fn, mz = sz / fn;
  const float cx = (float)coors[v * 4 + 3] * vx + x_off;
  const float cy = (float)coors[v * 4 + 2] * vy + y_off;
  const float cz = (float)coors[v * 4 + 1] * vz + z_off;
  const int width = dim + 7;
  const int64_t r0 = row_start[v];
  for (int j = j8; j < n; j += 8) {
    const float* p = points + (int64_t)pp[j] * dim;
    float* o = rows + (r0 + j) * width;
    const float x = p[0], y = p[1], z = p[2];
    const float fx = x - cx, fy = y - cy, fz = z - cz;
    o[0] = fx;
    o[1] = fy;
    o[2] = fz;
    for (int k = 3; k < dim; ++k) o[k] = p[k];
    o[dim + 0] = x - mx;
    o[dim + 1] = y - my;
    o[dim + 2] = z - mz;
    o[dim + 3] = fx;
    o[dim + 4] = fy;
    o[dim + 5] = fz;
    o[dim + 6] = sqrtf(fx * fx + fy * fy + fz * fz);
    row_pillar[r0 + j] = v;
  }
}

struct VoxWorkspace {
  uint32_t *keys_a, *keys_b, *vals_a, *vals_b, *hist, *flags, *fscan, *partials;
  int32_t* pillar_base;
  size_t bytes;
};

VoxWorkspace carve_voxelize(void* ws, int64_t n, int batch) {
  MbvCarver c(ws);
  VoxWorkspace w;
  const int64_t nblocks = (n + kSortTile - 1) / kSortTile;
  w.keys_a = c.take<uint32_t>(n);
  w.keys_b = c.take<uint32_t>(n);
  w.vals_a = c.take<uint32_t>(n);
  w.vals_b = c.take<uint32_t>(n);
  w.hist = c.take<uint32_t>(256 * (nblocks > 0 ? nblocks : 1));
  w.flags = c.take<uint32_t>(n + 1);
  w.fscan = c.take<uint32_t>(n + 1);
  w.partials = c.take<uint32_t>((n + 1 + kScanTile - 1) / kScanTile + 1);
  w.pillar_base = c.take<int32_t>(batch + 1);
  w.bytes = c.off;
  return w;
}

}  // namespace

extern "C" int mbv_abi_version(void) { return 58; }

extern "C" size_t mbv_voxelize_workspace_bytes(int64_t total_points, int32_t batch, int64_t /*cells_per_scan*/) {
  if (total_points < 0 || batch < 0) return 0;
  return carve_voxelize(nullptr, total_points, batch).bytes;
}

extern "C" int mbv_voxelize(const float* points, int32_t point_dim, int64_t total_points,
                            const int32_t* scan_offsets, int32_t batch, float x_min, float y_min, float z_min,
                            float x_max, float y_max, float z_max, float vx, float vy, float vz, int32_t gx,
                            int32_t gy, int32_t gz, int32_t prefilter, int32_t max_points, int32_t max_voxels,
                            int64_t pillar_capacity, int32_t* coors, int32_t* num_points, int32_t* pillar_points,
                            int32_t* row_start, int32_t* cell_to_pillar, int32_t* counts, void* workspace,
                            size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (batch <= 0 || point_dim < 3 || total_points < 0 || max_points <= 0 || gx <= 0 || gy <= 0 || gz <= 0 ||
      pillar_capacity < 0 || pillar_capacity > total_points)
    return MBV_ERR_BAD_ARG;
  const int64_t cells64 = (int64_t)gx * gy * gz;
  if (cells64 * batch >= 0xFFFFFFFFll || total_points >= 0x7FFFFFFFll) return MBV_ERR_UNSUPPORTED;
  if (!points || !scan_offsets || !coors || !num_points || !pillar_points || !row_start || !cell_to_pillar || !counts)
    return MBV_ERR_BAD_ARG;
  const int64_t n = total_points;
  VoxWorkspace w = carve_voxelize(workspace, n, batch);
  if (!workspace || workspace_bytes < w.bytes) return MBV_ERR_WORKSPACE;
  const uint32_t cells = (uint32_t)cells64;
  const uint32_t invalid_key = cells * (uint32_t)batch;  // sorts after every real key

  MBV_CHECK_HIP(mbv_fill_async(cell_to_pillar, 0xff, sizeof(int32_t) * cells64 * batch, stream));
  MBV_CHECK_HIP(mbv_fill_async(num_points, 0, sizeof(int32_t) * pillar_capacity, stream));
  MBV_CHECK_HIP(mbv_fill_async(w.flags, 0, sizeof(uint32_t) * (n + 1), stream));

  if (n > 0) {
    // 1. keys.  The y dimension of the grid walks the scans; x covers the longest scan.
    VoxelGeom g{x_min, y_min, z_min, x_max, y_max, z_max, vx, vy, vz, gx, gy, gz};
    // upper bound on points per scan is n; blocks past a scan's end exit immediately
    const unsigned bx = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_point_keys, dim3(bx, batch), dim3(256), 0, stream, points, point_dim, scan_offsets, g,
                       prefilter, cells, invalid_key, w.keys_a, w.vals_a);
    MBV_CHECK_LAUNCH();

    // 2. stable radix sort on the bits that can be set
    int bits = 1;
    while (bits < 32 && (invalid_key >> bits) != 0u) ++bits;
    const int passes = (bits + 7) / 8;
    const int nblocks = (int)((n + kSortTile - 1) / kSortTile);
    uint32_t *ka = w.keys_a, *kb = w.keys_b, *va = w.vals_a, *vb = w.vals_b;
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kSortThreads), 0, stream, ka, n, p * 8, w.hist, nblocks);
      MBV_CHECK_LAUNCH();
      int rc = launch_exclusive_scan(w.hist, w.hist, (int64_t)256 * nblocks, w.partials, stream);
      if (rc) return rc;
      hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(kSortThreads), 0, stream, ka, va, kb, vb, n, p * 8,
                         w.hist, nblocks);
      MBV_CHECK_LAUNCH();
      uint32_t* t = ka; ka = kb; kb = t;
      t = va; va = vb; vb = t;
    }

    // 3. rank pillars by first appearance
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_mark_first, dim3(nb), dim3(256), 0, stream, ka, va, n, invalid_key, w.flags);
    MBV_CHECK_LAUNCH();
    int rc = launch_exclusive_scan(w.flags, w.fscan, n + 1, w.partials, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_bases, dim3(1), dim3(64), 0, stream, w.fscan, scan_offsets, batch, max_voxels,
                       pillar_capacity, w.pillar_base, counts);
    MBV_CHECK_LAUNCH();

    // 4. outputs
    hipLaunchKernelGGL(k_emit_pillars, dim3(nb), dim3(256), 0, stream, ka, va, n, invalid_key, cells, gx, gy, w.fscan,
                       scan_offsets, w.pillar_base, counts, max_points, max_voxels, coors, num_points, pillar_points,
                       cell_to_pillar);
    MBV_CHECK_LAUNCH();
  } else {
    MBV_CHECK_HIP(mbv_fill_async(counts, 0, sizeof(int32_t) * (batch + 2), stream));
  }
  // row_start = exclusive scan of num_points (zero beyond V), K = row_start[V]
  // (num_points has pillar_capacity entries, row_start one more: the scan's last input does not exist)
  int rc = launch_exclusive_scan(reinterpret_cast<const uint32_t*>(num_points),
                                 reinterpret_cast<uint32_t*>(row_start), pillar_capacity + 1, w.partials, stream,
                                 pillar_capacity);
  if (rc) return rc;
  hipLaunchKernelGGL(k_total_rows, dim3(1), dim3(64), 0, stream, row_start, counts, batch);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_gather_voxels(const float* points, int32_t point_dim, const int32_t* pillar_points,
                                 int64_t num_pillars, int32_t max_points, float* voxels, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_pillars < 0 || max_points <= 0 || point_dim <= 0) return MBV_ERR_BAD_ARG;
  const int64_t slots = num_pillars * max_points;
  if (slots == 0) return MBV_OK;
  if (!points || !pillar_points || !voxels) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_gather_voxels, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, points, point_dim,
                     pillar_points, slots, voxels);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_pfn_decorate(const float* points, int32_t point_dim, const int32_t* pillar_points,
                                const int32_t* num_points, const int32_t* row_start, const int32_t* coors,
                                int64_t num_pillars, int32_t max_points, float vx, float vy, float vz, float x_off,
                                float y_off, float z_off, float* rows, int64_t* row_pillar, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_pillars < 0 || max_points <= 0 || point_dim < 3) return MBV_ERR_BAD_ARG;
  if (num_pillars == 0) return MBV_OK;
  if (!points || !pillar_points || !num_points || !row_start || !coors || !rows || !row_pillar) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_pfn_decorate8, dim3((unsigned)((num_pillars * 8 + 255) / 256)), dim3(256), 0, stream, points,
                     point_dim, pillar_points, num_points, row_start, coors, num_pillars, max_points, vx, vy, vz, x_off,
                     y_off, z_off, rows, row_pillar);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
