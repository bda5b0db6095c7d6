// K19 — row-local stage chains for the transformer decoder's query side (B*Q = 400 tokens x 256 channels), gfx950.
//
// Replaces, per decoder layer, the ≈ 28 forward and ≈ 30 backward launches of the reference's
//   cross-attn out-proj + residual + LayerNorm -> self-attn q/k/v projections -> out-proj + residual + LayerNorm ->
//   FFN 256-2048-256 (ReLU) + residual + LayerNorm -> post-norm + class / mask-embed heads -> next layer's query projection
// (mask_bev/models/networks/mask2former_head/mask2former_head.py:535-560 and :428-472; mmdet
// Mask2FormerTransformerDecoderLayer / mmcv MultiheadAttention, FFN) — few-row GEMMs of 5-8 us each with the chip idle,
// LayerNorms, clamps, adds and casts — with ONE launch per stretch between two attention kernels.
//
// Everything on the query side between two attention calls is ROW-LOCAL: a workgroup owns 16 token rows and walks a
// small program of stages over them (mbv_rowchain_run, include/maskbev_hip.h): LOAD / STORE rows, GEMM against a weight
// matrix streamed straight from L2 into MFMA B fragments (the M <= 16 weight-streaming form: the operand is used once
// per workgroup, an LDS round trip would be pure overhead), LayerNorm forward / backward, ADD, column partial sums.
// Activations live in LDS "slots" of 16 x 256 f32 between stages; nothing of a chain touches HBM except what the
// backward pass or an attention kernel needs.  The program is passed BY VALUE in the kernel arguments (no upload, and a
// captured HIP graph bakes it in), built by the host side (mask_bev_amd/decoder_fused.py).
//
// GEMM operand types (`wdtype`): f32 weights -> exact-f32 MFMA (v_mfma_f32_16x16x4_f32), bf16 / fp16 weights -> the
// activations are rounded to that type as MFMA A fragments, f32 accumulation (v_mfma_f32_16x16x32_{bf16,f16}).
// Backward data gradients dX = dY W are NT products against TRANSPOSED weight copies (mbv_transpose_group, one launch
// per step for the whole decoder), so that the reduction index is the contiguous one in every stream.
#include <stdlib.h>

#include "common.hpp"

namespace {

constexpr int RC_ROWS = 16, RC_MAXC = 256, RC_LD = RC_MAXC + 4, RC_SLOTS = 7, RC_NT = 512, RC_NW = RC_NT / 64;
constexpr int RC_MAX_STAGES = 64;

struct RowProgram {
  int num_stages, rows, q_mod, wdtype;
  float eps;
  int pad[3];
  MbvRowStage st[RC_MAX_STAGES];
};

// Pointers that arrive inside the program are GLOBAL memory.  Left generic, every access compiled to a FLAT instruction,
// which counts in lgkmcnt as well as vmcnt: each LDS wait (and the stage barrier's) then also drained the prefetched
// loads.  Everything below goes through address-space-1 pointers (global_load / global_store: vmcnt only).
#define G1 __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const G1 T* gp(const void* p) { return (const G1 T*)p; }
template <typename T>
__device__ __forceinline__ G1 T* gpw(const void* p) { return (G1 T*)const_cast<void*>(p); }
// 16- / 8-byte global accesses through native vector types (HIP's uint4 / float4 classes cannot bind to address space 1)
typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t;
typedef unsigned __attribute__((ext_vector_type(2))) u32x2_t;
__device__ __forceinline__ uint4 gld16(const G1 void* p) {
  const u32x4_t v = *reinterpret_cast<const G1 u32x4_t*>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gld8(const G1 void* p) {
  const u32x2_t v = *reinterpret_cast<const G1 u32x2_t*>(p);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void gst16(G1 void* p, uint4 v) { *reinterpret_cast<G1 u32x4_t*>(p) = u32x4_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gst8(G1 void* p, uint2 v) { *reinterpret_cast<G1 u32x2_t*>(p) = u32x2_t{v.x, v.y}; }

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ float cvt_in(const void* p, int64_t i, int dt) {
  if (dt == MBV_DT_F32) return reinterpret_cast<const float*>(p)[i];
  const unsigned short h = reinterpret_cast<const unsigned short*>(p)[i];
  if (dt == MBV_DT_BF16) return __uint_as_float((unsigned)h << 16);
  return (float)__builtin_bit_cast(_Float16, h);
}

__device__ __forceinline__ unsigned short f32_to_16(float f, int dt) {
  if (dt == MBV_DT_BF16) return f32_to_bf16_rne(f);
  return __builtin_bit_cast(unsigned short, (_Float16)f);
}

// ---- stages ------------------------------------------------------------------------------------------------------
// A workgroup's chain is a sequence of short dependent stages; left alone, every LOAD and every GEMM would expose one
// full global-memory round trip (1-2 us against 0.1-0.3 us of work).  All global READS of a program are of data no
// stage of the same program writes (weights, and activations earlier kernels produced — the host side keeps to that),
// so they can be requested early: the NEXT load / GEMM stage's operands are fetched into registers while the current
// stages run, and the stage barriers wait for LDS traffic only (a plain __syncthreads() would drain vmcnt, i.e. wait
// for the prefetches).
struct LoadPre {
  uint4 v[2];             // raw bytes of up to two pieces per thread of the 16 x 256 block (16 B f32 / 8 B 16-bit)
  float4 a[2];            // the same pieces of the added f32 operand
};

// Every load below is UNCONDITIONAL (addresses are clamped into the tensor, validity is applied when the registers are
// consumed): a predicated load compiles to a branch with `s_waitcnt vmcnt(0)` behind it, which serialises the prefetch.
__device__ __forceinline__ void load_prefetch(const MbvRowStage& s, LoadPre& pre, int row0, int rows, int q_mod) {
  const int dt = s.flags & 3, n4 = s.n >> 2;
  const bool has0 = s.p0 != nullptr, has1 = s.p1 != nullptr;
  const G1 char* base0 = gp<char>(has0 ? s.p0 : s.p1);      // some readable address either way
  const G1 float* base1 = gp<float>(has1 ? s.p1 : s.p0);
  const int ld0 = has0 ? s.ld : 0, ld1 = has1 ? s.ld2 : 0;
  const int es = dt == MBV_DT_F32 ? 4 : 2;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int i = threadIdx.x + u * RC_NT;
    i = i < RC_ROWS * n4 ? i : RC_ROWS * n4 - 1;
    const int r = i / n4, c = (i - r * n4) * 4;
    int row = row0 + r;
    row = row < rows ? row : rows - 1;
    const G1 char* p = base0 + ((int64_t)row * ld0 + (has0 ? c : 0)) * (has0 ? es : 4);
    if (dt == MBV_DT_F32 || !has0) {
      pre.v[u] = gld16(p);
    } else {
      const uint2 t = gld8(p);
      pre.v[u].x = t.x; pre.v[u].y = t.y; pre.v[u].z = 0u; pre.v[u].w = 0u;
    }
    const int prow = q_mod > 0 ? row % q_mod : row;
    pre.a[u] = __builtin_bit_cast(float4, gld16(base1 + (int64_t)prow * ld1 + (has1 ? c : 0)));
  }
}

__device__ __forceinline__ void st_load(const MbvRowStage& s, float* slots, const LoadPre& pre, int row0, int rows) {
  float* dst = slots + s.dst * (RC_ROWS * RC_LD);
  const int dt = s.flags & 3, n4 = s.n >> 2;                // n % 4 == 0 (host-checked)
  const bool has0 = s.p0 != nullptr, has1 = s.p1 != nullptr;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = threadIdx.x + u * RC_NT;
    if (i < RC_ROWS * n4) {
      const int r = i / n4, c = (i - r * n4) * 4;
      const bool live = row0 + r < rows;
      float4 v;
      const uint4 raw = pre.v[u];
      if (!has0) {                                           // base operand from a slot (e.g. x + positions)
        v = *reinterpret_cast<const float4*>(__builtin_assume_aligned(slots + s.src * (RC_ROWS * RC_LD) + r * RC_LD + c, 16));
      } else if (dt == MBV_DT_F32) {
        v = __builtin_bit_cast(float4, raw);
      } else if (dt == MBV_DT_BF16) {
        v = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                        __uint_as_float(raw.y & 0xffff0000u));
      } else {
        v = make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(raw.x & 0xffff)),
                        (float)__builtin_bit_cast(_Float16, (unsigned short)(raw.x >> 16)),
                        (float)__builtin_bit_cast(_Float16, (unsigned short)(raw.y & 0xffff)),
                        (float)__builtin_bit_cast(_Float16, (unsigned short)(raw.y >> 16)));
      }
      if (has1) { v.x += pre.a[u].x; v.y += pre.a[u].y; v.z += pre.a[u].z; v.w += pre.a[u].w; }
      if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(__builtin_assume_aligned(dst + r * RC_LD + c, 16)) = v;
    }
  }
}

__device__ __forceinline__ void st_store(const MbvRowStage& s, const float* slots, int row0, int rows, int sidx) {
  const float* src = slots + s.src * (RC_ROWS * RC_LD);
  const int dt = s.flags & 3, n = s.n;
  const bool accum = (s.flags & MBV_RC_ACCUM) != 0;
  const int64_t part_off = (s.flags & MBV_RC_SPLIT) ? (int64_t)sidx * s.ld2 : 0;   // (elements) this workgroup's part
  if ((n & 3) == 0) {
    const int n4 = n >> 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {                            // 16 x 256 / 4 = 1024 pieces over 512 threads: no loop
      const int i = threadIdx.x + u * RC_NT;
      if (i >= RC_ROWS * n4) continue;
      const int r = i / n4, c = (i - r * n4) * 4;
      const int row = row0 + r;
      if (row >= rows) continue;
      const float4 v = *reinterpret_cast<const float4*>(__builtin_assume_aligned(src + r * RC_LD + c, 16));
      const int64_t o = part_off + (int64_t)row * s.ld + c;
      if (dt == MBV_DT_F32) {
        G1 float* p = gpw<float>(s.p0) + o;
        if (accum) {
          const float4 old = __builtin_bit_cast(float4, gld16(p));
          gst16(p, __builtin_bit_cast(uint4, make_float4(old.x + v.x, old.y + v.y, old.z + v.z, old.w + v.w)));
        } else {
          gst16(p, __builtin_bit_cast(uint4, v));
        }
      } else {
        uint2 u2;
        u2.x = (unsigned)f32_to_16(v.x, dt) | ((unsigned)f32_to_16(v.y, dt) << 16);
        u2.y = (unsigned)f32_to_16(v.z, dt) | ((unsigned)f32_to_16(v.w, dt) << 16);
        gst8(gpw<unsigned short>(s.p0) + o, u2);
      }
    }
  } else {                                                 // narrow outputs (the class head: n = classes + 1)
    for (int i = threadIdx.x; i < RC_ROWS * n; i += RC_NT) {
      const int r = i / n, c = i - r * n;
      const int row = row0 + r;
      if (row >= rows) continue;
      const float v = src[r * RC_LD + c];
      const int64_t o = part_off + (int64_t)row * s.ld + c;
      if (dt == MBV_DT_F32) {
        G1 float* p = gpw<float>(s.p0) + o;
        *p = accum ? *p + v : v;
      } else {
        gpw<unsigned short>(s.p0)[o] = f32_to_16(v, dt);
      }
    }
  }
}

template <int WDT>
struct GemmPre {
  static constexpr int KB = WDT == MBV_DT_F32 ? RC_MAXC / 16 : RC_MAXC / 32;
  uint4 b[WDT == MBV_DT_F32 ? 1 : KB];            // the B fragments of the wave's FIRST tile of a GEMM stage (16-bit weights)
};

// Weight operands of the 16-bit programs come in FRAGMENT-MAJOR copies (mbv_fragment_group): the 64 lanes' 16-byte
// B fragments of (16-row tile t, 32-column block kb) are 1 KB of consecutive memory at ((t * KBN + kb) * 64 + lane) * 16.
// Read from the row-major matrix the same fragments are 16 rows x 64 bytes — 16 cache lines per load instruction, and
// the texture-address unit works per line: 1.8 us per tile with all eight waves loading (measured; unchanged with the
// weights resident in L1).  One instruction now covers 8 full lines.
template <int WDT>
__device__ __forceinline__ void gemm_load_tile(const MbvRowStage& s, int t, uint4* b) {
  constexpr int KB = GemmPre<WDT>::KB;
  constexpr int KSTEP = WDT == MBV_DT_F32 ? 16 : 32, ES = WDT == MBV_DT_F32 ? 4 : 2;
  const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
  const int tiles = (s.n + 15) >> 4;
  t = t < tiles ? t : tiles - 1;                       // a wave without this tile re-reads the last one (unused)
  if (s.flags & MBV_RC_FRAG) {
    const G1 char* base = gp<char>(s.p0) + ((int64_t)t * s.ld * 64 + lane) * 16;       // ld = k blocks per fragment row
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int ko = kb * KSTEP < s.k ? kb : 0;        // k blocks beyond k re-read block 0 and are never multiplied
      b[kb] = gld16(base + ko * 1024);
    }
    return;
  }
  int col = t * 16 + m;
  col = col < s.n ? col : s.n - 1;                     // clamped, not predicated: columns >= n are never stored
  const G1 char* wrow = gp<char>(s.p0) + ((int64_t)col * s.ld + (KSTEP / 4) * g) * ES;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int ko = kb * KSTEP < s.k ? kb * KSTEP : 0;  // k blocks beyond k re-read block 0 and are never multiplied
    b[kb] = gld16(wrow + ko * ES);
  }
}

// (f32 weights: 16 float4 fragments per tile — three live sets would spill; the f32 program, which is not the
// throughput path, loads each tile at its stage instead)
template <int WDT>
__device__ __forceinline__ void gemm_prefetch(const MbvRowStage& s, GemmPre<WDT>& pre) {
  if constexpr (WDT != MBV_DT_F32) gemm_load_tile<WDT>(s, threadIdx.x >> 6, pre.b);
}

typedef __attribute__((ext_vector_type(8))) float f32x8;

// The A fragments of a stage: the SAME for every output tile, so each wave reads and converts them ONCE per stage —
// all LDS reads issued back to back as 16-byte accesses (the slot rows are 16-byte aligned; without the alignment
// promise the compiler split them into b96 + b32 pieces with a wait behind each: 1.1 us per tile), then the products
// of both tiles.  NKB k-blocks, compile-time: a run-time bound put every block in its own basic block.
template <int WDT, int NKB>
__device__ __forceinline__ void gemm_tiles(const float* X, const uint4* b0, const uint4* b1, bool two, f32x4* acc) {
  const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
  if constexpr (WDT == MBV_DT_F32) {
    // k permuted consistently in A and B: MFMA j of a 16-wide k block sums k = kb + 4 g' + j, g' = 0..3
    const float4* xr = reinterpret_cast<const float4*>(__builtin_assume_aligned(X + m * RC_LD + 4 * g, 16));
    float4 a[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) a[kb] = xr[kb * 4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      if (tt == 1 && !two) break;
      const uint4* b = tt == 0 ? b0 : b1;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float4 bb = __builtin_bit_cast(float4, b[kb]);
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].x, bb.x, acc[tt], 0, 0, 0);
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].y, bb.y, acc[tt], 0, 0, 0);
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].z, bb.z, acc[tt], 0, 0, 0);
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].w, bb.w, acc[tt], 0, 0, 0);
      }
    }
  } else {
    const float4* xr = reinterpret_cast<const float4*>(__builtin_assume_aligned(X + m * RC_LD + 8 * g, 16));
    float4 r0[NKB], r1[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      r0[kb] = xr[kb * 8];
      r1[kb] = xr[kb * 8 + 1];
    }
    uint4 a[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const f32x8 v = {r0[kb].x, r0[kb].y, r0[kb].z, r0[kb].w, r1[kb].x, r1[kb].y, r1[kb].z, r1[kb].w};
      if constexpr (WDT == MBV_DT_BF16) a[kb] = __builtin_bit_cast(uint4, __builtin_convertvector(v, bf16x8));
      else a[kb] = __builtin_bit_cast(uint4, __builtin_convertvector(v, f16x8));
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      if (tt == 1 && !two) break;
      const uint4* b = tt == 0 ? b0 : b1;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if constexpr (WDT == MBV_DT_BF16)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kb]), __builtin_bit_cast(bf16x8, b[kb]),
                                                            acc[tt], 0, 0, 0);
        else
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[kb]), __builtin_bit_cast(f16x8, b[kb]),
                                                           acc[tt], 0, 0, 0);
      }
    }
  }
}

// any k (a multiple of the block width): k-blocks tested one by one — the slow form, used by odd test widths only
template <int WDT>
__device__ __forceinline__ void gemm_tiles_any(const float* X, const uint4* b0, const uint4* b1, bool two, int k, f32x4* acc) {
  constexpr int KB = GemmPre<WDT>::KB, KSTEP = WDT == MBV_DT_F32 ? 16 : 32;
  const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    if (tt == 1 && !two) break;
    const uint4* b = tt == 0 ? b0 : b1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (kb * KSTEP < k) {
        if constexpr (WDT == MBV_DT_F32) {
          const float4 a = *reinterpret_cast<const float4*>(__builtin_assume_aligned(X + m * RC_LD + kb * 16 + 4 * g, 16));
          const float4 bb = __builtin_bit_cast(float4, b[kb]);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, acc[tt], 0, 0, 0);
        } else {
          const float4* xr = reinterpret_cast<const float4*>(__builtin_assume_aligned(X + m * RC_LD + kb * 32 + 8 * g, 16));
          const float4 r0 = xr[0], r1 = xr[1];
          const f32x8 v = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          if constexpr (WDT == MBV_DT_BF16)
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_convertvector(v, bf16x8),
                                                              __builtin_bit_cast(bf16x8, b[kb]), acc[tt], 0, 0, 0);
          else
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_convertvector(v, f16x8),
                                                             __builtin_bit_cast(f16x8, b[kb]), acc[tt], 0, 0, 0);
        }
      }
    }
  }
}

// dst[r][j] = act((ACCUM ? dst[r][j] : 0) + sum_k src[r][k] W[j][k] + bias[j]) [* (src2[r][j] > 0)], j < n.
// One wave owns the 16-column output tiles t = wave and wave + 8 (n <= 256); its B fragments come straight from global
// memory (16 rows of W x 64 contiguous bytes per load instruction).  `pre` holds the first tile's fragments, requested
// while the previous stages ran.  The stage requests its second tile's fragments first, multiplies the first tile
// (already here), then the second, and only then — `pre` is dead — requests the NEXT GEMM stage's first tile into the
// same registers, so that its round trip runs underneath the epilogue, the barrier and whatever stages follow.
// (Two alternating register sets would hide more, but across this kernel's stage switch the register allocator splits
// their live ranges and copies them — behind an `s_waitcnt vmcnt(0)`.)
template <int WDT>
__device__ __forceinline__ void st_gemm(const MbvRowStage& s, bool has_next, const MbvRowStage& next, float* slots,
                                        GemmPre<WDT>& pre) {
  constexpr int KB = GemmPre<WDT>::KB;
  const float* X = slots + s.src * (RC_ROWS * RC_LD);
  float* Y = slots + s.dst * (RC_ROWS * RC_LD);
  const float* Mk = (s.flags & MBV_RC_MASK) ? slots + s.src2 * (RC_ROWS * RC_LD) : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int n = s.n, k = s.k;
  const int tiles = (n + 15) >> 4;
  const bool accum = (s.flags & MBV_RC_ACCUM) != 0, relu = (s.flags & MBV_RC_RELU) != 0;
  const G1 float* bias = gp<float>(s.p1);
  const G1 float* bp = bias ? bias : gp<float>(s.p0);     // unconditional loads, selected below
  uint4 b2[KB];
  if constexpr (WDT != MBV_DT_F32) gemm_load_tile<WDT>(s, wave + RC_NW, b2);
  float bvs[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int c = (wave + tt * RC_NW) * 16 + m;
    bvs[tt] = bp[c < n ? c : n - 1];
  }
  f32x4 acc[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int col = (wave + tt * RC_NW) * 16 + m;
    acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (accum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[tt][i] = Y[(4 * g + i) * RC_LD + (col < n ? col : 0)];
    }
  }
  const bool two = wave + RC_NW < tiles;               // wave-uniform
  if (wave < tiles) {
    constexpr int KSTEP = WDT == MBV_DT_F32 ? 16 : 32;
    if constexpr (WDT == MBV_DT_F32) {                 // (f32: the tiles' fragments are loaded here, one at a time)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (tt == 1 && !two) break;
        gemm_load_tile<WDT>(s, wave + tt * RC_NW, b2);
        if (k == KB * KSTEP) gemm_tiles<WDT, KB>(X, b2, b2, false, acc + tt);
        else if (k * 2 == KB * KSTEP) gemm_tiles<WDT, KB / 2>(X, b2, b2, false, acc + tt);
        else gemm_tiles_any<WDT>(X, b2, b2, false, k, acc + tt);
      }
    } else {
      if (k == KB * KSTEP) gemm_tiles<WDT, KB>(X, pre.b, b2, two, acc);
      else if (k * 2 == KB * KSTEP) gemm_tiles<WDT, KB / 2>(X, pre.b, b2, two, acc);
      else gemm_tiles_any<WDT>(X, pre.b, b2, two, k, acc);
    }
  }
  __builtin_amdgcn_sched_barrier(0);                 // the refill of `pre` must not be hoisted above its last use
  if (has_next) gemm_prefetch<WDT>(next, pre);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = wave + tt * RC_NW;
    const int col = t * 16 + m;
    if (t < tiles && col < n) {
      const float bv = bias ? bvs[tt] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[tt][i] + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (Mk) v = Mk[(4 * g + i) * RC_LD + col] > 0.f ? v : 0.f;
        Y[(4 * g + i) * RC_LD + col] = v;
      }
    }
  }
}

// 32 lanes per row: dst = LN(src [+ src2]) * gamma + beta; optional: src <- the sum, stats (mean, rstd) to p2
__device__ __forceinline__ void st_ln(const MbvRowStage& s, float* slots, int row0, int rows, float eps) {
  float* A = slots + s.src * (RC_ROWS * RC_LD);
  const float* B = s.src2 >= 0 ? slots + s.src2 * (RC_ROWS * RC_LD) : nullptr;
  float* Y = slots + s.dst * (RC_ROWS * RC_LD);
  const int r = threadIdx.x >> 5, l = threadIdx.x & 31, n = s.n;
  const G1 float* gamma = gp<float>(s.p0);
  const G1 float* beta = gp<float>(s.p1);
  float v[RC_MAXC / 32], gm[RC_MAXC / 32], bt[RC_MAXC / 32];
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {               // unconditional (clamped) loads, requested before the statistics
    const int c = l + 32 * j, cc = c < n ? c : n - 1;
    gm[j] = gamma[cc];
    bt[j] = beta[cc];
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    float x = 0.f;
    if (c < n) {
      x = A[r * RC_LD + c];
      if (B) x += B[r * RC_LD + c];
    }
    v[j] = x;
    sum += x;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)n;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    if (c < n) { const float d = v[j] - mean; var += d * d; }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
  const float rstd = rsqrtf(var / (float)n + eps);
  const bool save = (s.flags & MBV_RC_SAVE_SUM) != 0;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    if (c < n) {
      if (save) A[r * RC_LD + c] = v[j];
      Y[r * RC_LD + c] = (v[j] - mean) * rstd * gm[j] + bt[j];
    }
  }
  if (s.p2 && l == 0 && row0 + r < rows) {
    G1 float* st = gpw<float>(s.p2) + (int64_t)(row0 + r) * 2;
    st[0] = mean;
    st[1] = rstd;
  }
}

// dst = d(sum) of y = LN(sum) gamma + beta given g = src (dL/dy), sum = src2, stats p2; per-block partial
// d(gamma) / d(beta) rows to p1 [block][2 n] (rows beyond `rows` hold zero gradients: the LOAD stage zero-fills)
__device__ __forceinline__ void st_ln_bwd(const MbvRowStage& s, float* slots, float* red, int row0, int rows, int rb) {
  const float* G = slots + s.src * (RC_ROWS * RC_LD);
  const float* S = slots + s.src2 * (RC_ROWS * RC_LD);
  float* D = slots + s.dst * (RC_ROWS * RC_LD);
  const int r = threadIdx.x >> 5, l = threadIdx.x & 31, n = s.n;
  const G1 float* gamma = gp<float>(s.p0);
  const bool live = row0 + r < rows;
  const G1 float* st = gp<float>(s.p2) + (int64_t)(live ? row0 + r : rows - 1) * 2;
  const float mean = st[0], rstd = st[1];
  float gw[RC_MAXC / 32], xh[RC_MAXC / 32], gr[RC_MAXC / 32], gm[RC_MAXC / 32];
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    gm[j] = gamma[c < n ? c : n - 1];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    gw[j] = xh[j] = gr[j] = 0.f;
    if (c < n && live) {
      const float g = G[r * RC_LD + c];
      xh[j] = (S[r * RC_LD + c] - mean) * rstd;
      gr[j] = g;
      gw[j] = g * gm[j];
      s1 += gw[j];
      s2 += gw[j] * xh[j];
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  const float inv = 1.f / (float)n;
#pragma unroll
  for (int j = 0; j < RC_MAXC / 32; ++j) {
    const int c = l + 32 * j;
    if (c < n) D[r * RC_LD + c] = live ? rstd * (gw[j] - inv * s1 - xh[j] * inv * s2) : 0.f;
  }
  // parameter gradients: column sums over this block's 16 rows through an LDS image [2][16][n] reused from `red`
  if (s.p1) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RC_MAXC / 32; ++j) {
      const int c = l + 32 * j;
      if (c < n) {
        red[r * RC_MAXC + c] = gr[j] * xh[j];
        red[RC_ROWS * RC_MAXC + r * RC_MAXC + c] = gr[j];
      }
    }
    __syncthreads();
    G1 float* out = gpw<float>(s.p1) + (int64_t)rb * 2 * n;
    const int i = threadIdx.x;                                // 2 n <= 512 threads
    if (i < 2 * n) {
      const int which = i / n, c = i - which * n;
      float a = 0.f;
#pragma unroll
      for (int rr = 0; rr < RC_ROWS; ++rr) a += red[which * RC_ROWS * RC_MAXC + rr * RC_MAXC + c];
      out[i] = a;
    }
  }
}

__device__ __forceinline__ void st_add(const MbvRowStage& s, float* slots) {
  const float* A = slots + s.src * (RC_ROWS * RC_LD);
  const float* B = slots + s.src2 * (RC_ROWS * RC_LD);
  float* Y = slots + s.dst * (RC_ROWS * RC_LD);
  const int n4 = s.n >> 2;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = threadIdx.x + u * RC_NT;
    if (i < RC_ROWS * n4) {
      const int r = i / n4, c = (i - r * n4) * 4;
      const float4 a = *reinterpret_cast<const float4*>(__builtin_assume_aligned(A + r * RC_LD + c, 16));
      const float4 b = *reinterpret_cast<const float4*>(__builtin_assume_aligned(B + r * RC_LD + c, 16));
      *reinterpret_cast<float4*>(__builtin_assume_aligned(Y + r * RC_LD + c, 16)) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
  }
}

// p0[block * ld + c] = sum over the block's rows of src[r][c]  (rows beyond `rows` are zero by construction)
__device__ __forceinline__ void st_colsum(const MbvRowStage& s, const float* slots, int rb) {
  const float* A = slots + s.src * (RC_ROWS * RC_LD);
  G1 float* out = gpw<float>(s.p0) + (int64_t)rb * s.ld;
  const int c = threadIdx.x;                                  // n <= 256 < 512 threads
  if (c < s.n) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < RC_ROWS; ++r) a += A[r * RC_LD + c];
    out[c] = a;
  }
}

// ---- MBV_RC_FFN: the two GEMMs of an MLP (forward: h = relu(x W1^T + b1), y = h W2^T; backward: dh = (dy W2) * (h > 0),
// dx = dh W1) with the hidden dimension cut into 256-wide chunks that the eight waves own IN PARALLEL.
//
// As a chain of stages the MLP is 16 dependent GEMM stages, each exposing a global-memory round trip in front of a
// barrier (4 us per stage measured, 64 us for the pair — slower than two library GEMM launches).  Here a wave streams
// its own 2 x 128 KB of weights through a private software pipeline — no barrier, no other wave's schedule to wait
// for: phase 1 produces the chunk's 16 hidden tiles (epilogue in registers, the tile leaves for global memory and, in
// the weight dtype, for the wave's private LDS image), phase 2 multiplies that image with the second weight and adds
// its 16 x 256 partial result into a shared f64 image with LDS atomics (`ds_add_f64`, 9-16 clocks per wave instruction;
// `ds_add_f32` serialises its lanes).  16-bit weights only (the f32 program keeps the staged form).
//   p0 = first weight  (F rows x E: W1, or W2^T in backward), ld;  p1 = its bias (forward) or NULL
//   p2 = second weight (E rows x F: W2, or W1^T in backward), ld2
//   src = input slot (16 x E), dst = output slot (16 x E, f32, without the output bias), src2 = FIRST of the scratch
//   slots (n_chunks_in_flight x 8.3 KB of wave-private bf16 images: 5 slots for eight waves)
//   n = E (<= 256), k = F (multiple of 256);  flags: MBV_RC_MASK = backward (mask by the stored activations)
// The hidden activations / their gradients go to / come from HID = header pointer q (forward: written; backward:
// activations read from `hid_in`, gradients written to `hid_out`), passed through a second, LOAD-typed pseudo stage
// that directly FOLLOWS the FFN stage: p0 = activations (M x F f32), p1 = gradient out (backward) or NULL, p2 = per-block
// column partials of the gradient (backward, (blocks, ld2) f32 at column offset 0) or NULL, ld = F.
template <int WDT>
__device__ __forceinline__ f32x4 ffn_mma(const uint4* a, const uint4* b, f32x4 acc) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    if constexpr (WDT == MBV_DT_BF16)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[kb]), __builtin_bit_cast(bf16x8, b[kb]), acc, 0, 0, 0);
    else
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[kb]), __builtin_bit_cast(f16x8, b[kb]), acc, 0, 0, 0);
  }
  return acc;
}

template <int WDT>
__device__ __forceinline__ void st_ffn(const MbvRowStage& s, const MbvRowStage& io, float* slots, float* red, int row0,
                                       int rows, int rb) {
  if constexpr (WDT == MBV_DT_F32) {
    return;
  } else {
    constexpr int HB = 264;                            // row stride (16-bit elements) of a wave's hidden image: 528 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int E = s.n, F = s.k;
    const bool bwd = (s.flags & MBV_RC_MASK) != 0;
    const float* X = slots + s.src * (RC_ROWS * RC_LD);
    unsigned short* himg = reinterpret_cast<unsigned short*>(slots + s.src2 * (RC_ROWS * RC_LD)) + wave * (RC_ROWS * HB);
    double* acc64 = reinterpret_cast<double*>(red);    // [16][256] f64: 32 KB, the LayerNorm-backward scratch
    for (int i = threadIdx.x; i < RC_ROWS * RC_MAXC; i += RC_NT) acc64[i] = 0.0;
    __syncthreads();                                   // acc64 zeroed
    const G1 char* w1 = gp<char>(s.p0);
    const G1 char* w2 = gp<char>(s.p2);
    const G1 float* bias1 = gp<float>(s.p1);
    const G1 float* hid_in = gp<float>(io.p0);
    G1 float* hid_out = gpw<float>(bwd ? io.p1 : io.p0);
    G1 float* part = gpw<float>(io.p2);
    const int e_tiles = (E + 15) >> 4;
    for (int c = wave; c * 256 < F; c += RC_NW) {
      // ---- phase 1: 16 hidden tiles of this chunk, fragments of tile t + 1 requested before tile t is multiplied
      uint4 bA[8], bB[8];
      const int hrow0 = c * 256;
      // A fragments of the input (all E <= 256 columns; blocks beyond E are zero) — per chunk, so that they are dead
      // in phase 2 (its own A fragments take the registers)
      uint4 ax[8];
      {
        const float4* xr = reinterpret_cast<const float4*>(__builtin_assume_aligned(X + m * RC_LD + 8 * g, 16));
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (kb * 32 < E) {
            const float4 r0 = xr[kb * 8], r1 = xr[kb * 8 + 1];
            v = f32x8{r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          }
          if constexpr (WDT == MBV_DT_BF16) ax[kb] = __builtin_bit_cast(uint4, __builtin_convertvector(v, bf16x8));
          else ax[kb] = __builtin_bit_cast(uint4, __builtin_convertvector(v, f16x8));
        }
      }
      // (k blocks beyond E: the fragment rows are E*2 bytes long; clamp the offset, the A blocks there are zero)
      // both weights in fragment-major layout (see gemm_load_tile): first weight F rows x E (ld = E / 32 blocks per row)
      auto load1 = [&](int t, uint4* b) {
        const G1 char* base = w1 + ((int64_t)(c * 16 + t) * s.ld * 64 + lane) * 16;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) b[kb] = gld16(base + (kb * 32 < E ? kb : 0) * 1024);
      };
      load1(0, bA);
#pragma unroll 1
      for (int t = 0; t < 16; t += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int tt = t + half;
          uint4* cur = half == 0 ? bA : bB;
          uint4* nxt = half == 0 ? bB : bA;
          const int hcol = hrow0 + tt * 16 + m;                  // hidden unit of this lane's column
          // the epilogue's own operands are requested BEFORE the next tile's fragments: loads return in order, so
          // waiting for them must not mean waiting for the fragments issued behind them
          float hv[4] = {0.f, 0.f, 0.f, 0.f};
          if (bwd) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              int row = row0 + 4 * g + i;
              row = row < rows ? row : rows - 1;
              hv[i] = hid_in[(int64_t)row * io.ld + hcol];
            }
          }
          const G1 float* bp1 = bias1 ? bias1 : reinterpret_cast<const G1 float*>(w1);
          float bv = bp1[hcol];
          bv = (!bwd && bias1) ? bv : 0.f;
          if (tt + 1 < 16) load1(tt + 1, nxt);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = ffn_mma<WDT>(ax, cur, acc);
          float colsum = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = acc[i] + bv;
            if (bwd) v = hv[i] > 0.f ? v : 0.f;
            else v = fmaxf(v, 0.f);
            const int row = row0 + 4 * g + i;
            if (row >= rows) v = 0.f;
            colsum += v;
            himg[(4 * g + i) * HB + tt * 16 + m] = f32_to_16(v, WDT);
            if (row < rows) hid_out[(int64_t)row * io.ld + hcol] = v;
          }
          if (bwd && part) {                                     // column partial of d(hidden): sum over the 16 rows
            colsum += __shfl_xor(colsum, 16, 64);
            colsum += __shfl_xor(colsum, 32, 64);
            if (g == 0) part[(int64_t)rb * io.ld2 + hcol] = colsum;
          }
        }
      }
      // ---- phase 2: partial output = (16 x 256 chunk image) x second weight[:, chunk]
      uint4 ah[8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)                             // the wave's own writes: same-wave LDS ordering
        ah[kb] = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(himg + m * HB + kb * 32 + 8 * g, 16));
      // second weight E rows x F (ld2 = F / 32 blocks per row): this chunk is blocks c * 8 .. c * 8 + 7
      auto load2 = [&](int t, uint4* b) {
        const G1 char* base = w2 + (((int64_t)t * s.ld2 + c * 8) * 64 + lane) * 16;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) b[kb] = gld16(base + kb * 1024);
      };
      load2(0, bA);
#pragma unroll 1
      for (int t = 0; t < e_tiles; t += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int tt = t + half;
          if (tt >= e_tiles) break;
          uint4* cur = half == 0 ? bA : bB;
          uint4* nxt = half == 0 ? bB : bA;
          if (tt + 1 < e_tiles) load2(tt + 1, nxt);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = ffn_mma<WDT>(ah, cur, acc);
          const int col = tt * 16 + m;
          if (col < E) {
#pragma unroll
            // columns rotated by 16 per row group: the four groups of a wave instruction fall on different banks
            for (int i = 0; i < 4; ++i) atomicAdd(&acc64[(4 * g + i) * RC_MAXC + ((col + 16 * g) & (RC_MAXC - 1))], (double)acc[i]);
          }
        }
      }
    }
    __syncthreads();
    float* Y = slots + s.dst * (RC_ROWS * RC_LD);
    const G1 float* bias_out = bwd ? nullptr : gp<float>(io.p1);
    for (int i = threadIdx.x; i < RC_ROWS * RC_MAXC; i += RC_NT) {
      const int r = i >> 8, cc = i & 255;
      if (cc < E) Y[r * RC_LD + cc] = (float)acc64[r * RC_MAXC + ((cc + 16 * (r >> 2)) & (RC_MAXC - 1))] + (bias_out ? bias_out[cc] : 0.f);
    }
  }
}

// ---- MBV_RC_SUM: dst = sum over k parts of p0[j * ld2 + row * ld + c] (f32; the partial results that the workgroups of a
// split launch stored with MBV_RC_SPLIT), in part order — the same sum whatever the launch's timing.
__device__ __forceinline__ void st_sum(const MbvRowStage& s, float* slots, int row0, int rows) {
  float* dst = slots + s.dst * (RC_ROWS * RC_LD);
  const int n4 = s.n >> 2;
  const G1 float* base = gp<float>(s.p0);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = threadIdx.x + u * RC_NT;
    if (i >= RC_ROWS * n4) continue;
    const int r = i / n4, c = (i - r * n4) * 4;
    int row = row0 + r;
    const bool live = row < rows;
    row = live ? row : rows - 1;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const G1 float* p = base + (int64_t)row * s.ld + c;
#pragma unroll 8
    for (int j = 0; j < s.k; ++j) {
      const float4 v = __builtin_bit_cast(float4, gld16(p + (int64_t)j * s.ld2));
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (!live) a = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(__builtin_assume_aligned(dst + r * RC_LD + c, 16)) = a;
  }
}

// ---- MBV_RC_FFN | MBV_RC_SLICE: the workgroup `slice` of a split launch owns the 256 hidden units [256 slice, 256 slice
// + 256) — wave w the 32 units behind 256 slice + 32 w — and leaves its PARTIAL output in dst (slice 0: plus the output
// bias).  A wave's share is two 16-unit tiles of the first product (K = E) and one 32-deep k block of the second:
// 2 x 8 + 16 fragment loads, ALL requested before the first multiply — one memory round trip per workgroup where the
// whole-MLP form walks 32 dependent tiles per wave (21 us for 2048 hidden units; the other 231 CUs idle meanwhile).
// Measured: 13.6 us for the stage — 5.9 us to get the 33 loads of every wave issued, 4.3 us more until the last fragment
// has arrived: a CU draws ~25 GB/s of weights from L2 whatever the layout (k-major or strided made no difference), and
// the 25 row blocks x 2 MB of the layer are 4.8 TB/s in aggregate.  Fewer bytes would need more rows per workgroup.
template <int WDT>
__device__ __forceinline__ void st_ffn_slice(const MbvRowStage& s, const MbvRowStage& io, float* slots, float* red, int row0,
                                             int rows, int rb, int slice) {
  if constexpr (WDT == MBV_DT_F32) {
    return;
  } else {
    constexpr int HB = 40;                             // row stride (16-bit elements) of a wave's 16 x 32 hidden image: 80 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int E = s.n;
    const bool bwd = (s.flags & MBV_RC_MASK) != 0;
    const float* X = slots + s.src * (RC_ROWS * RC_LD);
    unsigned short* himg = reinterpret_cast<unsigned short*>(slots + s.src2 * (RC_ROWS * RC_LD)) + wave * (RC_ROWS * HB);
    double* acc64 = reinterpret_cast<double*>(red);
    for (int i = threadIdx.x; i < RC_ROWS * RC_MAXC; i += RC_NT) acc64[i] = 0.0;
    const G1 char* w1 = gp<char>(s.p0);
    const G1 char* w2 = gp<char>(s.p2);
    const G1 float* bias1 = gp<float>(s.p1);
    const G1 float* hid_in = gp<float>(io.p0);
    G1 float* hid_out = gpw<float>(bwd ? io.p1 : io.p0);
    G1 float* part = gpw<float>(io.p2);
    const int e_tiles = (E + 15) >> 4;
    const int h0 = slice * 256 + wave * 32;            // the wave's first hidden unit
    // the epilogue's operands first (loads return in order), then every weight fragment of the wave
    float hv[2][4], bv[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int hcol = h0 + tt * 16 + m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = row0 + 4 * g + i;
        row = row < rows ? row : rows - 1;
        hv[tt][i] = bwd ? hid_in[(int64_t)row * io.ld + hcol] : 0.f;
      }
      const G1 float* bp1 = bias1 ? bias1 : reinterpret_cast<const G1 float*>(w1);
      const float b = bp1[hcol];
      bv[tt] = (!bwd && bias1) ? b : 0.f;
    }
    uint4 b1[2][8], b2[16];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const G1 char* base = w1 + ((int64_t)((h0 >> 4) + tt) * s.ld * 64 + lane) * 16;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) b1[tt][kb] = gld16(base + (kb * 32 < E ? kb : 0) * 1024);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int tc = t < e_tiles ? t : e_tiles - 1;
      b2[t] = gld16(w2 + (((int64_t)(h0 >> 5) * e_tiles + tc) * 64 + lane) * 16);       // k-major copy (E / 16 == e_tiles tiles per k block)
    }
    uint4 ax[8];
    {
      const float4* xr = reinterpret_cast<const float4*>(__builtin_assume_aligned(X + m * RC_LD + 8 * g, 16));
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (kb * 32 < E) {
          const float4 r0 = xr[kb * 8], r1 = xr[kb * 8 + 1];
          v = f32x8{r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        }
        if constexpr (WDT == MBV_DT_BF16) ax[kb] = __builtin_bit_cast(uint4, __builtin_convertvector(v, bf16x8));
        else ax[kb] = __builtin_bit_cast(uint4, __builtin_convertvector(v, f16x8));
      }
    }
    __syncthreads();                                   // acc64 zeroed
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int hcol = h0 + tt * 16 + m;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = ffn_mma<WDT>(ax, b1[tt], acc);
      float colsum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[i] + bv[tt];
        if (bwd) v = hv[tt][i] > 0.f ? v : 0.f;
        else v = fmaxf(v, 0.f);
        const int row = row0 + 4 * g + i;
        if (row >= rows) v = 0.f;
        colsum += v;
        himg[(4 * g + i) * HB + tt * 16 + m] = f32_to_16(v, WDT);
        if (row < rows) hid_out[(int64_t)row * io.ld + hcol] = v;
      }
      if (bwd && part) {
        colsum += __shfl_xor(colsum, 16, 64);
        colsum += __shfl_xor(colsum, 32, 64);
        if (g == 0) part[(int64_t)rb * io.ld2 + hcol] = colsum;
      }
    }
    // the wave's own LDS writes: same-wave ordering
    const uint4 ah = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(himg + m * HB + 8 * g, 16));
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t >= e_tiles) break;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (WDT == MBV_DT_BF16)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, b2[t]), acc, 0, 0, 0);
      else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b2[t]), acc, 0, 0, 0);
      const int col = t * 16 + m;
      if (col < E) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&acc64[(4 * g + i) * RC_MAXC + ((col + 16 * g) & (RC_MAXC - 1))], (double)acc[i]);
      }
    }
    __syncthreads();
    float* Y = slots + s.dst * (RC_ROWS * RC_LD);
    const G1 float* bias_out = (bwd || slice != 0) ? nullptr : gp<float>(io.p1);
    for (int i = threadIdx.x; i < RC_ROWS * RC_MAXC; i += RC_NT) {
      const int r = i >> 8, cc = i & 255;
      if (cc < E) Y[r * RC_LD + cc] = (float)acc64[r * RC_MAXC + ((cc + 16 * (r >> 2)) & (RC_MAXC - 1))] + (bias_out ? bias_out[cc] : 0.f);
    }
  }
}

// barrier between stages: LDS traffic only (the prefetched global loads stay in flight across it)
__device__ __forceinline__ void stage_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The program arrives in the kernel-argument segment, which lives in host-visible memory: read stage by stage with
// scalar loads, EVERY stage paid a fresh miss of several microseconds (measured: 3.3 us per stage whatever its type,
// 150 us for the 44-stage FFN chain).  The workgroup therefore copies the whole program into LDS with one round of
// vector loads and decodes each stage from there (broadcast LDS reads + v_readfirstlane: the fields stay scalar).
constexpr int RC_PROG_WORDS = (int)(sizeof(RowProgram) / 4);
constexpr int RC_STAGE_WORDS = (int)(sizeof(MbvRowStage) / 4), RC_HEAD_WORDS = 8;
static_assert(sizeof(MbvRowStage) == 56 && sizeof(RowProgram) == 32 + RC_MAX_STAGES * 56, "program layout");

__device__ __forceinline__ MbvRowStage rc_stage(const uint32_t* sprog, int i) {
  uint32_t w[RC_STAGE_WORDS];
#pragma unroll
  for (int j = 0; j < RC_STAGE_WORDS; ++j)
    w[j] = __builtin_amdgcn_readfirstlane(sprog[RC_HEAD_WORDS + i * RC_STAGE_WORDS + j]);
  MbvRowStage s;
  __builtin_memcpy(&s, w, sizeof(s));
  return s;
}

template <int WDT>
__global__ void __launch_bounds__(RC_NT) k_rowchain(const RowProgram P_unused) {
  extern __shared__ __attribute__((aligned(16))) float rc_lds[];
  __shared__ uint32_t sprog[RC_PROG_WORDS];
  float* slots = rc_lds;
  float* red = rc_lds + RC_SLOTS * RC_ROWS * RC_LD;       // [2][16][256] scratch of the LayerNorm backward
  {
    const __attribute__((address_space(4))) uint32_t* kp =
        (const __attribute__((address_space(4))) uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int w = threadIdx.x; w < RC_PROG_WORDS; w += RC_NT) sprog[w] = kp[w];
  }
  __syncthreads();
  const int num_stages = __builtin_amdgcn_readfirstlane(sprog[0]);
  const int rows = __builtin_amdgcn_readfirstlane(sprog[1]);
  const int q_mod = __builtin_amdgcn_readfirstlane(sprog[2]);
  const float eps = __uint_as_float(__builtin_amdgcn_readfirstlane(sprog[4]));
  int ng = (int)__builtin_amdgcn_readfirstlane(sprog[5]);        // first GEMM / first LOAD stage (host-computed, -1: none)
  int nl = (int)__builtin_amdgcn_readfirstlane(sprog[6]);
  // split launches: `split` consecutive workgroups share a row block.  Stages without an owner (flags bits 8..15 == 0)
  // run in all of them — redundantly, on CUs that would idle otherwise — the others only in workgroup owner - 1.
  const int split = (int)__builtin_amdgcn_readfirstlane(sprog[7]);
  const int rb = (int)blockIdx.x / split, sidx = (int)blockIdx.x - rb * split;
  const int row0 = rb * RC_ROWS;
  GemmPre<WDT> gpre;
  LoadPre lpre;
  if (nl >= 0) {
    const MbvRowStage nx = rc_stage(sprog, nl);
    load_prefetch(nx, lpre, row0, rows, q_mod);
  }
  if (ng >= 0) {
    const MbvRowStage nx = rc_stage(sprog, ng);
    gemm_prefetch<WDT>(nx, gpre);
  }
  for (int i = 0; i < num_stages; ++i) {
    const MbvRowStage s = rc_stage(sprog, i);
    // `reserved`: the next GEMM stage (low byte) and the next LOAD stage (high byte) after this one, 0xff = none
    const int nxt_g = s.reserved & 0xff, nxt_l = (s.reserved >> 8) & 0xff;
    const int owner = (s.flags >> 8) & 0xff;
    if (owner != 0 && owner - 1 != sidx) {
      // another workgroup's stage: only keep the prefetch chain consistent (the operands in flight are this stage's)
      if (s.op == MBV_RC_LOAD && nxt_l != 0xff) {
        const MbvRowStage nx = rc_stage(sprog, nxt_l);
        load_prefetch(nx, lpre, row0, rows, q_mod);
      } else if (s.op == MBV_RC_GEMM && nxt_g != 0xff) {
        const MbvRowStage nx = rc_stage(sprog, nxt_g);
        gemm_prefetch<WDT>(nx, gpre);
      }
      continue;
    }
    switch (s.op) {
      case MBV_RC_LOAD:
        st_load(s, slots, lpre, row0, rows);
        __builtin_amdgcn_sched_barrier(0);                   // refill `lpre` only after its last use
        if (nxt_l != 0xff) {
          const MbvRowStage nx = rc_stage(sprog, nxt_l);
          load_prefetch(nx, lpre, row0, rows, q_mod);
        }
        break;
      case MBV_RC_STORE: st_store(s, slots, row0, rows, sidx); break;
      case MBV_RC_SUM: st_sum(s, slots, row0, rows); break;
      case MBV_RC_GEMM: {
        const MbvRowStage nx = rc_stage(sprog, nxt_g != 0xff ? nxt_g : i);
        st_gemm<WDT>(s, nxt_g != 0xff, nx, slots, gpre);
        break;
      }
      case MBV_RC_LN: st_ln(s, slots, row0, rows, eps); break;
      case MBV_RC_LN_BWD: st_ln_bwd(s, slots, red, row0, rows, rb); break;
      case MBV_RC_ADD: st_add(s, slots); break;
      case MBV_RC_COLSUM: st_colsum(s, slots, rb); break;
      case MBV_RC_FFN: {
        // the prefetch registers are dead across this stage (the host routes the prefetches AROUND an FFN stage: the
        // stage before it requests nothing, this stage requests the next GEMM / LOAD operands when it is done)
        {
          GemmPre<WDT> undef_g;
          LoadPre undef_l;
          gpre = undef_g;
          lpre = undef_l;
        }
        const MbvRowStage io = rc_stage(sprog, i + 1);
        if (s.flags & MBV_RC_SLICE) st_ffn_slice<WDT>(s, io, slots, red, row0, rows, rb, sidx);
        else st_ffn<WDT>(s, io, slots, red, row0, rows, rb);
        if (nxt_l != 0xff) {
          const MbvRowStage nx = rc_stage(sprog, nxt_l);
          load_prefetch(nx, lpre, row0, rows, q_mod);
        }
        if (nxt_g != 0xff) {
          const MbvRowStage nx = rc_stage(sprog, nxt_g);
          gemm_prefetch<WDT>(nx, gpre);
        }
        ++i;                                               // the descriptor stage that follows is consumed here
        break;
      }
      default: break;
    }
    stage_barrier();
  }
}

// ---- grouped 2-D transpose (the decoder's weight matrices, once per step) ----------------------------------------
struct TrEntry {
  const void* src;
  void* dst;
  int rows, cols;          // src is (rows, cols) row-major, dst (cols, rows)
  int tile_begin;          // first 64 x 64 tile of this entry in the launch
  int tiles_c;
};
struct TrArgs {
  int n, esize;
  TrEntry e[MBV_TR_MAX];
};

template <typename T>
__global__ void __launch_bounds__(256) k_transpose_group(const TrArgs A) {
  __shared__ T tile[64][65];
  int ei = 0;
  for (int i = 1; i < A.n; ++i)
    if ((int)blockIdx.x >= A.e[i].tile_begin) ei = i;
  const TrEntry& e = A.e[ei];
  const int t = (int)blockIdx.x - e.tile_begin;
  const int tr = t / e.tiles_c, tc = t - tr * e.tiles_c;
  const T* src = reinterpret_cast<const T*>(e.src);
  T* dst = reinterpret_cast<T*>(e.dst);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = tr * 64 + j, c = tc * 64 + tx;
    if (r < e.rows && c < e.cols) tile[j][tx] = src[(int64_t)r * e.cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = tc * 64 + j, r = tr * 64 + tx;
    if (r < e.rows && c < e.cols) dst[(int64_t)c * e.rows + r] = tile[tx][j];
  }
}

// ---- fragment-major weight copies (16-bit): dst[((t * KBN + kb) * 64 + lane) * 8 + j] = W[t * 16 + lane % 16][kb * 32 +
// 8 * (lane / 16) + j] for the logical (rows x cols) matrix W; `transposed`: W[r][c] is read from src[c * ld + r] (the
// data-gradient operand W^T of a row-major src), else from src[r * ld + c].  Rows beyond `rows` are zero.
struct FragEntry {
  const void* src;
  void* dst;
  int rows, cols, ld, mode, tile_begin;       // mode: bit 0 = transposed, bit 1 = 16-byte loads allowed
};
struct FragArgs {
  int n;
  FragEntry e[MBV_TR_MAX];
};

// One workgroup = one 64 x 64 tile of the logical matrix (4 fragment rows x 2 k blocks), staged through LDS so that BOTH
// orientations read full 128-byte lines of src (the first version gathered the transposed operand as eight 2-byte reads
// per lane, 32 contiguous bytes per 16 lanes: 104 us per step for the decoder's 130 copies).
__global__ void __launch_bounds__(256) k_fragment_group(const FragArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];
  const int tid = threadIdx.x;
  int ei = 0;
  for (int i = 1; i < A.n; ++i)
    if ((int)blockIdx.x >= A.e[i].tile_begin) ei = i;
  const FragEntry e = A.e[ei];
  const int tl = blockIdx.x - e.tile_begin;
  const int tiles_c = (e.cols + 63) >> 6;
  const int tr = tl / tiles_c, tc = tl - tr * tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const unsigned short* src = reinterpret_cast<const unsigned short*>(e.src);
  const bool transposed = e.mode & 1, vec = e.mode & 2;
  const int line = tid >> 3, piece = (tid & 7) * 8;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int a = line + 32 * it;                      // the src row within the tile
    unsigned short v[8];
    if (!transposed) {
      const int r = r0 + a, c = c0 + piece;
      const bool in = r < e.rows && c < e.cols;        // cols % 32 == 0: a piece is inside or outside as a whole
      if (in && vec) {
        const uint4 q = *reinterpret_cast<const uint4*>(src + (int64_t)r * e.ld + c);
        *reinterpret_cast<uint4*>(&tile[a][piece]) = q;
        continue;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = in ? src[(int64_t)r * e.ld + c + j] : (unsigned short)0;
      uint4 q;
      q.x = v[0] | ((unsigned)v[1] << 16); q.y = v[2] | ((unsigned)v[3] << 16);
      q.z = v[4] | ((unsigned)v[5] << 16); q.w = v[6] | ((unsigned)v[7] << 16);
      *reinterpret_cast<uint4*>(&tile[a][piece]) = q;
    } else {
      const int c = c0 + a, r = r0 + piece;            // W[r + j][c] = src[c * ld + r + j]
      if (c < e.cols && r + 7 < e.rows && vec) {
        const uint4 q = *reinterpret_cast<const uint4*>(src + (int64_t)c * e.ld + r);
        v[0] = q.x & 0xffff; v[1] = q.x >> 16; v[2] = q.y & 0xffff; v[3] = q.y >> 16;
        v[4] = q.z & 0xffff; v[5] = q.z >> 16; v[6] = q.w & 0xffff; v[7] = q.w >> 16;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c < e.cols && r + j < e.rows) ? src[(int64_t)c * e.ld + r + j] : (unsigned short)0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[piece + j][a] = v[j];
    }
  }
  __syncthreads();
  const int lane = tid & 63, kbn = e.cols >> 5, tn = (e.rows + 15) >> 4;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int f = (tid >> 6) + 4 * it, tt = f & 3, kbb = f >> 2;
    const int t = tr * 4 + tt, kb = tc * 2 + kbb;
    if (t >= tn || kb >= kbn) continue;
    const uint4 q = *reinterpret_cast<const uint4*>(&tile[tt * 16 + (lane & 15)][kbb * 32 + 8 * (lane >> 4)]);
    // mode bit 2: k-major — the 16-row tiles of ONE k block are consecutive (the second weight of a sliced MLP stage: a
    // wave multiplies one k block against every tile, 16 KB in one piece instead of sixteen at a 64 KB stride)
    const int64_t blk = (e.mode & 4) ? (int64_t)kb * tn + t : (int64_t)t * kbn + kb;
    reinterpret_cast<uint4*>(e.dst)[blk * 64 + lane] = q;
  }
}

}  // namespace

extern "C" int mbv_rowchain_max_stages(void) { return RC_MAX_STAGES; }
extern "C" int mbv_rowchain_slots(void) { return RC_SLOTS; }

extern "C" int mbv_rowchain_run(const MbvRowStage* stages, int32_t num_stages, int32_t rows, int32_t q_mod, float eps,
                                int32_t wdtype, void* stream_) {
  return mbv_rowchain_run_split(stages, num_stages, rows, q_mod, eps, wdtype, 1, stream_);
}

extern "C" int mbv_rowchain_run_split(const MbvRowStage* stages, int32_t num_stages, int32_t rows, int32_t q_mod, float eps,
                                      int32_t wdtype, int32_t split, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!stages || num_stages <= 0 || num_stages > RC_MAX_STAGES || rows <= 0 || q_mod < 0 || split < 1 || split > 64)
    return MBV_ERR_BAD_ARG;
  if (wdtype != MBV_DT_F32 && wdtype != MBV_DT_BF16 && wdtype != MBV_DT_F16) return MBV_ERR_BAD_ARG;
  RowProgram P;
  P.num_stages = num_stages; P.rows = rows; P.q_mod = q_mod; P.wdtype = wdtype; P.eps = eps;
  P.pad[0] = P.pad[1] = -1;                            // first GEMM / first LOAD stage, filled below
  P.pad[2] = split;
  const int kq = wdtype == MBV_DT_F32 ? 16 : 32;          // reduction granularity of the MFMA k blocks
  const int wes = wdtype == MBV_DT_F32 ? 4 : 2;
  for (int i = 0; i < num_stages; ++i) {
    const MbvRowStage& s = stages[i];
    const int dt = s.flags & 3;
    auto slot_ok = [](int v) { return v >= 0 && v < RC_SLOTS; };
    if (((s.flags >> 8) & 0xff) > split) return MBV_ERR_BAD_ARG;            // owned by a workgroup the launch does not have
    switch (s.op) {
      case MBV_RC_LOAD:
        if (!slot_ok(s.dst) || (!s.p0 && (!slot_ok(s.src) || !s.p1)) || s.n <= 0 || s.n > RC_MAXC || (s.n & 3) || dt > 2 ||
            (s.p0 && ((s.ld & 3) || (reinterpret_cast<size_t>(s.p0) & (dt == MBV_DT_F32 ? 15 : 7)))) ||
            (s.p1 && ((reinterpret_cast<size_t>(s.p1) & 15) || (s.ld2 & 3))))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_STORE:
        if (!slot_ok(s.src) || !s.p0 || s.n <= 0 || s.n > RC_MAXC || dt > 2 || ((s.flags & MBV_RC_ACCUM) && dt != MBV_DT_F32))
          return MBV_ERR_BAD_ARG;
        if ((s.n & 3) == 0 && ((s.ld & 3) || (reinterpret_cast<size_t>(s.p0) & (dt == MBV_DT_F32 ? 15 : 7))))
          return MBV_ERR_BAD_ARG;
        if ((s.flags & MBV_RC_SPLIT) && (s.ld2 <= 0 || (s.ld2 & 3) || (s.flags & MBV_RC_ACCUM))) return MBV_ERR_BAD_ARG;
        // an unowned accumulating store of a split launch would be added once per workgroup of the row block (and race)
        if ((s.flags & MBV_RC_ACCUM) && split > 1 && ((s.flags >> 8) & 0xff) == 0) return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_SUM:
        if (!slot_ok(s.dst) || !s.p0 || s.n <= 0 || s.n > RC_MAXC || (s.n & 3) || s.k < 1 || s.k > 64 || (s.ld & 3) ||
            (s.ld2 & 3) || (reinterpret_cast<size_t>(s.p0) & 15))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_GEMM:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || s.dst == s.src || !s.p0 || s.n <= 0 || s.n > RC_MAXC || s.k <= 0 ||
            s.k > RC_MAXC || (s.k % kq) || (!(s.flags & MBV_RC_FRAG) && ((int64_t)s.ld * wes) % 16) ||
            ((s.flags & MBV_RC_FRAG) && (wdtype == MBV_DT_F32 || s.ld * 32 < s.k)) || (reinterpret_cast<size_t>(s.p0) & 15) ||
            ((s.flags & MBV_RC_MASK) && (!slot_ok(s.src2) || s.src2 == s.dst)))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_LN:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || (s.src2 >= 0 && !slot_ok(s.src2)) || !s.p0 || !s.p1 || s.n <= 0 ||
            s.n > RC_MAXC)
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_LN_BWD:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || !slot_ok(s.src2) || !s.p0 || !s.p2 || s.n <= 0 || s.n > RC_MAXC)
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_ADD:
        if (!slot_ok(s.dst) || !slot_ok(s.src) || !slot_ok(s.src2) || s.n <= 0 || s.n > RC_MAXC || (s.n & 3))
          return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_COLSUM:
        if (!slot_ok(s.src) || !s.p0 || s.n <= 0 || s.n > RC_MAXC || s.ld < s.n) return MBV_ERR_BAD_ARG;
        break;
      case MBV_RC_FFN: {
        // [FFN stage][io descriptor]: 16-bit weights; scratch slots src2 .. src2 + 4 (8 wave images of 16 x 264 x 2 B)
        if (wdtype == MBV_DT_F32 || i + 1 >= num_stages || stages[i + 1].op != MBV_RC_FFN_IO) return MBV_ERR_BAD_ARG;
        const MbvRowStage& io = stages[i + 1];
        const bool bwd = (s.flags & MBV_RC_MASK) != 0;
        if (!slot_ok(s.dst) || !slot_ok(s.src) || !slot_ok(s.src2) || s.src2 + 5 > RC_SLOTS || s.dst == s.src ||
            (s.src >= s.src2 && s.src < s.src2 + 5) || (s.dst >= s.src2 && s.dst < s.src2 + 5) || !s.p0 || !s.p2 ||
            s.n <= 0 || s.n > RC_MAXC || (s.n % 32) || s.k <= 0 || (s.k % 256) || s.ld * 32 < s.n || s.ld2 * 32 < s.k || (reinterpret_cast<size_t>(s.p0) & 15) || (reinterpret_cast<size_t>(s.p2) & 15) || !io.p0 ||
            io.ld < s.k || (bwd && !io.p1) || (io.p2 && io.ld2 < s.k))
          return MBV_ERR_BAD_ARG;
        if (((s.flags >> 8) & 0xff) || ((s.flags & MBV_RC_SLICE) && s.k != 256 * split)) return MBV_ERR_BAD_ARG;   // (MLP stages have no owner)
        P.st[i] = s;
        ++i;
        P.st[i] = stages[i];
        continue;
      }
      case MBV_RC_FFN_IO:
        return MBV_ERR_BAD_ARG;                            // only valid directly behind an FFN stage (consumed above)
      default:
        return MBV_ERR_BAD_ARG;
    }
    P.st[i] = s;
  }
  // stage i's `reserved`: the next GEMM (low byte) / LOAD (high byte) stage after i, 0xff = none; header: the first ones
  int next_g = 0xff, next_l = 0xff;
  for (int i = num_stages - 1; i >= 0; --i) {
    P.st[i].reserved = (int16_t)(next_g | (next_l << 8));
    if (P.st[i].op == MBV_RC_GEMM) next_g = i;
    if (P.st[i].op == MBV_RC_LOAD) next_l = i;
    if (P.st[i].op == MBV_RC_FFN) next_g = next_l = 0xff;  // nothing is prefetched across an FFN stage (it re-arms both)
  }
  P.pad[0] = next_g == 0xff ? -1 : next_g;
  P.pad[1] = next_l == 0xff ? -1 : next_l;
  const size_t lds = (size_t)(RC_SLOTS * RC_ROWS * RC_LD + 2 * RC_ROWS * RC_MAXC) * sizeof(float);
  static bool attr_done = false;       // idempotent attribute of the code objects, not library state
  if (!attr_done) {
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowchain<MBV_DT_F32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowchain<MBV_DT_BF16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MBV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowchain<MBV_DT_F16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const dim3 grid((unsigned)((rows + RC_ROWS - 1) / RC_ROWS) * (unsigned)split), block(RC_NT);
  if (wdtype == MBV_DT_F32)
    hipLaunchKernelGGL(k_rowchain<MBV_DT_F32>, grid, block, lds, stream, P);
  else if (wdtype == MBV_DT_BF16)
    hipLaunchKernelGGL(k_rowchain<MBV_DT_BF16>, grid, block, lds, stream, P);
  else
    hipLaunchKernelGGL(k_rowchain<MBV_DT_F16>, grid, block, lds, stream, P);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}

extern "C" int mbv_fragment_group(const void* const* src, void* const* dst, const int32_t* rows, const int32_t* cols,
                                  const int32_t* ld, const int32_t* transposed, int32_t n, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !dst || !rows || !cols || !ld || !transposed || n < 0) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < n; base += MBV_TR_MAX) {
    FragArgs A;
    A.n = n - base < MBV_TR_MAX ? n - base : MBV_TR_MAX;
    int tiles = 0;
    for (int i = 0; i < A.n; ++i) {
      const int j = base + i;
      if (!src[j] || !dst[j] || rows[j] <= 0 || cols[j] <= 0 || (cols[j] % 32) || ld[j] <= 0 ||
          (reinterpret_cast<size_t>(dst[j]) & 15) || (reinterpret_cast<size_t>(src[j]) & 1))
        return MBV_ERR_BAD_ARG;
      A.e[i].src = src[j]; A.e[i].dst = dst[j]; A.e[i].rows = rows[j]; A.e[i].cols = cols[j]; A.e[i].ld = ld[j];
      const bool vec = (ld[j] % 8) == 0 && (reinterpret_cast<size_t>(src[j]) & 15) == 0;
      A.e[i].mode = ((transposed[j] & 1) ? 1 : 0) | (vec ? 2 : 0) | ((transposed[j] & 2) ? 4 : 0);
      A.e[i].tile_begin = tiles;
      tiles += ((rows[j] + 63) / 64) * ((cols[j] + 63) / 64);
    }
    hipLaunchKernelGGL(k_fragment_group, dim3((unsigned)tiles), dim3(256), 0, stream, A);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

// ---- grouped contiguous copies: the pieces of several small concatenations in one launch -----------------------------------
namespace {
constexpr int MBV_CP_MAX = 48;
constexpr int MBV_CP_CHUNK = 16384;          // bytes per workgroup
struct CpEntry {
  const void* src;
  void* dst;
  long bytes;
  int block_begin;
};
struct CpArgs {
  int n;
  CpEntry e[MBV_CP_MAX];
};
__global__ void __launch_bounds__(256) k_copy_group(const CpArgs A) {
  int ei = 0;
  for (int i = 1; i < A.n; ++i)
    if ((int)blockIdx.x >= A.e[i].block_begin) ei = i;
  const CpEntry& e = A.e[ei];
  const long off = (long)((int)blockIdx.x - e.block_begin) * MBV_CP_CHUNK;
  const long len = e.bytes - off < MBV_CP_CHUNK ? e.bytes - off : MBV_CP_CHUNK;
  const char* s = reinterpret_cast<const char*>(e.src) + off;
  char* d = reinterpret_cast<char*>(e.dst) + off;
  if (((reinterpret_cast<size_t>(s) | reinterpret_cast<size_t>(d)) & 15) == 0) {
    const long n16 = len >> 4;
    for (long i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (long i = (n16 << 4) + threadIdx.x; i < len; i += 256) d[i] = s[i];
  } else {
    for (long i = threadIdx.x; i < len; i += 256) d[i] = s[i];
  }
}
}  // namespace

extern "C" int mbv_copy_group(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !dst || !bytes || n < 0) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < n; base += MBV_CP_MAX) {
    CpArgs A;
    A.n = n - base < MBV_CP_MAX ? n - base : MBV_CP_MAX;
    long blocks = 0;
    for (int i = 0; i < A.n; ++i) {
      const int j = base + i;
      if (!src[j] || !dst[j] || bytes[j] <= 0) return MBV_ERR_BAD_ARG;
      A.e[i].src = src[j]; A.e[i].dst = dst[j]; A.e[i].bytes = bytes[j];
      A.e[i].block_begin = (int)blocks;
      blocks += (bytes[j] + MBV_CP_CHUNK - 1) / MBV_CP_CHUNK;
      if (blocks > 0x7fffffffL) return MBV_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_copy_group, dim3((unsigned)blocks), dim3(256), 0, stream, A);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}

extern "C" int mbv_transpose_group(const void* const* src, void* const* dst, const int32_t* rows, const int32_t* cols,
                                   int32_t n, int32_t elem_size, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !dst || !rows || !cols || n < 0 || (elem_size != 2 && elem_size != 4)) return MBV_ERR_BAD_ARG;
  for (int base = 0; base < n; base += MBV_TR_MAX) {
    TrArgs A;
    A.n = n - base < MBV_TR_MAX ? n - base : MBV_TR_MAX;
    A.esize = elem_size;
    int tiles = 0;
    for (int i = 0; i < A.n; ++i) {
      const int j = base + i;
      if (!src[j] || !dst[j] || rows[j] <= 0 || cols[j] <= 0) return MBV_ERR_BAD_ARG;
      A.e[i].src = src[j]; A.e[i].dst = dst[j]; A.e[i].rows = rows[j]; A.e[i].cols = cols[j];
      A.e[i].tile_begin = tiles;
      A.e[i].tiles_c = (cols[j] + 63) / 64;
      tiles += A.e[i].tiles_c * ((rows[j] + 63) / 64);
    }
    if (elem_size == 2)
      hipLaunchKernelGGL(k_transpose_group<unsigned short>, dim3((unsigned)tiles), dim3(256), 0, stream, A);
    else
      hipLaunchKernelGGL(k_transpose_group<float>, dim3((unsigned)tiles), dim3(256), 0, stream, A);
    MBV_CHECK_LAUNCH();
  }
  return MBV_OK;
}
