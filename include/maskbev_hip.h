/*
 * libmaskbev_hip.so — C ABI of the MI355X (gfx950) hot-path kernels of mask_bev_amd.
 *
 * Conventions (SURVEY.md §8b "Native (C-ABI) layer"):
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - the caller owns every buffer, including `workspace`; the library never allocates, frees or
 *     retains pointers, keeps no mutable global state and is re-entrant;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no hidden device sync;
 *   - return value: 0 = success, < 0 = an `mbv_status` code: bad argument, > 0 = hipError_t of a failed launch;
 *   - no exceptions, no abort, no stdout.
 *
 * Each entry point cites the reference interface it replaces (file:line under the reference tree).
 */
#ifndef MASKBEV_HIP_H_
#define MASKBEV_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MBV_OK = 0,
  MBV_ERR_BAD_ARG = -1,
  MBV_ERR_WORKSPACE = -2,
  MBV_ERR_UNSUPPORTED = -3
} mbv_status;

/* Library/ABI version; bumped whenever a signature below changes. */
int mbv_abi_version(void);

/* Storage type of an activation tensor.  Every `is_bf16` / `*_bf16` / `*_dtype` flag below takes one of these (the
 * flags were 0 / 1 before fp16 existed, hence their names): 2 selects IEEE half, computed by the same kernels
 * instantiated for `_Float16` (v_mfma_f32_32x32x16_f16; conversions round to nearest even, overflow -> inf).
 * Accumulation, softmax, LayerNorm statistics and every gradient of a parameter stay f32 for all three. */
#define MBV_DT_F32 0
#define MBV_DT_BF16 1
#define MBV_DT_F16 2

/* ------------------------------------------------------------------------------------------------
 * K1 — range filter + hard voxelisation of a batch of scans.
 * Replaces: MaskBevEncoder._filter_in_range (mask_bev/models/encoders/mask_bev_encoders.py:113-117)
 *           + MaskBevEncoder.voxelize (:95-111) → mmcv.ops.Voxelization (:69,100; mmcv 2.0.0
 *           hard_voxelize_forward, deterministic).
 * Semantics (bit-exact with the CPU reference): per point c = floor((p - min) / vs) in f32 with IEEE
 * division; pillar ids in order of FIRST APPEARANCE in the scan's point list; a pillar keeps its first
 * `max_points` points in input order; pillars beyond `max_voxels` per scan are dropped.
 *
 * points        (total_points, point_dim) f32, scans concatenated
 * scan_offsets  (batch + 1) i32, scan b owns points [scan_offsets[b], scan_offsets[b+1])
 * range/vs/grid x,y,z bounds (already rounded to f32), voxel sizes, grid sizes
 * prefilter     1 → apply the strict `min < p < max` test of :113-117 first
 * pillar_capacity  rows available in the outputs (>= min(total_points, batch*max_voxels))
 * coors         (pillar_capacity, 4) i32  (b, z, y, x)
 * num_points    (pillar_capacity) i32
 * pillar_points (pillar_capacity, max_points) i32: index into `points` of each kept point, -1 padded
 * row_start     (pillar_capacity + 1) i32: exclusive prefix sum of num_points (compact row of slot 0)
 * cell_to_pillar (batch, gz*gy*gx) i32: pillar id of every BEV cell, -1 if empty
 * counts        (batch + 2) i32: pillars per scan, then total pillars V, then total kept points K
 */
size_t mbv_voxelize_workspace_bytes(int64_t total_points, int32_t batch, int64_t cells_per_scan);

int mbv_voxelize(const float* points, int32_t point_dim, int64_t total_points,
                 const int32_t* scan_offsets, int32_t batch,
                 float x_min, float y_min, float z_min, float x_max, float y_max, float z_max,
                 float vx, float vy, float vz, int32_t gx, int32_t gy, int32_t gz,
                 int32_t prefilter, int32_t max_points, int32_t max_voxels, int64_t pillar_capacity,
                 int32_t* coors, int32_t* num_points, int32_t* pillar_points, int32_t* row_start,
                 int32_t* cell_to_pillar, int32_t* counts,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Dense (V, max_points, point_dim) zero-padded voxel tensor, i.e. the first return value of
 * mmcv.ops.Voxelization as exposed by MaskBevEncoder.voxelize (mask_bev_encoders.py:95-111). */
int mbv_gather_voxels(const float* points, int32_t point_dim, const int32_t* pillar_points,
                      int64_t num_pillars, int32_t max_points, float* voxels, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2a — pillar decoration over REAL points only (no zero padding).
 * Replaces the input stage of mmdet3d PillarFeatureNet.forward(legacy=True, with_distance=True)
 * reached from MaskBevEncoder.encode (mask_bev_encoders.py:70-72,119-120).
 * rows  (K, point_dim + 7) f32 : [fc_x, fc_y, fc_z, extra.., cl_x, cl_y, cl_z, fc_x, fc_y, fc_z, |fc|]
 *        (fc = offset from the pillar centre — the legacy aliasing — cl = offset from the pillar mean)
 * row_pillar (K) i64 : pillar of each compact row
 */
int mbv_pfn_decorate(const float* points, int32_t point_dim, const int32_t* pillar_points,
                     const int32_t* num_points, const int32_t* row_start, const int32_t* coors,
                     int64_t num_pillars, int32_t max_points,
                     float vx, float vy, float vz, float x_off, float y_off, float z_off,
                     float* rows, int64_t* row_pillar, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2b — the per-pillar (non-GEMM) parts of the PFNLayer stack, on real points + one representative padded row
 * per pillar with multiplicity max_points - n (identical to mmdet3d's dense zero-padded evaluation).
 * Replaces: mmdet3d PFNLayer.forward (Linear → BatchNorm1d(eps 1e-3, momentum 0.01) → ReLU → max over the point
 * slots → concat) reached from MaskBevEncoder.encode (mask_bev/models/encoders/mask_bev_encoders.py:70-72,
 * 119-120).  The Linear parts stay library GEMMs (y = a_prev W_a^T, t = max_prev W_b^T, y_pad = a_pad_prev W_a^T).
 * All tensors f32; `units` <= 128 channels; rows of pillar v are [row_start[v], row_start[v] + num_points[v]).
 * With 32 / 64 / 128 units and 16-byte aligned tensors the walks (apply_max, bwd_route, bwd_bn) give a lane four adjacent
 * channels: a wave instruction covers 64 / (units / 4) whole rows, and the wave's next pillar is requested while this one is
 * computed (pfn.hip k_pfn_*_v4); other unit counts take the lane = channel kernels.  Same results either way.
 * PRECONDITION of every mbv_pfn_* entry and of mbv_pfn_decorate: 1 <= num_points[v] <= max_points for every pillar v <
 * num_pillars (K1 never emits an empty pillar: a pillar exists because a point fell into its cell).  The walks request a
 * pillar's first rows unconditionally on a clamped index and the decoration reads slot num_points - 1; num_points lives on
 * the device, so the library cannot check it per call — an empty pillar is the caller's error (reads one row past the
 * compact rows for an empty LAST pillar).
 *   mbv_pfn_stats       y[r] += t[v], y_pad[v] += t[v] (when t != NULL); sums[0:U] = sum y, sums[U:2U] = sum y^2
 *                       over all V * max_points rows (padded rows weighted by their multiplicity), f64
 *   mbv_pfn_bn_finalize batch (training != 0) or running statistics → scale, shift, mean, rstd; updates the
 *                       running buffers in training mode (momentum, unbiased variance)
 *   mbv_pfn_apply_max   a = relu(y*scale+shift) (K, U), a_pad (V, U), m[v] = max over the pillar's rows and, if
 *                       n < max_points, its padded row; a / a_pad may be NULL for the last layer
 *   mbv_pfn_bwd_route   dz = relu'(.) * (dA + dM routed to the first maximal row [padded row last]); dz_pad holds
 *                       the SUM over the padded copies; sums = (sum dz, sum dz*xhat) f64 → d beta, d gamma
 *   mbv_pfn_bwd_bn      BatchNorm backward in place (dz → dy, dz_pad → summed dy_pad), dt[v] = sum_r dy[r] + dy_pad[v]
 */
/* K2c — the PFNLayer's `nn.Linear(in, out, bias=False)` (same reference lines) as a streaming exact-f32 GEMM for M >> C, N:
 * weight_is_nk != 0: y (m, n) = x (m, c) . w^T, w (n, c) with row stride ldw (forward; ldw > c reads a column block of a wider
 * weight: the [a | max] halves of a layer);  == 0: y (m, n) = x (m, c) . w, w (c, n) with row stride ldw (data gradient).
 * x, y contiguous f32; c <= 128, n in {32, 64, 96, 128} (mbv_skinny_gemm_f32_supported); c % 4 == 0 needs a 16-byte aligned x.
 * v_mfma_f32_32x32x2_f32 with the weight held in registers and 32-row tiles of x through wave-private LDS. */
int mbv_skinny_gemm_f32_supported(int64_t m, int32_t contraction, int32_t out_cols);
int mbv_skinny_gemm_f32(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n, int32_t ldw,
                        int32_t weight_is_nk, void* stream);
/* y (m, n) = x (m, c) . w^T + add[index[row]] (add (*, n) f32, index (m) i64): the forward form with a gathered row addend. */
int mbv_skinny_gemm_f32_addrows(const float* x, const float* w, float* y, int64_t m, int32_t c, int32_t n, int32_t ldw,
                                const float* add, const int64_t* index, void* stream);

/* The forward of ALL PFN layers behind one call (the launches above + K2c, issued from inside): the eager section in front of
 * the captured step pays the host's time per launch.  rows (num_rows, in_features) f32 = the decorated compact rows;
 * weights[l] (units[l], in_features) for l = 0 and (units[l], 2 * units[l-1]) behind it ([a | max], mmdet3d PFNLayer);
 * gammas / betas / running_means / running_vars: the layer's BatchNorm1d (running statistics updated in place when training).
 * Every tensor of the pass lives in `workspace` (f32) at the offsets (in floats) mbv_pfn_forward_layout writes — 11 per
 * layer: y, y_pad, t, sums (2 units doubles), scale, shift, mean, rstd, a, a_pad, m; -1 = none — and returns the total for;
 * the last layer's m (num_pillars, units) is the result.  units[l] in {32, 64, 96, 128}, in_features <= 128, <= 8 layers.
 * row_pillar (num_rows) i64, nullable: the pillar of every row (mbv_pfn_decorate writes it).  Given, a layer's pillar term
 * W_b . max is added to y inside the Linear's launch (mbv_skinny_gemm_f32_addrows) and the BatchNorm statistics are a
 * streaming column-sum pass over y; without it the per-pillar walk of mbv_pfn_stats does both. */
int64_t mbv_pfn_forward_layout(int64_t num_rows, int64_t num_pillars, const int32_t* units, int32_t num_layers,
                               int64_t* offsets);
int mbv_pfn_forward(const float* rows, int32_t in_features, const int32_t* row_start, const int32_t* num_points,
                    const int64_t* row_pillar, int64_t num_rows, int64_t num_pillars, int32_t max_points,
                    const float* const* weights,
                    const float* const* gammas, const float* const* betas, float* const* running_means,
                    float* const* running_vars, const int32_t* units, int32_t num_layers, float eps, float momentum,
                    int32_t training, float* workspace, int64_t workspace_floats, void* stream);

int mbv_pfn_stats(float* y, const float* t, float* y_pad, const int32_t* row_start, const int32_t* num_points,
                  int64_t num_pillars, int32_t units, int32_t max_points, double* sums, void* stream);

int mbv_pfn_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                        float momentum, int32_t training, float* running_mean, float* running_var, int32_t units,
                        float* scale, float* shift, float* mean, float* rstd, void* stream);

int mbv_pfn_apply_max(const float* y, const float* y_pad, const float* scale, const float* shift,
                      const int32_t* row_start, const int32_t* num_points, int64_t num_pillars, int32_t units,
                      int32_t max_points, float* a, float* a_pad, float* m, void* stream);

int mbv_pfn_bwd_route(const float* y, const float* y_pad, const float* scale, const float* shift,
                      const float* mean, const float* rstd, float* dz, int32_t has_da, const float* sum_da_pad,
                      const float* dm, const int32_t* row_start, const int32_t* num_points, int64_t num_pillars,
                      int32_t units, int32_t max_points, float* dz_pad, double* sums, void* stream);

int mbv_pfn_bwd_bn(const float* y, const float* y_pad, float* dz, float* dz_pad, const float* mean,
                   const float* rstd, const float* gamma, const double* sums, double count, int32_t training,
                   const int32_t* row_start, const int32_t* num_points, int64_t num_pillars, int32_t units,
                   int32_t max_points, float* dt, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3 — PointPillarsScatter fused with the (C, H, W) LayerNorm; the dense canvas is never built.
 * Replaces: MaskBevEncoder.middle_encode (mask_bev_encoders.py:122-123 → mmdet3d PointPillarsScatter)
 *           + nn.LayerNorm([C, ny, nx], eps) (:75, :92).
 * feats (V, C) f32; pillar_batch_start (batch+1) i32 prefix of pillars per scan (scan b owns pillars
 * [start[b], start[b+1])); cell_to_pillar (batch, cells) i32; weight/bias (C, cells) f32;
 * out (batch, C, cells) f32; stats (batch, 2) f32 = (mean, rstd), saved for backward.
 * workspace: mbv_scatter_layernorm_workspace_bytes(batch).
 * ev_start / ev_stop: optional hipEvent_t (NULL = none) recorded on `stream` immediately before and
 * after the dominant streaming kernel of the call (the apply kernel / the dense backward kernel), so
 * that a caller can time exactly that kernel (bench.py's roofline figure).
 */
size_t mbv_scatter_layernorm_workspace_bytes(int32_t batch);

/* `patch` selects the layout of `out` (forward) and `grad_out` (backward):
 *   0  (batch, C, ny, nx) f32 — the reference's NCHW pseudo-image;
 *   4  (batch, ny/4, nx/4, 16*C) patch tokens in bf16 or half (`patch_dtype` = MBV_DT_BF16 / MBV_DT_F16), element (y%4)*4C + c*4 + x%4 of row (b, y/4, x/4): the input
 *      rows of the backbone's 4 x 4 non-overlapping patch projection (mmdet PatchEmbed built at
 *      mask_bev/models/backbones/swin.py:579-586), which then is one GEMM with no layout or cast pass.
 *      Needs mbv_scatter_layernorm_patch_supported(C, ny, nx, 4) (C % 32 == 0, ny % 4 == 0, nx % 4 == 0). */
int mbv_scatter_layernorm_patch_supported(int32_t channels, int32_t ny, int32_t nx, int32_t patch);

int mbv_scatter_layernorm_fwd(const float* feats, const int32_t* pillar_batch_start,
                              const int32_t* cell_to_pillar, const float* weight, const float* bias,
                              int32_t batch, int32_t channels, int32_t ny, int32_t nx, float eps,
                              int32_t patch, int32_t patch_dtype, void* out, float* stats, void* workspace,
                              size_t workspace_bytes, void* stream, void* ev_start, void* ev_stop);
/* ... with amax_out: an optional absmax record (64 zeroed words, mbv_f32_absmax_group's format) of the f32 (B, C, ny, nx) map
 * (patch == 0), max-combined by one atomic per workgroup: the K20 patch projection behind it (fp32 compute) then needs no pass
 * over the 0.5 GB map. */
int mbv_scatter_layernorm_fwd2(const float* feats, const int32_t* pillar_batch_start, const int32_t* cell_to_pillar,
                               const float* weight, const float* bias, int32_t batch, int32_t channels, int32_t ny, int32_t nx,
                               float eps, int32_t patch, int32_t patch_dtype, void* out, float* stats, void* workspace,
                               size_t workspace_bytes, uint32_t* amax_out, void* stream, void* ev_start, void* ev_stop);

/* Backward: grad_out (layout per `patch`) → grad_feats (V, C), grad_weight / grad_bias (C, cells).
 * `accumulate` != 0 adds into grad_weight / grad_bias instead of overwriting them. */
int mbv_scatter_layernorm_bwd(const void* grad_out, int32_t patch, int32_t patch_dtype, const float* feats,
                              const int32_t* pillar_batch_start, const int32_t* cell_to_pillar,
                              const float* weight, const float* stats,
                              int32_t batch, int32_t channels, int32_t ny, int32_t nx, int64_t num_pillars,
                              float* grad_feats, float* grad_weight, float* grad_bias, int32_t accumulate,
                              void* workspace, size_t workspace_bytes, void* stream,
                              void* ev_start, void* ev_stop);

/* The same backward with the AdamW update of the two (C, ny, nx) affine parameters fused into it (one GPU, one backward
 * per optimizer step): their gradients are complete in registers inside this kernel (the batch sum is per thread), so they
 * never go to memory — no grad_weight / grad_bias traffic, no read of them by mbv_adamw_step, no zero fill.  `weight` and
 * `bias` (C, cells) f32 are read-modify-written together with their moments exp_avg_* / exp_avg_sq_* and, when given, their
 * 16-bit shadows (`shadow_dtype` = MBV_DT_BF16 / MBV_DT_F16); the arithmetic is mbv_adamw_step's (torch/optim/adamw.py's
 * single-tensor order; `step` >= 1 = the optimizer's step count INCLUDING this update; grad_scale 1, no loss scaling).
 * Replaces: mmdet3d / torch nn.LayerNorm([C, ny, nx]) backward (mask_bev_encoders.py:75,92) + the optimizer step of those
 * two parameters (mask_bev_module.py:131-166). */
int mbv_scatter_layernorm_bwd_adamw(const void* grad_out, int32_t patch, int32_t patch_dtype, const float* feats,
                                    const int32_t* pillar_batch_start, const int32_t* cell_to_pillar, float* weight,
                                    float* bias, const float* stats, int32_t batch, int32_t channels, int32_t ny, int32_t nx,
                                    int64_t num_pillars, float* grad_feats, float* exp_avg_w, float* exp_avg_sq_w,
                                    float* exp_avg_b, float* exp_avg_sq_b, void* shadow_w, void* shadow_b,
                                    int32_t shadow_dtype, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    int64_t step, int32_t decoupled, void* workspace, size_t workspace_bytes, void* stream,
                                    void* ev_start, void* ev_stop);

/* ------------------------------------------------------------------------------------------------
 * K5 — multi-scale deformable attention of the pixel decoder, forward / backward.
 * Replaces: mmcv MultiScaleDeformableAttention's `ms_deform_attn_forward/backward` CUDA op (or its
 * grid_sample fallback), configured at mask_bev/models/head/mask_bev_panoptic_head.py:127-136 and run inside
 * `self.pixel_decoder(x)` (mask_bev/models/networks/mask2former_head/mask2former_head.py:500).
 * value (B, Nv, H, D) f32; spatial_shapes (L, 2) i64 (h, w); level_start (L) i64; sampling_loc
 * (B, Nq, H, L, P, 2) f32 in [0, 1] (x, y); attn_weight (B, Nq, H, L, P) f32; out (B, Nq, H*D) f32.
 * Bilinear sampling at loc*size - 0.5 with zero padding (grid_sample align_corners=False).
 * head_dim must be a power of two <= 64.  spatial_shapes_host (nullable): the same (L, 2) shapes in HOST memory.
 * Backward, three forms: (a) head_dim == 32, host shapes given, every level map <= 4096 pixels — no global atomics:
 * d(value) is accumulated per (batch, head, 4-channel group) in a whole-map f64 LDS image and stored once (bitwise
 * reproducible up to the order of the LDS adds), d(location) / d(weight) are a separate gather; mbv_ms_deform_attn_bwd_split
 * says whether that form applies, and `part` (1 = the value part, 2 = the location / weight part, 3 = both) lets a
 * caller enqueue the two independent parts on two streams.  (b) host shapes given and num_query == num_value
 * (self-attention over the multi-scale map): bands of every level's map in LDS (f64), flushed bands and out-of-band
 * corners as global f32 atomics.  (c) otherwise: global f32 atomics.  (b) and (c) zero-fill grad_value themselves and
 * take only part == 3.  In form (a) with part == 1, bits 2.. of `part` may carry a mask of the levels whose d(value) this
 * call produces (0 = all): the per-level launches are independent (A/B experiments put them on different streams).
 */
int mbv_ms_deform_attn_fwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                           const float* sampling_loc, const float* attn_weight,
                           int32_t batch, int32_t num_value, int32_t num_heads, int32_t head_dim,
                           int32_t num_levels, int32_t num_query, int32_t num_points,
                           float* out, void* stream);

/* The same forward, and the d(location) / d(weight) part of the split backward on its own, with the value map stored in
 * value_dtype (MBV_DT_F32 / _BF16 / _F16; head_dim % 4 == 0 resp. == 32): a 16-bit map halves the bytes of every bilinear tap
 * — what the two kernels are bound by — and is what the reference's value projection produces under 16-bit autocast. */
int mbv_ms_deform_attn_fwd_v(const void* value, int32_t value_dtype, const int64_t* spatial_shapes, const int64_t* level_start,
                             const float* sampling_loc, const float* attn_weight, int32_t batch, int32_t num_value,
                             int32_t num_heads, int32_t head_dim, int32_t num_levels, int32_t num_query, int32_t num_points,
                             float* out, void* stream);
int mbv_ms_deform_attn_bwd_locattn(const float* grad_out, const void* value, int32_t value_dtype, const int64_t* spatial_shapes,
                                   const int64_t* level_start, const float* sampling_loc, const float* attn_weight,
                                   int32_t batch, int32_t num_value, int32_t num_heads, int32_t head_dim, int32_t num_levels,
                                   int32_t num_query, int32_t num_points, float* grad_loc, float* grad_attn, void* stream);

int mbv_ms_deform_attn_bwd_split(int32_t head_dim, int32_t num_levels, const int64_t* spatial_shapes_host);

/* d(value) of form (a) with packed fixed-point LDS accumulators (16-bit compute modes): two adjacent channels of a
 * contribution are rounded to signed 32-bit fixed point and added as ONE `ds_add_u64` — half the LDS-atomic instructions
 * of the f64 form, all levels in one launch, order-independent (integer) sums.  The scale is taken per block from
 * L1 = sum over queries of (|attention weights| of the level) x (largest |grad_out| of the block's channels), an upper
 * bound of every pixel's sum for any sampling pattern, so the 32-bit halves cannot wrap; addends are rounded to
 * 2^-30 of that bound's power of two.  head_dim == 32, num_points == 4, every level map <= 16 384 pixels (one 128 KB plane of
 * packed pairs per block in dynamic LDS; the location / weight part the caller runs next is mbv_ms_deform_attn_bwd_locattn,
 * gathers without a map-size limit), host shapes required.  grad_value is written in `out_dtype` (MBV_DT_F32 / BF16 / F16) with `out_ld` elements between
 * consecutive (batch, value) rows — e.g. straight into the first H*D columns of the 16-bit matrix
 * [d value | d offsets | d logits] whose product with [Wv; Wo; Wa] is d(x).  Every element of that block is written
 * (no zero fill needed).  `_supported` returns 1 when the shape qualifies.  A first launch
 * re-lays grad_out (by 4-channel group), the locations and the weights (by level) into the caller's workspace
 * (`_workspace_bytes`), so that every block of the accumulation kernel reads contiguous streams (full cache lines).
 * Replaces the same mmcv backward as mbv_ms_deform_attn_bwd (value part). */
int mbv_ms_deform_attn_bwd_value_packed_supported(int32_t head_dim, int32_t num_levels, int32_t num_points,
                                                  int32_t num_query, const int64_t* spatial_shapes_host);
size_t mbv_ms_deform_attn_bwd_value_packed_workspace_bytes(int32_t batch, int32_t num_heads, int32_t num_levels,
                                                           int32_t num_query);
int mbv_ms_deform_attn_bwd_value_packed(const float* grad_out, const float* sampling_loc, const float* attn_weight,
                                        int32_t batch, int32_t num_value, int32_t num_heads, int32_t head_dim,
                                        int32_t num_levels, int32_t num_query, int32_t num_points,
                                        const int64_t* spatial_shapes_host, void* grad_value, int32_t out_dtype,
                                        int64_t out_ld, void* workspace, size_t workspace_bytes, void* stream);

int mbv_ms_deform_attn_bwd(const float* grad_out, const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start, const float* sampling_loc, const float* attn_weight,
                           int32_t batch, int32_t num_value, int32_t num_heads, int32_t head_dim,
                           int32_t num_levels, int32_t num_query, int32_t num_points,
                           const int64_t* spatial_shapes_host, float* grad_value, float* grad_loc, float* grad_attn,
                           int32_t part, void* stream);

/* K16 — the element-wise chain in front of K5, fused (mmcv MultiScaleDeformableAttention.forward:
 * `attention_weights.softmax(-1)`, `offset_normalizer = stack([W_l, H_l])`, `sampling_locations =
 * reference_points + sampling_offsets / offset_normalizer`; same call site as K5).
 * offsets (B, Nq, H, L, P, 2) and logits (B, Nq, H, L*P): both f32 (is_bf16 = 0) or both bf16 (is_bf16 = 1, the
 * autocast projections; the quotient is then rounded to bf16 before the f32 add, as torch's type promotion does);
 * ref_points (Nq, 2) f32 (x, y) in [0, 1]; spatial_shapes_host (L, 2) int64 (h, w) in HOST memory.
 * Outputs loc (B, Nq, H, L, P, 2) f32 and attn (B, Nq, H, L*P) f32 = softmax over the L*P samples of a head.
 * Backward: grad_loc / grad_attn (f32, from K5) and attn → grad_offsets / grad_logits in the input dtype.
 * Supported: L <= 8, L*P <= 16 (mbv_msda_prepare_supported); the caller composes torch ops otherwise. */
int mbv_msda_prepare_supported(int32_t num_levels, int32_t num_points);

/* The 16-bit GEMM inputs of the query side in one pass: x_lo = lo(x) (value projection input) and q_lo = lo(x + pos)
 * (offset / attention-weight projection input) for x (rows, C) f32 and pos (pos_rows, C) f32 repeating every pos_rows
 * rows — `query = query + query_pos` of mmcv MultiScaleDeformableAttention.forward under autocast (same call site).
 * dtype: MBV_DT_BF16 / MBV_DT_F16; C % 4 == 0. */
int mbv_msda_query_inputs(const float* x, const float* pos, int64_t rows, int64_t pos_rows, int32_t C, int32_t dtype,
                          void* x_lo, void* q_lo, void* stream);

int mbv_msda_prepare_fwd(const void* offsets, const void* logits, int32_t is_bf16, const float* ref_points,
                         const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query, int32_t num_heads,
                         int32_t num_levels, int32_t num_points, float* loc, float* attn, void* stream);

/* Same with (batch, query) row strides (elements) for the two inputs — they can then be column blocks of ONE projection
 * GEMM's output, [offsets | logits] — and optional f32 biases (H*L*P*2) / (H*L*P) added to the loaded values in f32, for a
 * GEMM that left them out.  With L*P % 4 == 0 both pointers and strides must keep 4-element alignment. */
int mbv_msda_prepare_fwd_ld(const void* offsets, int64_t ld_offsets, const void* logits, int64_t ld_logits,
                            const float* bias_offsets, const float* bias_logits, int32_t is_bf16, const float* ref_points,
                            const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query, int32_t num_heads,
                            int32_t num_levels, int32_t num_points, float* loc, float* attn, void* stream);

int mbv_msda_prepare_bwd(const float* grad_loc, const float* grad_attn, const float* attn,
                         const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query, int32_t num_heads,
                         int32_t num_levels, int32_t num_points, int32_t out_bf16, void* grad_offsets,
                         void* grad_logits, void* stream);

/* Same with (batch, query) row strides for the two outputs (elements): they can then be columns of one wider gradient
 * matrix — next to the gradient of the value projection — that a single data-gradient GEMM consumes. */
int mbv_msda_prepare_bwd_ld(const float* grad_loc, const float* grad_attn, const float* attn,
                            const int64_t* spatial_shapes_host, int32_t batch, int32_t num_query, int32_t num_heads,
                            int32_t num_levels, int32_t num_points, int32_t out_bf16, void* grad_offsets,
                            int64_t ld_offsets, void* grad_logits, int64_t ld_logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4 — fused shifted-window multi-head attention (between the qkv and the output projection).
 * Replaces: ShiftWindowMSA.forward (mask_bev/models/networks/swin/swin.py:179-253: pad, roll, window
 * partition, mask, reverse, un-roll, crop) + WindowMSA.forward (:80-118: head split, q k^T * scale +
 * relative-position bias [+ shift mask], softmax, . v) + window_partition/reverse (:255-284).
 * qkv (B, H, W, 3C) channels-last, f32 (is_bf16 = 0: exact-f32 MFMA) or bf16 (is_bf16 = 1); tokens the
 * reference pads in take qkv_bias (3C) f32; bias_table ((2ws-1)^2, heads) f32; out (B, H, W, C) same dtype
 * as qkv; lse: mbv_window_attn_lse_elems(...) f32, saved for backward.  head_dim = C / heads must be
 * 16, 32 or 64 and ws <= 11.
 * Backward zero-fills grad_table ((2ws-1)^2, heads) and grad_qkv_bias (3C: gradient reaching the bias
 * through padded tokens; with full_bias_grad != 0 also the column sums of grad_qkv over the real tokens, i.e.
 * the whole bias gradient of the qkv Linear, which then skips its own pass over grad_qkv) itself, then accumulates them
 * with f32 atomics — unless accumulate != 0: then they are gradients to be added INTO (parameter-arena views), no fill.
 * grad_qkv (B, H, W, 3C) is written in full.
 */
int64_t mbv_window_attn_lse_elems(int32_t batch, int32_t H, int32_t W, int32_t heads, int32_t ws);

int mbv_window_attn_fwd(const void* qkv, const float* qkv_bias, const float* bias_table, int32_t is_bf16,
                        int32_t batch, int32_t H, int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift,
                        void* out, float* lse, void* stream);

int mbv_window_attn_bwd(const void* qkv, const float* qkv_bias, const float* bias_table, const void* out,
                        const void* grad_out, const float* lse, int32_t is_bf16,
                        int32_t batch, int32_t H, int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift,
                        void* grad_qkv, float* grad_table, float* grad_qkv_bias, int32_t full_bias_grad,
                        int32_t accumulate, void* stream);

/* K4 on f32 tensors in the SPLIT mode (fp32 compute; same tensors, results and reference lines as mbv_window_attn_fwd / _bwd with
 * is_bf16 = 0): every f32 operand element is split into an IEEE-half pair while its tile is staged (x 2^e = hi + lo: 22
 * significant bits) and every product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with f32 accumulation — K20's arithmetic
 * (mbv_gemm32s_*) inside the attention — instead of v_mfma_f32_32x32x2_f32 at the f32 vector rate out of word-wise LDS images.
 * amax_qkv / amax_grad_out: absmax records of the tensors (64 words each, mbv_f32_absmax_group or a producing kernel's epilogue;
 * NULL = unscaled); amax_grad_qkv (optional): the record of what the backward stores, max-combined by one atomic per workgroup.
 * Requires head dims 16 / 32 / 64, C % 4 == 0, 16-byte aligned tensors (mbv_window_attn_split_supported; else
 * MBV_ERR_UNSUPPORTED and the caller keeps the exact-f32 form).  Accuracy: <= 2e-6 of the result's maximum (tests). */
int mbv_window_attn_split_supported(int32_t C, int32_t heads, int32_t ws);
int mbv_window_attn_split_fwd(const float* qkv, const float* qkv_bias, const float* bias_table, int32_t batch, int32_t H,
                              int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift, const uint32_t* amax_qkv,
                              float* out, float* lse, void* stream);
int mbv_window_attn_split_bwd(const float* qkv, const float* qkv_bias, const float* bias_table, const float* out,
                              const float* grad_out, const float* lse, int32_t batch, int32_t H, int32_t W, int32_t C,
                              int32_t heads, int32_t ws, int32_t shift, const uint32_t* amax_qkv,
                              const uint32_t* amax_grad_out, float* grad_qkv, float* grad_table, float* grad_qkv_bias,
                              int32_t full_bias_grad, int32_t accumulate, uint32_t* amax_grad_qkv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8 — indexed bilinear point sampling of mask maps (loss and Hungarian targets).
 * Replaces: mmcv point_sample (= F.grid_sample, align_corners=False, zeros) at
 * mask_bev/models/networks/mask2former_head/mask2former_head.py:194-200,402-410 and the tensor gathers
 * feeding it (:227 `gt_masks[pos_assigned_gt_inds]`, :393 `mask_preds[mask_weights > 0]`).
 * out[g][p] = bilinear(src[src_index[g]] (H, W), coords[coord_index[g]][p] (x, y in [0, 1])).
 * src (num_src_maps, H, W) f32; src_index, coord_index (num_rows) i32; coords (*, num_points, 2) f32;
 * out / grad_out (num_rows, num_points) f32.
 * Backward zero-fills grad_src (num_src_maps, H, W) itself (skipped when every map is sampled: num_rows == num_src_maps).  For H*W <= 16384 each row's gradient tile is
 * accumulated in LDS and stored once, which requires src_index to hold no duplicates; larger maps use
 * global f32 atomics.
 */
int mbv_point_sample_fwd(const float* src, const int32_t* src_index, const float* coords,
                         const int32_t* coord_index, int32_t num_rows, int32_t num_points, int32_t H, int32_t W,
                         float* out, void* stream);

int mbv_point_sample_bwd(const float* grad_out, const int32_t* src_index, const float* coords,
                         const int32_t* coord_index, int32_t num_rows, int32_t num_points, int32_t H, int32_t W,
                         int64_t num_src_maps, float* grad_src, void* stream);

/* The same backward for a STACK of maps, (outer, inner, rows) of them, each sampled exactly once (num_rows == outer * inner *
 * rows, H*W <= 16384): the gradient of map (o, n, r) is stored at row (n, o, r) of grad_src — the two leading axes exchanged
 * — in f32, bf16 or fp16 (out_dtype).  The stacked mask logits of the D decoder outputs, (D, B, Q, H, W), hand their
 * gradient to the batched backward of the prediction heads sample-major and in the GEMM's operand type: no permute + cast
 * pass over the 262 MB gradient (mask2former_head.py:406-424 → :459). */
int mbv_point_sample_bwd_stack(const float* grad_out, const int32_t* src_index, const float* coords,
                               const int32_t* coord_index, int32_t num_rows, int32_t num_points, int32_t H, int32_t W,
                               int32_t outer, int32_t inner, int32_t rows, void* grad_src, int32_t out_dtype, void* stream);

/* Binary ({0, 1}-valued) maps packed 32 pixels per word, and K8's forward on the packed form: a whole
 * 512 x 512 ground-truth mask is 32 KB and is staged in LDS by the workgroup that samples it.
 * The batch contract of the reference makes GT masks float32 {0, 1}
 * (mask_bev/datasets/semantic_kitti/semantic_kitti_transforms.py:77-81,95-106); bit = (value != 0).
 * packed: (num_maps, mbv_packed_mask_words(H, W)) u32.  H*W <= 1024*1024. */
int64_t mbv_packed_mask_words(int32_t H, int32_t W);

int mbv_pack_binary_masks(const float* src, int64_t num_maps, int32_t H, int32_t W, uint32_t* packed, void* stream);

int mbv_point_sample_packed_fwd(const uint32_t* packed, const int32_t* src_index, const float* coords,
                                const int32_t* coord_index, int32_t num_rows, int32_t num_points,
                                int32_t H, int32_t W, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K10 — importance sampling of the mask loss: the k points with the smallest |logit| of each row.
 * Replaces: torch.topk(-|logits|, k) + coordinate gather of mmdet's get_uncertain_point_coords_with_randomness,
 * called at mask_bev/models/networks/mask2former_head/mask2former_head.py:401-404.
 * logits (rows, n) f32; coords (rows, n, 2) f32; out_coords (rows, k, 2) f32: the coordinates of the selected
 * points in ascending index order (same SET as top-k; ties at the threshold resolved towards lower indices).
 */
int mbv_select_uncertain_points(const float* logits, const float* coords, int64_t rows, int32_t n, int32_t k,
                                float* out_coords, void* stream);

/* Fused K8 + K10: sample the n candidate points of each row from its source map and keep the k most uncertain,
 * without materialising the (rows, n) sampled logits (the whole of get_uncertain_point_coords_with_randomness,
 * mask2former_head.py:401-404).  src (N, H, W) f32 with H*W <= 16384; src_index (rows) i32 → source map of a row;
 * n <= 40960 candidates per row, k <= 16384 of them kept, given EITHER as coords (rows, n, 2) f32 in [0, 1] (seed NULL) OR generated in the
 * kernel from the device-resident 64-bit *seed (coords NULL): point p of row r is mbv_uniform_points' value, so
 * the 1.2 GB candidate tensor of a training step is never written or read.  rand_coords (rows, n_rand, 2) f32
 * (nullable when n_rand == 0): the uniform tail, copied behind the selected points; out_coords
 * (rows, k + n_rand, 2).  Same selection, order and tie rule as mbv_point_sample_fwd followed by
 * mbv_select_uncertain_points.  MBV_ERR_UNSUPPORTED outside those limits (callers use the two-kernel form). */
int mbv_sample_select_uncertain(const float* src, const int32_t* src_index, const float* coords, const int64_t* seed,
                                int64_t rows, int32_t n, int32_t k, int32_t H, int32_t W, const float* rand_coords,
                                int32_t n_rand, float* out_coords, void* stream);

/* The generator behind the seed form above, written out: out_coords (rows, n, 2) f32 uniform in [0, 1), a pure
 * function of (*seed, row, point) — replaces torch.rand of mmdet's get_uncertain_point_coords_with_randomness. */
int mbv_uniform_points(const int64_t* seed, int64_t rows, int32_t n, float* out_coords, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K11 — parameter-arena kernels of the optimisation step.
 * Replaces: torch.optim.AdamW / Adam built by MaskBevModule.configure_optimizers
 * (mask_bev/mask_bev_module.py:131-166; update order of torch/optim/adamw.py `_single_tensor_adamw`) and the
 * bias-gradient reductions of every nn.Linear on the path.
 * mbv_adamw_step: ONE pass over a flat f32 arena of n parameters: param, grad, exp_avg, exp_avg_sq (all 16-byte
 *   aligned, n elements).  step >= 1 is the 1-based update count (bias corrections are computed on the host in
 *   f64).  decoupled=1 → AdamW (param *= 1 - lr*wd), 0 → Adam (grad += wd*param).  grad is multiplied by
 *   grad_scale first (1/world_size after a SUM all-reduce).  shadow_bf16 (nullable, 8-byte aligned) receives the
 *   updated parameters rounded to nearest-even bf16 or half (shadow_dtype = MBV_DT_BF16 / MBV_DT_F16) — the copy the
 *   16-bit GEMMs read.  zero_grad=1 clears grad in the same pass.
 *   fp16 loss scaling, without a host round trip: loss_scale (device f32 scalar, nullable) is the factor the loss was
 *   multiplied by — grad is divided by it on the fly; skip_flag (device i32, nullable) non-zero = the gradient held
 *   inf / nan: parameters, moments and shadow are left as they are and only zero_grad is honoured.  applied_steps
 *   (device i32, nullable): the count of updates applied so far — the bias corrections then use t = *applied_steps + 1
 *   instead of `step` (torch.amp.GradScaler skips optimizer.step() on an overflow: Adam's count must not advance).
 * mbv_grad_nonfinite: *flag |= 1 if any of grad[0..n) is inf or nan (torch.amp.GradScaler's found_inf).
 * mbv_loss_scale_update: GradScaler.update() on the device — flag set: scale = max(scale * backoff, 1), streak = 0;
 *   else streak += 1 and scale *= growth every growth_interval clean steps (and ++*applied_steps, nullable); clears
 *   the flag.
 * mbv_refresh_shadow: shadow[i] = bf16 / half (param[i]) (after load_state_dict / broadcast).
 * mbv_colsum_accum: out[c] += sum_r g[r, c] for row-major g (rows, n), bf16 (is_bf16=1) or f32; f32 atomics.
 */
int mbv_adamw_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int32_t shadow_dtype,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   float grad_scale, int32_t decoupled, int32_t zero_grad, const float* loss_scale,
                   const int32_t* skip_flag, const int32_t* applied_steps, void* stream);

int mbv_grad_nonfinite(const float* grad, int64_t n, int32_t* flag, void* stream);

int mbv_loss_scale_update(float* loss_scale, int32_t* clean_steps, int32_t* flag, float growth, float backoff,
                          int32_t growth_interval, int32_t* applied_steps, void* stream);

int mbv_refresh_shadow(const float* param, void* shadow_bf16, int32_t shadow_dtype, int64_t n, void* stream);

int mbv_colsum_accum(const void* g, int32_t is_bf16, int64_t rows, int32_t n, float* out, void* stream);
/* count independent column sums out[i] (n[i]) += sum_r g[i][r, 0:n[i]] with row stride ld[i] (elements) and storage
 * dtype[i] (MBV_DT_*), one launch per 64.  All array arguments are HOST arrays of length count.  Bias gradients and
 * LayerNorm-parameter partials are nobody's input: a caller may collect them during the backward and issue them
 * together at its end. */
int mbv_colsum_accum_group(const void* const* g, const int32_t* dtype, const int64_t* rows, const int32_t* n,
                           const int64_t* ld, float* const* out, int32_t count, void* stream);

/* Exact-f32 weight gradient of a few-row Linear (the decoder's B*Q tokens): acc (O, I) += g (T, O)^T x (T, I) with
 * v_mfma_f32_32x32x2_f32 from global memory, f32 atomics into the (arena) gradient; bias_acc (O) nullable += column
 * sums of g.  mbv_wgrad_small_f32_group: n such products in one launch per 48 — every array argument is a HOST array
 * of length n (device pointers / sizes per product; bias_acc may be NULL or hold NULL entries).  These gradients are
 * nobody's input, so a caller may collect them during the backward and issue them together at its end. */
int mbv_wgrad_small_f32_group(const float* const* g, const float* const* x, float* const* acc, float* const* bias_acc,
                              const int32_t* T, const int32_t* O, const int32_t* I, int32_t n, void* stream);

/* Activation backward fused with the bias gradient of the Linear in front of it (the fc1 layers of mmcv FFN,
 * mask_bev/models/networks/swin/swin.py:347-355, mask_bev_panoptic_head.py:137-142): grad_pre = grad_act * act'(pre_act)
 * with act = ReLU (kind 0) or erf-GELU (kind 1), all (rows, n) f32 or bf16, n % 4 == 0;
 * bias_acc (n) f32 (nullable) += column sums of grad_pre. */
int mbv_act_bwd_colsum(const void* grad_act, const void* pre_act, int32_t is_bf16, int32_t kind, int64_t rows, int32_t n,
                       void* grad_pre, float* bias_acc, void* stream);

/* Weight (and bias) gradient of a Linear applied to few tokens (the decoder's B*Q query rows), exact f32:
 * acc (O, I) += g^T x with g (T, O), x (T, I) row-major f32; bias_acc (O) += column sums of g when not NULL;
 * f32 MFMA, f32 atomics into acc / bias_acc. */
int mbv_wgrad_small_f32(const float* g, const float* x, int32_t T, int32_t O, int32_t I, float* acc, float* bias_acc,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * K13 — row sums of the point-sampled mask losses and their gradient.
 * Replaces: the elementwise chains of mmdet DiceLoss(naive_dice=True, eps=1) and CrossEntropyLoss(use_sigmoid=True)
 * on the sampled points (mask_bev/models/networks/mask2former_head/mask2former_head.py:406-424).
 * logits, targets (rows, points) f32.  out_sums (rows, 4) f32 = [sum sigmoid(x)*t, sum sigmoid(x), sum t,
 * sum bce_with_logits(x, t)].  bwd: grad_sums (rows, 4) → grad_logits (rows, points); the gradient of sum t
 * (column 2) is ignored (targets carry no gradient).
 */
int mbv_mask_loss_rows_fwd(const float* logits, const float* targets, int64_t rows, int32_t points, float* out_sums,
                           void* stream);

int mbv_mask_loss_rows_bwd(const float* logits, const float* targets, const float* grad_sums, int64_t rows,
                           int32_t points, float* grad_logits, void* stream);

/* The dice / BCE algebra on K13's row sums, per decoder output — mmdet DiceLoss(naive_dice, eps 1) and
 * CrossEntropyLoss(use_sigmoid) reduced with their avg_factor as at mask2former_head.py:406-424, for `outputs` decoder outputs of
 * rows / outputs rows each:  den = S1 + S2 + 1, dice = (2 S0 + 1) / den;
 *   loss_dice[d] = c_dice * sum_rows(1 - dice),  loss_mask[d] = c_mask * sum_rows(S3)
 * (c_dice = loss weight / avg_factor, c_mask likewise).  coef (rows, 3) f32 receives the gradient of the two losses with respect
 * to each row's sums for unit upstream gradients: (-2 c_dice / den, c_dice dice / den, c_mask).  sums 16-byte aligned. */
int mbv_dice_bce_reduce(const float* sums, int64_t rows, int32_t outputs, float c_dice, float c_mask, float* loss_dice,
                        float* loss_mask, float* coef, void* stream);

/* mbv_mask_loss_rows_bwd with the per-row gradient of the sums assembled in the kernel from mbv_dice_bce_reduce's `coef` and
 * the upstream gradients of the row's decoder output: grad_dice / grad_mask (outputs,) f32 with element strides
 * stride_dice / stride_mask (0 = one broadcast value; NULL = zero gradient). */
int mbv_mask_loss_rows_bwd_coef(const float* logits, const float* targets, const float* coef, const float* grad_dice,
                                int32_t stride_dice, const float* grad_mask, int32_t stride_mask, int64_t rows,
                                int32_t outputs, int32_t points, float* grad_logits, void* stream);

/* Matching-cost terms (mmdet CrossEntropyLossCost(use_sigmoid) + DiceCost on the sampled points,
 * mask2former_head.py:199-205): logits (groups, queries, points) f32 → terms (groups, 3, queries, points) f32 =
 * [softplus(-x), softplus(x), sigmoid(x)] — one batched GEMM against the sampled ground truth then gives all three
 * cost matrices — and row_sums (groups*queries, 2) = [sum softplus(x), sum sigmoid(x)].  ones_row != 0: a group has
 * 3 * queries + 1 rows, the last one all ones, so that the same GEMM also returns the targets' row sums. */
int mbv_match_cost_terms(const float* logits, int64_t groups, int32_t queries, int32_t points, int32_t ones_row,
                         float* terms, float* row_sums, void* stream);

/* The matching cost matrices from those products (HungarianAssigner's ClassificationCost 2.0 + CrossEntropyLossCost 5.0 +
 * DiceCost 5.0 as configured at mask_bev/models/head/mask_bev_panoptic_head.py and evaluated at mask2former_head.py:199-210):
 * cls (groups, queries, classes_plus_one) f32 logits, labels (batch, targets) i64, prod (groups, 3 * queries + 1, targets)
 * f32 = terms . sampled targets (ones row last), row_sums as above; group = (decoder output, image), image fastest.
 * cost (groups, queries, targets) f32. */
int mbv_match_cost(const float* cls, const int64_t* labels, const float* prod, const float* row_sums, int64_t groups,
                   int32_t queries, int32_t targets, int32_t classes_plus_one, int32_t batch, int32_t points, float* cost,
                   void* stream);

/* The same costs without the term planes and without a library GEMM (K13c, csrc/match_products.hip): the products
 * x . t and sigmoid(x) . t and the row sums come from ONE kernel that evaluates the terms per 32-point chunk, splits them
 * into IEEE-half pairs (22 significant bits) and contracts them on v_mfma_f32_32x32x16_f16 with f32 accumulation.
 * logits (groups, queries, points) f32 = the sampled mask logits, targets (groups, targets_n, points) f32 = the sampled
 * ground truth (any values in [0, 1]); both 16-byte aligned.  A group's points are cut into `splits` slices (one workgroup
 * each: choose groups * splits ~ 2 workgroups per CU; 1 <= splits <= ceil(points / 32)); slice s of group n writes
 *   prod[n][s] (2 * queries + 1, targets_n + 1) f32 = [x ; sigmoid(x) ; ones] . [t ; ones]^T   and
 *   neg_sums[n][s] (queries) f32 = sum softplus(x),
 * which mbv_match_cost_split adds in slice order (no atomics: bit-reproducible) into cost (groups, queries, targets) f32 =
 * -2 softmax(cls)[label] + 5 (sum softplus(x) - x . t) / points + 5 (1 - (2 sigmoid(x) . t + 1) / (sum sigmoid(x) + sum t + 1)).
 * Supported (mbv_match_products_supported != 0): 2 * queries + 1 <= 224, targets_n + 1 <= 128, points % 8 == 0; otherwise
 * MBV_ERR_UNSUPPORTED and the caller keeps mbv_match_cost_terms + a GEMM + mbv_match_cost. */
int mbv_match_products_supported(int32_t queries, int32_t targets_n, int32_t points);
int mbv_match_products(const float* logits, const float* targets, int64_t groups, int32_t queries, int32_t targets_n,
                       int32_t points, int32_t splits, float* prod, float* neg_sums, void* stream);
int mbv_match_cost_split(const float* cls, const int64_t* labels, const float* prod, const float* neg_sums, int64_t groups,
                         int32_t queries, int32_t targets, int32_t classes_plus_one, int32_t batch, int32_t points,
                         int32_t splits, float* cost, void* stream);

/* Classification loss of all decoder outputs (mmdet CrossEntropyLoss(class_weight), avg_factor = sum of the targets' class
 * weights: mask2former_head.py:393-404): cls (outputs, batch * queries, classes_plus_one) f32, assigned (outputs, batch,
 * queries) i32 = ground-truth column or -1 (→ label classes_plus_one - 1, "no object"), labels (batch, targets) i64.
 * loss (outputs) f32, weight_sum (outputs) f32 (kept for the backward); grad_cls has the shape of cls. */
int mbv_cls_loss_fwd(const float* cls, const int32_t* assigned, const int64_t* labels, const float* class_weight,
                     int32_t outputs, int32_t batch, int32_t queries, int32_t targets, int32_t classes_plus_one,
                     float loss_weight, float eps, float* loss, float* weight_sum, void* stream);
int mbv_cls_loss_bwd(const float* cls, const int32_t* assigned, const int64_t* labels, const float* class_weight,
                     const float* weight_sum, const float* grad_loss, int32_t outputs, int32_t batch, int32_t queries,
                     int32_t targets, int32_t classes_plus_one, float loss_weight, float eps, float* grad_cls, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K12 — fused residual-add + LayerNorm over the last (channel) axis of token-major activations.
 * Replaces: the `x + f(x)` → nn.LayerNorm(C) pairs of SwinBlock.forward (mask_bev/models/networks/swin/swin.py:
 * 357-377), of the pixel-decoder and masked-attention decoder layers (mmdet, configured at
 * mask_bev/models/head/mask_bev_panoptic_head.py:119-176) and the decoder's post_norm
 * (mask2former_head.py:448), forward and backward.
 * a, b: (rows, C) f32 or bf16 (`*_bf16` flags; b nullable = plain LayerNorm); gamma, beta (C) f32; C % 4 == 0,
 * C <= 2048 (mbv_add_layernorm_supported; the patch-merging form below takes 4 c <= 3072).  fwd writes sum_out = a + b (f32, the tensor the backward needs; may be
 * NULL only for a lone f32 `a`, which then serves as the saved input), y in f32 or bf16, mean / rstd (rows) f32.
 * bwd: dy (rows, C) f32/bf16, ds nullable (gradient reaching the sum from the residual path), s = the saved sum;
 * writes dx (rows, C) f32 (the gradient of a and of b), optionally the same in 16 bits (dx_lo, of dx_lo_dtype), and
 * dgamma / dbeta (C) f32 — overwritten, or accumulated into when accumulate != 0 (parameter-arena gradients).
 * dbranch_bias (C) f32, nullable: += the column sums of dx — the bias gradient of the Linear that produced the
 * residual branch b (its output gradient IS dx), saving that layer a pass over dx.
 * partial_ws: mbv_add_layernorm_bwd_blocks(rows, C) * 3 * C floats.
 */
int mbv_add_layernorm_supported(int32_t C);
int64_t mbv_add_layernorm_bwd_blocks(int64_t rows, int32_t C);
int mbv_add_layernorm_fwd(const void* a, int32_t a_bf16, const void* b, int32_t b_bf16, const float* gamma,
                          const float* beta, int64_t rows, int32_t C, float eps, float* sum_out, void* y, int32_t y_bf16,
                          float* mean, float* rstd, void* stream);
/* Same forward with a second copy of y in another storage type (y2 nullable): a post-LN layer's output has two consumers
 * — the next residual add takes it in f32, the branch GEMM that follows in 16 bits — and the cast launch between them goes.
 * The two copies hold the same f32 value rounded to their types; their gradients return through dy / dy2 of the backward.
 * b_rows > 0 (a divisor of rows): b has only b_rows rows and repeats — a per-sample map added to every sample of the batch
 * (the gradient of such a b is the batch sum of dx: the caller's to take).  0 or rows: b has `rows` rows. */
int mbv_add_layernorm_fwd2(const void* a, int32_t a_bf16, const void* b, int32_t b_bf16, int64_t b_rows, const float* gamma,
                           const float* beta, int64_t rows, int32_t C, float eps, float* sum_out, void* y, int32_t y_bf16,
                           void* y2, int32_t y2_dtype, float* mean, float* rstd, void* stream);

/* acc (C, R) f32 += the batch sum of g (batch, R, C) f32, transposed: the gradient of a per-sample token map that was added to
 * every sample, accumulated into the (1, C, H, W) parameter it is a transposed view of — the backbone's absolute position
 * embedding (mask_bev/models/networks/swin/swin.py:579-586, added at :750-760); one pass instead of a batch reduction and a
 * transposed accumulate. */
int mbv_transposed_batch_sum_accum(const float* g, int32_t batch, int64_t R, int32_t C, float* acc, void* stream);
int mbv_add_layernorm_bwd(const void* dy, int32_t dy_bf16, const void* ds, int32_t ds_bf16, const float* s,
                          const float* mean, const float* rstd, const float* gamma, int64_t rows, int32_t C, float* dx,
                          void* dx_lo, int32_t dx_lo_dtype, float* dgamma, float* dbeta, int32_t accumulate,
                          float* dbranch_bias, float* partial_ws, int32_t defer_reduce, void* stream);
/* mbv_add_layernorm_bwd with a second gradient of y (dy2, nullable, dy2_dtype = MBV_DT_*) that is added to dy on load:
 * the output of a post-LN layer feeds both the next residual add and the next branch (mask_bev_panoptic_head.py:119-176),
 * and its two gradients arrive separately instead of through an element-wise add launch. */
int mbv_add_layernorm_bwd2(const void* dy, int32_t dy_bf16, const void* dy2, int32_t dy2_dtype, const void* ds,
                           int32_t ds_bf16, const float* s, const float* mean, const float* rstd, const float* gamma,
                           int64_t rows, int32_t C, float* dx, void* dx_lo, int32_t dx_lo_dtype, float* dgamma,
                           float* dbeta, int32_t accumulate, float* dbranch_bias, float* partial_ws, int32_t defer_reduce,
                           void* stream);
/* mbv_add_layernorm_bwd2 with amax_dx: an optional absmax record (64 zeroed words, mbv_f32_absmax_group's format) that receives the
 * bits of max|dx| — one max-combine per block — so that the K20 products of the Linear backward that takes dx as its output
 * gradient need no pass over it (fp32 compute). */
int mbv_add_layernorm_bwd3(const void* dy, int32_t dy_bf16, const void* dy2, int32_t dy2_dtype, const void* ds,
                           int32_t ds_bf16, const float* s, const float* mean, const float* rstd, const float* gamma,
                           int64_t rows, int32_t C, float* dx, void* dx_lo, int32_t dx_lo_dtype, float* dgamma, float* dbeta,
                           int32_t accumulate, float* dbranch_bias, float* partial_ws, int32_t defer_reduce, uint32_t* amax_dx,
                           void* stream);
/* 1 = the backward of (rows, C) adds the parameter gradients from inside its one kernel; 0 = it writes per-block
 * partial rows (nblk = mbv_add_layernorm_bwd_blocks, layout [nblk][np][C], np = 3 with dbranch_bias else 2) and reduces
 * them with a second launch — unless defer_reduce != 0 (accumulating callers only), in which case the caller adds
 * the partials' column sums itself, typically for many layers at once with mbv_colsum_accum_group. */
int mbv_add_layernorm_bwd_direct(int64_t rows, int32_t C);

/* Patch merging without the unfolded copy: y (batch, h/2, w/2, 4c) = LayerNorm_{4c}(unfold_{2x2, stride 2}(x)) for a
 * channels-last f32 token map x (batch, h, w, c), h and w even, channel order c*4 + kh*2 + kw — the nn.Unfold order of
 * mmdet's PatchMerging as built at mask_bev/models/networks/swin/swin.py:611-616 (its `sampler` + `norm`; the `reduction`
 * Linear follows as a GEMM on y).  Element 4v + k of row (b, oh, ow) is channel v of pixel (2oh + k/2, 2ow + k%2): K12's
 * kernels gather / scatter with that addressing, so the (batch, h/2, w/2, 4c) copy, whose elements land 4 bytes at a
 * time, is never written and the backward's dx arrives in x's layout (every pixel belongs to exactly one row).
 * y: MBV_DT_F32 / BF16 / F16 (y_dtype); mean / rstd (batch*h/2*w/2) f32 are saved for the backward; dgamma / dbeta (4c),
 * accumulate, partial_ws (mbv_add_layernorm_bwd_blocks(rows, 4c) * 2 * 4c floats) and defer_reduce as in
 * mbv_add_layernorm_bwd. */
int mbv_merge_layernorm_supported(int32_t h, int32_t w, int32_t c);
int mbv_merge_layernorm_fwd(const float* x, int64_t batch, int32_t h, int32_t w, int32_t c, const float* gamma,
                            const float* beta, float eps, void* y, int32_t y_dtype, float* mean, float* rstd,
                            void* stream);
int mbv_merge_layernorm_bwd(const void* dy, int32_t dy_dtype, const float* x, const float* mean, const float* rstd,
                            const float* gamma, int64_t batch, int32_t h, int32_t w, int32_t c, float* dx, float* dgamma,
                            float* dbeta, int32_t accumulate, float* partial_ws, int32_t defer_reduce, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K14 — batch producer: instance-id map → instance ids → per-instance binary masks.
 * Replaces: FilterSmallMasks + MaskToLabelInstanceMasks of the reference's data pipeline
 * (mask_bev/datasets/semantic_kitti/semantic_kitti_transforms.py:11-26, 66-81) and the host→device copy of the
 * dense (B, Q, ny, nx) f32 masks they produce.
 * instance_map (B, nx, ny) i32, 0 = background (the layout of the reference's mask cache; the transform's `.T` is
 * applied here).  mbv_instance_ids: ids (B, Q) i32 = the ids with >= min_pixels pixels in ascending order, -1
 * padded (the reference enumerates a Python set: same set, unspecified order — the Hungarian matcher makes the
 * loss independent of it); counts (B) i32; *status (device i32) bit 0 = a scan had more than Q instances (the
 * reference raises IndexError; here the Q smallest ids are kept), bit 1 = more than 4096 distinct ids.
 * mbv_expand_instance_masks: masks_f32 (B, Q, ny, nx) f32 {0,1} and / or masks_packed (B*Q, words) u32 in the
 * bit-packed layout of mbv_pack_binary_masks (either may be NULL): masks[b, q, y, x] = (map[b, x, y] == ids[b, q]).
 */
int mbv_instance_ids(const int32_t* instance_map, int32_t batch, int32_t nx, int32_t ny, int32_t num_queries,
                     int32_t min_pixels, int32_t* ids, int32_t* counts, int32_t* status, void* stream);

int mbv_expand_instance_masks(const int32_t* instance_map, const int32_t* ids, int32_t batch, int32_t nx, int32_t ny,
                              int32_t num_queries, float* masks_f32, uint32_t* masks_packed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K15 — mask IoU of matched (prediction, ground truth) pairs at ground-truth resolution.
 * Replaces: the upsample → sigmoid > 0.5 → batched_mask_iou chain of MaskBevPanopticHead.update_mAP_metrics
 * (mask_bev/models/head/mask_bev_panoptic_head.py:74-85; mask_bev/evaluation/average_precision.py:78-81).
 * logits (N, h, w) f32 with h*w <= 16384; pred_row / gt_row (pairs) i32: rows of `logits` and of the bit-packed
 * ground truth (mbv_pack_binary_masks layout, maps of H x W pixels); gt_row < 0 marks an unmatched prediction.
 * inter / uni (pairs) i32 pixel counts; IoU = inter / (uni + 1e-12) like the reference.
 */
int mbv_matched_mask_iou(const float* logits, const int32_t* pred_row, const uint32_t* gt_packed,
                         const int32_t* gt_row, int32_t num_pairs, int32_t h, int32_t w, int32_t H, int32_t W,
                         int32_t* inter, int32_t* uni, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K9 — batched linear-sum assignment on the device (one wavefront per cost matrix).
 * Replaces: mmdet HungarianAssigner → scipy.optimize.linear_sum_assignment on the host, reached from
 * Mask2FormerHead._get_targets_single (mask_bev/models/networks/mask2former_head/mask2former_head.py:207-210).
 * cost (batch, num_rows, num_cols) f32 (finite); row_to_col (batch, num_rows) i32 = assigned column of each
 * row, -1 for rows left unassigned when num_rows > num_cols.  Up to 128 x 128 the cost matrix lives in LDS (the
 * 100-query configurations); up to 320 x 320 (200 / 300 queries) it is read from global memory one coalesced row
 * per search step, which needs num_rows <= num_cols: for num_rows > num_cols > 128 call mbv_hungarian_wide_t with
 * the materialised transpose cost_t (batch, num_cols, num_rows) — same row_to_col (batch, num_rows) output.
 */
int mbv_hungarian(const float* cost, int32_t batch, int32_t num_rows, int32_t num_cols,
                  int32_t* row_to_col, void* stream);

/* The same for matrices (rows = predictions) x (cols = ground-truth slots, rows <= cols <= 320; above 128 columns the
 * wide kernel with the real columns' block staged in LDS) whose trailing columns
 * are identical padding — the dataset pads the instance list to num_queries with all-zero masks of label 0
 * (semantic_kitti_transforms.py:66-81), so for a given prediction those columns hold one and the same cost.
 * real_cols (batch) i32 ON THE DEVICE: the number of leading real columns of each problem.  Solves the equivalent
 * rectangular problem of the real columns (the padded columns' cost enters as the start value of the predictions'
 * duals) — the same optimum, the same real pairs whenever it is unique, at about (real / rows)^2 of the search steps;
 * the predictions left over take the padded columns in ascending order.  Problems without padding (real_cols ==
 * cols) and non-square problems (a real column may then stay unmatched) are solved like mbv_hungarian. */
int mbv_hungarian_padded(const float* cost, int32_t batch, int32_t num_rows, int32_t num_cols, const int32_t* real_cols,
                         int32_t* row_to_col, void* stream);
int mbv_hungarian_wide_t(const float* cost_t, int32_t batch, int32_t num_rows, int32_t num_cols,
                         int32_t* row_to_col, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K7 — per-query mask logits on MFMA, and the boolean cross-attention mask of the next decoder layer.
 * Replaces: torch.einsum('bqc,bchw->bqhw', mask_embed, mask_feature)
 * (mask_bev/models/networks/mask2former_head/mask2former_head.py:459) and :460-470 + :538-539
 * (F.interpolate bilinear → sigmoid() < 0.5 → repeat over heads; a row that would block every key is unblocked).
 * mask_embed (B, Q, C), mask_feature (B, C, pixels), logits (B, Q, pixels): all f32 (is_bf16 = 0, exact-f32 MFMA,
 * C even) or all bf16 (is_bf16 = 1, C % 16 == 0); with bf16 inputs, logits_f32 != 0 stores the f32 accumulators
 * as f32 logits (the loss consumes f32; saves the cast pass).
 * mbv_attn_mask_from_logits: logits (rows, H, W) → blocked (rows, h*w) u8 (1 = may not attend), kept once per
 * query (the reference materialises it 8x, once per head).
 */
int mbv_mask_logits_fwd(const void* mask_embed, const void* mask_feature, int32_t is_bf16, int32_t batch,
                        int32_t num_queries, int32_t channels, int64_t pixels, void* logits, int32_t logits_f32,
                        void* stream);

int mbv_attn_mask_from_logits(const void* logits, int32_t is_bf16, int64_t rows, int32_t H, int32_t W,
                              int32_t h, int32_t w, uint8_t* blocked, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6 — multi-head (masked) attention of the transformer decoder, forward / backward on MFMA.
 * Replaces: the nn.MultiheadAttention cores (via mmcv MultiheadAttention) of the decoder loop at
 * mask_bev/models/networks/mask2former_head/mask2former_head.py:542-553 (masked cross-attention over the
 * L = h*w memory tokens of one level, then query self-attention); heads/ffn configured at
 * mask_bev/models/head/mask_bev_panoptic_head.py:150-176.
 * q (B, Q, heads*D), k, v (B, L, heads*D): projected inputs, all f32 (is_bf16 = 0) or all bf16; the softmax
 * scale 1/sqrt(D) is applied inside.  blocked (B, Q, L) u8, 1 = may not attend, or NULL (self-attention);
 * every row must keep at least one attendable key (mbv_attn_mask_from_logits guarantees it).
 * out (B, Q, heads*D) same dtype; lse (B, heads, Q) f32 saved for backward.  head_dim in {16, 32, 64}.
 * Backward writes f32 gradients grad_q (B, Q, E), grad_k / grad_v (B, L, E) in full.
 */
size_t mbv_attn_workspace_bytes(int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim);

int mbv_attn_fwd(const void* q, const void* k, const void* v, const uint8_t* blocked, int32_t is_bf16,
                 int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim,
                 void* out, float* lse, void* workspace, size_t workspace_bytes, void* stream);

int mbv_attn_bwd(const void* q, const void* k, const void* v, const uint8_t* blocked, const void* out,
                 const void* grad_out, const float* lse, int32_t is_bf16,
                 int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim,
                 float* grad_q, float* grad_k, float* grad_v, void* stream);

/* Same kernels with strided key / value operands: k and v point at columns of a wider row-major matrix whose rows are
 * `ld_kv` elements apart (the key / value projections of SEVERAL decoder layers that attend to the same memory are
 * then produced side by side by one GEMM — the three layers of a level in mask2former_head.py:535-560 — and the
 * gradient of that matrix is assembled in place: grad_k / grad_v rows are `ld_grad_kv` apart, optionally stored as
 * bf16, ready to be the operand of the batched data / weight-gradient GEMMs).  Strided or bf16 key / value gradients
 * need num_queries <= 128. */
int mbv_attn_fwd_ld(const void* q, const void* k, const void* v, int32_t ld_kv, const uint8_t* blocked,
                    int32_t is_bf16, int32_t batch, int32_t num_queries, int32_t num_keys, int32_t heads,
                    int32_t head_dim, void* out, float* lse, void* workspace, size_t workspace_bytes, void* stream);

int mbv_attn_bwd_ld(const void* q, const void* k, const void* v, int32_t ld_kv, const uint8_t* blocked,
                    const void* out, const void* grad_out, const float* lse, int32_t is_bf16, int32_t batch,
                    int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim, float* grad_q,
                    void* grad_k, void* grad_v, int32_t ld_grad_kv, int32_t grad_kv_bf16, void* stream);

/* K6 on f32 tensors in the SPLIT mode (fp32 compute; the arguments, results and reference lines of mbv_attn_fwd_ld / _bwd_ld with
 * is_bf16 = 0, f32 key / value gradients): every f32 operand element of a block's tiles is split into an IEEE-half pair while the
 * tile is staged and every product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 (K20's arithmetic, K4's split mode), with
 * one power-of-two scale PER TILE found in registers between the loads and the LDS stores.  head_dim 16 / 32 / 64, ld_kv % 4 == 0,
 * 16-byte aligned tensors (mbv_attn_split_supported; else MBV_ERR_UNSUPPORTED and the caller keeps the exact-f32 form). */
int mbv_attn_split_supported(int32_t heads, int32_t head_dim, int32_t ld_kv);
int mbv_attn_split_fwd_ld(const float* q, const float* k, const float* v, int32_t ld_kv, const uint8_t* blocked, int32_t batch,
                          int32_t num_queries, int32_t num_keys, int32_t heads, int32_t head_dim, float* out, float* lse,
                          void* workspace, size_t workspace_bytes, void* stream);
int mbv_attn_split_bwd_ld(const float* q, const float* k, const float* v, int32_t ld_kv, const uint8_t* blocked, const float* out,
                          const float* grad_out, const float* lse, int32_t batch, int32_t num_queries, int32_t num_keys,
                          int32_t heads, int32_t head_dim, float* grad_q, float* grad_k, float* grad_v, int32_t ld_grad_kv,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * K17 — 16-bit (bf16 / fp16) MFMA GEMMs of the token-major Linear layers, with the element-wise work around them
 * fused into the epilogue.  dtype: 0 = bf16, 1 = fp16 (inputs; accumulation is f32).
 * Replaces the cuBLAS/ATen GEMMs behind nn.Linear in WindowMSA.qkv / .proj (mask_bev/models/networks/swin/swin.py:
 * 89-116), the mmcv FFN of SwinBlock (swin.py:347-377: Linear -> GELU -> Linear), PatchEmbed / PatchMerging
 * (swin.py:579-586, 611), the Linears of the pixel decoder and the transformer decoder
 * (mask_bev/models/head/mask_bev_panoptic_head.py:119-175) and their autograd backward.
 * Every matrix is row-major with leading dimension ld* (elements, multiples of 8; pointers 16-byte aligned);
 * `batch` independent problems are `stride_*` elements apart.  Shapes the kernels do not take return
 * MBV_ERR_UNSUPPORTED (callers use the library GEMM for those).
 *
 *   mbv_gemm16_nt   out (m, n) = act(x (m, k) . w (n, k)^T + bias);  act: 0 none, 1 ReLU, 2 GELU (erf form);
 *                   out_pre (optional, act != 0) receives the pre-activation; out / out_pre are 16-bit or f32.
 *   mbv_gemm16_nn   out (m, k) = act'(aux) * (g (m, n) . w (n, k));  act: 0 none, 1 ReLU' with aux = the activation
 *                   output, 2 GELU' with aux = the pre-activation (aux (m, k) 16-bit, ld = ldaux, batch stride of
 *                   out); colsum (k) f32 (optional) += column sums of the stored out (the bias gradient of the
 *                   Linear in front of the activation; needs `workspace` of mbv_gemm16_nn_workspace_bytes).
 *   mbv_gemm16_tn   accumulate != 0: dw (n, k) f32 += g (m, n)^T . x (m, k), the sum over m split over `splits`
 *                   workgroups per tile (0 = choose).  With a `workspace` of mbv_gemm16_tn_workspace_bytes (dw
 *                   contiguous, batch 1) the parts are stored and added by their owner thread (no atomics, bit-
 *                   reproducible); without it they add with f32 atomics (dw is the parameter arena's gradient);
 *                   accumulate == 0: dw = g^T . x stored once (16-bit or f32).
 */
int mbv_gemm16_supported(int32_t layout, int64_t m, int64_t n, int64_t k);

int mbv_gemm16_nt(const void* x, const void* w, const float* bias, void* out, void* out_pre, int64_t m, int64_t n,
                  int64_t k, int64_t ldx, int64_t ldw, int64_t ldo, int32_t dtype, int32_t out_f32, int32_t act,
                  int32_t batch, int64_t stride_x, int64_t stride_w, int64_t stride_o, void* stream);

/* acc (m, n) f32 += x (m, k) . w (n, k)^T with the sum over k split over `splits` workgroups per tile (0 = choose)
 * that add with f32 atomics: few-row products with a long contraction — d(mask_embed) = d(logits) . mask_feature^T of
 * torch.einsum('bqc,bchw->bqhw') (mask_bev/models/networks/mask2former_head/mask2former_head.py:459), 1000 x 256
 * outputs over 16 384 pixels per sample. */
int mbv_gemm16_nt_acc(const void* x, const void* w, float* acc, int64_t m, int64_t n, int64_t k, int64_t ldx,
                      int64_t ldw, int64_t ldacc, int32_t dtype, int32_t splits, int32_t batch, int64_t stride_x,
                      int64_t stride_w, int64_t stride_acc, void* stream);

size_t mbv_gemm16_nn_workspace_bytes(int64_t m, int64_t k, int32_t batch);

int mbv_gemm16_nn(const void* g, const void* w, void* out, const void* aux, float* colsum, int64_t m, int64_t n,
                  int64_t k, int64_t ldg, int64_t ldw, int64_t ldo, int64_t ldaux, int32_t dtype, int32_t out_f32,
                  int32_t act, int32_t batch, int64_t stride_g, int64_t stride_w, int64_t stride_o, void* workspace,
                  size_t workspace_bytes, void* stream);

/* mbv_gemm16_nn with the bias gradient's reduction left to the caller: `parts` (mbv_gemm16_nn_part_rows(m, k, batch), k)
 * f32, contiguous, 16-byte aligned, parts_bytes >= mbv_gemm16_nn_workspace_bytes(m, k, batch), receives one partial
 * column-sum row per 64 output rows (every element written); the column sum is their sum over the rows.  A backward pass
 * hands the rows of all its fused data gradients to ONE mbv_colsum_accum_group launch at its end instead of a small
 * reduction launch behind every GEMM (same reference lines as mbv_gemm16_nn: the fc1 bias gradient of mmcv's FFN,
 * mask_bev/models/networks/swin/swin.py:347-355). */
int64_t mbv_gemm16_nn_part_rows(int64_t m, int64_t k, int32_t batch);

int mbv_gemm16_nn_parts(const void* g, const void* w, void* out, const void* aux, float* parts, size_t parts_bytes,
                        int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldw, int64_t ldo, int64_t ldaux,
                        int32_t dtype, int32_t out_f32, int32_t act, int32_t batch, int64_t stride_g, int64_t stride_w,
                        int64_t stride_o, void* stream);

size_t mbv_gemm16_tn_workspace_bytes(int64_t m, int64_t n, int64_t k);

int mbv_gemm16_tn(const void* g, const void* x, void* dw, int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldx,
                  int64_t lddw, int32_t dtype, int32_t accumulate, int32_t out_f32, int32_t splits, int32_t batch,
                  int64_t stride_g, int64_t stride_x, int64_t stride_dw, void* workspace, size_t workspace_bytes,
                  void* stream);

/* d(coarse map) of `F.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=False)` — the adjoint K18's fused
 * FPN step owes its added map (ATen upsample_bilinear2d_backward): grad_out (planes, h, w) → grad_in (planes, in_h, in_w), any
 * MBV_DT_* storage types; gather form, no atomics, every output written once. */
int mbv_upsample_bilinear_bwd(const void* grad_out, int32_t grad_dtype, int64_t planes, int32_t h, int32_t w, int32_t in_h,
                              int32_t in_w, void* grad_in, int32_t in_dtype, void* stream);

/* K18 — GroupNorm of NCHW maps fused with its surroundings in mmcv's ConvModule (conv -> GroupNorm(32) [-> ReLU]) as
 * the pixel decoder builds it (mask_bev/models/head/mask_bev_panoptic_head.py:119-123 -> mmdet MSDeformAttnPixelDecoder
 * input_convs / lateral_convs / output_convs) and the FPN step  lateral + F.interpolate(previous, bilinear,
 * align_corners=False)  between them.  x (batch, channels, h, w) in x_dtype (MBV_DT_*), h*w % 4 == 0, 16-byte aligned;
 *   y = GroupNorm(x; groups, gamma, beta, eps)  [+ up-sampled `add` (batch, channels, add_h, add_w), w % 4 == 0]  [ReLU]
 * stored in y_dtype; mean / rstd (batch * groups) f32 are saved for the backward; statistics in f64 (biased variance).
 * Backward: dx in dx_dtype, dgamma / dbeta (channels) stored or (accumulate != 0) added to; the ReLU gate is recomputed
 * from x with the forward's arithmetic; the gradient of `add` is the up-sampling's backward of dy (left to the caller).
 * plane_sums: batch * channels * 2 floats of scratch. */
int mbv_groupnorm_supported(int32_t channels, int32_t groups, int32_t h, int32_t w);
size_t mbv_groupnorm_workspace_bytes(int64_t batch, int32_t channels, int32_t groups, int32_t h, int32_t w);
int mbv_groupnorm_fwd(const void* x, int32_t x_dtype, int64_t batch, int32_t channels, int32_t h, int32_t w,
                      int32_t groups, const float* gamma, const float* beta, float eps, const void* add,
                      int32_t add_dtype, int32_t add_h, int32_t add_w, int32_t relu, void* y, int32_t y_dtype, float* mean,
                      float* rstd, void* workspace, size_t workspace_bytes, void* stream);
int mbv_groupnorm_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, int64_t batch, int32_t channels, int32_t h,
                      int32_t w, int32_t groups, int32_t relu, void* dx, int32_t dx_dtype, float* dgamma, float* dbeta,
                      int32_t accumulate, float* plane_sums, void* stream);

/* The weight gradients of `count` Linears in one GEMM launch (+ one parts-add launch) per 48 of them:
 *   dw[i] (n[i], k[i]) f32, contiguous  +=  g[i] (m[i], n[i])^T . x[i] (m[i], k[i])          for i < count.
 * Replaces the per-layer weight-gradient GEMMs of autograd's Linear backward (the same layers as above): a weight
 * gradient is nobody's input, so the caller collects the (g, x, dw) triples of a backward pass and issues them together
 * at its end — a 192 x 192 ... 2304 x 768 output alone cannot fill 256 CUs without cutting the token sum into slivers,
 * the tiles of all layers together do.  Every array argument is a HOST array of length count; g / x are 16-bit
 * (`dtype`), ld* in elements (multiples of 8), pointers 16-byte aligned, m[i] == 0 entries are skipped.  Token sums
 * deeper than the work-item depth are cut into parts stored in `workspace` (mbv_gemm16_tn_group_workspace_bytes) and
 * added to dw by their owner thread: no atomics, bit-reproducible. */
size_t mbv_gemm16_tn_group_workspace_bytes(const int64_t* m, const int64_t* n, const int64_t* k, int32_t count);

int mbv_gemm16_tn_group(const void* const* g, const void* const* x, float* const* dw, const int64_t* m, const int64_t* n,
                        const int64_t* k, const int64_t* ldg, const int64_t* ldx, int32_t count, int32_t dtype,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K19 — row-local stage chains of the transformer decoder's query side (csrc/rowchain.hip).
 * Replaces, per decoder layer, the few-row Linear / LayerNorm / ReLU / add launches between two attention calls of
 * mask_bev/models/networks/mask2former_head/mask2former_head.py:535-560 (mmdet Mask2FormerTransformerDecoderLayer:
 * cross-attn out-proj + LN, self-attn projections, out-proj + LN, FFN + LN) and the prediction-head MLPs of :428-472.
 * A workgroup owns 16 token rows and runs `num_stages` stages over them; activations stay in LDS slots
 * (mbv_rowchain_slots() slots of 16 x 256 f32) between stages.  Stage operations:
 *   MBV_RC_LOAD    dst <- rows of p0 (dtype = flags & 3, row stride ld) [+ f32 p1 at row (r % q_mod), stride ld2];
 *                  p0 == NULL: the base operand is slot `src` (then p1 is required).  Rows >= `rows` load as zeros.
 *   MBV_RC_STORE   rows of p0 <- src (dtype = flags & 3, stride ld); MBV_RC_ACCUM: p0 += src (f32; in a split launch only with an owner — BAD_ARG otherwise).
 *   MBV_RC_GEMM    dst = act((ACCUM ? dst : 0) + src (16 x k) . W^T + bias), W = p0 (n, k) row-major with row stride ld in
 *                  the program's weight dtype, bias = p1 f32 or NULL; MBV_RC_RELU; MBV_RC_MASK: result zeroed where
 *                  slot src2 <= 0 (ReLU backward).  f32 weights: exact f32 MFMA; 16-bit weights: activations rounded to
 *                  that type, f32 accumulation.  n <= 256, k <= 256, k % 16 (f32) / 32 (16-bit) == 0, dst != src.
 *                  MBV_RC_FRAG (16-bit): p0 is a fragment-major copy (mbv_fragment_group) positioned at the stage's first
 *                  (tile, block) and ld = its 32-column blocks per tile row (one load instruction = 8 full cache lines
 *                  instead of 16 half lines).  The FFN stage's weights are always fragment-major (ld / ld2 likewise).
 *   MBV_RC_LN      dst = LayerNorm(src [+ src2]) * p0 + p1 (src2 = -1: none); MBV_RC_SAVE_SUM: src <- the sum;
 *                  p2 (nullable): (rows, 2) f32 <- (mean, rstd).
 *   MBV_RC_LN_BWD  dst = d(sum) given src = d(output), src2 = the sum, p0 = gamma, p2 = stats; p1 (nullable):
 *                  (blocks, 2 n) f32 <- this block's partial d(gamma) | d(beta).
 *   MBV_RC_ADD     dst = src + src2.      MBV_RC_COLSUM  p0[block * ld + c] = sum over the block's rows of src.
 *   MBV_RC_FFN     (16-bit weights) the MLP pair with the hidden dimension cut into 256-wide chunks owned by the waves in
 *                  parallel: forward dst = relu(src . W1^T + b1) . W2^T with W1 = p0 (k x n, stride ld), b1 = p1, W2 = p2
 *                  (n x k, stride ld2); MBV_RC_MASK = backward: dst = ((src . p0^T) * (act > 0)) . p2^T with p0 = W2^T,
 *                  p2 = W1^T.  n = embed width (<= 256, % 32), k = hidden width (% 256); src2 = first of 5 scratch slots.
 *                  Directly followed by an MBV_RC_FFN_IO descriptor: p0 = hidden activations (rows x k f32, stride ld;
 *                  written forward, read backward), p1 = d(hidden) out (backward), p2 = (blocks, ld2) column partials of
 *                  d(hidden) (backward, nullable).  The output bias is left to the caller (e.g. a following LN's operand).
 *   MBV_RC_SUM     dst = sum over j < k of p0[j * ld2 + row * ld + c] (f32): the parts a split launch stored.
 * Split launches (mbv_rowchain_run_split, split = S > 1): S consecutive workgroups share a 16-row block, so that a
 * 400-row decoder layer occupies S x 25 CUs instead of 25.  A stage whose flags bits 8..15 hold v > 0 runs only in
 * workgroup v - 1 of its row block, every other stage in all S (redundantly: loads, the cheap products and norms).
 * MBV_RC_FFN | MBV_RC_SLICE (k == 256 S): workgroup j computes the hidden units [256 j, 256 j + 256) and leaves its
 * PARTIAL output in dst (workgroup 0 adds the output bias p1 of the descriptor; p2 is a K-MAJOR fragment copy and
 * n % 16 == 0); MBV_RC_STORE | MBV_RC_SPLIT writes to
 * p0 + j * ld2 elements.  A following launch adds the parts with MBV_RC_SUM — in part order, so the result does not
 * depend on timing.  Per-block partial outputs (MBV_RC_COLSUM, LN_BWD's p1, the descriptor's p2) are indexed by ROW
 * block in either form.
 * Data gradients dX = dY . W are MBV_RC_GEMM stages against transposed weight copies (mbv_transpose_group).
 * The program is copied into the kernel arguments (<= mbv_rowchain_max_stages() stages): nothing is retained. */
#define MBV_RC_LOAD 0
#define MBV_RC_STORE 1
#define MBV_RC_GEMM 2
#define MBV_RC_LN 3
#define MBV_RC_LN_BWD 4
#define MBV_RC_ADD 5
#define MBV_RC_COLSUM 6
#define MBV_RC_FFN 7
#define MBV_RC_FFN_IO 8
#define MBV_RC_SUM 9
#define MBV_RC_ACCUM 4
#define MBV_RC_RELU 8
#define MBV_RC_MASK 16
#define MBV_RC_SAVE_SUM 32
#define MBV_RC_SLICE 64  /* FFN stages */
#define MBV_RC_SPLIT 64  /* STORE stages */
#define MBV_RC_FRAG 128
#define MBV_RC_OWNER(j) (((j) + 1) << 8) /* stage of workgroup j of a split launch only */
#define MBV_TR_MAX 96
typedef struct MbvRowStage {
  int32_t op;
  int16_t dst, src, src2, reserved;
  int32_t n, k, flags, ld, ld2;
  const void* p0;
  const void* p1;
  const void* p2;
} MbvRowStage;
int mbv_rowchain_max_stages(void);
int mbv_rowchain_slots(void);
int mbv_rowchain_run(const MbvRowStage* stages, int32_t num_stages, int32_t rows, int32_t q_mod, float eps,
                     int32_t wdtype, void* stream);
int mbv_rowchain_run_split(const MbvRowStage* stages, int32_t num_stages, int32_t rows, int32_t q_mod, float eps,
                           int32_t wdtype, int32_t split, void* stream);
/* Fragment-major copies of 16-bit weight matrices for the 16 x 16 x 32 MFMA B operand: for the logical (rows, cols) matrix
 * W, dst[((t * (cols / 32) + kb) * 64 + lane) * 8 + j] = W[t * 16 + lane % 16][kb * 32 + 8 * (lane / 16) + j], rows padded
 * with zeros to a multiple of 16; transposed[i] & 1: W[r][c] = src[c * ld + r] (the data-gradient operand), else
 * src[r * ld + c].  transposed[i] & 2: k-major block order — block (t, kb) at (kb * ceil(rows / 16) + t) instead of
 * (t * (cols / 32) + kb) — the second weight of an MBV_RC_SLICE stage.  cols % 32 == 0.  dst holds
 * ceil(rows / 16) * 16 * cols elements. */
int mbv_fragment_group(const void* const* src, void* const* dst, const int32_t* rows, const int32_t* cols,
                       const int32_t* ld, const int32_t* transposed, int32_t n, void* stream);
/* dst[i] = src[i], bytes[i] bytes each: the pieces of several small concatenations (the k / v rows of three decoder layers'
 * packed in_proj parameters, mask2former_head.py:535-560 → nn.MultiheadAttention) in one launch. */
int mbv_copy_group(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);
/* dst[i] (cols, rows) = transpose of src[i] (rows, cols), n matrices of elem_size 2 or 4 bytes, <= MBV_TR_MAX per launch. */
int mbv_transpose_group(const void* const* src, void* const* dst, const int32_t* rows, const int32_t* cols, int32_t n,
                        int32_t elem_size, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K20 — f32 GEMMs on the 16-bit matrix pipe (csrc/gemm_f32s.hip): every f32 operand element is split at staging time into
 * an IEEE-half pair  x 2^e = hi + lo  (22 significant bits) and the product is hi.hi + hi.lo + lo.hi with f32
 * accumulation (error <= 2^-22 sum|a||b| + K 2^-37 max|a| max|b|: the level of an f32 dot product; the dropped lo.lo
 * term is 2^-22 relative).  Replaces, in the reference-precision (fp32) step, the f32 library GEMMs behind nn.Linear
 * forward / backward of the token-major layers: Swin qkv / proj / FFN / patch merging
 * (/root/reference: mask_bev/models/networks/swin/swin.py:89-116, 347-355, 611-616) and the pixel decoder's Linears
 * (mask_bev/models/head/mask_bev_panoptic_head.py:119-146) — `precision=32` in train_mask_bev.py:96.
 * All matrices f32, row-major, ld* in elements (multiples of 4), pointers 16-byte aligned, n and k multiples of 8.
 * amax_*: absmax RECORDS — 64 consecutive device words (256 bytes) whose maximum is the BITS of max|operand| or of an upper
 * bound of it (mbv_f32_absmax_group, or a K20 product that max-combined its outputs into the record: amax_out here; 64 slots, and a
 * workgroup adds only when its maximum exceeds what its slot already holds: a few atomics per launch) —
 * from which the kernel derives the power of two that puts the operand's largest magnitude in [2^13, 2^14); a bound up
 * to 2^4 too large costs nothing (the full-accuracy window is 18 binades deep); NULL = unscaled (operands whose magnitudes lie
 * within [2^-3, 2^15] then keep full accuracy, smaller elements lose one bit per binade).  `act`: 0 none, 1 ReLU,
 * 2 GELU (erf form); out_pre (optional) receives the pre-activation.  Batched: `batch` products with element strides. */
int mbv_gemm32s_supported(int32_t layout, int64_t m, int64_t n, int64_t k);

/* max|x| of `count` f32 tensors (rows[i], cols[i]) with row stride ld[i] (cols % 4 == 0), as the BITS of the maximum,
 * max-combined into the 64-word record at out[i] (slot = workgroup mod 64): the record must hold zeros (or earlier partial
 * maxima of the same tensor) when the launch starts.  Every array argument is a HOST array of length count; one launch per 64 tensors. */
int mbv_f32_absmax_group(const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ld,
                         uint32_t* const* out, int32_t count, void* stream);

/* Absmax BOUNDS of LayerNorm outputs without a pass over them: y = xhat gamma + beta with |xhat| <= sqrt(C - 1), so word 0 of
 * records[i] (64 words, cleared by the caller once) receives the bits of sqrt(c[i]) max|gamma[i]| + max|beta[i]| (beta[i] may be
 * NULL) — ~ 3x above the true maximum of a 65 536 x 192 map, well inside the 18 binades a K20 operand keeps at full precision.
 * One workgroup per LayerNorm, 96 per launch; HOST arrays of length count. */
int mbv_ln_bound_group(const float* const* gamma, const float* const* beta, const int32_t* c, uint32_t* const* records,
                       int32_t count, void* stream);

/* out (m, n) = act(x (m, k) . w (n, k)^T + bias)                                   (nn.Linear forward) */
int mbv_gemm32s_nt(const float* x, const float* w, const float* bias, float* out, float* out_pre, int64_t m, int64_t n,
                   int64_t k, int64_t ldx, int64_t ldw, int64_t ldo, const uint32_t* amax_x, const uint32_t* amax_w,
                   uint32_t* amax_out, int32_t act, int32_t batch, int64_t stride_x, int64_t stride_w, int64_t stride_o,
                   void* stream);

/* out (m, k) = g (m, n) . w (n, k)                                                 (nn.Linear data gradient) */
int mbv_gemm32s_nn(const float* g, const float* w, float* out, int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldw,
                   int64_t ldo, const uint32_t* amax_g, const uint32_t* amax_w, uint32_t* amax_out, int32_t batch,
                   int64_t stride_g, int64_t stride_w, int64_t stride_o, void* stream);

/* dw (n, k) contiguous += g (m, n)^T . x (m, k)                                     (nn.Linear weight gradient)
 * The token sum is cut into parts that are stored to `workspace` (mbv_gemm32s_tn_workspace_bytes) and added into dw by
 * their owner thread: no atomics, bit-reproducible. */
size_t mbv_gemm32s_tn_workspace_bytes(int64_t m, int64_t n, int64_t k);
int mbv_gemm32s_tn_acc(const float* g, const float* x, float* dw, int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldx,
                       const uint32_t* amax_g, const uint32_t* amax_x, void* workspace, size_t workspace_bytes,
                       void* stream);

/* The 4 x 4 non-overlapping patch projection of the backbone (mmdet PatchEmbed = Conv2d(C, E, 4, stride 4) on the encoder's
 * (B, C, h, w) f32 pseudo-image; /root/reference: mask_bev/models/networks/swin/swin.py:579-586) on K20 WITHOUT materialising
 * the (tokens, 16 C) row matrix: element (token, k' = c 16 + dy 4 + dx) of the rows is image element (b, c, 4 oy + dy, 4 ox + dx),
 * gathered (forward, weight gradient) or scattered (image gradient) by the GEMM's own 16-byte operand pieces.  Replaces, in fp32
 * compute, MIOpen's convolution forward / backward (0.73 + 1.78 ms per step at 512 x 512, batch 4).  weight (E, 16 C) = the
 * (E, C, 4, 4) parameter as it lies in memory; out (B * h/4 * w/4, E) token-major; w % 128 == 0, h % 4 == 0, C even, E % 8 == 0. */
int mbv_patch_embed32_supported(int64_t batch, int64_t channels, int64_t h, int64_t w, int64_t embed);
int mbv_patch_embed32_fwd(const float* image, const float* weight, const float* bias, float* out, int64_t batch,
                          int64_t channels, int64_t h, int64_t w, int64_t embed, const uint32_t* amax_image,
                          const uint32_t* amax_w, void* stream);
int mbv_patch_embed32_bwd_image(const float* d_out, const float* weight, float* d_image, int64_t batch, int64_t channels,
                                int64_t h, int64_t w, int64_t embed, const uint32_t* amax_g, const uint32_t* amax_w,
                                void* stream);
size_t mbv_patch_embed32_bwd_weight_workspace_bytes(int64_t batch, int64_t channels, int64_t h, int64_t w, int64_t embed);
int mbv_patch_embed32_bwd_weight(const float* d_out, const float* image, float* d_weight, int64_t batch, int64_t channels,
                                 int64_t h, int64_t w, int64_t embed, const uint32_t* amax_g, const uint32_t* amax_image,
                                 void* workspace, size_t workspace_bytes, void* stream);


/* The weight gradients of `count` f32 Linears in ONE K20 launch (+ one parts-add launch) per 48 of them:
 * dw[i] (n[i], k[i]) f32, contiguous  +=  g[i] (m[i], n[i])^T . x[i] (m[i], k[i]);  the dw[i] of one call must not overlap.
 * A weight gradient is nobody's input: the Linears of a backward pass hand theirs over and the pass issues them together at its
 * end (fp32 compute; the 16-bit modes' mbv_gemm16_tn_group).  Few-row products (the decoder's 400 query rows: a 256 x 256 ...
 * 2048 x 256 output is 4 ... 32 tiles) fill the chip together; token-major ones (1 024 - 65 536 tokens) are cut into ranges of
 * ~ 4 096 tokens instead of the ~ 40 slivers each needs alone.  One range: the tile is added to dw in place by its owner
 * workgroup; several: partial tiles go to `workspace` (mbv_gemm32s_tn_group_workspace_bytes; 16-byte aligned) and are added by
 * their owner — no atomics, bit-reproducible.  amax_g[i] / amax_x[i]: absmax records of the operands (NULL array or entry =
 * unscaled).  Replaces, in fp32 compute, mbv_wgrad_small_f32_group (exact-f32 MFMA + atomics) for shapes with n, k multiples
 * of 8, and the per-layer mbv_gemm32s_tn_acc launches.  HOST arrays of length count. */
size_t mbv_gemm32s_tn_group_workspace_bytes(const int64_t* m, const int64_t* n, const int64_t* k, int32_t count);
int mbv_gemm32s_tn_group(const float* const* g, const float* const* x, float* const* dw, const int64_t* m, const int64_t* n,
                         const int64_t* k, const int64_t* ldg, const int64_t* ldx, const uint32_t* const* amax_g,
                         const uint32_t* const* amax_x, int32_t count, void* workspace, size_t workspace_bytes, void* stream);

/* The pixel decoder's 3 x 3 convolution in-tree (fp32 compute; replaces MIOpen's f32 convolution forward / backward-data /
 * backward-weight for /root/reference: mask_bev/models/head/mask_bev_panoptic_head.py:119-146 — mmdet
 * MSDeformAttnPixelDecoder.output_convs: ConvModule(256, 256, 3, padding 1, bias off) + GroupNorm + ReLU on the (B, 256, 128, 128)
 * map).  The map is turned once into a zero-bordered channels-last ROWS buffer — (mbv_conv_rows(B, H, W), C): G = W + 3 guard
 * rows, then the (B, H + 2, W + 2) padded positions, then G guard rows; everything but the interior pixels zero — on which the
 * convolution is ONE token-major K20 product over k = (tap, channel): the row of position m for tap t = 3 dy + dx is row
 * m + (dy - 1)(W + 2) + (dx - 1) of the same buffer (a scalar offset per K-step, no im2col).
 *   mbv_conv_pad_rows:   dst rows (cleared by the caller) <- interior pixels of src (B, C, H, W); elem_size 4 (f32) or 2
 *   mbv_conv_unpad_rows: dst (B, C, H, W) <- interior pixels of the rows buffer src
 *   mbv_conv3x3_gemm32s: out_rows (rows buffer, cout channels; its interior positions hold the result) from rows (C channels)
 *                        and wm (cout, 9 C) f32 with wm[co][t C + ci] = weight[co][ci][dy][dx].  The DATA gradient is the same call
 *                        on the output gradient's rows with wm[ci][t cout + co] = weight[co][ci][2 - dy][2 - dx]; the WEIGHT
 *                        gradient is nine entries of mbv_gemm32s_tn_group (g = the output gradient's rows, x = rows shifted by
 *                        the tap).  C % 32 == 0, cout % 8 == 0, 16-byte aligned buffers; amax_*: absmax records (NULL = unscaled). */
int64_t mbv_conv_rows(int64_t batch, int64_t H, int64_t W);
int mbv_conv_pad_rows(const void* src, void* dst, int64_t batch, int64_t C, int64_t H, int64_t W, int32_t elem_size, void* stream);
int mbv_conv_unpad_rows(const void* src, void* dst, int64_t batch, int64_t C, int64_t H, int64_t W, int32_t elem_size,
                        void* stream);
int mbv_conv3x3_gemm32s(const float* rows, const float* wm, float* out_rows, int64_t batch, int64_t H, int64_t W, int64_t C,
                        int64_t cout, const uint32_t* amax_rows, const uint32_t* amax_w, uint32_t* amax_out, void* stream);
/* the 16-bit compute modes' form (K17; dtype 0 = bf16, 1 = fp16; out_rows 16-bit or f32): the same buffers and weight layout */
int mbv_conv3x3_gemm16(const void* rows, const void* wm, void* out_rows, int64_t batch, int64_t H, int64_t W, int64_t C,
                       int64_t cout, int32_t dtype, int32_t out_f32, void* stream);

/* The fp32 FFN's backward in one K20 launch: out (m, k) = act'(pre (m, k)) * (g (m, n) . w (n, k)) — the data gradient of the
 * output layer times the activation's derivative (act: 1 ReLU, 2 erf-GELU) — and, into `parts`
 * ((mbv_gemm32s_nn_part_rows(m, 1), k) f32, every element written), the partial column sums of `out`: their sum over the rows is
 * the bias gradient of the input layer.  Replaces the f32 library GEMM + mbv_act_bwd_colsum pass of mmcv FFN's backward
 * (/root/reference: mask_bev/models/networks/swin/swin.py:347-355). */
int64_t mbv_gemm32s_nn_part_rows(int64_t m, int32_t batch);
int mbv_gemm32s_nn_act(const float* g, const float* w, float* out, const float* pre, float* parts, size_t parts_bytes,
                       int64_t m, int64_t n, int64_t k, int64_t ldg, int64_t ldw, int64_t ldo, int64_t ldpre,
                       const uint32_t* amax_g, const uint32_t* amax_w, uint32_t* amax_out, int32_t act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKBEV_HIP_H_ */
