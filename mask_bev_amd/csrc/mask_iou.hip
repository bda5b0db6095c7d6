// K15 — mask IoU of matched (prediction, ground-truth) pairs at ground-truth resolution (metrics path, §8f-3).
//
// Replaces, inside MaskBevPanopticHead.update_mAP_metrics (mask_bev/models/head/mask_bev_panoptic_head.py:74-85):
//   F.interpolate(pred_masks, gt_size, mode='bilinear', align_corners=False) → sigmoid() > 0.5 →
//   batched_mask_iou(mask_targets, pred)     (mask_bev/evaluation/average_precision.py:78-81)
// which materialises (Q, ny, nx) f32 upsampled logits, a bool copy and two more f32 maps per sample and drags
// the whole thing through the metric objects.  Here one workgroup owns one pair: the (h, w) logit map sits in a
// 64 KB LDS tile, every ground-truth-resolution pixel is interpolated on the fly with PyTorch's
// upsample_bilinear2d arithmetic (source index max(scale·(dst+0.5)−0.5, 0), neighbour clamped at the border),
// thresholded (sigmoid(v) > 0.5 ⇔ v > 0) and compared with the bit-packed ground-truth mask; intersection and
// union are popcounts.  Integer / bit work, one pass over 32 KB of mask bits + 64 KB of logits per pair.
#include "common.hpp"

namespace {

constexpr int kThreads = 512;

__global__ void __launch_bounds__(kThreads) k_matched_mask_iou(const float* __restrict__ logits,
                                                               const int32_t* __restrict__ pred_row,
                                                               const uint32_t* __restrict__ gt_packed,
                                                               int64_t words_per_map,
                                                               const int32_t* __restrict__ gt_row, int h, int w, int H,
                                                               int W, int32_t* __restrict__ inter,
                                                               int32_t* __restrict__ uni) {
  __shared__ __attribute__((aligned(16))) float tile[16384];
  __shared__ int red[2][kThreads / 64];
  const int pair = blockIdx.x;
  const int gr = gt_row[pair];
  if (gr < 0) {                                   // unmatched prediction
    if (threadIdx.x == 0) { inter[pair] = 0; uni[pair] = 0; }
    return;
  }
  const float* src = logits + (int64_t)pred_row[pair] * h * w;
  for (int i = threadIdx.x; i < h * w; i += kThreads) tile[i] = src[i];
  __syncthreads();
  const uint32_t* bits = gt_packed + (int64_t)gr * words_per_map;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  int n_inter = 0, n_union = 0;
  const int64_t cells = (int64_t)H * W;
  for (int64_t p0 = 0; p0 < cells; p0 += kThreads) {
    const int64_t p = p0 + threadIdx.x;
    bool pred = false, gt = false;
    if (p < cells) {
      const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
      const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.f);
      const int y0 = (int)fy, x0 = (int)fx;
      const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float hy = 1.f - ly, hx = 1.f - lx;
      const float v = hy * (hx * tile[y0 * w + x0] + lx * tile[y0 * w + x0 + xp]) +
                      ly * (hx * tile[(y0 + yp) * w + x0] + lx * tile[(y0 + yp) * w + x0 + xp]);
      pred = v > 0.f;
      gt = (bits[p >> 5] >> (p & 31)) & 1u;
    }
    n_inter += __popcll(__ballot(pred && gt));     // wave-uniform counts (every lane holds the same number)
    n_union += __popcll(__ballot(pred || gt));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = n_inter; red[1][wave] = n_union; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0, b = 0;
    for (int i = 0; i < kThreads / 64; ++i) { a += red[0][i]; b += red[1][i]; }
    inter[pair] = a;
    uni[pair] = b;
  }
}

}  // namespace

extern "C" int mbv_matched_mask_iou(const float* logits, const int32_t* pred_row, const uint32_t* gt_packed,
                                    const int32_t* gt_row, int32_t num_pairs, int32_t h, int32_t w, int32_t H,
                                    int32_t W, int32_t* inter, int32_t* uni, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_pairs < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return MBV_ERR_BAD_ARG;
  if ((int64_t)h * w > 16384) return MBV_ERR_UNSUPPORTED;
  if (num_pairs == 0) return MBV_OK;
  if (!logits || !pred_row || !gt_packed || !gt_row || !inter || !uni) return MBV_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_matched_mask_iou, dim3(num_pairs), dim3(kThreads), 0, stream, logits, pred_row, gt_packed,
                     mbv_packed_mask_words(H, W), gt_row, h, w, H, W, inter, uni);
  MBV_CHECK_LAUNCH();
  return MBV_OK;
}
