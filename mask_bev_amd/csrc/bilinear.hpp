// Bilinear sampling set-up shared by K8 (point_sample.hip) and the fused importance sampling of K10
// (select_points.hip): both must produce bit-identical sampled logits.
#pragma once

struct Bil {
  int o[4];
  float w[4];
};

// pixel = coord * size - 0.5 (grid_sample(2p - 1), align_corners=False), zero padding
__device__ __forceinline__ void bil_setup(float px, float py, int H, int W, Bil& b) {
  const float x = px * (float)W - 0.5f, y = py * (float)H - 0.5f;
  const float xf = floorf(x), yf = floorf(y);
  const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
  const float lx = x - xf, ly = y - yf;
  b.w[0] = (1.f - ly) * (1.f - lx);
  b.w[1] = (1.f - ly) * lx;
  b.w[2] = ly * (1.f - lx);
  b.w[3] = ly * lx;
  const bool xv0 = x0 >= 0 && x0 < W, xv1 = x1 >= 0 && x1 < W, yv0 = y0 >= 0 && y0 < H, yv1 = y1 >= 0 && y1 < H;
  b.o[0] = (yv0 && xv0) ? y0 * W + x0 : -1;
  b.o[1] = (yv0 && xv1) ? y0 * W + x1 : -1;
  b.o[2] = (yv1 && xv0) ? y1 * W + x0 : -1;
  b.o[3] = (yv1 && xv1) ? y1 * W + x1 : -1;
}


// The same sample with every tap index clamped into the map and the weight of an out-of-range tap set to zero instead
// of a -1 index the caller has to test: for coordinates in [0, 1) only x0 = -1 / x1 = W (and likewise y) can fall
// outside, so two compares per axis replace eight compares, four ANDs and the per-tap tests of the gather loop.  The
// sampled value is the same finite sum (an out-of-range tap adds w * v = 0 * finite).  Used where the taps are READ
// (K10's fused sampler, a VALU-bound kernel); the backward kernels keep the form above and skip invalid taps.
__device__ __forceinline__ void bil_setup_clamped(float px, float py, int H, int W, Bil& b) {
  const float x = px * (float)W - 0.5f, y = py * (float)H - 0.5f;
  const float xf = floorf(x), yf = floorf(y);
  const int x0 = (int)xf, y0 = (int)yf;
  const float lx = x - xf, ly = y - yf;
  const float wx0 = x0 >= 0 ? 1.f - lx : 0.f, wx1 = x0 + 1 < W ? lx : 0.f;
  const float wy0 = y0 >= 0 ? 1.f - ly : 0.f, wy1 = y0 + 1 < H ? ly : 0.f;
  // (24-bit multiplies: full rate, v_mul_lo_u32 is a quarter; the tile is at most 16 384 pixels)
  const int xa = max(x0, 0), xb = min(x0 + 1, W - 1), ya = __mul24(max(y0, 0), W), yb = __mul24(min(y0 + 1, H - 1), W);
  b.w[0] = wy0 * wx0; b.w[1] = wy0 * wx1; b.w[2] = wy1 * wx0; b.w[3] = wy1 * wx1;
  b.o[0] = ya + xa; b.o[1] = ya + xb; b.o[2] = yb + xa; b.o[3] = yb + xb;
}
