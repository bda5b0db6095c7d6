/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the hard-voxelisation algorithm that the reference
 * reaches through `mmcv.ops.Voxelization` (mmcv==2.0.0, pinned at
 * /root/reference/Dockerfile:25; constructed at
 * /root/reference/mask_bev/models/encoders/mask_bev_encoders.py:69 and called
 * per scan at :98-103) preceded by the strict range pre-filter of
 * mask_bev_encoders.py:113-117.
 *
 * mmcv is NOT vendored under /root/reference, so this follows the published
 * algorithm of mmcv 2.0.0 `hard_voxelize_forward_cpu_kernel` /
 * `dynamic_voxelize_forward_cpu_kernel`:
 *   - per point, per dim j: c = floor((p[j] - range_min[j]) / voxel_size[j]) in
 *     f32 arithmetic; reject the point if c < 0 or c >= grid[j];
 *   - coordinates stored reversed (z, y, x);
 *   - voxel ids are handed out in order of first appearance in the point list;
 *   - a voxel keeps the first `max_points` points in input order;
 *   - once `max_voxels` voxels exist, points opening a new voxel are dropped.
 * PARITY UNPINNED for the upstream part: no reference test holds values for it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this. Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/*
 * points      : (n, dim) f32 row-major, dim >= 3 (x, y, z, ...)
 * range6      : x_min, y_min, z_min, x_max, y_max, z_max  (already rounded to f32)
 * vsize3      : voxel size x, y, z (f32)
 * grid3       : grid size x, y, z
 * prefilter   : 1 -> apply the strict `min < p < max` test of
 *               mask_bev_encoders.py:113-117 first
 * voxels      : (max_voxels, max_points, dim) f32, must be zero-filled by caller
 * coors       : (max_voxels, 3) i32 (z, y, x)
 * num_points  : (max_voxels,) i32, zero-filled by caller
 * point_voxel : optional (n,) i32 -> voxel id the point was stored in, -1 if dropped
 * point_slot  : optional (n,) i32 -> slot inside the voxel, -1 if dropped
 * returns the number of voxels.
 */
int mbv_oracle_hard_voxelize(const float* points, int64_t n, int dim,
                             const float* range6, const float* vsize3,
                             const int32_t* grid3, int max_points, int max_voxels,
                             int prefilter, float* voxels, int32_t* coors,
                             int32_t* num_points, int32_t* point_voxel,
                             int32_t* point_slot) {
  const int64_t gx = grid3[0], gy = grid3[1], gz = grid3[2];
  int32_t* lut = (int32_t*)malloc(sizeof(int32_t) * (size_t)(gx * gy * gz));
  if (!lut) return -1;
  memset(lut, 0xff, sizeof(int32_t) * (size_t)(gx * gy * gz)); /* -1 */
  int voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = points + i * dim;
    if (point_voxel) point_voxel[i] = -1;
    if (point_slot) point_slot[i] = -1;
    if (prefilter) {
      if (!(range6[0] < p[0] && p[0] < range6[3] && range6[1] < p[1] &&
            p[1] < range6[4] && range6[2] < p[2] && p[2] < range6[5]))
        continue;
    }
    int c[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      /* f32 subtract, f32 IEEE divide, floor: as the upstream kernel */
      float q = (p[j] - range6[j]) / vsize3[j];
      int cj = (int)floorf(q);
      if (cj < 0 || cj >= grid3[j]) {
        failed = 1;
        break;
      }
      c[j] = cj;
    }
    if (failed) continue;
    const int64_t cell = ((int64_t)c[2] * gy + c[1]) * gx + c[0];
    int vid = lut[cell];
    if (vid == -1) {
      if (max_voxels != -1 && voxel_num >= max_voxels) continue;
      vid = voxel_num++;
      lut[cell] = vid;
      coors[vid * 3 + 0] = c[2];
      coors[vid * 3 + 1] = c[1];
      coors[vid * 3 + 2] = c[0];
    }
    int num = num_points[vid];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((int64_t)vid * max_points + num) * dim, p,
             sizeof(float) * (size_t)dim);
      num_points[vid] = num + 1;
      if (point_voxel) point_voxel[i] = vid;
      if (point_slot) point_slot[i] = num;
    }
  }
  free(lut);
  return voxel_num;
}
