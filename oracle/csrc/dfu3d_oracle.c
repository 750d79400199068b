/*
 * dfu3d_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle, CPU).
 *
 * Plain-C restatement of the loop-heavy / rounding-critical leaves of DFU3D's
 * pseudo-box path (reference: tools/PENet/dataloaders/my_loader.py,
 * calibration_kitti.py, rectangle_fitting/rectangle_fitting.py and the
 * third-party leaves they call: Open3D, spconv, torch).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (dfu3d_amd/) never does.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -shared -fPIC (see oracle/build.py).
 * -ffp-contract=off: every rounding below is written out explicitly; fmaf()
 * is used exactly where the reference's BLAS sgemm fuses (measured: numpy
 * 2.2 / OpenBLAS 0.3.29 sgemm == sequential-k FMA chain).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- a3: calibration_kitti.py:104-112 lidar_to_rect (fp32 sgemm) ----------
 * out[i][j] = fma(1, M[3][j], fma(z, M[2][j], fma(y, M[1][j], x*M[0][j])))
 * M43 is the (4,3) row-major product V2C^T @ R0^T.                            */
void orc_lidar_to_rect_f32(const float *xyz, int64_t stride, const float *M43,
                           float *out, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    const float x = xyz[i * stride + 0], y = xyz[i * stride + 1],
                z = xyz[i * stride + 2];
    for (int j = 0; j < 3; j++) {
      float acc = x * M43[0 * 3 + j];
      acc = fmaf(y, M43[1 * 3 + j], acc);
      acc = fmaf(z, M43[2 * 3 + j], acc);
      acc = fmaf(1.0f, M43[3 * 3 + j], acc);
      out[i * 3 + j] = acc;
    }
  }
}

/* ---- a3: calibration_kitti.py:114-123 rect_to_img (fp32) ------------------
 * h = [rect,1] @ P2^T (FMA chain), uv = h[0:2] / rect_z, depth = h[2]-P2[2][3] */
void orc_rect_to_img_f32(const float *rect, const float *P2, float *uv,
                         float *depth, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    const float x = rect[i * 3 + 0], y = rect[i * 3 + 1], z = rect[i * 3 + 2];
    float h[3];
    for (int j = 0; j < 3; j++) {
      float acc = x * P2[j * 4 + 0];
      acc = fmaf(y, P2[j * 4 + 1], acc);
      acc = fmaf(z, P2[j * 4 + 2], acc);
      acc = fmaf(1.0f, P2[j * 4 + 3], acc);
      h[j] = acc;
    }
    uv[i * 2 + 0] = h[0] / z;
    uv[i * 2 + 1] = h[1] / z;
    depth[i] = h[2] - P2[2 * 4 + 3];
  }
}

/* ---- a3/a7: float64 points through a float32-valued (4,m) matrix -----------
 * calibration_kitti.py:89-102 rect_to_lidar and the float64 path of :104-112:
 * np.dot([xyz,1] float64, M) is a dgemm; measured here (numpy 2.2 / OpenBLAS
 * 0.3.29) it is the same sequential-k FMA chain as the sgemm, bit for bit
 * (tests/test_oracle_golden.py::test_fp64_chain_matches_numpy_dgemm).
 * M: (4,ld) row-major float32, the first m columns are used.                   */
void orc_hom_dot_f64(const double *xyz, const float *M, int64_t ld, int64_t m,
                     double *out, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    const double x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    for (int64_t j = 0; j < m; j++) {
      double acc = x * (double)M[0 * ld + j];
      acc = fma(y, (double)M[1 * ld + j], acc);
      acc = fma(z, (double)M[2 * ld + j], acc);
      acc = fma(1.0, (double)M[3 * ld + j], acc);
      out[i * m + j] = acc;
    }
  }
}

/* (4,3) = V2C^T (4,3... as (4,3)x(3,3)) product, sequential-k FMA chain, the
 * arithmetic numpy's sgemm performs for np.dot(V2C.T, R0.T)
 * (calibration_kitti.py:110).  V2C is (3,4) row-major, R0 (3,3) row-major.   */
void orc_m43_f32(const float *V2C, const float *R0, float *M43) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 3; j++) {
      /* A[i][k] = V2C[k][i]; B[k][j] = R0[j][k] */
      float acc = V2C[0 * 4 + i] * R0[j * 3 + 0];
      acc = fmaf(V2C[1 * 4 + i], R0[j * 3 + 1], acc);
      acc = fmaf(V2C[2 * 4 + i], R0[j * 3 + 2], acc);
      M43[i * 3 + j] = acc;
    }
}

/* ---- a8: spconv Point2VoxelCPU3d.point_to_voxel (my_loader.py:22-83) ------
 * Points in input order; c_j = (int)floor((p_j - range_min_j) / vsize_j) with
 * float-valued vsize/range constants promoted to double; point skipped when
 * any c_j is outside [0, grid_j); a new voxel id is handed out on first sight
 * until max_voxels; a point is appended while its voxel holds < max_points.
 * Outputs: vox_of_pt[i] = voxel id the point was STORED in (or -1),
 * n_vox.  Voxel ids are in first-seen order.  `table` is a caller-provided
 * int32 scratch of grid[0]*grid[1]*grid[2] entries filled with -1 on entry and
 * restored to -1 on exit.                                                     */
int64_t orc_voxelize(const double *coords, int64_t stride, int64_t n,
                     const double *vsize, const double *rmin,
                     const int32_t *grid, int32_t max_points,
                     int32_t max_voxels, int32_t *table, int32_t *vox_of_pt,
                     int32_t *vox_count, int64_t *vox_cell) {
  int64_t nv = 0;
  for (int64_t i = 0; i < n; i++) {
    int ok = 1;
    int64_t cell = 0;
    for (int j = 0; j < 3; j++) {
      double q = floor((coords[i * stride + j] - rmin[j]) / vsize[j]);
      /* NaN / out of int range -> fails the range test like the C++ int cast */
      if (!(q >= 0.0 && q < (double)grid[j])) { ok = 0; break; }
      cell = cell * grid[j] + (int64_t)q;
    }
    vox_of_pt[i] = -1;
    if (!ok) continue;
    int32_t v = table[cell];
    if (v < 0) {
      if (nv >= max_voxels) continue;
      v = (int32_t)nv++;
      table[cell] = v;
      vox_count[v] = 0;
      vox_cell[v] = cell;
    }
    if (vox_count[v] < max_points) {
      vox_count[v]++;
      vox_of_pt[i] = v;
    }
  }
  for (int64_t v = 0; v < nv; v++) table[vox_cell[v]] = -1;
  return nv;
}

/* representative per voxel = first argmin of key over the points stored in the
 * voxel (my_loader.py:255-258, 270-273: np.argmin over voxel[:pt_n, 10]).     */
void orc_voxel_argmin(const int32_t *vox_of_pt, const double *key, int64_t n,
                      int64_t nv, int64_t *rep) {
  for (int64_t v = 0; v < nv; v++) rep[v] = -1;
  for (int64_t i = 0; i < n; i++) {
    int32_t v = vox_of_pt[i];
    if (v < 0) continue;
    if (rep[v] < 0 || key[i] < key[rep[v]]) rep[v] = i;
  }
}

/* ---- a10: Open3D remove_radius_outlier (my_loader.py:581-599) -------------
 * keep[i] = #{j : ((dx^2)+dy^2)+dz^2 < r^2, j==i included} > nb_points.       */
void orc_radius_outlier(const double *p, int64_t n, double radius,
                        int32_t nb_points, uint8_t *keep) {
  const double r2 = radius * radius;
  for (int64_t i = 0; i < n; i++) {
    int32_t cnt = 0;
    const double x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    for (int64_t j = 0; j < n && cnt <= nb_points; j++) {
      const double dx = x - p[j * 3], dy = y - p[j * 3 + 1],
                   dz = z - p[j * 3 + 2];
      double d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      if (d < r2) cnt++;
    }
    keep[i] = cnt > nb_points;
  }
}

/* ---- a11: Open3D remove_statistical_outlier (my_loader0.py:735, dormant) --
 * mean_d[i] = mean over the k nearest (self included) of sqrt(d2).            */
static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}
void orc_knn_mean_dist(const double *p, int64_t n, int32_t k, double *mean_d) {
  /* the k smallest squared distances of every point, ascending (what a full sort of all n would put in front; kept in
   * a k-entry insertion list so that a 50 000-point cloud takes seconds, not minutes), summed in that order */
  const int64_t kk = k < n ? k : n;
  double *best = (double *)malloc(sizeof(double) * (size_t)(kk > 0 ? kk : 1));
  for (int64_t i = 0; i < n; i++) {
    int64_t nb = 0;
    for (int64_t j = 0; j < n; j++) {
      const double dx = p[i * 3] - p[j * 3], dy = p[i * 3 + 1] - p[j * 3 + 1],
                   dz = p[i * 3 + 2] - p[j * 3 + 2];
      double d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      if (nb < kk || d < best[nb - 1]) {
        int64_t q = nb < kk ? nb : kk - 1;
        while (q > 0 && best[q - 1] > d) { best[q] = best[q - 1]; q--; }
        best[q] = d;
        if (nb < kk) nb++;
      }
    }
    double s = 0.0;
    for (int64_t j = 0; j < nb; j++) s += sqrt(best[j]);
    mean_d[i] = nb > 0 ? s / (double)nb : -1.0;
  }
  free(best);
}

/* ---- a12: BallQuery (my_loader.py:489-494) --------------------------------
 * keep[i] = min_j sqrt(((dx^2)+dy^2)+dz^2) < C   (roi_max_dim == 0, H12).     */
void orc_ball_query(const double *p1, int64_t n1, const double *p2, int64_t n2,
                    double C, uint8_t *keep) {
  for (int64_t i = 0; i < n1; i++) {
    double best = INFINITY;
    for (int64_t j = 0; j < n2; j++) {
      const double dx = p1[i * 3] - p2[j * 3], dy = p1[i * 3 + 1] - p2[j * 3 + 1],
                   dz = p1[i * 3 + 2] - p2[j * 3 + 2];
      double d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      d = sqrt(d);
      if (d < best) best = d;
    }
    keep[i] = best < C;
  }
}

/* ---- a13: _adoptive_range_segmentation (rectangle_fitting.py:161-191) -----
 * C_i = {j : sqrt(dx^2+dy^2) <= R0 + Rd*sqrt(x_i^2+y_i^2)}; merging all
 * intersecting sets == connected components of the graph with an edge i--j
 * whenever j in C_i or i in C_j.  label[i] = smallest index of i's component
 * (the reference's output order is ascending smallest index).      */
static int32_t uf_find(int32_t *parent, int32_t a) {
  while (parent[a] != a) {
    parent[a] = parent[parent[a]];
    a = parent[a];
  }
  return a;
}
void orc_range_cluster(const double *x, const double *y, int64_t n, double R0,
                       double Rd, int32_t *label) {
  double *R = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) {
    label[i] = (int32_t)i;
    R[i] = R0 + Rd * sqrt(x[i] * x[i] + y[i] * y[i]);
  }
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < i; j++) {
      const double dx = x[i] - x[j], dy = y[i] - y[j];
      const double d = sqrt(dx * dx + dy * dy);
      if (d <= R[i] || d <= R[j]) {
        int32_t a = uf_find(label, (int32_t)i), b = uf_find(label, (int32_t)j);
        if (a < b) label[b] = a;
        else if (b < a) label[a] = b;
      }
    }
  for (int64_t i = 0; i < n; i++) label[i] = uf_find(label, (int32_t)i);
  free(R);
}
