// fit_stage.hip -- per-instance clustering (adaptive range segmentation) and
// L-shape rectangle search + KITTI box assembly.
//
// a13  rectangle_fitting.py:161-191: the reference builds C_i = {j : d_ij <= R_i}
//      and merges intersecting sets until a fixed point.  i is in C_i and j in
//      C_i links i and j, so the result is the set of connected components of
//      the graph {i--j : d_ij <= R_i or d_ij <= R_j}; the merge loop leaves them
//      ordered by their smallest index.  Here: one workgroup per instance, a
//      lock-free union-find whose roots are always the smallest index of their
//      set (atomicMin link), x/y tiles staged through LDS.
// a14  rectangle_fitting.py:83-159: 89 candidate headings; one wave per heading,
//      lanes stride the cluster's points, three sweeps (extent, mean, variance)
//      with fp64 wave reductions; first strict maximum wins.
// a15  my_loader.py:633-702: box from the rectangle, fp64.
#include "common.hpp"

namespace {

constexpr int CT = 512;            // threads per clustering workgroup
constexpr int TJ = 1024;           // j-tile
constexpr int LDS_PARENT = 12288;  // parents kept in LDS up to this many points

__device__ __forceinline__ int ld_parent(const int *p, int i) {
  return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int *parent, int a) {
  int p = ld_parent(parent, a);
  while (p != a) { a = p; p = ld_parent(parent, a); }
  return a;
}
// returns the root of the merged set as seen by this thread
__device__ __forceinline__ int uf_unite(int *parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return a;
    if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
    const int old = atomicMin(parent + a, b);
    if (old == a) return b;
    a = old;                                        // someone re-linked a: retry
  }
}

__global__ __launch_bounds__(CT) void k_range_cluster(
    const double *__restrict__ px, const double *__restrict__ py,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, double R0,
    double Rd, int *__restrict__ label) {
  __shared__ double sxj[TJ], syj[TJ], sRj[TJ];
  __shared__ int s_parent[LDS_PARENT];
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  if (n == 0) return;
  const long long base = seg_base[s];
  int *glabel = label + base;
  int *parent = (n <= LDS_PARENT) ? s_parent : glabel;
  for (int i = threadIdx.x; i < n; i += CT) parent[i] = i;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += CT) {
    const int i = c0 + threadIdx.x;
    const bool valid = i < n;
    double xi = 0.0, yi = 0.0, Ri = 0.0;
    if (valid) {
      xi = px[base + i];
      yi = py[base + i];
      Ri = R0 + Rd * sqrt(xi * xi + yi * yi);       // rectangle_fitting.py:167
    }
    int ri = valid ? uf_find(parent, i) : -1;
    const int jend = min(n, c0 + CT);               // only j < i matter
    for (int t0 = 0; t0 < jend; t0 += TJ) {
      const int m = min(TJ, jend - t0);
      __syncthreads();
      for (int k = threadIdx.x; k < m; k += CT) {
        const double xj = px[base + t0 + k], yj = py[base + t0 + k];
        sxj[k] = xj;
        syj[k] = yj;
        sRj[k] = R0 + Rd * sqrt(xj * xj + yj * yj);
      }
      __syncthreads();
      if (valid) {
        const int lim = min(m, i - t0);
        for (int k = 0; k < lim; k++) {
          const double dx = xi - sxj[k], dy = yi - syj[k];
          const double d = sqrt(dx * dx + dy * dy);   // rectangle_fitting.py:169
          if (d <= Ri || d <= sRj[k]) {
            const int j = t0 + k;
            if (ld_parent(parent, j) != ri) ri = uf_unite(parent, ri, j);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += CT) {
    const int r = uf_find(parent, i);
    if (parent == s_parent) glabel[i] = r;
    else atomicMin(glabel + i, r);                  // compress towards the root
  }
}

// ---------------------------------------------------------------- a14/a15
constexpr int FT = 256;
constexpr int FW = FT / 64;
constexpr int MAXTH = 128;
constexpr int LDS_MEMBERS = 3072;   // cluster members cached in LDS (48 KB)

struct Ext { double c1min, c1max, c2min, c2max; };

__device__ __forceinline__ void cross_point(double a0, double a1, double b0, double b1,
                                            double c0, double c1, double &x, double &y) {
  // my_loader.py:699-702
  x = (b0 * -c1 - b1 * -c0) / (a0 * b1 - a1 * b0);
  y = (a1 * -c0 - a0 * -c1) / (a0 * b1 - a1 * b0);
}

__global__ __launch_bounds__(FT) void k_lshape_fit(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const int *__restrict__ label,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, int max_inst,
    const ViewCalib *__restrict__ calib, const int *__restrict__ inst_class,
    const int *__restrict__ inst_is_car, const float *__restrict__ inst_box,
    const float *__restrict__ inst_score, int n_theta, double dtheta, double car_aspect_max,
    double *__restrict__ gsx, double *__restrict__ gsy, int *__restrict__ sroot, int cap_rows,
    double *__restrict__ rows, int *__restrict__ n_rows, uint32_t *__restrict__ status) {
  __shared__ double lx[LDS_MEMBERS], ly[LDS_MEMBERS];
  __shared__ double s_cost[MAXTH];
  __shared__ double s_red[FW];
  __shared__ double s_ext[FW][4];
  __shared__ int s_w[FW];
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  if (n == 0) return;
  const long long base = seg_base[s];
  const int v = s / max_inst, jinst = s - v * max_inst;
  const int wave = threadIdx.x >> 6, lane = lane_id();

  // max z over ALL instance points (my_loader.py:647-648)
  double zm = -INFINITY;
  for (int i = threadIdx.x; i < n; i += FT) zm = fmax(zm, pz[base + i]);
  zm = wave_max_d(zm);
  if (lane == 0) s_red[wave] = zm;
  __syncthreads();
  const double zmax = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();

  // ordered list of cluster roots
  int nroots = 0;
  for (int t0 = 0; t0 < n; t0 += FT) {
    const int i = t0 + threadIdx.x;
    const bool f = (i < n) && (label[base + i] == i);
    int tot;
    const int r = block_rank<FW>(f, s_w, tot);
    if (f) sroot[base + nroots + r] = i;
    nroots += tot;
  }
  __syncthreads();

  for (int kc = 0; kc < nroots; kc++) {
    const int root = sroot[base + kc];
    // gather the cluster's members in index order
    int m = 0;
    for (int t0 = 0; t0 < n; t0 += FT) {
      const int i = t0 + threadIdx.x;
      const bool f = (i < n) && (label[base + i] == root);
      int tot;
      const int r = block_rank<FW>(f, s_w, tot);
      if (f) {
        const int d = m + r;
        const double x = px[base + i], y = py[base + i];
        gsx[base + d] = x;
        gsy[base + d] = y;
        if (d < LDS_MEMBERS) { lx[d] = x; ly[d] = y; }
      }
      m += tot;
    }
    __syncthreads();
    const bool in_lds = m <= LDS_MEMBERS;
    const double *mx = in_lds ? lx : gsx + base;
    const double *my = in_lds ? ly : gsy + base;

    // 89 headings, one wave each (rectangle_fitting.py:119-136)
    for (int th = wave; th < n_theta; th += FW) {
      const double theta = (double)th * dtheta;
      const double ct = cos(theta), st = sin(theta);
      double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
      for (int i = lane; i < m; i += 64) {
        const double x = mx[i], y = my[i];
        const double c1 = x * ct + y * st;
        const double c2 = x * (-st) + y * ct;
        a0 = fmin(a0, c1); a1 = fmax(a1, c1);
        b0 = fmin(b0, c2); b1 = fmax(b1, c2);
      }
      a0 = wave_min_d(a0); a1 = wave_max_d(a1);
      b0 = wave_min_d(b0); b1 = wave_max_d(b1);
      // rectangle_fitting.py:89-99: D1/D2, split into E1/E2
      double s1 = 0.0, s2 = 0.0;
      int n1 = 0, n2 = 0;
      for (int i = lane; i < m; i += 64) {
        const double x = mx[i], y = my[i];
        const double c1 = x * ct + y * st;
        const double c2 = x * (-st) + y * ct;
        const double d1 = fmin(fabs(a1 - c1), fabs(c1 - a0));
        const double d2 = fmin(fabs(b1 - c2), fabs(c2 - b0));
        if (d1 < d2) { s1 += d1; n1++; } else { s2 += d2; n2++; }
      }
      s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
      n1 = wave_sum_i(n1); n2 = wave_sum_i(n2);
      const double m1 = n1 ? s1 / (double)n1 : 0.0, m2 = n2 ? s2 / (double)n2 : 0.0;
      double q1 = 0.0, q2 = 0.0;
      for (int i = lane; i < m; i += 64) {
        const double x = mx[i], y = my[i];
        const double c1 = x * ct + y * st;
        const double c2 = x * (-st) + y * ct;
        const double d1 = fmin(fabs(a1 - c1), fabs(c1 - a0));
        const double d2 = fmin(fabs(b1 - c2), fabs(c2 - b0));
        if (d1 < d2) { const double u = d1 - m1; q1 += u * u; }
        else { const double u = d2 - m2; q2 += u * u; }
      }
      q1 = wave_sum_d(q1); q2 = wave_sum_d(q2);
      double V1 = 0.0, V2 = 0.0;
      if (n1) V1 = -(q1 / (double)n1);
      if (n2) V2 = -(q2 / (double)n2);
      if (lane == 0) s_cost[th] = V1 + V2;
    }
    __syncthreads();
    // first strict maximum (rectangle_fitting.py:135-136)
    int best = 0;
    {
      double bc = -INFINITY;
      bool have = false;
      for (int th = 0; th < n_theta; th++) {
        const double c = s_cost[th];
        if (bc < c) { bc = c; best = th; have = true; }
      }
      (void)have;
    }
    // extents at the best heading (rectangle_fitting.py:139-157)
    const double thb = (double)best * dtheta;
    const double sin_s = sin(thb), cos_s = cos(thb);
    {
      double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
      for (int i = threadIdx.x; i < m; i += FT) {
        const double x = mx[i], y = my[i];
        const double c1 = x * cos_s + y * sin_s;
        const double c2 = x * (-sin_s) + y * cos_s;
        a0 = fmin(a0, c1); a1 = fmax(a1, c1);
        b0 = fmin(b0, c2); b1 = fmax(b1, c2);
      }
      a0 = wave_min_d(a0); a1 = wave_max_d(a1);
      b0 = wave_min_d(b0); b1 = wave_max_d(b1);
      if (lane == 0) { s_ext[wave][0] = a0; s_ext[wave][1] = a1; s_ext[wave][2] = b0; s_ext[wave][3] = b1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double c1min = s_ext[0][0], c1max = s_ext[0][1], c2min = s_ext[0][2], c2max = s_ext[0][3];
      for (int w = 1; w < FW; w++) {
        c1min = fmin(c1min, s_ext[w][0]); c1max = fmax(c1max, s_ext[w][1]);
        c2min = fmin(c2min, s_ext[w][2]); c2max = fmax(c2max, s_ext[w][3]);
      }
      const double a[4] = {cos_s, -sin_s, cos_s, -sin_s};
      const double b[4] = {sin_s, cos_s, sin_s, cos_s};
      const double c[4] = {c1min, c2min, c1max, c2max};
      double cx[4], cy[4];                         // my_loader.py:686-697
      cross_point(a[0], a[1], b[0], b[1], c[0], c[1], cx[0], cy[0]);
      cross_point(a[1], a[2], b[1], b[2], c[1], c[2], cx[1], cy[1]);
      cross_point(a[2], a[3], b[2], b[3], c[2], c[3], cx[2], cy[2]);
      cross_point(a[3], a[0], b[3], b[0], c[3], c[0], cx[3], cy[3]);
      double center_x = (cx[0] + cx[2]) / 2.0;
      double center_y = (cy[0] + cy[2]) / 2.0;
      const double center_z = zmax / 2.0 - 1.5;
      const double height = zmax;
      const double e03x = cx[0] - cx[3], e03y = cy[0] - cy[3];
      const double e01x = cx[0] - cx[1], e01y = cy[0] - cy[1];
      const double l1 = sqrt(e03x * e03x + e03y * e03y);
      const double l2 = sqrt(e01x * e01x + e01y * e01y);
      bool skip = false;
      if (inst_is_car[s] && (l1 / l2 > car_aspect_max || l2 / l1 > car_aspect_max)) skip = true;
      double length = 0.0, width = 0.0, rotation = 0.0;
      if (l1 >= l2) {
        length = l1; width = l2;
        rotation = atan((cy[3] - cy[0]) / (cx[3] - cx[0] + 1e-8));
      } else if (l1 < l2) {
        length = l2; width = l1;
        rotation = atan((cy[1] - cy[0]) / (cx[1] - cx[0] + 1e-8));
      } else {
        skip = true;                               // NaN extents
      }
      if (!skip) {
        const double kPi = 3.141592653589793;
        rotation = -rotation - kPi / 2.0;
        const double theta = atan(-center_x / (center_y + 1e-8));
        const double alpha = rotation - theta;
        const float *M = calib[v].M43;             // calibration_kitti.py:104-112 (fp64 in)
        const double rx = ((center_x * (double)M[0] + center_y * (double)M[3]) + center_z * (double)M[6]) + (double)M[9];
        const double ry = ((center_x * (double)M[1] + center_y * (double)M[4]) + center_z * (double)M[7]) + (double)M[10];
        const double rz = ((center_x * (double)M[2] + center_y * (double)M[5]) + center_z * (double)M[8]) + (double)M[11];
        const int slot = atomicAdd(n_rows, 1);
        if (slot < cap_rows) {
          double *o = rows + (size_t)slot * DFU3D_ROW_DOUBLES;
          o[0] = (double)v; o[1] = (double)jinst; o[2] = (double)kc;
          o[3] = (double)inst_class[s]; o[4] = alpha;
          o[5] = (double)inst_box[s * 4 + 0]; o[6] = (double)inst_box[s * 4 + 1];
          o[7] = (double)inst_box[s * 4 + 2]; o[8] = (double)inst_box[s * 4 + 3];
          o[9] = height; o[10] = width; o[11] = length;
          o[12] = rx; o[13] = ry; o[14] = rz; o[15] = rotation;
          o[16] = (double)inst_score[s]; o[17] = (double)m;
        } else {
          atomicOr(status, DFU3D_ST_ROW_OVERFLOW);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int dfu3d_range_cluster(const double *px, const double *py, const int64_t *seg_base,
                                   const int32_t *seg_cnt, int32_t S, double R0, double Rd,
                                   int32_t *label, void *stream) {
  if (!px || !py || !seg_base || !seg_cnt || !label) return DFU3D_EINVAL;
  if (S <= 0) return DFU3D_EINVAL;
  hipLaunchKernelGGL(k_range_cluster, dim3(S), dim3(CT), 0, (hipStream_t)stream, px, py,
                     (const long long *)seg_base, seg_cnt, R0, Rd, label);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_lshape_fit(const double *px, const double *py, const double *pz,
                                const int32_t *label, const int64_t *seg_base,
                                const int32_t *seg_cnt, int32_t S, int32_t max_inst,
                                const float *calib, const int32_t *inst_class,
                                const int32_t *inst_is_car, const float *inst_box,
                                const float *inst_score, int32_t n_theta, double dtheta,
                                double car_aspect_max, double *sx, double *sy, int32_t *sroot,
                                int32_t cap_rows, double *rows, int32_t *n_rows,
                                uint32_t *status, void *stream) {
  if (!px || !py || !pz || !label || !seg_base || !seg_cnt || !calib || !inst_class ||
      !inst_is_car || !inst_box || !inst_score || !sx || !sy || !sroot || !rows || !n_rows ||
      !status)
    return DFU3D_EINVAL;
  if (S <= 0 || max_inst <= 0 || cap_rows <= 0 || n_theta <= 0) return DFU3D_EINVAL;
  if (n_theta > MAXTH) return DFU3D_ERANGE;
  hipLaunchKernelGGL(k_lshape_fit, dim3(S), dim3(FT), 0, (hipStream_t)stream, px, py, pz, label,
                     (const long long *)seg_base, seg_cnt, max_inst, (const ViewCalib *)calib,
                     inst_class, inst_is_car, inst_box, inst_score, n_theta, dtheta,
                     car_aspect_max, sx, sy, sroot, cap_rows, rows, n_rows, status);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_version(void) { return DFU3D_VERSION; }

extern "C" const char *dfu3d_strerror(int code) {
  switch (code) {
    case DFU3D_OK: return "ok";
    case DFU3D_EINVAL: return "invalid argument";
    case DFU3D_ELAUNCH: return "kernel launch failed";
    case DFU3D_ERANGE: return "size exceeds a compiled-in limit";
    default: return "unknown error";
  }
}
