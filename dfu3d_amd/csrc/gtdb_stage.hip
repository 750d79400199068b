// gtdb_stage.hip -- SURVEY.md §8 row f-2: the step right after the pseudo labels,
// OpenPCDet's ground-truth database (pcdet/datasets/kitti/kitti_dataset.py:284-331):
// for every labelled box the LiDAR points inside it, relative to the box centre.
//
// Point-in-box test: pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-140
// (points_in_boxes_cpu, the variant kitti_dataset.py:311 calls): float32 box and
// point, |z - cz| <= dz/2 compared in double, rotation by -heading in float32,
// |local| < d/2 + MARGIN (MARGIN = float32 1e-2) compared in double.
// cos/sin of the heading: evaluated in double and rounded to float32 (the
// reference's `cos(-rot_angle)` resolves to libm's cosf or cos depending on the
// headers in scope; both are within one float32 ulp of this choice -- hazard noted
// in DESIGN.md, row f-2).
#include "common.hpp"

namespace {

struct BoxF {
  float cx, cy, cz, cosa, sina;
  double hz, hx, hy;          // dz/2, dx/2 + MARGIN, dy/2 + MARGIN (double, as the reference compares)
  double c64x, c64y, c64z;    // float64 centre: gt_points[:, :3] -= gt_boxes[i, :3] (kitti_dataset.py:319)
};

__device__ __forceinline__ BoxF load_box(const double *b) {
  BoxF q;
  q.c64x = b[0]; q.c64y = b[1]; q.c64z = b[2];
  q.cx = (float)b[0]; q.cy = (float)b[1]; q.cz = (float)b[2];          // boxes.float()
  const float dx = (float)b[3], dy = (float)b[4], dz = (float)b[5], rz = (float)b[6];
  const float MARGIN = 1e-2f;
  q.cosa = (float)cos((double)(-rz));
  q.sina = (float)sin((double)(-rz));
  q.hz = (double)dz / 2.0;
  q.hx = (double)dx / 2.0 + (double)MARGIN;
  q.hy = (double)dy / 2.0 + (double)MARGIN;
  return q;
}

__device__ __forceinline__ bool pt_in_box(const BoxF &q, float x, float y, float z) {
  if ((double)fabsf(z - q.cz) > q.hz) return false;
  const float sx = x - q.cx, sy = y - q.cy;
  const float lx = sx * q.cosa + sy * (-q.sina);
  const float ly = sx * q.sina + sy * q.cosa;
  return ((double)fabsf(lx) < q.hx) && ((double)fabsf(ly) < q.hy);
}

// (B, n) int32 indicator, the return value of roiaware_pool3d_utils.points_in_boxes_cpu
__global__ __launch_bounds__(256) void k_pib_mask(const float *__restrict__ pts, int n, int stride,
                                                  const double *__restrict__ boxes, int B,
                                                  int *__restrict__ out) {
  const int b = blockIdx.y;
  const BoxF q = load_box(boxes + (size_t)b * 7);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float *p = pts + (size_t)i * stride;
    out[(size_t)b * n + i] = pt_in_box(q, p[0], p[1], p[2]) ? 1 : 0;
  }
}

constexpr int GT = 1024;   // threads per box workgroup
constexpr int GE = 4;      // consecutive points per thread

__global__ __launch_bounds__(GT) void k_pib_count(const float4 *__restrict__ pts,
                                                  const int *__restrict__ pt_off,
                                                  const int *__restrict__ box_frame,
                                                  const double *__restrict__ boxes,
                                                  int *__restrict__ cnt) {
  __shared__ int s_w[GT / 64];
  const int b = blockIdx.x;
  const int f = box_frame[b];
  const int p0 = pt_off[f], n = pt_off[f + 1] - p0;
  const BoxF q = load_box(boxes + (size_t)b * 7);
  int c = 0;
  for (int i = threadIdx.x; i < n; i += GT) {
    const float4 p = pts[p0 + i];
    c += pt_in_box(q, p.x, p.y, p.z) ? 1 : 0;
  }
  c = wave_sum_i(c);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < GT / 64; w++) t += s_w[w];
    cnt[b] = t;
  }
}

__global__ __launch_bounds__(1024) void k_pib_scan(int B, const int *__restrict__ cnt,
                                                   long long *__restrict__ off, long long cap,
                                                   uint32_t *__restrict__ status) {
  __shared__ int s_w[16];
  long long running = 0;
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const int v = (b < B) ? cnt[b] : 0;
    int tot;
    const int ex = block_excl_scan<16>(v, s_w, tot);
    if (b < B) off[b] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) {
    off[B] = running;
    if (running > cap) atomicOr(status, DFU3D_ST_POOL_OVERFLOW);
  }
}

__global__ __launch_bounds__(GT) void k_pib_fill(const float4 *__restrict__ pts,
                                                 const int *__restrict__ pt_off,
                                                 const int *__restrict__ box_frame,
                                                 const double *__restrict__ boxes,
                                                 const long long *__restrict__ off, long long cap,
                                                 int *__restrict__ idx_out,
                                                 float4 *__restrict__ gt_pts) {
  __shared__ int s_w[GT / 64];
  const int b = blockIdx.x;
  const int f = box_frame[b];
  const int p0 = pt_off[f], n = pt_off[f + 1] - p0;
  const BoxF q = load_box(boxes + (size_t)b * 7);
  const long long o = off[b];
  if (off[b + 1] == o) return;                         // nothing inside: no second sweep
  int running = 0;
  for (int t0 = 0; t0 < n; t0 += GT * GE) {
    const int i0 = t0 + threadIdx.x * GE;
    float4 p[GE];
    bool in[GE];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < GE; k++) {
      p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      in[k] = false;
      if (i0 + k < n) {
        p[k] = pts[p0 + i0 + k];
        in[k] = pt_in_box(q, p[k].x, p[k].y, p[k].z);
      }
      mine += in[k] ? 1 : 0;
    }
    int tot;
    int r = block_excl_scan<GT / 64>(mine, s_w, tot);
#pragma unroll
    for (int k = 0; k < GE; k++) {
      if (in[k]) {
        const long long d = o + running + r;
        if (d < cap) {
          idx_out[d] = i0 + k;
          // float32 array minus float64 centre, computed in double, stored as float32
          gt_pts[d] = make_float4((float)((double)p[k].x - q.c64x), (float)((double)p[k].y - q.c64y),
                                  (float)((double)p[k].z - q.c64z), p[k].w);
        }
        r++;
      }
    }
    running += tot;
  }
}

}  // namespace

extern "C" int dfu3d_points_in_boxes_mask(const float *pts, int32_t n, int32_t pt_stride,
                                          const double *boxes, int32_t B, int32_t *out,
                                          void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!pts || !boxes || !out) return DFU3D_EINVAL;
  if (n < 0 || B < 0 || pt_stride < 3) return DFU3D_EINVAL;
  if (n == 0 || B == 0) return DFU3D_OK;
  if (B > 65535) return DFU3D_ERANGE;
  const int gx = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(k_pib_mask, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, pts, n,
                     pt_stride, boxes, B, out);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_gt_database(const float *points, const int32_t *pt_off,
                                 const int32_t *box_frame, const double *boxes, int32_t Bt,
                                 int32_t *box_cnt, int64_t *box_off, int64_t cap_out,
                                 int32_t *idx_out, float *gt_pts, uint32_t *status, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!points || !pt_off || !box_frame || !boxes || !box_cnt || !box_off || !idx_out || !gt_pts ||
      !status)
    return DFU3D_EINVAL;
  if (Bt <= 0 || cap_out <= 0) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pib_count, dim3(Bt), dim3(GT), 0, st, (const float4 *)points, pt_off,
                     box_frame, boxes, box_cnt);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pib_scan, dim3(1), dim3(1024), 0, st, Bt, box_cnt, (long long *)box_off,
                     (long long)cap_out, status);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pib_fill, dim3(Bt), dim3(GT), 0, st, (const float4 *)points, pt_off,
                     box_frame, boxes, (const long long *)box_off, (long long)cap_out, idx_out,
                     (float4 *)gt_pts);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}
