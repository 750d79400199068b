// iou_stage.hip -- SURVEY.md §8 row f-3: overlap of rotated rectangles in the ground plane (BEV),
// the IoU criteria built on it, and rotated non-maximum suppression.
//
// What it stands in for in the reference: pcdet/ops/iou3d_nms (boxes_overlap / boxes_iou_bev /
// nms kernels, iou3d_nms.cpp:120-177 host walk over the mask) and the AP evaluator's
// rotate_iou_gpu_eval (pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py:262-330, numba-CUDA,
// no ROCm target).  This is NOT their algorithm (edge-pair intersections + corner-in-box tests with a
// 1e-2 margin + angular sort of up to 16 vertices): the overlap here is the exact area of the
// intersection polygon --
//   * rectangle A is expressed in the frame of rectangle B, where B is the axis-aligned box
//     |u| <= hu, |v| <= hv;
//   * A's quadrilateral is clipped against B's four sides one after the other (Sutherland-Hodgman);
//     a convex polygon stays convex and ordered under clipping, so there is nothing to sort;
//   * the polygon (at most 8 vertices) lives in LDS as [vertex][thread], which makes a run-time
//     vertex index free of register spills and of bank conflicts beyond 2-way;
//   * pairs whose circumscribed circles do not meet leave at once (almost all pairs of an NMS).
// Differences to the reference's numbers: its margin lets a corner that is up to 1 cm OUTSIDE the other
// box count as a vertex, so it over-estimates the overlap by a thin sliver in those configurations; the
// exact polygon never does.  Parity for this row is therefore against geometry (float64 clipper +
// analytic cases), not against the reference's rounding -- stated in DESIGN.md.
#include "common.hpp"
#include "rect_overlap.hpp"

namespace {

// (n, m) matrix: workgroup = 8 rows x 32 columns of pairs (128-byte output rows)
__global__ __launch_bounds__(IB) void k_pair_matrix(const float *__restrict__ a, int n,
                                                    const float *__restrict__ b, int m, int fmt,
                                                    float *__restrict__ out, int crit) {
  __shared__ float s_poly[4][MAXV * IB];
  __shared__ Rect s_a[8], s_b[32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.y * 8, j0 = blockIdx.x * 32;
  if (threadIdx.x < 8 && i0 + threadIdx.x < n) s_a[threadIdx.x] = make_rect(a + (size_t)(i0 + threadIdx.x) * fmt, fmt);
  if (threadIdx.x >= 32 && threadIdx.x < 64 && j0 + threadIdx.x - 32 < m)
    s_b[threadIdx.x - 32] = make_rect(b + (size_t)(j0 + threadIdx.x - 32) * fmt, fmt);
  __syncthreads();
  const int i = i0 + ty, j = j0 + tx;
  if (i >= n || j >= m) return;
  const Rect A = s_a[ty], B = s_b[tx];
  const float inter = overlap_area(A, B, s_poly[0] + threadIdx.x, s_poly[1] + threadIdx.x, s_poly[2] + threadIdx.x,
                                   s_poly[3] + threadIdx.x);
  out[(size_t)i * m + j] = criterion_value(inter, A, B, crit);
}

// row i against row i of the other list ("aligned" / "paired" variants of the reference)
__global__ __launch_bounds__(IB) void k_pair_list(const float *__restrict__ a, const float *__restrict__ b, int n,
                                                  int fmt, float *__restrict__ out, int crit) {
  __shared__ float s_poly[4][MAXV * IB];
  const int i = blockIdx.x * IB + threadIdx.x;
  if (i >= n) return;
  const Rect A = make_rect(a + (size_t)i * fmt, fmt), B = make_rect(b + (size_t)i * fmt, fmt);
  const float inter = overlap_area(A, B, s_poly[0] + threadIdx.x, s_poly[1] + threadIdx.x, s_poly[2] + threadIdx.x,
                                   s_poly[3] + threadIdx.x);
  out[i] = criterion_value(inter, A, B, crit);
}

// Suppression mask, one 64-bit word per (box i, block of 64 later boxes): ONE WAVE per word -- lane l tests box
// 64*cb + l against box i and the wave's ballot IS the word.  Blocks that lie wholly before i are never read by
// the walk below and are skipped.  normal != 0: axis-aligned IoU (the reference's nms_normal).
__global__ __launch_bounds__(IB) void k_suppress_mask(const float *__restrict__ boxes, int n, float thresh,
                                                      int normal, unsigned long long *__restrict__ mask) {
  __shared__ float s_poly[4][MAXV * IB];
  const int nblk = (n + 63) >> 6;
  const int i = blockIdx.y * (IB / 64) + (threadIdx.x >> 6);
  const int cb = blockIdx.x;
  if (i >= n || cb * 64 + 63 <= i) return;                  // uniform per wave
  const int lane = lane_id();
  const int j = cb * 64 + lane;
  bool over = false;
  if (j < n && j > i) {
    const Rect A = make_rect(boxes + (size_t)i * 7, 7), B = make_rect(boxes + (size_t)j * 7, 7);
    float inter;
    if (normal) {                                           // headings ignored
      const float w = fminf(A.cx + A.hu, B.cx + B.hu) - fmaxf(A.cx - A.hu, B.cx - B.hu);
      const float h = fminf(A.cy + A.hv, B.cy + B.hv) - fmaxf(A.cy - A.hv, B.cy - B.hv);
      inter = fmaxf(w, 0.0f) * fmaxf(h, 0.0f);
    } else {
      inter = overlap_area(A, B, s_poly[0] + threadIdx.x, s_poly[1] + threadIdx.x, s_poly[2] + threadIdx.x,
                           s_poly[3] + threadIdx.x);
    }
    over = criterion_value(inter, A, B, 1) > thresh;
  }
  const unsigned long long word = __ballot(over);
  if (lane == 0) mask[(size_t)i * nblk + cb] = word;
}

// Greedy walk over the boxes in score order on the device (the reference copies the mask to the host,
// iou3d_nms.cpp:158-172): one wave, lane l owns the words l, l+64, ... of the running "suppressed" set.
__global__ __launch_bounds__(64) void k_nms_reduce(int n, const unsigned long long *__restrict__ mask,
                                                   long long *__restrict__ keep, int *__restrict__ num_keep) {
  const int col_blocks = (n + 63) / 64;
  const int lane = threadIdx.x;
  constexpr int MAXW = 8;                       // up to 64 * 64 * 8 = 32768 boxes
  unsigned long long remv[MAXW];
#pragma unroll
  for (int w = 0; w < MAXW; w++) remv[w] = 0ull;
  int nk = 0;
  for (int i = 0; i < n; i++) {
    const int nblock = i >> 6, inblock = i & 63;
    // the word of block nblock lives in lane nblock % 64, slot nblock / 64
    unsigned long long word = 0ull;
#pragma unroll
    for (int w = 0; w < MAXW; w++)
      if (w == (nblock >> 6)) word = remv[w];
    const unsigned lo = __shfl((unsigned)(word & 0xFFFFFFFFull), nblock & 63, 64);
    const unsigned hi = __shfl((unsigned)(word >> 32), nblock & 63, 64);
    word = ((unsigned long long)hi << 32) | lo;
    if (!(word & (1ULL << inblock))) {
      if (lane == 0) keep[nk] = i;
      nk++;
      const unsigned long long *p = mask + (size_t)i * col_blocks;
#pragma unroll
      for (int w = 0; w < MAXW; w++) {
        const int j = w * 64 + lane;
        if (j >= nblock && j < col_blocks) remv[w] |= p[j];
      }
    }
  }
  if (lane == 0) *num_keep = nk;
}

int launch_matrix(const float *a, int n, const float *b, int m, int fmt, float *out, int crit, hipStream_t st) {
  if ((n + 7) / 8 > 65535) return DFU3D_ERANGE;
  hipLaunchKernelGGL(k_pair_matrix, dim3((m + 31) / 32, (n + 7) / 8), dim3(IB), 0, st, a, n, b, m, fmt, out, crit);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

}  // namespace

extern "C" int dfu3d_boxes_bev(const float *boxes_a, int32_t n, const float *boxes_b, int32_t m,
                               float *out, int32_t mode, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!boxes_a || !boxes_b || !out) return DFU3D_EINVAL;
  if (n < 0 || m < 0 || (mode != 0 && mode != 1)) return DFU3D_EINVAL;
  if (n == 0 || m == 0) return DFU3D_OK;
  return launch_matrix(boxes_a, n, boxes_b, m, 7, out, mode, (hipStream_t)stream);
}

extern "C" int dfu3d_boxes_bev_paired(const float *boxes_a, const float *boxes_b, int32_t n, float *out,
                                      int32_t mode, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!boxes_a || !boxes_b || !out) return DFU3D_EINVAL;
  if (n < 0 || (mode != 0 && mode != 1)) return DFU3D_EINVAL;
  if (n == 0) return DFU3D_OK;
  hipLaunchKernelGGL(k_pair_list, dim3((n + IB - 1) / IB), dim3(IB), 0, (hipStream_t)stream, boxes_a, boxes_b, n, 7, out,
                     mode);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_rotate_iou_eval(const float *boxes, int32_t n, const float *query_boxes, int32_t k,
                                     float *out, int32_t criterion, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!boxes || !query_boxes || !out) return DFU3D_EINVAL;
  if (n < 0 || k < 0) return DFU3D_EINVAL;
  if (n == 0 || k == 0) return DFU3D_OK;
  // rotate_iou.py:247-255: -1 -> IoU, 0 -> / area(box), 1 -> / area(query box), anything else -> the overlap itself
  const int crit = criterion == -1 ? 1 : (criterion == 0 ? 2 : (criterion == 1 ? 3 : 0));
  return launch_matrix(boxes, n, query_boxes, k, 5, out, crit, (hipStream_t)stream);
}

static int nms_impl(const float *boxes, int32_t n, float thresh, int normal, uint64_t *mask, int64_t *keep,
                    int32_t *num_keep, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!boxes || !mask || !keep || !num_keep) return DFU3D_EINVAL;
  if (n < 0) return DFU3D_EINVAL;
  if (n > 32768) return DFU3D_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(num_keep, 0, sizeof(int32_t), st) != hipSuccess) return DFU3D_ELAUNCH;
    return DFU3D_OK;
  }
  const int cb = (n + 63) / 64;
  hipLaunchKernelGGL(k_suppress_mask, dim3(cb, (n + IB / 64 - 1) / (IB / 64)), dim3(IB), 0, st, boxes, n, thresh,
                     normal, (unsigned long long *)mask);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(64), 0, st, n, (const unsigned long long *)mask,
                     (long long *)keep, num_keep);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_nms_bev(const float *boxes, int32_t n, float thresh, uint64_t *mask,
                             int64_t *keep, int32_t *num_keep, void *stream) {
  return nms_impl(boxes, n, thresh, 0, mask, keep, num_keep, stream);
}

extern "C" int dfu3d_nms_normal_bev(const float *boxes, int32_t n, float thresh, uint64_t *mask,
                                    int64_t *keep, int32_t *num_keep, void *stream) {
  return nms_impl(boxes, n, thresh, 1, mask, keep, num_keep, stream);
}
