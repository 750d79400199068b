// iou_stage.hip -- SURVEY.md §8 row f-3: rotated BEV overlap / IoU and rotated NMS
// (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu; host loop of iou3d_nms.cpp:139-177).
// The per-pair arithmetic is iou_common.inc (float32, the reference's operation order).
#include "common.hpp"

namespace {

#define IOU_FN __device__ __forceinline__
#include "iou_common.inc"

// (N, M) overlap areas (mode 0: boxes_overlap_kernel) or BEV IoUs (mode 1: boxes_iou_bev_kernel)
__global__ __launch_bounds__(256) void k_boxes_bev(const float *__restrict__ a, int n,
                                                   const float *__restrict__ b, int m,
                                                   float *__restrict__ out, int mode) {
  __shared__ float sb[16 * 7];
  __shared__ float sa[16 * 7];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tx, i = blockIdx.y * 16 + ty;
  if (threadIdx.x < 16 * 7) {
    const int bj = blockIdx.x * 16 + threadIdx.x / 7;
    sb[threadIdx.x] = (bj < m) ? b[(size_t)blockIdx.x * 16 * 7 + threadIdx.x] : 0.f;
    const int ai = blockIdx.y * 16 + threadIdx.x / 7;
    sa[threadIdx.x] = (ai < n) ? a[(size_t)blockIdx.y * 16 * 7 + threadIdx.x] : 0.f;
  }
  __syncthreads();
  if (i >= n || j >= m) return;
  float ba[7], bb[7];
#pragma unroll
  for (int k = 0; k < 7; k++) { ba[k] = sa[ty * 7 + k]; bb[k] = sb[tx * 7 + k]; }
  out[(size_t)i * m + j] = mode ? iou_bev(ba, bb) : iou_box_overlap(ba, bb);
}

// mask[i * col_blocks + c] bit t: iou_bev(box i, box 64c + t) > thresh, for 64c + t > i
// (iou3d_nms_kernel.cu:295-339; column blocks left of the diagonal are never read by the
// reduction and are not computed)
__global__ __launch_bounds__(64) void k_nms_mask(int n, float thresh, const float *__restrict__ boxes,
                                                 unsigned long long *__restrict__ mask) {
  const int row_start = blockIdx.y, col_start = blockIdx.x;
  if (col_start < row_start) return;
  const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
  __shared__ float block_boxes[64 * 7];
  if ((int)threadIdx.x < col_size) {
#pragma unroll
    for (int k = 0; k < 7; k++)
      block_boxes[threadIdx.x * 7 + k] = boxes[(size_t)(64 * col_start + threadIdx.x) * 7 + k];
  }
  __syncthreads();
  if ((int)threadIdx.x < row_size) {
    const int cur = 64 * row_start + threadIdx.x;
    float cb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) cb[k] = boxes[(size_t)cur * 7 + k];
    unsigned long long t = 0;
    const int start = (row_start == col_start) ? (int)threadIdx.x + 1 : 0;
    for (int i = start; i < col_size; i++) {
      float ob[7];
#pragma unroll
      for (int k = 0; k < 7; k++) ob[k] = block_boxes[i * 7 + k];
      if (iou_bev(cb, ob) > thresh) t |= 1ULL << i;
    }
    const int col_blocks = (n + 63) / 64;
    mask[(size_t)cur * col_blocks + col_start] = t;
  }
}

// The reference copies the mask to the host and walks it there (iou3d_nms.cpp:158-172);
// here one wave does the same walk on the device: lane l owns the `removed` words l, l+64, ...
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

}  // namespace

extern "C" int dfu3d_boxes_bev(const float *boxes_a, int32_t n, const float *boxes_b, int32_t m,
                               float *out, int32_t mode, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!boxes_a || !boxes_b || !out) return DFU3D_EINVAL;
  if (n < 0 || m < 0 || (mode != 0 && mode != 1)) return DFU3D_EINVAL;
  if (n == 0 || m == 0) return DFU3D_OK;
  if ((n + 15) / 16 > 65535) return DFU3D_ERANGE;
  hipLaunchKernelGGL(k_boxes_bev, dim3((m + 15) / 16, (n + 15) / 16), dim3(256), 0,
                     (hipStream_t)stream, boxes_a, n, boxes_b, m, out, mode);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_nms_bev(const float *boxes, int32_t n, float thresh, uint64_t *mask,
                             int64_t *keep, int32_t *num_keep, void *stream) {
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
  hipLaunchKernelGGL(k_nms_mask, dim3(cb, cb), dim3(64), 0, st, n, thresh, boxes,
                     (unsigned long long *)mask);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(64), 0, st, n, (const unsigned long long *)mask,
                     (long long *)keep, num_keep);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}
