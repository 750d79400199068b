// eval_stage.hip -- SURVEY.md §8 row f-3, AP-evaluation half: the KITTI evaluator the self-evolution loop scores its
// detector with (pcdet/datasets/kitti/kitti_object_eval_python/eval.py, reached from kitti_dataset.py:421-431).
//
// The reference walks frames, ground truths and detections in numba-jitted CPU loops (compute_statistics_jit, eval.py:
// 161-290), once per frame to collect the scores of the matched detections and then once per frame AND score threshold
// (up to 41) for every (class, difficulty, min_overlap) cell, on top of dense cross-frame IoU matrices cut into parts
// (calculate_iou_partly, :336-417) whose rotated-box kernel is numba-CUDA (no ROCm target).  Here:
//   * k_eval_overlaps: one thread per (detection, ground truth) pair OF THE SAME FRAME only (the cross-frame blocks
//     of the reference's matrices are never read), all frames in one launch, float64 results laid out
//     [frame][ground truth][detection];
//   * k_eval_match<FP>: the greedy assignment is sequential in the ground truths of a frame and its outcome depends on
//     the score threshold, so ONE LANE runs the reference's state machine for ONE threshold, a wave takes the (up to
//     64) thresholds of one frame and one (class, difficulty, min_overlap) cell, and the grid is frames x cells.  All
//     lanes of a wave walk the same (ground truth, detection) sequence, so overlaps, scores and ignore flags are
//     wave-uniform loads; what differs per lane is the threshold, the set of assigned detections (a bit set per lane in
//     LDS) and the running choice.  clean_data (:30-92) -- the per-class / per-difficulty ignore rules the reference
//     re-derives in Python for every cell -- is evaluated in the kernel from per-box codes.
// Counts (tp, fp, fn) are integers and exact; the orientation similarity is summed per frame in match order and over
// frames by the caller (the reference adds frame by frame: same terms, different association, ~1e-16 relative).
#include "common.hpp"
#include "rect_overlap.hpp"

namespace {

constexpr int EV_MAX_DET = 2048;             // detections per frame (bit set of a lane: 64 words)
constexpr int EV_WORDS = EV_MAX_DET / 32;
constexpr double EV_NONE = -10000000.0;      // eval.py:186 NO_DETECTION

// eval.py:33-35: MIN_HEIGHT [40, 25, 25], MAX_OCCLUSION [0, 1, 2], MAX_TRUNCATION [0.15, 0.3, 0.5] by difficulty 0..2
__device__ __forceinline__ double min_height(int difficulty) { return difficulty == 0 ? 40.0 : 25.0; }
__device__ __forceinline__ double max_truncation(int difficulty) {
  return difficulty == 0 ? 0.15 : (difficulty == 1 ? 0.3 : 0.5);
}

struct Combo { int cls, difficulty; double min_overlap; };
static_assert(sizeof(Combo) == sizeof(dfu3d_eval_combo), "combo");

// frame of flat pair index p: largest f with ov_off[f] <= p
__device__ __forceinline__ int frame_of(const long long *ov_off, int F, long long p) {
  int lo = 0, hi = F;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ov_off[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// eval.py:95-124 for one pair, float64.  crit -1 IoU, 0 / area(a), 1 / area(b)
__device__ __forceinline__ double image_overlap(const double *a, const double *b, int crit) {
  const double iw = fmin(a[2], b[2]) - fmax(a[0], b[0]);
  if (!(iw > 0)) return 0.0;
  const double ih = fmin(a[3], b[3]) - fmax(a[1], b[1]);
  if (!(ih > 0)) return 0.0;
  const double sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  const double ua = crit == -1 ? (sa + sb - iw * ih) : (crit == 0 ? sa : (crit == 1 ? sb : 1.0));
  return iw * ih / ua;
}

// metric 0: 2-D boxes; 1: footprints in the camera's x-z plane (IoU); 2: volumes (eval.py:126-158)
__global__ __launch_bounds__(IB) void k_eval_overlaps(int metric, int F, const long long *__restrict__ gt_off,
                                                      const long long *__restrict__ dt_off,
                                                      const long long *__restrict__ ov_off,
                                                      const double *__restrict__ gt_bbox,
                                                      const double *__restrict__ dt_bbox,
                                                      const double *__restrict__ gt_cam,
                                                      const double *__restrict__ dt_cam, double *__restrict__ ov,
                                                      long long n_pairs) {
  __shared__ float s_poly[4][MAXV * IB];
  const long long p = (long long)blockIdx.x * IB + threadIdx.x;
  if (p >= n_pairs) return;
  const int f = frame_of(ov_off, F, p);
  const long long D = dt_off[f + 1] - dt_off[f];
  const long long local = p - ov_off[f];
  const long long i = local / D, j = local - i * D;            // ground truth i, detection j
  const long long g = gt_off[f] + i, d = dt_off[f] + j;
  if (metric == 0) {
    ov[p] = image_overlap(dt_bbox + 4 * d, gt_bbox + 4 * g, -1);
    return;
  }
  // rotate_iou_gpu_eval works on float32 [x, z, l, w, ry] (eval.py:154-155, 367-381)
  const double *bd = dt_cam + 7 * d, *bg = gt_cam + 7 * g;
  const float a5[5] = {(float)bd[0], (float)bd[2], (float)bd[3], (float)bd[5], (float)bd[6]};
  const float b5[5] = {(float)bg[0], (float)bg[2], (float)bg[3], (float)bg[5], (float)bg[6]};
  const Rect A = make_rect(a5, 5), B = make_rect(b5, 5);
  const float inter = overlap_area(A, B, s_poly[0] + threadIdx.x, s_poly[1] + threadIdx.x, s_poly[2] + threadIdx.x,
                                   s_poly[3] + threadIdx.x);
  if (metric == 1) {
    ov[p] = (double)criterion_value(inter, A, B, 1);
    return;
  }
  float r = inter;                                             // the float32 matrix d3_box_overlap_kernel updates
  if (r > 0.0f) {
    const double iw = fmin(bd[1], bg[1]) - fmax(bd[1] - bd[4], bg[1] - bg[4]);     // y points down, y = box bottom
    if (iw > 0) {
      const double inc = iw * (double)r;
      const double ua = bd[3] * bd[4] * bd[5] + bg[3] * bg[4] * bg[5] - inc;
      r = (float)(inc / ua);
    } else {
      r = 0.0f;
    }
  }
  ov[p] = (double)r;
}

// clean_data (eval.py:30-92) for one ground truth: 0 counted, 1 neutral, -1 of no concern to this class
__device__ __forceinline__ int ignored_gt(int code, const double *bbox, int occluded, double truncated, const Combo &c) {
  const int valid_class = code == c.cls ? 1 : (code == 1000 + c.cls ? 0 : -1);
  bool hard = false;
  if (c.difficulty < 3)
    hard = occluded > c.difficulty || truncated > max_truncation(c.difficulty) ||
           (bbox[3] - bbox[1]) <= min_height(c.difficulty);
  if (valid_class == 1 && !hard) return 0;
  if (valid_class == 0 || (hard && valid_class == 1)) return 1;
  return -1;
}
__device__ __forceinline__ int ignored_dt(int code, const double *bbox, const Combo &c) {
  const double h = fabs(bbox[3] - bbox[1]);
  if (c.difficulty >= 0 && c.difficulty < 3 && h < min_height(c.difficulty)) return 1;
  return code == c.cls ? 0 : -1;
}

// FP false: eval.py:497-511 (thresh 0, compute_fp False) -- lane 0 only, scores of the matched detections out.
// FP true : fused_compute_statistics (:304-333) -- lane = threshold.
template <bool FP>
__global__ __launch_bounds__(64) void k_eval_match(
    int metric, int F, const long long *__restrict__ gt_off, const long long *__restrict__ dt_off,
    const long long *__restrict__ ov_off, const double *__restrict__ ov, const int *__restrict__ gt_code,
    const int *__restrict__ gt_dontcare, const double *__restrict__ gt_bbox, const double *__restrict__ gt_alpha,
    const int *__restrict__ gt_occluded, const double *__restrict__ gt_truncated, const int *__restrict__ dt_code,
    const double *__restrict__ dt_bbox, const double *__restrict__ dt_alpha, const double *__restrict__ dt_score,
    const Combo *__restrict__ combos, const double *__restrict__ thresholds, const int *__restrict__ n_thresh,
    int t_stride, int compute_aos, long long G_total, double *__restrict__ matched, int *__restrict__ n_valid,
    unsigned long long *__restrict__ pr, double *__restrict__ sim) {
  __shared__ uint32_t s_assigned[EV_WORDS * 64];
  __shared__ signed char s_idt[EV_MAX_DET];
  const int f = blockIdx.x, c = blockIdx.z, lane = threadIdx.x;
  const Combo cb = combos[c];
  const long long g0 = gt_off[f], d0 = dt_off[f];
  const int G = (int)(gt_off[f + 1] - g0), D = (int)(dt_off[f + 1] - d0);
  const double *ovf = ov + ov_off[f];
  if (D > EV_MAX_DET || D < 0 || G < 0) return;                // (uniform) offsets that contradict max_dt: touch nothing
  const int W = (D + 31) >> 5;
  for (int w = 0; w < W; w++) s_assigned[w * 64 + lane] = 0u;
  for (int j = lane; j < D; j += 64) s_idt[j] = (signed char)ignored_dt(dt_code[d0 + j], dt_bbox + 4 * (d0 + j), cb);
  __syncthreads();
  const int t = blockIdx.y * 64 + lane;
  const int nT = FP ? n_thresh[c] : 1;
  if (t >= nT) return;                                         // (no barrier below)
  const double thr = FP ? thresholds[(size_t)c * t_stride + t] : 0.0;
  const double mo = cb.min_overlap;
  auto is_assigned = [&](int j) { return (s_assigned[(j >> 5) * 64 + lane] >> (j & 31)) & 1u; };
  auto assign = [&](int j) { s_assigned[(j >> 5) * 64 + lane] |= 1u << (j & 31); };
  long long tp = 0, fp = 0, fn = 0;
  double sim_sum = 0.0;
  int valid_gt = 0;
  for (int i = 0; i < G; i++) {
    const int ig = ignored_gt(gt_code[g0 + i], gt_bbox + 4 * (g0 + i), gt_occluded[g0 + i], gt_truncated[g0 + i], cb);
    if (ig == 0) valid_gt++;
    if (ig == -1) continue;
    int pick = -1;
    double valid = EV_NONE, best = 0.0;
    bool picked_ignored = false;
    const double *row = ovf + (size_t)i * D;
    for (int j = 0; j < D; j++) {
      const int idt = s_idt[j];
      if (idt == -1) continue;
      const double sc = dt_score[d0 + j];
      if (is_assigned(j) || (FP && sc < thr)) continue;
      const double o = row[j];
      if (!FP) {
        if (o > mo && sc > valid) { pick = j; valid = sc; }
      } else if (o > mo && (o > best || picked_ignored) && idt == 0) {
        best = o; pick = j; valid = 1.0; picked_ignored = false;
      } else if (o > mo && valid == EV_NONE && idt == 1) {
        pick = j; valid = 1.0; picked_ignored = true;
      }
    }
    if (valid == EV_NONE) {
      if (ig == 0) fn++;
    } else if (ig == 1 || s_idt[pick] == 1) {
      assign(pick);
    } else {
      if (!FP) matched[(size_t)c * G_total + g0 + tp] = dt_score[d0 + pick];
      tp++;
      if (FP && compute_aos) sim_sum += (1.0 + cos(gt_alpha[g0 + i] - dt_alpha[d0 + pick])) / 2.0;
      assign(pick);
    }
  }
  if (!FP) {
    for (long long k = tp; k < G; k++) matched[(size_t)c * G_total + g0 + k] = __longlong_as_double(0x7FF8000000000000ll);
    n_valid[(size_t)c * F + f] = valid_gt;
    return;
  }
  for (int j = 0; j < D; j++) {
    const int idt = s_idt[j];
    if (!(is_assigned(j) || idt == -1 || idt == 1 || dt_score[d0 + j] < thr)) fp++;
  }
  if (metric == 0) {                                           // detections lying on DontCare regions are no false positives
    long long stuff = 0;
    for (int i = 0; i < G; i++) {
      if (!gt_dontcare[g0 + i]) continue;
      const double *dc = gt_bbox + 4 * (g0 + i);
      for (int j = 0; j < D; j++) {
        const int idt = s_idt[j];
        if (is_assigned(j) || idt == -1 || idt == 1 || dt_score[d0 + j] < thr) continue;
        if (image_overlap(dt_bbox + 4 * (d0 + j), dc, 0) > mo) { assign(j); stuff++; }
      }
    }
    fp -= stuff;
  }
  unsigned long long *cell = pr + ((size_t)c * t_stride + t) * 3;
  if (tp) atomicAdd(cell + 0, (unsigned long long)tp);
  if (fp) atomicAdd(cell + 1, (unsigned long long)fp);
  if (fn) atomicAdd(cell + 2, (unsigned long long)fn);
  if (sim) sim[((size_t)c * F + f) * t_stride + t] = (tp > 0 || fp > 0) ? sim_sum : 0.0;
}

bool sizes_ok(int32_t F, int64_t n_gt, int64_t n_dt) { return F > 0 && n_gt >= 0 && n_dt >= 0; }

}  // namespace

extern "C" int dfu3d_eval_overlaps(int32_t metric, int32_t F, const int64_t *gt_off, const int64_t *dt_off,
                                   const int64_t *ov_off, const double *gt_bbox, const double *dt_bbox,
                                   const double *gt_cam, const double *dt_cam, double *ov, int64_t n_pairs,
                                   void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!gt_off || !dt_off || !ov_off) return DFU3D_EINVAL;
  if (metric < 0 || metric > 2 || F <= 0 || n_pairs < 0) return DFU3D_EINVAL;
  if (n_pairs == 0) return DFU3D_OK;
  if (!ov || (metric == 0 && (!gt_bbox || !dt_bbox)) || (metric != 0 && (!gt_cam || !dt_cam))) return DFU3D_EINVAL;
  if (n_pairs > (int64_t)0x7FFFFFFF * IB) return DFU3D_ERANGE;
  hipLaunchKernelGGL(k_eval_overlaps, dim3((unsigned)((n_pairs + IB - 1) / IB)), dim3(IB), 0, (hipStream_t)stream, metric,
                     F, (const long long *)gt_off, (const long long *)dt_off, (const long long *)ov_off, gt_bbox, dt_bbox,
                     gt_cam, dt_cam, ov, (long long)n_pairs);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_eval_match_scores(int32_t metric, int32_t F, int32_t max_dt, const int64_t *gt_off,
                                       const int64_t *dt_off, const int64_t *ov_off, const double *ov,
                                       const int32_t *gt_code, const int32_t *gt_dontcare, const double *gt_bbox,
                                       const double *gt_alpha, const int32_t *gt_occluded, const double *gt_truncated,
                                       const int32_t *dt_code, const double *dt_bbox, const double *dt_alpha,
                                       const double *dt_score, const dfu3d_eval_combo *combos, int32_t n_combo,
                                       int64_t n_gt, double *matched, int32_t *n_valid, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!gt_off || !dt_off || !ov_off || !combos || !n_valid) return DFU3D_EINVAL;
  if (metric < 0 || metric > 2 || n_combo <= 0 || n_combo > 65535 || !sizes_ok(F, n_gt, 0)) return DFU3D_EINVAL;
  if (max_dt < 0 || max_dt > EV_MAX_DET) return DFU3D_ERANGE;
  if (n_gt > 0 && (!matched || !gt_code || !gt_dontcare || !gt_bbox || !gt_alpha || !gt_occluded || !gt_truncated))
    return DFU3D_EINVAL;
  hipLaunchKernelGGL(k_eval_match<false>, dim3(F, 1, n_combo), dim3(64), 0, (hipStream_t)stream, metric, F,
                     (const long long *)gt_off, (const long long *)dt_off, (const long long *)ov_off, ov, gt_code,
                     gt_dontcare, gt_bbox, gt_alpha, gt_occluded, gt_truncated, dt_code, dt_bbox, dt_alpha, dt_score,
                     (const Combo *)combos, (const double *)nullptr, (const int *)nullptr, 1, 0, (long long)n_gt, matched,
                     n_valid, (unsigned long long *)nullptr, (double *)nullptr);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_eval_match_stats(int32_t metric, int32_t F, int32_t max_dt, const int64_t *gt_off,
                                      const int64_t *dt_off, const int64_t *ov_off, const double *ov,
                                      const int32_t *gt_code, const int32_t *gt_dontcare, const double *gt_bbox,
                                      const double *gt_alpha, const int32_t *gt_occluded, const double *gt_truncated,
                                      const int32_t *dt_code, const double *dt_bbox, const double *dt_alpha,
                                      const double *dt_score, const dfu3d_eval_combo *combos, int32_t n_combo,
                                      const double *thresholds, const int32_t *n_thresh, int32_t t_stride,
                                      int32_t compute_aos, int64_t *pr, double *sim, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!gt_off || !dt_off || !ov_off || !combos || !thresholds || !n_thresh || !pr) return DFU3D_EINVAL;
  if (metric < 0 || metric > 2 || n_combo <= 0 || n_combo > 65535 || F <= 0 || t_stride <= 0) return DFU3D_EINVAL;
  if (max_dt < 0 || max_dt > EV_MAX_DET) return DFU3D_ERANGE;
  if (compute_aos && !sim) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(pr, 0, sizeof(int64_t) * 3 * (size_t)n_combo * t_stride, st) != hipSuccess) return DFU3D_ELAUNCH;
  if (sim && hipMemsetAsync(sim, 0, sizeof(double) * (size_t)n_combo * F * t_stride, st) != hipSuccess)
    return DFU3D_ELAUNCH;
  hipLaunchKernelGGL(k_eval_match<true>, dim3(F, (t_stride + 63) / 64, n_combo), dim3(64), 0, st, metric, F,
                     (const long long *)gt_off, (const long long *)dt_off, (const long long *)ov_off, ov, gt_code,
                     gt_dontcare, gt_bbox, gt_alpha, gt_occluded, gt_truncated, dt_code, dt_bbox, dt_alpha, dt_score,
                     (const Combo *)combos, thresholds, n_thresh, t_stride, compute_aos, 0ll, (double *)nullptr,
                     (int *)nullptr, (unsigned long long *)pr, compute_aos ? sim : (double *)nullptr);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}
