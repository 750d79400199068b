// sampling_stage.hip -- SURVEY.md §8 row f-4: the consumer of the virtual points inside OpenPCDet's ground-truth
// sampling augmentor, `la_sampling` (pcdet/datasets/augmentor/database_sampler_virtual.py:307-351): the float32 points of
// one sampled object are binned by (theta // vert_res, fan // hor_res) of their spherical coordinates, every bin keeps
// the point with the smallest theta (first one on ties), bins come out in first-seen order; an object that would keep
// fewer than five points is returned unchanged.
//
// The reference walks the points in a Python loop and keys a dict by str(vert) + '_' + str(hor).  Here a batch of
// objects is one launch, one workgroup per object.  All arithmetic is float32 in NumPy's operation order:
//   r     = sqrt((x*x + y*y) + z*z)                     np.linalg.norm(points[:, 0:3], ord=2, axis=-1)   (:308)
//   theta = arccos(z / r),  fan = arctan(y / x)         (:309-310)
//   key   = (theta // vert_res, fan // hor_res)         NumPy's float floor division (npy_divmodf)       (:330-331)
// arccos / arctan are evaluated in fp64 and rounded to float32 (the correctly rounded float32 value); NumPy's float32
// loops may differ from that by one ulp, which matters only for a theta / fan within an ulp of a bin edge or of its
// bin's minimum -- the parity test states and checks exactly that.
// Two keys are the same bin iff their strings are: the float32 patterns are compared, with every NaN made one pattern
// ('nan'), and +0.0 / -0.0 kept apart ('0.0' / '-0.0').
#include "common.hpp"

namespace {

// numpy/core/src/npymath/npy_math_internal.h.src: npy_divmod (float32), the quotient only
__device__ __forceinline__ float np_floor_divide_f32(float a, float b) {
  if (b == 0.0f) return a / b;
  float mod = fmodf(a, b);
  float div = (a - mod) / b;
  if (mod != 0.0f) {
    if ((b < 0.0f) != (mod < 0.0f)) div -= 1.0f;
  }
  if (div != 0.0f) {
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.0f;
    return fl;
  }
  return copysignf(0.0f, a / b);
}

__device__ __forceinline__ uint32_t key_bits(float v) {
  return (v != v) ? 0x7FC00000u : __float_as_uint(v);
}

// np.argmin order: a NaN is the minimum (the first NaN wins), otherwise the smaller value, the smaller index on ties
__device__ __forceinline__ bool beats(float tj, int j, float ti, int i) {
  const bool nj = tj != tj, ni = ti != ti;
  if (nj || ni) return nj && (!ni || j < i);
  return tj < ti || (tj == ti && j < i);
}

constexpr int LT = 256;

// scratch per point: key u64 | theta f32 | first i32 | rank i32  (24 bytes, arrays over all points of the batch)
__global__ __launch_bounds__(LT) void k_la_sampling(const float *__restrict__ pts, int n_cols,
                                                    const long long *__restrict__ obj_off, float vert_res,
                                                    float hor_res, float *__restrict__ out, int *__restrict__ out_cnt,
                                                    unsigned long long *__restrict__ s_key, float *__restrict__ s_theta,
                                                    int *__restrict__ s_first, int *__restrict__ s_rank) {
  __shared__ int s_w[LT / 64];
  const int b = blockIdx.x;
  const long long p0 = obj_off[b];
  const int n = (int)(obj_off[b + 1] - p0);
  if (n <= 0) { if (threadIdx.x == 0) out_cnt[b] = 0; return; }
  unsigned long long *key = s_key + p0;
  float *theta = s_theta + p0;
  int *first = s_first + p0, *rank = s_rank + p0;
  const float *P = pts + (size_t)p0 * n_cols;
  float *O = out + (size_t)p0 * n_cols;
  for (int i = threadIdx.x; i < n; i += LT) {
    const float x = P[(size_t)i * n_cols], y = P[(size_t)i * n_cols + 1], z = P[(size_t)i * n_cols + 2];
    float s = x * x + y * y;
    s = s + z * z;
    const float r = sqrtf(s);
    const float th = (float)acos((double)(z / r));
    const float fan = (float)atan((double)(y / x));
    const float vc = np_floor_divide_f32(th, vert_res), hc = np_floor_divide_f32(fan, hor_res);
    key[i] = ((unsigned long long)key_bits(vc) << 32) | key_bits(hc);
    theta[i] = th;
  }
  __syncthreads();                     // (global scratch written and read by this workgroup only)
  __threadfence();
  // first index of every point's bin, and whether the point is its bin's representative
  int running = 0;
  for (int i0 = 0; i0 < n; i0 += LT) {
    const int i = i0 + threadIdx.x;
    bool leader = false;
    if (i < n) {
      const unsigned long long ki = key[i];
      int f = i;
      for (int j = 0; j < i; j++)
        if (key[j] == ki) { f = j; break; }
      first[i] = f;
      leader = (f == i);
    }
    int tot;
    const int r = block_rank<LT / 64>(leader, s_w, tot);
    if (leader) rank[i] = running + r;            // bins in first-seen order
    running += tot;
  }
  __syncthreads();
  __threadfence();
  const int K = running;
  if (K < 5) {                                     // database_sampler_virtual.py:348-349: the object as it came
    for (size_t e = threadIdx.x; e < (size_t)n * n_cols; e += LT) O[e] = P[e];
    if (threadIdx.x == 0) out_cnt[b] = n;
    return;
  }
  for (int i = threadIdx.x; i < n; i += LT) {
    const unsigned long long ki = key[i];
    const float ti = theta[i];
    bool rep = true;
    for (int j = 0; j < n && rep; j++)
      if (j != i && key[j] == ki && beats(theta[j], j, ti, i)) rep = false;
    if (rep) {
      float *o = O + (size_t)rank[first[i]] * n_cols;
      for (int c = 0; c < n_cols; c++) o[c] = P[(size_t)i * n_cols + c];
    }
  }
  if (threadIdx.x == 0) out_cnt[b] = K;
}

}  // namespace

extern "C" int dfu3d_la_sampling(const float *points, int32_t n_cols, const int64_t *obj_off, int32_t B,
                                 float vert_res, float hor_res, float *out, int32_t *out_cnt, void *scratch,
                                 int64_t n_points, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!points || !obj_off || !out || !out_cnt || !scratch) return DFU3D_EINVAL;
  if (B <= 0 || n_cols < 3 || n_points <= 0) return DFU3D_EINVAL;
  if ((uintptr_t)scratch & 7u) return DFU3D_EINVAL;
  if (points == out) return DFU3D_EINVAL;                     // rows move: not in place
  unsigned long long *key = (unsigned long long *)scratch;
  float *theta = (float *)(key + n_points);
  int *first = (int *)(theta + n_points);
  int *rank = first + n_points;
  hipLaunchKernelGGL(k_la_sampling, dim3(B), dim3(LT), 0, (hipStream_t)stream, points, n_cols,
                     (const long long *)obj_off, vert_res, hor_res, out, out_cnt, key, theta, first, rank);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}
