// common.hpp -- shared device helpers for libdfu3d_hip (gfx950 / CDNA4, wave64).
//
// Arithmetic contract: this library is compiled with
// -ffp-contract=off; every fused multiply-add below is written explicitly and
// exists only where the reference's BLAS sgemm fuses (measured: sequential-k
// FMA chain).  fp32 division / sqrt are correctly rounded (hipcc default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfu3d.h"

#define DFU3D_WAVE 64

struct ViewCalib {           // 48 floats, see dfu3d.h
  float M43[12];
  float P2[12];
  float cu, cv, fu, fv, tx, ty;
  float Minv[12];
  float pad[6];
};
static_assert(sizeof(ViewCalib) == DFU3D_CALIB_FLOATS * 4, "calib record");

// ---- wave / block ordered-compaction helpers -------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// exclusive rank of this lane among lanes with flag set; total in wave_total
__device__ __forceinline__ int wave_rank(bool flag, int &wave_total) {
  const unsigned long long m = __ballot(flag);
  wave_total = __popcll(m);
  return __popcll(m & ((1ull << lane_id()) - 1ull));
}

// Block-wide exclusive rank for a flag.  NW = waves per block.  s_w: NW ints
// of LDS.  Contains two barriers; s_w is reusable on return.
template <int NW>
__device__ __forceinline__ int block_rank(bool flag, int *s_w, int &block_total) {
  int wt;
  const int r = wave_rank(flag, wt);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0) s_w[w] = wt;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int c = s_w[i];
    off += (i < w) ? c : 0;
    tot += c;
  }
  __syncthreads();
  block_total = tot;
  return off + r;
}

// Register moves between lanes without the LDS path: data-parallel-primitive modifiers.  A lane whose source lies
// outside its row (or whose row the row mask leaves out) gets `old`.  (__shfl_up / __shfl_xor are ds_bpermute: an LDS
// round trip each, and a scan or a reduction is six of them in a row.)
// PRECONDITION of every helper built on them (wave_incl_scan, wave_sum_i, wave_min_i_dpp, wave_or_u32_dpp, wave_min_d,
// wave_max_d, and nbr_f / nbr_u of the radius filter): ALL 64 lanes of the wave are active at the call and blockDim.x is a
// multiple of 64.  An inactive source lane leaves `old` in its reader, and if lane 63 is inactive the value read back
// from it is stale for the whole wave -- call them from wave-uniform control flow only (every call site is).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
// wave inclusive scan of an int: shifts by 1, 2, 4, 8 inside the rows of 16, then lane 15 of the row before into rows
// 1 and 3, then lane 31 into rows 2 and 3
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_move_i<0x111, 0xF>(0, v);               // row_shr:1
  v += dpp_move_i<0x112, 0xF>(0, v);               // row_shr:2
  v += dpp_move_i<0x114, 0xF>(0, v);               // row_shr:4
  v += dpp_move_i<0x118, 0xF>(0, v);               // row_shr:8
  v += dpp_move_i<0x142, 0xA>(0, v);               // row_bcast15
  v += dpp_move_i<0x143, 0xC>(0, v);               // row_bcast31
  return v;
}
// the wave's sum / bitwise OR, in every lane (integers: the order does not matter)
__device__ __forceinline__ int wave_sum_i_dpp(int v) {
  v += dpp_move_i<0xB1, 0xF>(v, v);                // quad_perm [1,0,3,2]
  v += dpp_move_i<0x4E, 0xF>(v, v);                // quad_perm [2,3,0,1]
  v += dpp_move_i<0x141, 0xF>(v, v);               // row_half_mirror
  v += dpp_move_i<0x140, 0xF>(v, v);               // row_mirror: every lane holds its row's sum
  v += dpp_move_i<0x142, 0xA>(0, v);               // rows 1, 3 += row before
  v += dpp_move_i<0x143, 0xC>(0, v);               // rows 2, 3 += rows 0 + 1: lane 63 holds the wave's
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i_dpp(int v) {
  v = min(v, dpp_move_i<0xB1, 0xF>(v, v));
  v = min(v, dpp_move_i<0x4E, 0xF>(v, v));
  v = min(v, dpp_move_i<0x141, 0xF>(v, v));
  v = min(v, dpp_move_i<0x140, 0xF>(v, v));
  v = min(v, dpp_move_i<0x142, 0xA>(v, v));
  v = min(v, dpp_move_i<0x143, 0xC>(v, v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_or_u32_dpp(uint32_t x) {
  int v = (int)x;
  v |= dpp_move_i<0xB1, 0xF>(v, v);
  v |= dpp_move_i<0x4E, 0xF>(v, v);
  v |= dpp_move_i<0x141, 0xF>(v, v);
  v |= dpp_move_i<0x140, 0xF>(v, v);
  v |= dpp_move_i<0x142, 0xA>(0, v);
  v |= dpp_move_i<0x143, 0xC>(0, v);
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// Block-wide exclusive scan of an int value.
template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int *s_w, int &block_total) {
  const int inc = wave_incl_scan(v);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 63) s_w[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int c = s_w[i];
    off += (i < w) ? c : 0;
    tot += c;
  }
  __syncthreads();
  block_total = tot;
  return off + inc - v;
}

// ---- fp64 wave reductions ---------------------------------------------------
__device__ __forceinline__ double shfl_xor_d(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
  return v;
}
// Minimum / maximum over the wave, the result in every lane.  Data-parallel-primitive moves inside the rows, two row
// broadcasts, one read of lane 63: twelve register moves at the latency of a vector instruction.  (The xor butterfly
// through ds_bpermute was twelve dependent LDS-latency operations per reduction, and the fit kernels do eight of them
// per pair of headings.)  A minimum does not depend on the order it is formed in: same bits as before.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double min_f64_raw(double a, double b) {     // (no canonicalising copy of the operands)
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_f64_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double lane63_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_min_d(double v) {
  v = min_f64_raw(v, dpp_move_d<0xB1, 0xF>(v));      // quad_perm [1,0,3,2]
  v = min_f64_raw(v, dpp_move_d<0x4E, 0xF>(v));      // quad_perm [2,3,0,1]
  v = min_f64_raw(v, dpp_move_d<0x141, 0xF>(v));     // row_half_mirror
  v = min_f64_raw(v, dpp_move_d<0x140, 0xF>(v));     // row_mirror: every lane of a row holds the row's minimum
  v = min_f64_raw(v, dpp_move_d<0x142, 0xA>(v));     // row_bcast15 into rows 1 and 3
  v = min_f64_raw(v, dpp_move_d<0x143, 0xC>(v));     // row_bcast31 into rows 2 and 3: lane 63 holds the wave's
  return lane63_d(v);
}
__device__ __forceinline__ double wave_max_d(double v) {
  v = max_f64_raw(v, dpp_move_d<0xB1, 0xF>(v));
  v = max_f64_raw(v, dpp_move_d<0x4E, 0xF>(v));
  v = max_f64_raw(v, dpp_move_d<0x141, 0xF>(v));
  v = max_f64_raw(v, dpp_move_d<0x140, 0xF>(v));
  v = max_f64_raw(v, dpp_move_d<0x142, 0xA>(v));
  v = max_f64_raw(v, dpp_move_d<0x143, 0xC>(v));
  return lane63_d(v);
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned int lo = (unsigned int)__shfl_xor((int)(v & 0xFFFFFFFFull), m, 64);
    const unsigned int hi = (unsigned int)__shfl_xor((int)(v >> 32), m, 64);
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) { return wave_sum_i_dpp(v); }

// ---- calibration arithmetic -------------------------------------------------
// calibration_kitti.py:104-112, float32: sequential-k FMA chain (== sgemm).
__device__ __forceinline__ void lidar_to_rect_f32(const float *M, float x, float y,
                                                  float z, float r[3]) {
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float acc = x * M[0 * 3 + j];
    acc = __fmaf_rn(y, M[1 * 3 + j], acc);
    acc = __fmaf_rn(z, M[2 * 3 + j], acc);
    acc = acc + M[3 * 3 + j];
    r[j] = acc;
  }
}
// calibration_kitti.py:114-123, float32.
__device__ __forceinline__ void rect_to_img_f32(const float *P2, const float r[3],
                                                float &u, float &v, float &depth) {
  float h[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float acc = r[0] * P2[j * 4 + 0];
    acc = __fmaf_rn(r[1], P2[j * 4 + 1], acc);
    acc = __fmaf_rn(r[2], P2[j * 4 + 2], acc);
    acc = acc + P2[j * 4 + 3];
    h[j] = acc;
  }
  u = h[0] / r[2];
  v = h[1] / r[2];
  depth = h[2] - P2[2 * 4 + 3];
}
// a / b for a divisor that is reused many times: r = RN(1/b) is formed once, then
//   q0 = a*r, rem = fma(-q0, b, a) (exact), q = fma(rem, r, q0)
// is the correctly rounded quotient (Markstein's theorem: q0 is a faithful
// approximation, r the correctly rounded reciprocal, b's significand is not all
// ones -- it is a float32 value here).  Checked against the hardware division on
// 3.2e9 random operands (numerics).  Zero / huge / non-finite quotients
// take the ordinary division (sign of zero, inf - inf in the residual).
struct Recip { double rfu, rfv; };
__device__ __forceinline__ Recip make_recip(const ViewCalib &c) {
  Recip r;
  r.rfu = 1.0 / (double)c.fu;
  r.rfv = 1.0 / (double)c.fv;
  return r;
}
__device__ __forceinline__ double div_reused(double a, double b, double r) {
  const double q0 = a * r;
  // the shortcut holds for 2^-900 <= |q0| < 2^900; zero, tiny, huge and non-finite quotients take the division.
  // One unsigned range test on the exponent field (32-bit integer work: fp64 compares issue at half rate)
  const uint32_t hi = (uint32_t)__double2hiint(q0) & 0x7FFFFFFFu;
  if (__builtin_expect(hi - 0x07B00000u >= 0x78400000u - 0x07B00000u, 0)) return a / b;
  const double rem = fma(-q0, b, a);
  return fma(rem, r, q0);
}
// calibration_kitti.py:134-144 + 89-102, fp64: pixel (col u, row v, depth d)
// -> LiDAR xyz.
__device__ __forceinline__ void pixel_to_lidar(const ViewCalib &c, const Recip &rc, int u, int v,
                                               float d, double &x, double &y,
                                               double &z) {
  const double dd = (double)d;
  const double xr = div_reused(((double)u - (double)c.cu) * dd, (double)c.fu, rc.rfu) + (double)c.tx;
  const double yr = div_reused(((double)v - (double)c.cv) * dd, (double)c.fv, rc.rfv) + (double)c.ty;
  const double zr = dd;
  // [xr,yr,zr,1] @ Minv: the reference's np.dot is a dgemm, i.e. a sequential-k FMA chain
  // (measured bit-exact: tests/test_oracle_golden.py::test_fp64_chain_matches_numpy_dgemm, golden G1)
  const float *M = c.Minv;   // (4,3) row-major
  x = fma(zr, (double)M[6], fma(yr, (double)M[3], xr * (double)M[0])) + (double)M[9];
  y = fma(zr, (double)M[7], fma(yr, (double)M[4], xr * (double)M[1])) + (double)M[10];
  z = fma(zr, (double)M[8], fma(yr, (double)M[5], xr * (double)M[2])) + (double)M[11];
}
// only the coordinate that serves as the voxel key (key_axis 1 = y, 2 = z)
// The column of Minv that gives one LiDAR coordinate (axis 0/1/2), read from the record in GLOBAL memory: indexing a
// private copy of the record with a run-time axis would put the whole record into scratch memory.
struct KeyCol { float m0, m3, m6, m9; };
__device__ __forceinline__ KeyCol load_key_col(const ViewCalib *c, int axis) {
  const float *M = c->Minv + axis;
  KeyCol k;
  k.m0 = M[0]; k.m3 = M[3]; k.m6 = M[6]; k.m9 = M[9];
  return k;
}
__device__ __forceinline__ double pixel_to_lidar_axis(const ViewCalib &c, const Recip &rc, const KeyCol &kc, int u,
                                                      int v, float d) {
  const double dd = (double)d;
  const double xr = div_reused(((double)u - (double)c.cu) * dd, (double)c.fu, rc.rfu) + (double)c.tx;
  const double yr = div_reused(((double)v - (double)c.cv) * dd, (double)c.fv, rc.rfv) + (double)c.ty;
  return fma(dd, (double)kc.m6, fma(yr, (double)kc.m3, xr * (double)kc.m0)) + (double)kc.m9;
}

// ---- instance masks -------------------------------------------------------------
// DFU3D_MASK_BYTES (0): uint8 planes (V, max_inst, H, W), bit j <- plane j > 0 for j < m.
// 1 / 2 / 4: ONE word of that many bytes per pixel (V, H, W), bit j = instance j -- what
// dfu3d_pack_masks writes; bits >= m are ignored.
__host__ __device__ inline bool mask_format_ok(int fmt, int max_inst) {
  return fmt == 0 || (fmt == 1 && max_inst <= 8) || (fmt == 2 && max_inst <= 16) || fmt == 4;
}
__device__ __forceinline__ uint32_t mask_bits_at(const void *masks, int fmt, int v, int max_inst, int m,
                                                 int HW, int pix) {
  if (fmt == 0) {
    const uint8_t *mb = (const uint8_t *)masks + (size_t)v * max_inst * HW + pix;
    uint32_t bits = 0u;
    for (int j = 0; j < m; j++) bits |= (mb[(size_t)j * HW] > 0) ? (1u << j) : 0u;
    return bits;
  }
  const size_t o = (size_t)v * HW + pix;
  const uint32_t w = fmt == 1 ? (uint32_t)((const uint8_t *)masks)[o]
                   : fmt == 2 ? (uint32_t)((const uint16_t *)masks)[o] : ((const uint32_t *)masks)[o];
  return m >= 32 ? w : (w & ((1u << m) - 1u));
}

// monotone map double -> uint64 (total order; -0.0 canonicalised by caller)
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// splitmix64 finaliser; must match oracle/penet_oracle.py:_mix64
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// a sticky error left behind by an earlier, unrelated HIP call of the process must
// not be mistaken for a failure of our launch
#define DFU3D_CLEAR_STALE_ERROR() (void)hipGetLastError()

#define DFU3D_LAUNCH_CHECK()                                      \
  do {                                                            \
    if (hipGetLastError() != hipSuccess) return DFU3D_ELAUNCH;    \
  } while (0)

// Zeroing (or any one word value) as an ordinary kernel.  hipMemsetAsync is a command of its own kind: in the kernel trace
// of a pass every one of its seven fills started 19 us after the kernel before it had ended (tools/gap_report.py), while
// kernel follows kernel without a gap -- 130 us per pass spent waiting in front of seven memsets.  Words of 4 bytes, p 4-byte aligned.
namespace {
__global__ __launch_bounds__(256) void k_fill_words(uint32_t *__restrict__ p, size_t n, uint32_t v) {
  const size_t head = ((16u - (unsigned)((uintptr_t)p & 15u)) & 15u) / 4u < n ? ((16u - (unsigned)((uintptr_t)p & 15u)) & 15u) / 4u : n;
  uint4 *q = (uint4 *)(p + head);                      // the body in 16-byte stores
  const size_t nq = (n - head) / 4;
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  for (size_t i = tid; i < nq; i += nth) q[i] = make_uint4(v, v, v, v);
  const size_t done = head + nq * 4;
  if (tid < head) p[tid] = v;
  if (tid < n - done) p[done + tid] = v;
}
// up to three small regions in ONE launch (the counters a call resets before its first kernel)
__global__ void k_fill_small(uint32_t *a, int na, uint32_t *b, int nb, uint32_t *c, int nc, uint32_t v) {
  for (int i = threadIdx.x; i < na; i += blockDim.x) a[i] = v;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = v;
  for (int i = threadIdx.x; i < nc; i += blockDim.x) c[i] = v;
}
}  // namespace
// bytes a multiple of 4; value: the byte every byte is set to (as memset)
static inline hipError_t dfu3d_fill_async(void *p, int value, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if (((uintptr_t)p & 3u) || (bytes & 3u)) return hipMemsetAsync(p, value, bytes, st);
  const uint32_t b = (uint32_t)(value & 0xFF), v = b | (b << 8) | (b << 16) | (b << 24);
  const size_t n = bytes / 4, quads = (n + 3) / 4;
  const unsigned grid = (unsigned)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192);
  hipLaunchKernelGGL(k_fill_words, dim3(grid ? grid : 1), dim3(256), 0, st, (uint32_t *)p, n, v);
  return hipGetLastError();
}
static inline hipError_t dfu3d_fill_small_async(void *a, size_t bytes_a, void *b, size_t bytes_b, void *c, size_t bytes_c, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_small, dim3(1), dim3(256), 0, st, (uint32_t *)a, (int)(bytes_a / 4), (uint32_t *)b, (int)(bytes_b / 4),
                     (uint32_t *)c, (int)(bytes_c / 4), 0u);
  return hipGetLastError();
}
