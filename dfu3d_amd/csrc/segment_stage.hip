// segment_stage.hip -- per-instance point sets and the noise-suppression
// filters that act on them.
//
// Layout: every instance segment s = v*max_inst + j owns a contiguous slice of
// a structure-of-arrays fp64 pool (px,py,pz): LiDAR rows first, voxel
// representatives directly behind them, so the reference's
// torch.cat((lidar, pseudo)) (my_loader.py:605) is a no-op in memory.
// Filters are brute force over LDS-staged 1024-point tiles (24 KB/tile, 8 B/lane
// coalesced SoA loads); query tiles of 256 points are spread over the chip by a
// prefix-summed tile list so that one huge instance cannot serialise a launch;
// a query starts its sweep at the tile it lives in and the workgroup leaves the
// sweep as soon as every query has its answer (nb_points = 1 makes that the
// first tile for almost every point).  Survivors are compacted in order with
// ballot/popcount ranks.
#include "common.hpp"

namespace {

constexpr int QT = 256;    // queries per workgroup tile
constexpr int PT = 1024;   // points per LDS tile

// last word of a shadow entry (see the radius filter below): segment << 16 | radius as a bfloat16
__device__ __forceinline__ uint32_t shadow_word(int seg, double radius) {
  // radius as the top 16 bits of its float32 value rounded toward zero: r' <= r, so "certainly
  // within r'" implies "within r".  r == 0 (no filter) stays 0; r < 0 (drop all) keeps its sign
  // bit, NaN its pattern; a positive radius never becomes 0.
  float rf = (float)radius;
  if (radius > 0.0 && (double)rf > radius) rf = __uint_as_float(__float_as_uint(rf) - 1u);   // (float) rounded up
  uint32_t hi = __float_as_uint(rf) >> 16;
  if (radius > 0.0 && hi == 0u) hi = 1u;
  if (radius != radius) hi = 0x7FC0u;
  return ((uint32_t)seg << 16) | hi;
}


// ---------------------------------------------------------------- build
// A view's items are cut into chunks of SEG_CH; every (view, chunk) is one workgroup in the counting and in the
// writing pass.  The workgroup walks its chunk in steps of SEG_SUB items, and inside a step every WAVE owns SEG_WI
// consecutive items: lane j of a wave keeps the wave's count of instance j in a register, so a step costs two
// barriers per 8192 items.  History: a workgroup per view stepping through 1024 items at a time with two barriers
// per step was bound by the latency of that chain (0.70 ms for 0.2 GB); a workgroup per 8192 items made 12 000
// workgroups pay the prologue (item count -> list sizes -> bases -> earlier chunks) and was slower still (1.05 ms).
constexpr int SEG_WAVES = 8, SEG_WI = 256, SEG_STEPS = SEG_WI / 64;
constexpr int SEG_SUB = SEG_WAVES * SEG_WI;         // 2048 items per step of a workgroup (4 per lane: 62 VGPRs, four workgroups per CU)
constexpr int SEG_CH = 16 * SEG_SUB;                // 32768 items per workgroup
static_assert(DFU3D_MAX_INST <= 32, "one lane per instance, instance bits in one 32-bit word");

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x |= (uint32_t)__shfl_xor((int)x, m, 64);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
}

// lane j: how many of the wave's items (bit words b[k], one per lane and step) carry instance j
__device__ __forceinline__ int wave_instance_counts(const uint32_t (&b)[SEG_STEPS], uint32_t (&wany)[SEG_STEPS]) {
  int mine = 0;
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < SEG_STEPS; k++) {
    wany[k] = wave_or_u32(b[k]);
    for (uint32_t w = wany[k]; w; w &= w - 1u) {     // uniform
      const int j = __ffs((int)w) - 1;
      const int c = __popcll(__ballot((b[k] >> j) & 1u));
      if (lane == j) mine += c;
    }
  }
  return mine;
}

__global__ __launch_bounds__(SEG_WAVES * 64) void k_seg_count(const uint32_t *__restrict__ bits,
                                                             const int *__restrict__ n_item,
                                                             int cap_item, int max_inst,
                                                             int *__restrict__ cnt, int *__restrict__ chunk_cnt) {
  __shared__ int s_c[DFU3D_MAX_INST];
  const int v = blockIdx.y, ch = blockIdx.x;
  const int lane = lane_id();
  const int n = min(min(n_item[v], cap_item), (ch + 1) * SEG_CH);
  int *cc = chunk_cnt + ((size_t)v * gridDim.x + ch) * DFU3D_MAX_INST;
  if (ch * SEG_CH >= n) {                          // (uniform) nothing in this chunk
    if (threadIdx.x < DFU3D_MAX_INST) cc[threadIdx.x] = 0;
    return;
  }
  if (threadIdx.x < DFU3D_MAX_INST) s_c[threadIdx.x] = 0;
  __syncthreads();
  int mine = 0;
  for (int sub = ch * SEG_CH; sub < n; sub += SEG_SUB) {            // uniform
    const int w0 = sub + (int)(threadIdx.x >> 6) * SEG_WI;
    uint32_t b[SEG_STEPS], wany[SEG_STEPS];
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) {
      const int t = w0 + k * 64 + lane;
      b[k] = (t < n) ? bits[(size_t)v * cap_item + t] : 0u;
    }
    mine += wave_instance_counts(b, wany);
  }
  if (lane < DFU3D_MAX_INST && mine) atomicAdd(&s_c[lane], mine);
  __syncthreads();
  if (threadIdx.x < DFU3D_MAX_INST) {
    const int c = s_c[threadIdx.x];
    cc[threadIdx.x] = c;
    if (c && (int)threadIdx.x < max_inst) atomicAdd(&cnt[v * max_inst + threadIdx.x], c);
  }
}

// joint view of the 2S lists for the one-pass radius filter: s < S the LiDAR lists, S + s the pseudo lists
struct JointSegs {
  long long *base;
  int *cnt;
  double *rad;
  const double *rad_a, *rad_b;
};

__global__ __launch_bounds__(1024) void k_seg_alloc(int S, int *__restrict__ cnt_a,
                                                    int *__restrict__ cnt_b,
                                                    long long *__restrict__ base_a,
                                                    long long *__restrict__ base_b,
                                                    long long pool_cap,
                                                    long long *__restrict__ cursor,
                                                    uint32_t *__restrict__ status, JointSegs J) {
  __shared__ int s_w[16];
  __shared__ long long s_end;              // end of the last segment that fitted (the pool is dense up to there)
  long long running = *cursor;
  bool over = false;
  if (threadIdx.x == 0) s_end = -1;
  __syncthreads();
  for (int b0 = 0; b0 < S; b0 += 1024) {
    const int s = b0 + threadIdx.x;
    const int ca = (s < S) ? cnt_a[s] : 0, cb = (s < S) ? cnt_b[s] : 0;
    int tot;
    const int ex = block_excl_scan<16>(ca + cb, s_w, tot);
    if (s < S) {
      const long long b = running + ex;
      if (b + ca + cb > pool_cap) {
        cnt_a[s] = 0;
        cnt_b[s] = 0;
        base_a[s] = 0;
        base_b[s] = 0;
        if (ca + cb > 0) {
          over = true;
          atomicMin((unsigned long long *)&s_end, (unsigned long long)b);     // -1 = "none" is the largest value
        }
      } else {
        base_a[s] = b;
        base_b[s] = b + ca;
      }
      if (J.base) {
        J.base[s] = base_a[s]; J.base[S + s] = base_b[s];
        J.cnt[s] = cnt_a[s]; J.cnt[S + s] = cnt_b[s];
        J.rad[s] = J.rad_a[s]; J.rad[S + s] = J.rad_b[s];
      }
    }
    running += tot;
  }
  if (over) atomicOr(status, DFU3D_ST_POOL_OVERFLOW);
  __syncthreads();
  // segments are allocated in index order, so the ones that fit form a prefix: [0, cursor) is exactly covered
  if (threadIdx.x == 0) *cursor = (s_end >= 0) ? s_end : (running < pool_cap ? running : pool_cap);
}

// One workgroup per (view, chunk) writes its part of the ordered lists of all the view's instances: a list starts
// where the earlier chunks of the view end (their counts), a wave's part of it where the earlier waves of the step
// end, and inside the wave ballots give the order.  The bit words stay in registers between the counting and the
// writing sweep of a step; the coordinates of all SEG_STEPS sub-steps are requested before the first is used.
#ifdef DFU3D_DBG_GRID_TIMING      /* dev build: cycles of thread 0 per phase of k_seg_write, summed over workgroups */
__device__ unsigned long long g_seg_dbg[16];
#define SEG_T(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); atomicAdd(&g_seg_dbg[k], (unsigned long long)(t_ - sg_t)); sg_t = t_; } } while (0)
#else
#define SEG_T(k) do {} while (0)
#endif
constexpr int SWT = SEG_WAVES * 64;
__global__ __launch_bounds__(SWT) void k_seg_write(
    const uint32_t *__restrict__ bits, const double *__restrict__ ix,
    const double *__restrict__ iy, const double *__restrict__ iz,
    const int *__restrict__ n_item, int cap_item, int max_inst,
    const long long *__restrict__ base, const int *__restrict__ cnt, const int *__restrict__ chunk_cnt,
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
    float4 *__restrict__ pq, const double *__restrict__ rad, int seg_off) {
  __shared__ int s_wc[SEG_WAVES][DFU3D_MAX_INST];     // per wave: items of instance j in this step
  __shared__ int s_run[DFU3D_MAX_INST];               // per instance: items of the earlier chunks
  const int v = blockIdx.y, ch = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = lane_id();
#ifdef DFU3D_DBG_GRID_TIMING
  long long sg_t = clock64();
  if (threadIdx.x == 0) atomicAdd(&g_seg_dbg[8], 1ull);
#endif
  // the first-level loads are requested together (one memory round trip instead of a chain)
  const int n_raw = n_item[v];
  const bool lj = lane < max_inst;                     // lane j < max_inst: the facts of instance j
  const int cj = lj ? cnt[v * max_inst + lane] : 0;    // (0 also for the lists that did not fit the pool)
  long long off = lj ? base[v * max_inst + lane] : 0;
  const double rj = (lj && pq) ? rad[v * max_inst + lane] : 0.0;
  const int *cc = chunk_cnt + (size_t)v * gridDim.x * DFU3D_MAX_INST;
  const int jj = threadIdx.x & (DFU3D_MAX_INST - 1), c_first = threadIdx.x / DFU3D_MAX_INST;
  int q_first = 0;                                     // chunks 0..31 in one go, the rest (if any) in the loop below
  if (c_first < ch) q_first = cc[(size_t)c_first * DFU3D_MAX_INST + jj];
  const int n = min(min(n_raw, cap_item), (ch + 1) * SEG_CH);
  if (ch * SEG_CH >= n) { SEG_T(0); return; }
  const uint32_t live = (uint32_t)__ballot(cj > 0);   // instances of this view with a non-empty list
  if (live == 0u) return;
#ifdef DFU3D_DBG_GRID_TIMING
  if (threadIdx.x == 0) atomicAdd(&g_seg_dbg[9], 1ull);
#endif
  const uint32_t sw = (lj && pq) ? shadow_word(seg_off + v * max_inst + lane, rj) : 0u;
  if (threadIdx.x < DFU3D_MAX_INST) s_run[threadIdx.x] = 0;
  __syncthreads();
  if (q_first) atomicAdd(&s_run[jj], q_first);
  for (int c = c_first + SWT / DFU3D_MAX_INST; c < ch; c += SWT / DFU3D_MAX_INST) {
    const int q = cc[(size_t)c * DFU3D_MAX_INST + jj];
    if (q) atomicAdd(&s_run[jj], q);
  }
  __syncthreads();
  if (lane < DFU3D_MAX_INST) off += s_run[lane];       // every wave: where the chunk's part of list `lane` starts
  SEG_T(1);
  for (int sub = ch * SEG_CH; sub < n; sub += SEG_SUB) {            // uniform
    const int w0 = sub + wave * SEG_WI;
    uint32_t b[SEG_STEPS], wany[SEG_STEPS];
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) {
      const int t = w0 + k * 64 + lane;
      b[k] = (t < n) ? (bits[(size_t)v * cap_item + t] & live) : 0u;
    }
    double x[SEG_STEPS], y[SEG_STEPS], z[SEG_STEPS];
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) {
      const size_t o = (size_t)v * cap_item + w0 + k * 64 + lane;
      x[k] = 0.0; y[k] = 0.0; z[k] = 0.0;
      if (b[k]) { x[k] = ix[o]; y[k] = iy[o]; z[k] = iz[o]; }
    }
    const int mine = wave_instance_counts(b, wany);
    SEG_T(2);
#ifdef DFU3D_DBG_GRID_TIMING
    if (threadIdx.x == 0) atomicAdd(&g_seg_dbg[10], 1ull);
#endif
    if (sub > ch * SEG_CH) __syncthreads();            // the previous step's s_wc has been read by every wave
    if (lane < DFU3D_MAX_INST) s_wc[wave][lane] = mine;
    __syncthreads();
    long long woff = off;                              // lane j: where THIS wave's part of list j starts in this step
    if (lane < DFU3D_MAX_INST) {
      int before = 0, total = 0;
#pragma unroll
      for (int ww = 0; ww < SEG_WAVES; ww++) {
        const int c = s_wc[ww][lane];
        before += (ww < wave) ? c : 0;
        total += c;
      }
      woff += before;
      off += total;
    }
    SEG_T(3);
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) {
      for (uint32_t w = wany[k]; w; w &= w - 1u) {     // uniform per wave
        const int j = __ffs((int)w) - 1;
        const bool has = (b[k] >> j) & 1u;
        const unsigned long long m = __ballot(has);
        const long long start = __shfl(woff, j, 64);
        const uint32_t swj = (uint32_t)__shfl((int)sw, j, 64);
        if (has) {
          const long long d = start + __popcll(m & ((1ull << lane) - 1ull));
          px[d] = x[k];
          py[d] = y[k];
          pz[d] = z[k];
          if (pq)                                      // float32 shadow for the radius filter
            pq[d] = make_float4((float)x[k], (float)y[k], (float)z[k], __uint_as_float(swj));
        }
        if (lane == j) woff += __popcll(m);
      }
    }
    SEG_T(4);
  }
}

#ifdef DFU3D_DBG_GRID_TIMING
extern "C" int dfu3d_debug_seg_timing(unsigned long long *out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_seg_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return DFU3D_ELAUNCH;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_seg_dbg), z, sizeof(z)) != hipSuccess) return DFU3D_ELAUNCH;
  }
  return DFU3D_OK;
}
#endif

// ---------------------------------------------------------------- tiles
__global__ __launch_bounds__(1024) void k_tile_scan(int S, const int *__restrict__ cnt,
                                                    int *__restrict__ tile_off, int qt) {
  __shared__ int s_w[16];
  int running = 0;
  for (int b0 = 0; b0 < S; b0 += 1024) {
    const int s = b0 + threadIdx.x;
    const int nt = (s < S) ? (cnt[s] + qt - 1) / qt : 0;
    int tot;
    const int ex = block_excl_scan<16>(nt, s_w, tot);
    if (s < S) tile_off[s] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) tile_off[S] = running;
}

// largest s with tile_off[s] <= t  (t < tile_off[S])
__device__ __forceinline__ int find_segment(const int *tile_off, int S, int t) {
  int lo = 0, hi = S;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_off[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------- a10 radius
// The filter works on a float32 SHADOW of the pool: pq[i] = (x, y, z, seg << 16 | radius
// as a truncated bfloat16) for pool position i -- one 16-byte load per point, the segment
// travels with the coordinates (k_seg_write produces it together with the pool).  Every decision
// taken from the shadow is a CERTAIN one: a neighbour is counted only when its float32 distance is
// below the radius by more than a bound on everything float32 rounding (of the coordinates and of
// the arithmetic) can do; whatever is not certain is decided from the fp64 pool with the
// reference's predicate d2 < r2 (Open3D / nanoflann, self included).
//
// Phase A (k_radius_flags) streams the shadow linearly over the used part of the pool, 2048
// positions per workgroup, all of a wave's loads issued up front.  A query is tested against
// itself and its two list neighbours (lane^1, lane^2 -- the lists are in pixel / sweep order,
// so list neighbours are spatial neighbours; this settles ~98 % for nb_points = 1), then, if a
// lane of the wave is still undecided, against eight of the wave's points broadcast as scalar
// operands (v_readlane).  What is still undecided is kept in an LDS list of the workgroup and
// tried pairwise against the other undecided points of the workgroup (an outlier's neighbours
// are usually other outliers a few dozen list positions away).  The rest -- isolated points and
// the rare uncertain comparisons -- goes to phase B through one global atomic per workgroup.
// Phase A also leaves, per wave range (512 consecutive pool positions), two bounding boxes: one of the
// points it decided and one of the points it could not decide (outliers: kept apart so that they do not
// blow up the first).  Phase B (k_radius_resolve): one wave per queued query tests its own 64-chunk, then
// sweeps the BOXES of its segment (64 ranges = 32 k positions per step, a lane each) and tests the points
// of the few ranges whose boxes come within the radius: float32 shadow with certain-hit / certain-miss
// bounds, fp64 for the pairs in between.  Every point of the segment lies in one of the two boxes of its
// range, so nothing is missed.
constexpr int RFB = 256;           // threads per phase-A workgroup
constexpr int RF_IT = 8;           // 64-point chunks per wave
constexpr int RF_WG = RFB * RF_IT; // pool positions per workgroup
constexpr int RF_STRIDE = 8;       // phase A tries lanes 0, 8, 16, ... of the wave
constexpr int RF_LIST = 512;       // undecided points a workgroup tries pairwise in LDS
constexpr uint32_t RF_NOSEG = 0xFFFFu;

// bound on |float32 distance - true distance| for a query at (x,y,z) and neighbours within ~r of it:
// each coordinate of either point carries <= 2^-24 relative rounding, the differences, squares and
// sums add a few ulp: 2^-20 * (|x|+|y|+|z| + 3r) is more than 10x that.
__device__ __forceinline__ float rf_bound(float x, float y, float z, float r) {
  return (fabsf(x) + fabsf(y) + fabsf(z) + 3.0f * r) * 9.5367431640625e-07f;
}
__device__ __forceinline__ float rf_certain_hit2(float x, float y, float z, float r) {
  const float t = r - rf_bound(x, y, z, r) - r * 2.4e-7f;
  return (t > 0.0f) ? t * t * 0.999999f : -1.0f;
}

// wave-wide min / max with DPP row operations (VALU, no LDS traffic); the result is valid in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min63(float v) {
  v = fminf(v, dpp_f<0xB1, 0xF>(v, v));      // quad_perm [1,0,3,2]
  v = fminf(v, dpp_f<0x4E, 0xF>(v, v));      // quad_perm [2,3,0,1]
  v = fminf(v, dpp_f<0x141, 0xF>(v, v));     // row_half_mirror
  v = fminf(v, dpp_f<0x140, 0xF>(v, v));     // row_mirror: every lane of a row holds the row's minimum
  v = fminf(v, dpp_f<0x142, 0xA>(v, v));     // row_bcast15 into rows 1 and 3
  v = fminf(v, dpp_f<0x143, 0xC>(v, v));     // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_max63(float v) {
  v = fmaxf(v, dpp_f<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f<0x143, 0xC>(v, v));
  return v;
}
constexpr int BOX_FLOATS = 12;               // per range: decided min xyz, max xyz | undecided min xyz, max xyz
constexpr int BOX_SHIFT = 9;                 // a range = the 64 * RF_IT = 512 positions one wave of phase A walks
constexpr float BOX_EMPTY = 3.0e38f;

__global__ __launch_bounds__(RFB) void k_radius_flags(
    const float4 *__restrict__ pq, const long long *__restrict__ n_used_ptr, long long n_max, int nb, int S,
    uint8_t *__restrict__ flags, int *__restrict__ queue, float *__restrict__ boxes) {
  __shared__ float4 s_pt[RF_LIST];
  __shared__ int s_pos[RF_LIST], s_cnt[RF_LIST], s_q[RF_LIST];
  __shared__ int s_n, s_nq, s_base;
  long long n_used = n_max;
  if (n_used_ptr) { const long long u = *n_used_ptr; n_used = u < n_max ? u : n_max; }
  const long long wg0 = (long long)blockIdx.x * RF_WG;
  if (wg0 >= n_used) return;
  if (threadIdx.x == 0) { s_n = 0; s_nq = 0; }
  __syncthreads();
  const int lane = lane_id();
  const long long w0 = wg0 + (long long)(threadIdx.x >> 6) * (64 * RF_IT);
  float4 p[RF_IT];
#pragma unroll
  for (int it = 0; it < RF_IT; it++) {
    const long long i = w0 + it * 64 + lane;
    p[it] = (i < n_used) ? pq[i] : make_float4(0.f, 0.f, 0.f, __uint_as_float(RF_NOSEG << 16));
  }
  float bx[BOX_FLOATS];                        // per lane, reduced over the wave once at the end
#pragma unroll
  for (int k = 0; k < BOX_FLOATS; k++) bx[k] = ((k % 6) < 3) ? BOX_EMPTY : -BOX_EMPTY;
#pragma unroll
  for (int it = 0; it < RF_IT; it++) {
    const long long i = w0 + it * 64 + lane;
    const float x = p[it].x, y = p[it].y, z = p[it].z;
    const uint32_t wb = __float_as_uint(p[it].w);
    const uint32_t seg = wb >> 16;
    const float r = __uint_as_float(wb << 16);
    const bool valid = seg < (uint32_t)S;           // "no segment" mark, or a slot nobody wrote: never a point
    if (__ballot(valid) == 0ull) continue;
    const bool active = valid && (r > 0.0f);
    if (valid && !active) flags[i] = (r == 0.0f) ? 1 : 0;     // r == 0: no filter; r < 0 / NaN: drop all
    const float thr2 = rf_certain_hit2(x, y, z, r);
    int cnt = 1;                                    // the query itself (d = 0 < r^2)
    {
      const float x1 = __shfl_xor(x, 1, 64), y1 = __shfl_xor(y, 1, 64), z1 = __shfl_xor(z, 1, 64);
      const float x2 = __shfl_xor(x, 2, 64), y2 = __shfl_xor(y, 2, 64), z2 = __shfl_xor(z, 2, 64);
      const uint32_t s1 = (uint32_t)__shfl_xor((int)seg, 1, 64), s2 = (uint32_t)__shfl_xor((int)seg, 2, 64);
      float dx = x - x1, dy = y - y1, dz = z - z1;
      cnt += (s1 == seg && dx * dx + dy * dy + dz * dz < thr2) ? 1 : 0;
      dx = x - x2; dy = y - y2; dz = z - z2;
      cnt += (s2 == seg && dx * dx + dy * dy + dz * dz < thr2) ? 1 : 0;
    }
    if (__ballot(active && cnt <= nb)) {
      const int l1 = lane ^ 1, l2 = lane ^ 2;
      for (int j = 0; j < 64; j += RF_STRIDE) {
        const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
        const float yj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), j));
        const float zj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), j));
        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)seg, j);
        const float dx = x - xj, dy = y - yj, dz = z - zj;
        if (sj == seg && dx * dx + dy * dy + dz * dz < thr2 && j != lane && j != l1 && j != l2) cnt++;
        if (__ballot(active && cnt <= nb) == 0ull) break;
      }
    }
    const bool pending = active && cnt <= nb;
    if (active && !pending) flags[i] = 1;
    if (valid) {   // the range's two boxes: points decided here | points left undecided
      const int o = pending ? 6 : 0;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int q = k * 6;
        if (q == o) {
          bx[q + 0] = fminf(bx[q + 0], x); bx[q + 1] = fminf(bx[q + 1], y); bx[q + 2] = fminf(bx[q + 2], z);
          bx[q + 3] = fmaxf(bx[q + 3], x); bx[q + 4] = fmaxf(bx[q + 4], y); bx[q + 5] = fmaxf(bx[q + 5], z);
        }
      }
    }
    const unsigned long long pm = __ballot(pending);
    if (pm) {                       // workgroup-local list (LDS), one LDS atomic per wave
      int slot0 = 0;
      if (lane == 0) slot0 = atomicAdd(&s_n, __popcll(pm));
      slot0 = __builtin_amdgcn_readfirstlane(slot0);
      if (pending) {
        const int slot = slot0 + __popcll(pm & ((1ull << lane) - 1ull));
        if (slot < RF_LIST) {
          s_pt[slot] = p[it];
          s_pos[slot] = (int)(i - wg0);
          s_cnt[slot] = cnt;
        } else {                     // list full (a pathological tile): straight to phase B
          const int g = atomicAdd(&queue[0], 1);
          queue[2 + g] = (int)i;
        }
      }
    }
  }
  {
#pragma unroll
    for (int k = 0; k < BOX_FLOATS; k++) bx[k] = ((k % 6) < 3) ? wave_min63(bx[k]) : wave_max63(bx[k]);
    if (lane == 63 && w0 < n_used) {
      float4 *o = (float4 *)(boxes + (size_t)(w0 >> BOX_SHIFT) * BOX_FLOATS);
      o[0] = make_float4(bx[0], bx[1], bx[2], bx[3]);
      o[1] = make_float4(bx[4], bx[5], bx[6], bx[7]);
      o[2] = make_float4(bx[8], bx[9], bx[10], bx[11]);
    }
  }
  __syncthreads();
  const int np = min(s_n, RF_LIST);
  if (np == 0) return;
  // pairwise among the workgroup's undecided points.  A partner that phase A already looked at
  // (same chunk: lane^1, lane^2 and the broadcast lanes 0, 8, ...) is skipped: counted or not, it
  // must not be counted twice.
  for (int a = threadIdx.x; a < np; a += RFB) {
    const float4 q = s_pt[a];
    const uint32_t wa = __float_as_uint(q.w);
    const uint32_t seg = wa >> 16;
    const float thr2 = rf_certain_hit2(q.x, q.y, q.z, __uint_as_float(wa << 16));
    int cnt = s_cnt[a];
    const int pa = s_pos[a];
    for (int b = 0; b < np && cnt <= nb; b++) {
      const float4 o = s_pt[b];
      if ((__float_as_uint(o.w) >> 16) != seg || b == a) continue;
      const int pb = s_pos[b];
      if ((pb >> 6) == (pa >> 6)) {
        const int lb = pb & 63, la = pa & 63;
        if (lb == (la ^ 1) || lb == (la ^ 2) || (lb & (RF_STRIDE - 1)) == 0) continue;
      }
      const float dx = q.x - o.x, dy = q.y - o.y, dz = q.z - o.z;
      if (dx * dx + dy * dy + dz * dz < thr2) cnt++;
    }
    if (cnt > nb) flags[wg0 + pa] = 1;
    else s_q[atomicAdd(&s_nq, 1)] = pa;
  }
  __syncthreads();
  const int nq = s_nq;
  if (nq == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&queue[0], nq);        // one global atomic per workgroup
  __syncthreads();
  for (int k = threadIdx.x; k < nq; k += RFB) queue[2 + s_base + k] = (int)(wg0 + s_q[k]);
}

// squared distance from q to an axis-aligned box (0 inside)
__device__ __forceinline__ float box_dist2(float qx, float qy, float qz, float lx, float ly, float lz,
                                           float hx, float hy, float hz) {
  const float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.0f), dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.0f),
              dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.0f);
  return dx * dx + dy * dy + dz * dz;
}

// Phase B: one wave per queued query.
#ifndef DFU3D_RF_RANGES_PER_STEP
#define DFU3D_RF_RANGES_PER_STEP 1
#endif
constexpr int RF_RQ = DFU3D_RF_RANGES_PER_STEP;   // candidate ranges whose points are requested together (tuning constant; 2 is the `rf2` dev build)
__global__ __launch_bounds__(256) void k_radius_resolve(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const float4 *__restrict__ pq, const float *__restrict__ boxes, const long long *__restrict__ seg_base,
    const int *__restrict__ seg_cnt, const double *__restrict__ radius, int nb, int S, long long pool_cap,
    uint8_t *__restrict__ flags, const int *__restrict__ queue) {
  const int nq = (int)min((long long)queue[0], pool_cap);
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  const int lane = lane_id();
  for (int e = wave; e < nq; e += nwaves) {
    const long long i = queue[2 + e];
    const float4 qf = pq[i];
    const int s = (int)(__float_as_uint(qf.w) >> 16);     // < S: phase A queues nothing else
    if (s >= S) continue;
    const int n = max(seg_cnt[s], 0);
    const long long base = seg_base[s], end = base + n;
    if (i < base || i >= end) continue;                   // cannot happen for a shadow built from this table
    const double r = radius[s], r2 = r * r;
    const double x = px[i], y = py[i], z = pz[i];
    // float32 screening: certainly inside below lo2, certainly outside above hi2, fp64 in between
    const float rf = (float)r;
    const float eb = rf_bound(qf.x, qf.y, qf.z, rf) + rf * 2.4e-7f;
    const float tl = rf - eb, th = rf + eb;
    const float lo2 = (tl > 0.0f) ? tl * tl * 0.999999f : -1.0f;
    const float hi2 = th * th * 1.000001f;
    int cnt = 0;
    // the points of chunk c (pool positions 64c .. 64c+63) that belong to the segment
    auto test_chunk = [&](long long c) {
      const long long g = (c << 6) + lane;
      const bool in = (g >= base) && (g < end);
      bool hit = false;
      if (in) {
        const float4 o = pq[g];
        const float dx = qf.x - o.x, dy = qf.y - o.y, dz = qf.z - o.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        hit = d2 < lo2;
        if (!hit && !(d2 > hi2)) {                   // too close to call in float32
          const double ex = x - px[g], ey = y - py[g], ez = z - pz[g];
          double d = ex * ex;
          d += ey * ey;
          d += ez * ez;
          hit = d < r2;
        }
      }
      cnt += __popcll(__ballot(hit));
    };
    const long long c_own = i >> 6;
    test_chunk(c_own);                               // the query itself is counted here (d = 0 < r2)
    const long long r_lo = base >> BOX_SHIFT, r_hi = (end - 1) >> BOX_SHIFT;
    for (long long r0 = r_lo; r0 <= r_hi && cnt <= nb; r0 += 64) {
      const long long rg = r0 + lane;
      bool cand = false;
      if (rg <= r_hi) {
        const float4 *bp = (const float4 *)(boxes + (size_t)rg * BOX_FLOATS);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2];
        cand = box_dist2(qf.x, qf.y, qf.z, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y) <= hi2 ||
               box_dist2(qf.x, qf.y, qf.z, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w) <= hi2;
      }
      unsigned long long m = __ballot(cand);
      while (m && cnt <= nb) {                       // uniform: the candidate ranges of this step, RF_RQ at a time
        // the eight chunks of a range are requested together (one memory round trip, not eight in a row: a query's
        // candidate ranges were a chain of dependent loads, and the kernel's 0.24 ms was that chain)
        constexpr int RC = 1 << (BOX_SHIFT - 6);
        long long c0[RF_RQ];
        bool have[RF_RQ];
#pragma unroll
        for (int q = 0; q < RF_RQ; q++) {
          have[q] = m != 0ull;
          const int k = have[q] ? __ffsll((long long)m) - 1 : 0;
          m &= m - 1ull;                             // (0 stays 0)
          c0[q] = (r0 + k) << (BOX_SHIFT - 6);
        }
        float4 o[RF_RQ * RC];
        bool in[RF_RQ * RC];
#pragma unroll
        for (int u = 0; u < RF_RQ * RC; u++) {
          const long long c = c0[u / RC] + (u % RC);
          const long long g = (c << 6) + lane;
          in[u] = have[u / RC] && (g >= base) && (g < end) && (c != c_own);
          o[u] = in[u] ? pq[g] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < RF_RQ * RC; u++) {
          bool hit = false;
          if (in[u]) {
            const float dx = qf.x - o[u].x, dy = qf.y - o[u].y, dz = qf.z - o[u].z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            hit = d2 < lo2;
            if (!hit && !(d2 > hi2)) {                 // too close to call in float32
              const long long g = ((c0[u / RC] + (u % RC)) << 6) + lane;
              const double ex = x - px[g], ey = y - py[g], ez = z - pz[g];
              double d = ex * ex;
              d += ey * ey;
              d += ez * ez;
              hit = d < r2;
            }
          }
          cnt += __popcll(__ballot(hit));
        }
      }
    }
    if (lane == 0) flags[i] = (cnt > nb) ? 1 : 0;
  }
}

// standalone use of the filter (no k_seg_write in front): float32 shadow of the given segments
__global__ __launch_bounds__(QT) void k_shadow_build(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt,
    const double *__restrict__ radius, int S, const int *__restrict__ tile_off, float4 *__restrict__ pq) {
  const int ntile = tile_off[S];
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int s = find_segment(tile_off, S, t);
    const int q = (t - tile_off[s]) * QT + threadIdx.x;
    if (q >= seg_cnt[s]) continue;
    const long long i = seg_base[s] + q;
    pq[i] = make_float4((float)px[i], (float)py[i], (float)pz[i], __uint_as_float(shadow_word(s, radius[s])));
  }
}

// in-place ordered compaction of SHORT lists (the per-instance LiDAR lists): one wave per segment
__global__ __launch_bounds__(256) void k_seg_compact_short(
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
    const long long *__restrict__ seg_base, int *__restrict__ seg_cnt, const uint8_t *__restrict__ flags, int S) {
  const int s = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (s >= S) return;
  const int n = seg_cnt[s];
  if (n == 0) return;
  const long long base = seg_base[s];
  const int lane = lane_id();
  int running = 0;
  for (int t0 = 0; t0 < n; t0 += 64) {
    const int i = t0 + lane;
    const bool f = (i < n) && flags[base + i];
    double x = 0.0, y = 0.0, z = 0.0;
    if (f) { x = px[base + i]; y = py[base + i]; z = pz[base + i]; }
    const unsigned long long m = __ballot(f);
    if (f) {                                        // dst <= src of every lane: loads above come first
      const long long d = base + running + __popcll(m & ((1ull << lane) - 1ull));
      px[d] = x; py[d] = y; pz[d] = z;
    }
    running += __popcll(m);
  }
  if (lane == 0) seg_cnt[s] = running;
}

// ---------------------------------------------------------------- a12 ball query
// exists-within-C with a spatial hash.  Coordinates are quantised to units of
// u = C(1+1e-5)/16 (17 bits per axis, +-409 m for C = 0.1); a cell is 32 units
// = 2C(1+1e-5) wide.  The instance's LiDAR points (at most 4096) are chained
// per cell in LDS: head[hash(cell)] -> node -> node ..., a node being ONE 64-bit
// word (3 x 17-bit quantised coordinate | 13-bit next link), so that a dense
// cell costs one LDS read per point and no probing.  A pseudo point q can only
// be within C of points whose cells meet [q-C', q+C'] (C' = C(1+1e-6)): at most
// two cells per axis, and because fp rounding of the quantisation is monotone no
// candidate is missed.  For every node on those chains the quantised
// coordinates give a lower bound of the distance that discards almost every
// non-neighbour without touching memory; the rest is decided by the
// reference's predicate on the fp64 coordinates (d2 < T <=> sqrt(d2) < C).
// A workgroup takes one query tile of one instance and builds that instance's table
// (one tile per workgroup measured best:
// more, smaller workgroups balance better than amortising the build).  Instances with more
// LiDAR points than the table holds, or beyond the quantised range, use the
// brute-force tile loop.
// Two builds of the kernel share the work by the size of the instance's LiDAR list: almost every instance has a
// few hundred LiDAR points, so its table fits 16 KB of LDS and 256-thread workgroups -- eight of them per compute
// unit instead of two, which is what hides the dependent loads at the start of a workgroup (segment search,
// segment facts, LiDAR points); the big build (64 KB, 1024 threads) takes the rest and the brute-force case.
constexpr int BT_BIG = 1024, BH_HEADS_BIG = 8192, BH_MAX_BIG = 4096;      // 32 KB + 32 KB of LDS; 12-bit node index
constexpr int BT_SMALL = 256, BH_HEADS_SMALL = 2048, BH_MAX_SMALL = 1024; // 8 KB + 8 KB of LDS

// query tiles of the segments whose LiDAR list has lo < cnt_a <= hi entries
__global__ __launch_bounds__(1024) void k_tile_scan_class(int S, const int *__restrict__ cnt, const int *__restrict__ cnt_a,
                                                          int lo, int hi, int *__restrict__ tile_off, int qt) {
  __shared__ int s_w[16];
  int running = 0;
  for (int b0 = 0; b0 < S; b0 += 1024) {
    const int s = b0 + threadIdx.x;
    int nt = 0;
    if (s < S) {
      const int na = cnt_a[s];
      if (na > lo && na <= hi) nt = (cnt[s] + qt - 1) / qt;
    }
    int tot;
    const int ex = block_excl_scan<16>(nt, s_w, tot);
    if (s < S) tile_off[s] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) tile_off[S] = running;
}

__device__ __forceinline__ uint32_t bh_hash(uint32_t ix, uint32_t iy, uint32_t iz) {
  uint32_t h = ix * 0x9E3779B1u ^ iy * 0x85EBCA77u ^ iz * 0xC2B2AE3Du;
  h ^= h >> 15;
  return h;
}

template <int BT, int BH_MAX, int BH_HEADS>
__global__ __launch_bounds__(BT) void k_ball_flags(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ base_a,
    const int *__restrict__ cnt_a, const long long *__restrict__ base_b,
    const int *__restrict__ cnt_b, double T, double C, int S, const int *__restrict__ tile_off,
    uint8_t *__restrict__ flags, int masked) {
  __shared__ unsigned long long s_node[BH_MAX];
  __shared__ uint32_t s_head[BH_HEADS];
  __shared__ int s_pending;
  const int ntile = tile_off[S];
  int t = blockIdx.x;
  if (t >= ntile) return;
  const int t_end = t + 1;
  int s = find_segment(tile_off, S, t);
  const double inv = 16.0 / (C * (1.0 + 1e-5));  // quantisation: 16 units per C
  const double Cq = C * (1.0 + 1e-6);
  const double OFF = 65536.0;                    // quantised coordinates are stored with this offset
  int hashed_s = -1;                             // segment whose LiDAR points are in the table
  bool hash_ok = false;
  uint32_t mask = 0;
  for (; t < t_end; t++) {
    while (tile_off[s + 1] <= t) s++;
    const int q0 = (t - tile_off[s]) * BT;
    const int nq = cnt_b[s], na = cnt_a[s];
    const long long bq = base_b[s], ba = base_a[s];
    const int q = q0 + threadIdx.x;
    // masked: flags hold the keep mask of a preceding filter that was not compacted;
    // a dropped point is not a query and stays dropped
    const bool valid = (q < nq) && (!masked || flags[bq + q]);
    if (na == 0) {                       // my_loader.py:602: fuse skipped
      if (valid) flags[bq + q] = 1;
      continue;
    }
    double x = 0.0, y = 0.0, z = 0.0;
    if (valid) { x = px[bq + q]; y = py[bq + q]; z = pz[bq + q]; }
    if (na <= BH_MAX && hashed_s != s) {         // (re)build the table -- uniform per workgroup
      int slots = 256;
      while (slots < 2 * na) slots <<= 1;
      mask = (uint32_t)slots - 1u;
      __syncthreads();                           // queries of the previous tile are done
      for (int i = threadIdx.x; i < slots; i += BT) s_head[i] = 0u;
      if (threadIdx.x == 0) s_pending = 0;
      __syncthreads();
      bool too_wide = false;
      for (int i = threadIdx.x; i < na; i += BT) {
        const double fx = floor(px[ba + i] * inv) + OFF, fy = floor(py[ba + i] * inv) + OFF,
                     fz = floor(pz[ba + i] * inv) + OFF;
        if (!(fx >= 64.0 && fy >= 64.0 && fz >= 64.0 && fx < 131000.0 && fy < 131000.0 && fz < 131000.0)) {
          too_wide = true;
          continue;
        }
        const uint32_t ix = (uint32_t)fx, iy = (uint32_t)fy, iz = (uint32_t)fz;
        const uint32_t prev = atomicExch(&s_head[bh_hash(ix >> 5, iy >> 5, iz >> 5) & mask], (uint32_t)i + 1u);
        s_node[i] = (unsigned long long)ix | ((unsigned long long)iy << 17) |
                    ((unsigned long long)iz << 34) | ((unsigned long long)prev << 51);
      }
      if (too_wide) s_pending = 1;
      __syncthreads();
      hash_ok = (s_pending == 0);
      hashed_s = s;
    }
    bool found = false;
    if (na <= BH_MAX && hash_ok) {
      if (valid) {
        // quantised range [q - C', q + C'] on each axis (NaN / far-away queries fail the range test)
        const double lx = floor((x - Cq) * inv) + OFF, hx = floor((x + Cq) * inv) + OFF;
        const double ly = floor((y - Cq) * inv) + OFF, hy = floor((y + Cq) * inv) + OFF;
        const double lz = floor((z - Cq) * inv) + OFF, hz = floor((z + Cq) * inv) + OFF;
        // stored coordinates lie in [64, 131000); a range that misses [0, 131071] entirely
        // (or is NaN / infinite) cannot contain one
        if (hx >= 0.0 && hy >= 0.0 && hz >= 0.0 && lx <= 131071.0 && ly <= 131071.0 && lz <= 131071.0) {
          const int x0 = (int)fmax(lx, 0.0) >> 5, x1 = (int)fmin(hx, 131071.0) >> 5;
          const int y0 = (int)fmax(ly, 0.0) >> 5, y1 = (int)fmin(hy, 131071.0) >> 5;
          const int z0 = (int)fmax(lz, 0.0) >> 5, z1 = (int)fmin(hz, 131071.0) >> 5;
          // the query's own quantised position; a query outside the quantised range skips the
          // lower bound and tests every node of its chains exactly
          const double fqx = floor(x * inv) + OFF, fqy = floor(y * inv) + OFF, fqz = floor(z * inv) + OFF;
          const bool qin = fqx >= 0.0 && fqy >= 0.0 && fqz >= 0.0 && fqx <= 131071.0 && fqy <= 131071.0 && fqz <= 131071.0;
          const int qx = qin ? (int)fqx : 0, qy = qin ? (int)fqy : 0, qz = qin ? (int)fqz : 0;
          // walk a chain from `node`, whose word `wv` the caller has read already
          auto walk_from = [&](uint32_t node, unsigned long long wv) {
            while (true) {
              const int ax = (int)(wv & 0x1FFFFull), ay = (int)((wv >> 17) & 0x1FFFFull),
                        az = (int)((wv >> 34) & 0x1FFFFull);
              // each quantised difference is within 1 (+3e-11) unit of the true one, so the
              // true distance is at least |max(|d|-1, 0)| units, and C is 16/(1+1e-5) < 16 units:
              // 258 > 16.06^2 leaves room for the rounding of the quantisation itself
              const int ex_ = max(abs(ax - qx) - 1, 0), ey_ = max(abs(ay - qy) - 1, 0),
                        ez_ = max(abs(az - qz) - 1, 0);
              if (!qin || ex_ * ex_ + ey_ * ey_ + ez_ * ez_ <= 257) {
                const int j = (int)node - 1;
                const double ex = x - px[ba + j], ey = y - py[ba + j], ez = z - pz[ba + j];
                double d = ex * ex;
                d += ey * ey;
                d += ez * ez;
                if (d < T) { found = true; return; }         // <=> sqrt(d) < C, see dfu3d_ballquery_fuse
              }
              node = (uint32_t)(wv >> 51);
              if (!node) return;
              wv = s_node[node - 1u];
            }
          };
          auto walk = [&](uint32_t node) { if (node) walk_from(node, s_node[node - 1u]); };
          if (x1 - x0 <= 1 && y1 - y0 <= 1 && z1 - z0 <= 1) {
            // the usual case, at most 2 x 2 x 2 cells: all eight heads are read before the first chain is walked
            // (one LDS round trip instead of eight in a row -- most cells are empty, the reads were the cost)
            uint32_t heads[8];
#pragma unroll
            for (int cidx = 0; cidx < 8; cidx++) {
              const int ux = x0 + (cidx & 1), uy = y0 + ((cidx >> 1) & 1), uz = z0 + (cidx >> 2);
              const bool there = ux <= x1 && uy <= y1 && uz <= z1;
              heads[cidx] = there ? s_head[bh_hash((uint32_t)ux, (uint32_t)uy, (uint32_t)uz) & mask] : 0u;
            }
#pragma unroll
            for (int cidx = 0; cidx < 8; cidx++)               // (reading the chains' first nodes ahead as well gained nothing)
              if (!found) walk(heads[cidx]);
          } else {
            for (int uz = z0; uz <= z1 && !found; uz++)
              for (int uy = y0; uy <= y1 && !found; uy++)
                for (int ux = x0; ux <= x1 && !found; ux++)
                  walk(s_head[bh_hash((uint32_t)ux, (uint32_t)uy, (uint32_t)uz) & mask]);
          }
        }
        flags[bq + q] = found ? 1 : 0;
      }
      continue;
    }
    // brute force over LDS tiles (more LiDAR points than the table holds, or a huge extent)
    hashed_s = -1;                               // the tiles below overwrite the table
    constexpr int PTB = BH_MAX / 3;              // points per tile: x | y | z in the nodes' LDS
    double *sx = (double *)s_node, *sy = sx + PTB, *sz = sy + PTB;
    for (int j0 = 0; j0 < na; j0 += PTB) {
      const int m = min(PTB, na - j0);
      __syncthreads();
      if (threadIdx.x == 0) s_pending = 0;
      for (int i = threadIdx.x; i < m; i += BT) {
        sx[i] = px[ba + j0 + i];
        sy[i] = py[ba + j0 + i];
        sz[i] = pz[ba + j0 + i];
      }
      __syncthreads();
      if (valid && !found) {
        for (int j = 0; j < m; j++) {
          const double dx = x - sx[j], dy = y - sy[j], dz = z - sz[j];
          double d = dx * dx;
          d += dy * dy;
          d += dz * dz;
          if (d < T) { found = true; break; }
        }
        if (!found) s_pending = 1;
      }
      __syncthreads();
      if (!s_pending) break;
    }
    if (valid) flags[bq + q] = found ? 1 : 0;
  }
}

// ---------------------------------------------------------------- compaction
// In-order compaction of segment s by flags.  dst = src (in place) or, when
// dst_after_base != nullptr, directly behind another segment
// (dst_after_base[s] + dst_after_cnt[s]); base_out[s] is updated then.
constexpr int CPT = 1024;   // threads per compaction workgroup
constexpr int CPE = 2;      // consecutive elements per thread (4 needed 76 VGPRs: one 1024-thread workgroup per CU)
__global__ __launch_bounds__(CPT) void k_seg_compact(
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
    long long *__restrict__ seg_base, int *__restrict__ seg_cnt,
    const uint8_t *__restrict__ flags, const long long *__restrict__ dst_after_base,
    const int *__restrict__ dst_after_cnt) {
  __shared__ int s_w[CPT / 64];
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  const long long src = seg_base[s];
  const long long dst = dst_after_base ? dst_after_base[s] + dst_after_cnt[s] : src;
  if (n == 0) {
    if (dst_after_base && threadIdx.x == 0) seg_base[s] = dst;
    return;
  }
  int running = 0;
  for (int t0 = 0; t0 < n; t0 += CPT * CPE) {
    const int i0 = t0 + threadIdx.x * CPE;
    bool f[CPE];
    double x[CPE], y[CPE], z[CPE];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < CPE; k++) {
      const int i = i0 + k;
      f[k] = (i < n) && flags[src + i];
      x[k] = y[k] = z[k] = 0.0;
      if (f[k]) { x[k] = px[src + i]; y[k] = py[src + i]; z[k] = pz[src + i]; }
      mine += f[k] ? 1 : 0;
    }
    int tot;
    int r = block_excl_scan<CPT / 64>(mine, s_w, tot);   // barriers: loads above complete first
#pragma unroll
    for (int k = 0; k < CPE; k++) {
      if (f[k]) {
        const long long d = dst + running + r;            // d <= src + i: never ahead of the reads
        px[d] = x[k]; py[d] = y[k]; pz[d] = z[k];
        r++;
      }
    }
    running += tot;
  }
  if (threadIdx.x == 0) {
    seg_cnt[s] = running;
    if (dst_after_base) seg_base[s] = dst;
  }
}

// ---------------------------------------------------------------- a11 statistical
constexpr int KMAX = 64;
__global__ __launch_bounds__(QT) void k_knn_mean(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ seg_base,
    const int *__restrict__ seg_cnt, const int *__restrict__ enable, int knn, int S,
    const int *__restrict__ tile_off, double *__restrict__ mean_d) {
  __shared__ double sx[PT], sy[PT], sz[PT];
  const int t = blockIdx.x;
  if (t >= tile_off[S]) return;
  const int s = find_segment(tile_off, S, t);
  if (!enable[s]) return;
  const int q0 = (t - tile_off[s]) * QT;
  const int n = seg_cnt[s];
  const long long base = seg_base[s];
  const int q = q0 + threadIdx.x;
  const bool valid = q < n;
  double x = 0.0, y = 0.0, z = 0.0;
  if (valid) { x = px[base + q]; y = py[base + q]; z = pz[base + q]; }
  const int kk = min(knn, n);
  double best[KMAX];                 // ascending squared distances
  int nbest = 0;
  for (int j0 = 0; j0 < n; j0 += PT) {
    const int m = min(PT, n - j0);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += QT) {
      sx[i] = px[base + j0 + i];
      sy[i] = py[base + j0 + i];
      sz[i] = pz[base + j0 + i];
    }
    __syncthreads();
    if (valid) {
      for (int j = 0; j < m; j++) {
        const double dx = x - sx[j], dy = y - sy[j], dz = z - sz[j];
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        if (nbest < kk || d < best[nbest - 1]) {
          int p = (nbest < kk) ? nbest : kk - 1;
          while (p > 0 && best[p - 1] > d) { best[p] = best[p - 1]; p--; }
          best[p] = d;
          if (nbest < kk) nbest++;
        }
      }
    }
  }
  if (valid) {
    double sum = 0.0;
    for (int i = 0; i < nbest; i++) sum += sqrt(best[i]);
    mean_d[base + q] = nbest > 0 ? sum / (double)nbest : -1.0;
  }
}

// per segment: mu, sigma (Bessel) over mean distances, then flags
__global__ __launch_bounds__(256) void k_stat_flags(
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt,
    const int *__restrict__ enable, double std_ratio, const double *__restrict__ mean_d,
    uint8_t *__restrict__ flags) {
  __shared__ double s_red[4];
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  const long long base = seg_base[s];
  if (!enable[s]) {
    for (int i = threadIdx.x; i < n; i += 256) flags[base + i] = 1;
    return;
  }
  if (n == 0) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = mean_d[base + i];
    acc += (v > 0.0) ? v : 0.0;
  }
  acc = wave_sum_d(acc);
  if (lane_id() == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const double mu = (((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]) / (double)n;
  __syncthreads();
  acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = mean_d[base + i];
    acc += (v > 0.0) ? (v - mu) * (v - mu) : 0.0;
  }
  acc = wave_sum_d(acc);
  if (lane_id() == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const double sq = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  const double sd = sqrt(sq / (double)(n - 1));     // n == 1 -> NaN -> nothing kept
  const double thr = mu + std_ratio * sd;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = mean_d[base + i];
    flags[base + i] = (v > 0.0 && v < thr) ? 1 : 0;
  }
}

inline int tile_grid(int64_t pool_cap, int S) {
  return (int)((pool_cap + QT - 1) / QT + S);
}

}  // namespace

extern "C" int64_t dfu3d_segments_scratch_words(int32_t V, int32_t a_cap, int32_t b_cap) {
  if (V <= 0 || a_cap <= 0 || b_cap <= 0) return DFU3D_EINVAL;
  return (int64_t)V * DFU3D_MAX_INST * ((a_cap + SEG_CH - 1) / SEG_CH + (b_cap + SEG_CH - 1) / SEG_CH);
}

extern "C" int dfu3d_segments_build(
    const uint32_t *a_bits, const double *a_x, const double *a_y, const double *a_z,
    const int32_t *a_n, int32_t a_cap, const uint32_t *b_bits, const double *b_x,
    const double *b_y, const double *b_z, const int32_t *b_n, int32_t b_cap, int32_t V,
    int32_t max_inst, int64_t pool_cap, int64_t *pool_cursor, double *px, double *py,
    double *pz, int64_t *base_a, int32_t *cnt_a, int64_t *base_b, int32_t *cnt_b,
    uint32_t *status, const double *rad_a, const double *rad_b, void *shadow, int64_t *base_ab,
    int32_t *cnt_ab, double *rad_ab, int32_t *chunk_cnt, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!a_bits || !a_x || !a_y || !a_z || !a_n || !b_bits || !b_x || !b_y || !b_z || !b_n ||
      !pool_cursor || !px || !py || !pz || !base_a || !cnt_a || !base_b || !cnt_b || !status || !chunk_cnt)
    return DFU3D_EINVAL;
  if (V <= 0 || max_inst <= 0 || a_cap <= 0 || b_cap <= 0 || pool_cap <= 0) return DFU3D_EINVAL;
  if (max_inst > DFU3D_MAX_INST) return DFU3D_ERANGE;
  if ((shadow || base_ab) && (!rad_a || !rad_b)) return DFU3D_EINVAL;
  if (base_ab && (!cnt_ab || !rad_ab)) return DFU3D_EINVAL;
  const int S = V * max_inst;
  if (shadow && 2 * (int64_t)S >= (int64_t)RF_NOSEG) return DFU3D_ERANGE;   // 16-bit segment ids in the shadow
  if (shadow && ((uintptr_t)shadow & 15u)) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nca = (a_cap + SEG_CH - 1) / SEG_CH, ncb = (b_cap + SEG_CH - 1) / SEG_CH;
  int32_t *cc_a = chunk_cnt, *cc_b = chunk_cnt + (size_t)V * nca * DFU3D_MAX_INST;
  if (hipMemsetAsync(cnt_a, 0, sizeof(int32_t) * S, st) != hipSuccess) return DFU3D_ELAUNCH;
  if (hipMemsetAsync(cnt_b, 0, sizeof(int32_t) * S, st) != hipSuccess) return DFU3D_ELAUNCH;
  hipLaunchKernelGGL(k_seg_count, dim3(nca, V), dim3(SEG_WAVES * 64), 0, st, a_bits, a_n, a_cap, max_inst, cnt_a, cc_a);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_count, dim3(ncb, V), dim3(SEG_WAVES * 64), 0, st, b_bits, b_n, b_cap, max_inst, cnt_b, cc_b);
  DFU3D_LAUNCH_CHECK();
  const JointSegs J = {(long long *)base_ab, cnt_ab, rad_ab, rad_a, rad_b};
  hipLaunchKernelGGL(k_seg_alloc, dim3(1), dim3(1024), 0, st, S, cnt_a, cnt_b,
                     (long long *)base_a, (long long *)base_b, (long long)pool_cap,
                     (long long *)pool_cursor, status, J);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_write, dim3(nca, V), dim3(SWT), 0, st, a_bits, a_x, a_y, a_z, a_n, a_cap,
                     max_inst, (const long long *)base_a, cnt_a, cc_a, px, py, pz, (float4 *)shadow, rad_a, 0);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_write, dim3(ncb, V), dim3(SWT), 0, st, b_bits, b_x, b_y, b_z, b_n, b_cap,
                     max_inst, (const long long *)base_b, cnt_b, cc_b, px, py, pz, (float4 *)shadow, rad_b, S);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_radius_filter(double *px, double *py, double *pz, const int64_t *seg_base,
                                   int32_t *seg_cnt, const double *radius, int32_t nb_points,
                                   int32_t S, int64_t pool_cap, const int64_t *n_used, void *shadow,
                                   int32_t *tile_off, uint8_t *flags, int32_t *queue, int32_t phases,
                                   void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!px || !py || !pz || !seg_base || !seg_cnt || !radius || !tile_off || !flags || !queue || !shadow)
    return DFU3D_EINVAL;
  if (S <= 0 || pool_cap <= 0 || nb_points < 0) return DFU3D_EINVAL;
  if (pool_cap >= (1ll << 31) - 2) return DFU3D_ERANGE;            // queue entries are int32 positions
  if ((int64_t)S >= (int64_t)RF_NOSEG) return DFU3D_ERANGE;         // 16-bit segment ids in the shadow
  if ((uintptr_t)shadow & 15u) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float4 *pq = (float4 *)shadow;
  float *boxes = (float *)(pq + pool_cap);           // 12 floats per 64 pool slots, behind the shadow proper
  if (phases & DFU3D_RF_SHADOW) {
    // positions outside the given segments carry the "no segment" mark (all bits set)
    if (hipMemsetAsync(pq, 0xFF, sizeof(float4) * (size_t)pool_cap, st) != hipSuccess) return DFU3D_ELAUNCH;
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, S, seg_cnt, tile_off, QT);
    DFU3D_LAUNCH_CHECK();
    const int g = tile_grid(pool_cap, S);
    hipLaunchKernelGGL(k_shadow_build, dim3(g < 4096 ? g : 4096), dim3(QT), 0, st, px, py, pz,
                       (const long long *)seg_base, seg_cnt, radius, S, tile_off, pq);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_FLAGS) {
    if (hipMemsetAsync(queue, 0, 2 * sizeof(int), st) != hipSuccess) return DFU3D_ELAUNCH;
    hipLaunchKernelGGL(k_radius_flags, dim3((unsigned)((pool_cap + RF_WG - 1) / RF_WG)), dim3(RFB), 0, st, pq,
                       (const long long *)n_used, (long long)pool_cap, nb_points, S, flags, queue, boxes);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_RESOLVE) {
    hipLaunchKernelGGL(k_radius_resolve, dim3(4096), dim3(256), 0, st, px, py, pz, pq, boxes,
                       (const long long *)seg_base, seg_cnt, radius, nb_points, S, (long long)pool_cap, flags, queue);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_COMPACT) {
    if (phases & DFU3D_RF_SHORT_LISTS) {
      hipLaunchKernelGGL(k_seg_compact_short, dim3((S + 3) / 4), dim3(256), 0, st, px, py, pz,
                         (const long long *)seg_base, seg_cnt, flags, S);
    } else {
      hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz,
                         (long long *)seg_base, seg_cnt, flags, (const long long *)nullptr,
                         (const int *)nullptr);
    }
    DFU3D_LAUNCH_CHECK();
  }
  return DFU3D_OK;
}

extern "C" int dfu3d_stat_filter(double *px, double *py, double *pz, const int64_t *seg_base,
                                 int32_t *seg_cnt, const int32_t *enable, int32_t nb_neighbors,
                                 double std_ratio, int32_t S, int64_t pool_cap,
                                 int32_t *tile_off, uint8_t *flags, double *mean_d,
                                 double *stats, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  (void)stats;
  if (!px || !py || !pz || !seg_base || !seg_cnt || !enable || !tile_off || !flags || !mean_d)
    return DFU3D_EINVAL;
  if (S <= 0 || pool_cap <= 0 || nb_neighbors < 1) return DFU3D_EINVAL;
  if (nb_neighbors > KMAX) return DFU3D_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, S, seg_cnt, tile_off, QT);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_knn_mean, dim3(tile_grid(pool_cap, S)), dim3(QT), 0, st, px, py, pz,
                     (const long long *)seg_base, seg_cnt, enable, nb_neighbors, S, tile_off,
                     mean_d);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_stat_flags, dim3(S), dim3(256), 0, st, (const long long *)seg_base,
                     seg_cnt, enable, std_ratio, mean_d, flags);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz,
                     (long long *)seg_base, seg_cnt, flags, (const long long *)nullptr,
                     (const int *)nullptr);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

static int ballquery_fuse_impl(double *px, double *py, double *pz, const int64_t *base_a,
                                    const int32_t *cnt_a, int64_t *base_b, int32_t *cnt_b,
                                    double C, int32_t S, int64_t pool_cap, int32_t *tile_off,
                                    uint8_t *flags, int masked, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!px || !py || !pz || !base_a || !cnt_a || !base_b || !cnt_b || !tile_off || !flags)
    return DFU3D_EINVAL;
  if (S <= 0 || pool_cap <= 0) return DFU3D_EINVAL;
  if (!(C > 0.0) || !(C < 1e150)) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // The reference tests sqrt(d2) < C (my_loader.py:490-493).  sqrt is monotone
  // and correctly rounded, so that is d2 < T with T the smallest double whose
  // rounded square root reaches C; T is found exactly here, once, on the host.
  double T = C * C;
  while (__builtin_sqrt(T) >= C) T = __builtin_nextafter(T, 0.0);
  while (__builtin_sqrt(T) < C) T = __builtin_nextafter(T, __builtin_inf());
  // small build: instances with at most BH_MAX_SMALL LiDAR points (and those with none: the fuse is skipped there);
  // big build: the others.  tile_off: two lists of S+1 entries.
  int32_t *tile_small = tile_off, *tile_big = tile_off + S + 1;
  hipLaunchKernelGGL(k_tile_scan_class, dim3(1), dim3(1024), 0, st, S, cnt_b, cnt_a, -1, BH_MAX_SMALL, tile_small, BT_SMALL);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_tile_scan_class, dim3(1), dim3(1024), 0, st, S, cnt_b, cnt_a, BH_MAX_SMALL, 0x7FFFFFFF, tile_big, BT_BIG);
  DFU3D_LAUNCH_CHECK();
  const int g_small = (int)((pool_cap + BT_SMALL - 1) / BT_SMALL + S);
  hipLaunchKernelGGL((k_ball_flags<BT_SMALL, BH_MAX_SMALL, BH_HEADS_SMALL>), dim3(g_small), dim3(BT_SMALL), 0, st, px, py, pz,
                     (const long long *)base_a, cnt_a, (const long long *)base_b, cnt_b, T, C, S,
                     tile_small, flags, masked);
  DFU3D_LAUNCH_CHECK();
  const int g_big = (int)((pool_cap + BT_BIG - 1) / BT_BIG + S);
  hipLaunchKernelGGL((k_ball_flags<BT_BIG, BH_MAX_BIG, BH_HEADS_BIG>), dim3(g_big), dim3(BT_BIG), 0, st, px, py, pz,
                     (const long long *)base_a, cnt_a, (const long long *)base_b, cnt_b, T, C, S,
                     tile_big, flags, masked);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz, (long long *)base_b,
                     cnt_b, flags, (const long long *)base_a, cnt_a);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_ballquery_fuse(double *px, double *py, double *pz, const int64_t *base_a,
                                    const int32_t *cnt_a, int64_t *base_b, int32_t *cnt_b,
                                    double C, int32_t S, int64_t pool_cap, int32_t *tile_off,
                                    uint8_t *flags, void *stream) {
  return ballquery_fuse_impl(px, py, pz, base_a, cnt_a, base_b, cnt_b, C, S, pool_cap, tile_off,
                             flags, 0, stream);
}

extern "C" int dfu3d_ballquery_fuse_masked(double *px, double *py, double *pz,
                                           const int64_t *base_a, const int32_t *cnt_a,
                                           int64_t *base_b, int32_t *cnt_b, double C, int32_t S,
                                           int64_t pool_cap, int32_t *tile_off, uint8_t *flags,
                                           void *stream) {
  return ballquery_fuse_impl(px, py, pz, base_a, cnt_a, base_b, cnt_b, C, S, pool_cap, tile_off,
                             flags, 1, stream);
}
