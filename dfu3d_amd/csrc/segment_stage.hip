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
// A view's items are cut into RANGES of SEG_WI = 256 consecutive items, one wave each, in three passes without a barrier
// or a word of LDS in the two that touch the items:
//   k_seg_count  wave per range: lane j counts the range's items of instance j (ballots) -> range_cnt[v][range][j]
//   k_seg_scan   workgroup per (view, side): exclusive prefix over the ranges, in place; the totals are the list lengths
//   k_seg_alloc  list bases in the pool
//   k_seg_write  wave per range: list j of the view starts at base + prefix for this wave; ballots give the order inside
// History: a workgroup per view stepping through 1024 items at a time with two barriers per step was bound by the
// latency of that chain (0.70 ms for 0.2 GB); a workgroup per 32768 items with lane j keeping the instance's running
// count across 16 steps of 2048 items (two barriers each) had 1152 workgroups of sixteen dependent steps on the
// bench workload -- 0.31 ms per launch, the time of that chain and not of the 1 GB it moved (round 2 .. mid round 3).
constexpr int SEG_WI = 256, SEG_STEPS = SEG_WI / 64;
constexpr int SEG_WPB = 4;                          // waves (ranges) per workgroup
constexpr int SEG_GX = 128;                         // workgroups per view at most (each loops over its ranges)
static_assert(DFU3D_MAX_INST == 32, "one lane per instance, instance bits in one 32-bit word; k_seg_scan: 32 x 32 threads");
inline int seg_ranges(int cap_item) { return (cap_item + SEG_WI - 1) / SEG_WI; }

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t x) { return wave_or_u32_dpp(x); }

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

__global__ __launch_bounds__(SEG_WPB * 64) void k_seg_count(const uint32_t *__restrict__ bits,
                                                            const int *__restrict__ n_item, int cap_item, int NR,
                                                            int *__restrict__ range_cnt) {
  const int v = blockIdx.y;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = min(n_item[v], cap_item);
  const int nr = (n + SEG_WI - 1) / SEG_WI;
  for (int r = blockIdx.x * SEG_WPB + wave; r < nr; r += gridDim.x * SEG_WPB) {       // uniform per wave
    uint32_t b[SEG_STEPS], wany[SEG_STEPS];
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) {
      const int t = r * SEG_WI + k * 64 + lane;
      b[k] = (t < n) ? bits[(size_t)v * cap_item + t] : 0u;
    }
    const int mine = wave_instance_counts(b, wany);
    if (lane < DFU3D_MAX_INST) range_cnt[((size_t)v * NR + r) * DFU3D_MAX_INST + lane] = mine;
  }
}

// blockIdx.y = 0: side a, 1: side b.  Thread (g, j) = (threadIdx.x / 32, threadIdx.x % 32) owns rows [g * per, (g + 1) * per)
// of column j of the view's nr x 32 matrix of range counts.
struct SegSide {
  const int *n_item;
  int cap_item, NR;
  int *range_cnt, *cnt;
};
__global__ __launch_bounds__(1024) void k_seg_scan(SegSide A, SegSide B, int max_inst) {
  __shared__ int s_g[32][DFU3D_MAX_INST + 1];
  const SegSide X = blockIdx.y ? B : A;
  const int v = blockIdx.x;
  const int j = threadIdx.x & (DFU3D_MAX_INST - 1), g = threadIdx.x / DFU3D_MAX_INST;
  const int n = min(X.n_item[v], X.cap_item);
  const int nr = (n + SEG_WI - 1) / SEG_WI;
  const int per = (nr + 31) / 32;
  int *col = X.range_cnt + (size_t)v * X.NR * DFU3D_MAX_INST + j;
  const int r0 = g * per, r1 = min(r0 + per, nr);
  int sum = 0;
  for (int r = r0; r < r1; r++) sum += col[(size_t)r * DFU3D_MAX_INST];
  s_g[g][j] = sum;
  __syncthreads();
  int run = 0, tot = 0;
#pragma unroll 8
  for (int q = 0; q < 32; q++) {
    const int c = s_g[q][j];
    run += (q < g) ? c : 0;
    tot += c;
  }
  for (int r = r0; r < r1; r++) {
    const int c = col[(size_t)r * DFU3D_MAX_INST];
    col[(size_t)r * DFU3D_MAX_INST] = run;
    run += c;
  }
  if (g == 0 && j < max_inst) X.cnt[v * max_inst + j] = tot;
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

// One wave per range writes its part of the ordered lists of all the view's instances: lane j knows where list j
// continues for this range (base + the prefix of k_seg_scan), and inside the wave ballots give the order.  The
// coordinates of all SEG_STEPS sub-steps are requested before the first is used.
__global__ __launch_bounds__(SEG_WPB * 64) void k_seg_write(
    const uint32_t *__restrict__ bits, const double *__restrict__ ix,
    const double *__restrict__ iy, const double *__restrict__ iz,
    const int *__restrict__ n_item, int cap_item, int max_inst, int NR,
    const long long *__restrict__ base, const int *__restrict__ cnt, const int *__restrict__ range_cnt,
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
    float4 *__restrict__ pq, const double *__restrict__ rad, int seg_off, int V) {
  // XCD-aware placement (as k_bp_vox): consecutive workgroups are dealt round the eight XCDs, and the ranges of ONE view
  // continue each other's runs in the instance lists -- a run of ~32 points starts and ends inside a line of every
  // plane.  With all workgroups of a view on one XCD the neighbouring runs meet in one L2 and leave as whole lines:
  // 0.318 -> 0.280 ms for the stage (tools/ab_builds.py).
  const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, turn = lin >> 3;
  const int v = (int)((turn / gridDim.x) * 8u + (lin & 7u)), bx = (int)(turn % gridDim.x);
  if (v >= V) return;                                  // (the grid's y is V rounded up to a multiple of 8)
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = min(n_item[v], cap_item);
  const int nr = (n + SEG_WI - 1) / SEG_WI;
  if (bx * SEG_WPB + wave >= nr) return;
  const bool lj = lane < max_inst;                     // lane j < max_inst: the facts of instance j
  const int cj = lj ? cnt[v * max_inst + lane] : 0;    // (0 also for the lists that did not fit the pool)
  const long long bj = lj ? base[v * max_inst + lane] : 0;
  const double rj = (lj && pq) ? rad[v * max_inst + lane] : 0.0;
  const uint32_t live = (uint32_t)__ballot(cj > 0);   // instances of this view with a non-empty list
  if (live == 0u) return;
  const uint32_t sw = (lj && pq) ? shadow_word(seg_off + v * max_inst + lane, rj) : 0u;
  for (int r = bx * SEG_WPB + wave; r < nr; r += gridDim.x * SEG_WPB) {       // uniform per wave
    const int w0 = r * SEG_WI;
    long long woff = bj + ((lane < DFU3D_MAX_INST) ? range_cnt[((size_t)v * NR + r) * DFU3D_MAX_INST + lane] : 0);
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
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) wany[k] = wave_or_u32(b[k]);
#pragma unroll
    for (int k = 0; k < SEG_STEPS; k++) {
      for (uint32_t w = wany[k]; w; w &= w - 1u) {     // uniform per wave
        const int j = __ffs((int)w) - 1;
        const bool has = (b[k] >> j) & 1u;
        const unsigned long long m = __ballot(has);
        // (j is the same in every lane: a read of lane j, not a shuffle through the LDS path)
        const long long start = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(woff >> 32), j) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)(woff & 0xFFFFFFFFll), j));
        const uint32_t swj = (uint32_t)__builtin_amdgcn_readlane((int)sw, j);
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
  }
}

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
// Phase A (k_rf_stream) is a pure stream over the used part of the pool: a wave takes a RANGE of 512 positions (all
// eight loads issued up front, 64 VGPRs = eight waves per SIMD, no LDS, no barrier, no returning atomic) and tests every
// point against its two LIST neighbours (the positions before and behind, inside rows of 16 lanes, DPP row shifts --
// the lists are in pixel / sweep order, so list neighbours are spatial neighbours).  A point with such a neighbour
// is COHERENT; for nb_points = 1 it is decided at once (~98 % of a dense cloud).  The coherent points of a range
// are summarised by one bounding box (they are surface points: the box is tight); every other point -- the
// INCOHERENT ones (depth outliers that became the representative of their voxel, row ends of a mask, sparse
// sweeps) and, for nb_points > 1, the coherent points that could not collect enough neighbours inside their
// chunk -- is LISTED: it goes to the range's U slots (uent / upos[range * 128 + k]), straight from the lane.
// Phase A' (k_rf_pair): a wave per range puts its U slots, the tail of the range before and the head of the range
// behind into LDS and counts, for every listed point of its range, the listed points within the radius among the 16
// before and the 16 behind it in the list.  A lower bound of the true count: above nb_points the point
// is kept, otherwise it is queued (one returning atomic per wave that has any, on one of 64 counters).  The wave also
// leaves a second box per range, around its incoherent points.
// Phase B (k_rf_resolve): one wave per queued point counts from scratch.  A short segment (the LiDAR lists) is
// simply read whole.  Otherwise: the incoherent points of the segment's ranges (the U slots of the ranges whose
// second box comes within the radius, nearest ranges first) plus the coherent points of the ranges whose first box
// comes within the radius (such a range is re-read and the coherence of its points re-derived with phase A's own
// function, so every point of the segment is counted exactly once: through U slots or through its range).  The
// first two such ranges are read by the wave itself, further ones become work items of k_rf_ranges.
// The lists are NOT compacted here in the engine's path: the flags go to dfu3d_ballquery_fuse_joint, whose compaction
// launch serves both lists and both filters.
// History, all measured on the bench pool (11.5 M points; tools/rf_timing.py, tools/rf_variants.py):
//  * round 2: phase B walked candidate ranges one after the other: 0.23 ms for 0.14 % of the points (a box that a
//    pair of neighbouring outliers has blown up is a candidate for every query of its segment: one query had 71);
//  * a workgroup-wide list, per-segment U lists (one returning atomic per workgroup and segment) and a pairwise stage
//    behind a barrier made a phase-A workgroup live 36 us (24 of them in a one-partner-per-step loop), 17 us once the
//    loop was spread over the waves: 186 / 86 us per pass with 2048 resident workgroups of 32 KB each;
//  * waves on their own with the pairwise stage and the queue atomic at their end: 65 us -- the stream alone takes
//    35 us, so the lists moved to a pass of their own that touches 2 % of the points;
//  * three quarters of the workgroups of a grid over the pool's CAPACITY started only to find nothing (14 us of
//    machine time): the grid is capped and a workgroup walks its tiles;
//  * wave_shr:1 / wave_shl:1 (neighbours across the rows of 16 lanes) made the chunk loop 40 % longer.
constexpr int RFB = 256;           // threads per phase-A workgroup
constexpr int RF_IT = 8;           // 64-point chunks per wave
constexpr int RF_WG = RFB * RF_IT; // pool positions per workgroup
constexpr int RF_STRIDE = 8;       // nb_points > 1: lanes 0, 8, 16, ... of the chunk are broadcast
constexpr int RF_OCC = 8;          // phase-A workgroups per compute unit the register allocation aims at
constexpr int RF_GRID = 8192;      // phase-A workgroups at most (2048 fit the chip at once; 2048 / 4096 measured no better)
constexpr int RF_WLIST = 128;      // U slots of a range
constexpr int RF_WIN = 16;         // the pairing looks at this many listed points before / behind a listed point
constexpr int RF_PLDS = RF_WLIST + 2 * RF_WIN;
constexpr uint32_t RF_NOSEG = 0xFFFFu;
constexpr int BOX_FLOATS = 16;     // per range: coherent min xyz, #listed | coherent max xyz, - | incoherent min xyz, - | incoherent max xyz, -
constexpr int BOX_SHIFT = 9;       // a range = the 64 * RF_IT = 512 positions one wave of phase A walks
constexpr float BOX_EMPTY = 3.0e38f;
constexpr int RF_SC = 4;           // chunks of a range / a short segment phase B has in flight
constexpr int RF_DIRECT = 1024;    // phase B reads segments up to this size whole
constexpr int RF_NEAR = 2;         // phase B looks at the U slots of its own range +- RF_NEAR first
constexpr int RF_QSHARDS = 64;     // the queue of undecided points has 64 parts, each with its counter on a line of its own
constexpr int RF_QHDR = RF_QSHARDS * 16;   // (one counter took every wave's atomic: ~90 per microsecond is all one word does)
constexpr int RF_LQ_CAP = 1 << 16;         // "long" queries: more candidate ranges than their wave scans itself
constexpr int RF_ITEM_CAP = 1 << 20;       // their (query, range) work items
constexpr int RF_INLINE_CAND = 2;
constexpr int RF_WORK_HDR = 32;            // ints: [0] long queries, [16] work items
constexpr uint32_t RF_INCOH = 0x80000000u; // upos: position | RF_INCOH

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

// v_min_f32 / v_max_f32 without the canonicalisation of both operands that fminf / fmaxf add (no NaN can occur here)
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// wave-wide min / max with DPP row operations (VALU, no LDS traffic); the result is valid in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min63(float v) {
  v = min_raw(v, dpp_f<0xB1, 0xF>(v, v));      // quad_perm [1,0,3,2]
  v = min_raw(v, dpp_f<0x4E, 0xF>(v, v));      // quad_perm [2,3,0,1]
  v = min_raw(v, dpp_f<0x141, 0xF>(v, v));     // row_half_mirror
  v = min_raw(v, dpp_f<0x140, 0xF>(v, v));     // row_mirror: every lane of a row holds the row's minimum
  v = min_raw(v, dpp_f<0x142, 0xA>(v, v));     // row_bcast15 into rows 1 and 3
  v = min_raw(v, dpp_f<0x143, 0xC>(v, v));     // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_max63(float v) {
  v = max_raw(v, dpp_f<0xB1, 0xF>(v, v));
  v = max_raw(v, dpp_f<0x4E, 0xF>(v, v));
  v = max_raw(v, dpp_f<0x141, 0xF>(v, v));
  v = max_raw(v, dpp_f<0x140, 0xF>(v, v));
  v = max_raw(v, dpp_f<0x142, 0xA>(v, v));
  v = max_raw(v, dpp_f<0x143, 0xC>(v, v));
  return v;
}

// value of the lane before / behind (DPP row_shr:1 / row_shl:1: inside the rows of 16 lanes; the lane without such
// a neighbour reads 0).  All 64 lanes must be active where this is called.
// (measured alternatives: quad permutes lane ^ 1 / lane ^ 2 -- same speed, 2.9 % of the bench pool listed instead of 2.5 %;
// wave_shr:1 / wave_shl:1 across the rows -- the chunk loop 40 % longer)
constexpr int RF_NB1 = 0x111, RF_NB2 = 0x101, RF_ND1 = -1, RF_ND2 = 1;    // row_shr:1, row_shl:1
// the lane whose point lane `lane` tested as its neighbour number d (-1 where the permute reads nothing)
__device__ __forceinline__ int rf_nbr_lane(int lane, int d) {
  const int t = lane + d;
  return ((t >> 4) == (lane >> 4)) ? t : -1;
}
template <int CTRL>
__device__ __forceinline__ float nbr_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t nbr_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// The list-neighbour test, shared by phase A and by phase B's re-derivation (the two must agree bit for bit:
// same expression, no contraction).  wb: the shadow word (segment | radius) -- equal words <=> same segment; a word
// of 0 (what a lane reads for a missing neighbour) is segment 0 WITHOUT a radius, i.e. never the word of an active
// point.  (First version: lane ^ 1 and lane ^ 2 -- 2.9 % of the bench pool incoherent against 2.2 %: one point of
// three at the end of a mask row had both partners in the next row.)
__device__ __forceinline__ void rf_list_neighbours(float x, float y, float z, uint32_t wb, float thr2, bool &h1, bool &h2) {
  const float x1 = nbr_f<RF_NB1>(x), y1 = nbr_f<RF_NB1>(y), z1 = nbr_f<RF_NB1>(z);
  const float x2 = nbr_f<RF_NB2>(x), y2 = nbr_f<RF_NB2>(y), z2 = nbr_f<RF_NB2>(z);
  const uint32_t w1 = nbr_u<RF_NB1>(wb), w2 = nbr_u<RF_NB2>(wb);
  float dx = x - x1, dy = y - y1, dz = z - z1;
  h1 = (w1 == wb) & (dx * dx + dy * dy + dz * dz < thr2);
  dx = x - x2; dy = y - y2; dz = z - z2;
  h2 = (w2 == wb) & (dx * dx + dy * dy + dz * dz < thr2);
}

struct RfScratch {
  uint8_t *flags;
  int *queue;               // [16 q]: length of part q; part q = queue + RF_QHDR + q * qcap
  long long qcap;
  float *rrec;              // BOX_FLOATS per range
  float4 *uent;             // (x, y, z, shadow word) of listed point k of range rg at rg * RF_WLIST + k, k < #listed (rrec[.][3])
  uint32_t *upos;           // its pool position | RF_INCOH
  float4 *ulist;            // (x, y, z, position) of incoherent points that found their range's slots full: per segment, at seg_base[s] + k, k < ucount[s]
  int *work;                // RF_WORK_HDR counters | ucount (65536) | long-query records (4 ints) | work items (2 ints)
  int *ucount;
  const long long *seg_base;
};
__host__ __device__ inline long long rf_queue_part_cap(long long pool_cap) {
  const long long nwg = (pool_cap + RF_WG - 1) / RF_WG;
  return ((nwg + RF_QSHARDS - 1) / RF_QSHARDS) * RF_WG;
}
// append to the part of the queue that belongs to `key` (any number; consecutive keys use different counters)
__device__ __forceinline__ void rf_queue_push(const RfScratch &W, long long key, unsigned long long mask, bool mine, uint32_t gpos) {
  const int lane = lane_id();
  const int qs = (int)(key % RF_QSHARDS);
  int g = 0;
  if (lane == 0) g = atomicAdd(&W.queue[16 * qs], __popcll(mask));
  g = __builtin_amdgcn_readfirstlane(g);
  if (mine) W.queue[RF_QHDR + qs * W.qcap + g + __popcll(mask & ((1ull << lane) - 1ull))] = (int)gpos;
}


template <bool NB1>
__global__ __launch_bounds__(RFB, NB1 ? RF_OCC : RF_OCC - 1) void k_rf_stream(
    const float4 *__restrict__ pq, const long long *__restrict__ n_used_ptr, long long n_max, int nb, int S, RfScratch W) {
  long long n_used = n_max;
  if (n_used_ptr) { const long long u = *n_used_ptr; n_used = u < n_max ? u : n_max; }
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: the wave's addresses are SGPR base + lane)
  // the grid is capped (a pool is sized for the worst case); a workgroup walks its tiles -- from the END of the pool
  // (DFU3D_RF_REVERSE, default): k_seg_write has just written the shadow front to back, so the memory-side cache
  // holds its tail; a walk from the front misses, and what it brings in pushes out exactly the lines it would have
  // hit next (44 -> 52 us when the writer became a sequential pass); from the end the most recent lines are read first
  const long long n_tiles = (n_used + RF_WG - 1) / RF_WG;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long wg0 = (n_tiles - 1 - tile) * RF_WG;
    const long long w0 = wg0 + (long long)wave * (64 * RF_IT);
    if (w0 >= n_used) continue;
    // (positions at or beyond n_used are read all the same -- the scratch allocation extends megabytes beyond the
    // shadow proper -- and marked "no segment" afterwards: eight loads, no branch between them)
    float4 p[RF_IT];
#pragma unroll
    for (int it = 0; it < RF_IT; it++) p[it] = pq[w0 + it * 64 + lane];
    const int lim = (int)min(n_used - w0, (long long)(64 * RF_IT));     // (scalar)
    const long long rg = w0 >> BOX_SHIFT;
    float4 *my_ent = W.uent + rg * RF_WLIST;
    uint32_t *my_pos = W.upos + rg * RF_WLIST;
    float lx = BOX_EMPTY, ly = BOX_EMPTY, lz = BOX_EMPTY, hx = -BOX_EMPTY, hy = -BOX_EMPTY, hz = -BOX_EMPTY;
    int run = 0;                                    // listed points of this wave so far (wave-uniform)
#pragma unroll
    for (int it = 0; it < RF_IT; it++) {
      const long long i = w0 + it * 64 + lane;
      const float x = p[it].x, y = p[it].y, z = p[it].z;
      const uint32_t wb = (it * 64 + lane < lim) ? __float_as_uint(p[it].w) : (RF_NOSEG << 16);
      const float r = __uint_as_float(wb << 16);
      const bool valid = (wb >> 16) < (uint32_t)S;    // "no segment" mark, or a slot nobody wrote: never a point
      const bool active = valid && (r > 0.0f);
      const float thr2 = rf_certain_hit2(x, y, z, r);
      bool h1, h2;
      rf_list_neighbours(x, y, z, wb, thr2, h1, h2);
      const bool coh = active && (h1 || h2);
      bool listed;
      if (NB1) {
        listed = active && !coh;
      } else {
        int cnt = 1 + (h1 ? 1 : 0) + (h2 ? 1 : 0);   // the point itself (d = 0 < r^2) and its list neighbours
        if (__ballot(active && cnt <= nb)) {
          const int l1 = rf_nbr_lane(lane, RF_ND1), l2 = rf_nbr_lane(lane, RF_ND2);
          for (int j = 0; j < 64; j += RF_STRIDE) {
            const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
            const float yj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), j));
            const float zj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), j));
            const uint32_t wj = (uint32_t)__builtin_amdgcn_readlane((int)wb, j);
            const float dx = x - xj, dy = y - yj, dz = z - zj;
            if (wj == wb && dx * dx + dy * dy + dz * dz < thr2 && j != lane && j != l1 && j != l2) cnt++;
            if (__ballot(active && cnt <= nb) == 0ull) break;
          }
        }
        listed = active && (cnt <= nb || !coh);
      }
      // decided here: inactive segments (r == 0: no filter; r < 0 / NaN: drop all) and the coherent points that
      // are not listed; a listed point gets its flag in phase A' (or B)
      if (valid && !listed) W.flags[i] = active ? 1 : ((r == 0.0f) ? 1 : 0);
      lx = min_raw(lx, coh ? x : BOX_EMPTY); ly = min_raw(ly, coh ? y : BOX_EMPTY); lz = min_raw(lz, coh ? z : BOX_EMPTY);
      hx = max_raw(hx, coh ? x : -BOX_EMPTY); hy = max_raw(hy, coh ? y : -BOX_EMPTY); hz = max_raw(hz, coh ? z : -BOX_EMPTY);
      const unsigned long long m = __ballot(listed);
      if (m) {
        if (listed) {
          const int slot = run + __popcll(m & ((1ull << lane) - 1ull));
          if (slot < RF_WLIST) {
            my_ent[slot] = make_float4(x, y, z, __uint_as_float(wb));
            my_pos[slot] = (uint32_t)i | (coh ? 0u : RF_INCOH);
          } else {                     // slots full (a pathological range): the segment's overflow list and the queue
            const uint32_t seg = wb >> 16;
            if (!coh) {
              const int k = atomicAdd(&W.ucount[seg], 1);
              W.ulist[W.seg_base[seg] + k] = make_float4(x, y, z, __uint_as_float((uint32_t)i));
            }
            const int qs = (int)(rg % RF_QSHARDS);
            const int g = atomicAdd(&W.queue[16 * qs], 1);
            W.queue[RF_QHDR + qs * W.qcap + g] = (int)i;
            W.flags[i] = 0;
          }
        }
        run += __popcll(m);
      }
    }
    lx = wave_min63(lx); ly = wave_min63(ly); lz = wave_min63(lz);
    hx = wave_max63(hx); hy = wave_max63(hy); hz = wave_max63(hz);
    if (lane == 63) {
      float4 *o = (float4 *)(W.rrec + (size_t)rg * BOX_FLOATS);
      o[0] = make_float4(lx, ly, lz, __int_as_float(min(run, RF_WLIST)));
      o[1] = make_float4(hx, hy, hz, 0.0f);
    }
  }
}

// Phase A': one wave per range.  The listed points of the range, the last RF_WIN of the range before and the first
// RF_WIN of the range behind stand in LDS in list order; listed point c is tested against the RF_WIN listed points
// before and the RF_WIN behind it (itself included: d = 0), a lane per point, every lane its own partner -- 2 RF_WIN + 1
// steps whatever the number of listed points.  Measured before this: a wave per FOUR ranges with all pairs in LDS
// (ranges of sparse or noisy lists have a hundred listed points: single waves ran for 60 us, the pass took 72 us);
// a wave per range, all pairs, partners broadcast with v_readlane (64 us, 43 us with the partners capped).
__global__ __launch_bounds__(256) void k_rf_pair(const long long *__restrict__ n_used_ptr, long long n_max, int nb, RfScratch W) {
  __shared__ float4 s_e[4][RF_PLDS];
  long long n_used = n_max;
  if (n_used_ptr) { const long long u = *n_used_ptr; n_used = u < n_max ? u : n_max; }
  const long long n_rg = (n_used + (1 << BOX_SHIFT) - 1) >> BOX_SHIFT;
  const int lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4 *my_e = s_e[wv];
  for (long long rg = (long long)blockIdx.x * 4 + wv; rg < n_rg; rg += (long long)gridDim.x * 4) {
    // lane l < 3: the number of listed points of range rg - 1 + l (0 outside the used ranges)
    int nl = 0;
    {
      const long long rgl = rg - 1 + lane;
      if (lane < 3 && rgl >= 0 && rgl < n_rg) nl = min(max(__float_as_int(W.rrec[(size_t)rgl * BOX_FLOATS + 3]), 0), RF_WLIST);
    }
    const int n_own = __builtin_amdgcn_readlane(nl, 1);
    float4 *rec = (float4 *)(W.rrec + (size_t)rg * BOX_FLOATS);
    if (n_own == 0) {                                // nothing listed here: only the (empty) second box
      if (lane == 0) {
        rec[2] = make_float4(BOX_EMPTY, BOX_EMPTY, BOX_EMPTY, 0.0f);
        rec[3] = make_float4(-BOX_EMPTY, -BOX_EMPTY, -BOX_EMPTY, 0.0f);
      }
      continue;
    }
    const int n_prev = __builtin_amdgcn_readlane(nl, 0), n_next = __builtin_amdgcn_readlane(nl, 2);
    const int t_prev = min(n_prev, RF_WIN), t_next = min(n_next, RF_WIN);     // what of the neighbours is taken
    // all loads first: the range (two halves), its positions, the tail before, the head behind
    const bool v0 = lane < n_own, v1 = lane + 64 < n_own;
    // (every slot of the range's U list is readable memory: a lane beyond the count loads all the same and drops what it
    // got -- a shared `none` value for the conditional loads went through scratch memory)
    float4 q0 = W.uent[rg * RF_WLIST + lane];
    uint32_t p0 = W.upos[rg * RF_WLIST + lane];
    if (!v0) { q0.x = 0.f; q0.y = 0.f; q0.z = 0.f; q0.w = 0.f; p0 = 0u; }
    float4 q1, eh;
    q1.x = q1.y = q1.z = q1.w = 0.f;
    eh.x = eh.y = eh.z = eh.w = 0.f;
    uint32_t p1 = 0u;
    if (n_own > 64) {                                // uniform
      if (v1) { q1 = W.uent[rg * RF_WLIST + 64 + lane]; p1 = W.upos[rg * RF_WLIST + 64 + lane]; }
    }
    // lanes 0 .. t_prev-1: the tail of the range before; lanes 32 .. 32+t_next-1: the head of the range behind
    if (lane < t_prev) eh = W.uent[(rg - 1) * RF_WLIST + n_prev - t_prev + lane];
    else if (lane >= 32 && lane - 32 < t_next) eh = W.uent[(rg + 1) * RF_WLIST + lane - 32];
    // LDS, list order: [RF_WIN - t_prev, RF_WIN) tail | [RF_WIN, RF_WIN + n_own) own | [.., + t_next) head
    __builtin_amdgcn_wave_barrier();
    if (v0) my_e[RF_WIN + lane] = q0;
    if (v1) my_e[RF_WIN + 64 + lane] = q1;
    if (lane < t_prev) my_e[RF_WIN - t_prev + lane] = eh;
    else if (lane >= 32 && lane - 32 < t_next) my_e[RF_WIN + n_own + lane - 32] = eh;
    {                                                // the box around the range's incoherent points
      const bool i0 = (p0 & RF_INCOH) != 0u, i1 = (p1 & RF_INCOH) != 0u;
      float ux = min_raw(i0 ? q0.x : BOX_EMPTY, i1 ? q1.x : BOX_EMPTY), uy = min_raw(i0 ? q0.y : BOX_EMPTY, i1 ? q1.y : BOX_EMPTY),
            uz = min_raw(i0 ? q0.z : BOX_EMPTY, i1 ? q1.z : BOX_EMPTY);
      float vx = max_raw(i0 ? q0.x : -BOX_EMPTY, i1 ? q1.x : -BOX_EMPTY), vy = max_raw(i0 ? q0.y : -BOX_EMPTY, i1 ? q1.y : -BOX_EMPTY),
            vz = max_raw(i0 ? q0.z : -BOX_EMPTY, i1 ? q1.z : -BOX_EMPTY);
      ux = wave_min63(ux); uy = wave_min63(uy); uz = wave_min63(uz);
      vx = wave_max63(vx); vy = wave_max63(vy); vz = wave_max63(vz);
      if (lane == 63) {
        rec[2] = make_float4(ux, uy, uz, 0.0f);
        rec[3] = make_float4(vx, vy, vz, 0.0f);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int lo = RF_WIN - t_prev, hi = RF_WIN + n_own + t_next;      // the filled part of the LDS list
    // one half (64 listed points) at a time; written as a function of the half's own registers (a loop over h that picks
    // q0 / q1 with h put both into scratch memory: 32 bytes per lane)
    auto half = [&](const bool v, const float4 q, const uint32_t p, const int ci) {
      const uint32_t wa = __float_as_uint(q.w);
      const float thr2 = rf_certain_hit2(q.x, q.y, q.z, __uint_as_float(wa << 16));
      int cnt = 0;
#pragma unroll 1
      for (int d0 = -RF_WIN; d0 <= RF_WIN; d0 += 4) {
        float4 o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = my_e[min(max(ci + d0 + u, 0), RF_PLDS - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = ci + d0 + u;
          const float dx = q.x - o[u].x, dy = q.y - o[u].y, dz = q.z - o[u].z;
          cnt += ((d0 + u <= RF_WIN) & (k >= lo) & (k < hi) & (__float_as_uint(o[u].w) == wa) & (dx * dx + dy * dy + dz * dz < thr2)) ? 1 : 0;
        }
        if (__ballot(v && cnt <= nb) == 0ull) break;
      }
      const bool pend = v && cnt <= nb;
      const uint32_t gpos = p & ~RF_INCOH;
      if (v) W.flags[gpos] = pend ? 0 : 1;
      const unsigned long long mp = __ballot(pend);
      if (mp) rf_queue_push(W, rg, mp, pend, gpos);
    };
    half(v0, q0, p0, RF_WIN + lane);                 // the point's own place in the LDS list
    if (n_own > 64) half(v1, q1, p1, RF_WIN + 64 + lane);      // (uniform)
  }
}

// squared distance from q to an axis-aligned box (0 inside)
__device__ __forceinline__ float box_dist2(float qx, float qy, float qz, float lx, float ly, float lz,
                                           float hx, float hy, float hz) {
  const float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.0f), dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.0f),
              dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.0f);
  return dx * dx + dy * dy + dz * dz;
}

// one queued point with everything its tests need
struct RfQuery {
  long long i;
  float4 qf;
  uint32_t wq;
  long long base, end;
  double x, y, z, r2;
  float lo2, hi2;       // float32 screening: certainly inside below lo2, certainly outside above hi2, fp64 in between
};
__device__ __forceinline__ bool rf_load_query(RfQuery &Q, long long i, long long n_max, const float4 *pq, const double *px,
                                              const double *py, const double *pz, const long long *seg_base,
                                              const int *seg_cnt, const double *radius, int S) {
  if (i < 0 || i >= n_max) return false;
  Q.i = i;
  Q.qf = pq[i];
  Q.wq = __float_as_uint(Q.qf.w);
  const int s = (int)(Q.wq >> 16);                       // < S: nothing else is ever queued
  if (s >= S) return false;
  const int n = max(seg_cnt[s], 0);
  Q.base = seg_base[s];
  Q.end = Q.base + n;
  if (i < Q.base || i >= Q.end) return false;            // cannot happen for a shadow built from this table
  const double r = radius[s];
  Q.r2 = r * r;
  Q.x = px[i]; Q.y = py[i]; Q.z = pz[i];
  const float rf = (float)r;
  const float eb = rf_bound(Q.qf.x, Q.qf.y, Q.qf.z, rf) + rf * 2.4e-7f;
  const float tl = rf - eb, th = rf + eb;
  Q.lo2 = (tl > 0.0f) ? tl * tl * 0.999999f : -1.0f;
  Q.hi2 = th * th * 1.000001f;
  // every lane holds the same query: keep it in scalar registers (twenty vector registers otherwise)
  auto sf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  auto sd = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  auto sl = [](long long v) {
    return (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                       (unsigned int)__builtin_amdgcn_readfirstlane((int)(v & 0xFFFFFFFFll)));
  };
  Q.i = sl(Q.i); Q.base = sl(Q.base); Q.end = sl(Q.end);
  Q.qf = make_float4(sf(Q.qf.x), sf(Q.qf.y), sf(Q.qf.z), sf(Q.qf.w));
  Q.wq = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q.wq);
  Q.x = sd(Q.x); Q.y = sd(Q.y); Q.z = sd(Q.z); Q.r2 = sd(Q.r2);
  Q.lo2 = sf(Q.lo2); Q.hi2 = sf(Q.hi2);
  return true;
}
// is the point with float32 coordinates o at pool position g within the radius of the query?
__device__ __forceinline__ bool rf_within(const RfQuery &Q, const float4 &o, long long g, const double *px, const double *py,
                                          const double *pz) {
  const float dx = Q.qf.x - o.x, dy = Q.qf.y - o.y, dz = Q.qf.z - o.z;
  const float d2 = dx * dx + dy * dy + dz * dz;
  bool hit = d2 < Q.lo2;
  if (!hit && !(d2 > Q.hi2)) {                     // too close to call in float32
    const double ex = Q.x - px[g], ey = Q.y - py[g], ez = Q.z - pz[g];
    double d = ex * ex;
    d += ey * ey;
    d += ez * ez;
    hit = d < Q.r2;
  }
  return hit;
}
// the COHERENT points of range rg (512 positions) within the radius of the query: the range is read as phase A's wave
// read it and the coherence of its points re-derived with phase A's own function.  Whole wave, uniform.
__device__ __forceinline__ int rf_scan_range(const RfQuery &Q, long long rg, const float4 *pq, long long n_used,
                                             const double *px, const double *py, const double *pz) {
  const int lane = lane_id();
  int cnt = 0;
#pragma unroll 1
  for (int half = 0; half < RF_IT / RF_SC; half++) {
    const long long c0 = (rg << (BOX_SHIFT - 6)) + half * RF_SC;
    float4 o[RF_SC];
#pragma unroll
    for (int u = 0; u < RF_SC; u++) {
      const long long g = ((c0 + u) << 6) + lane;
      o[u] = pq[g];                                   // (as in phase A: read, then marked if beyond n_used)
      if (g >= n_used) o[u].w = __uint_as_float(RF_NOSEG << 16);
    }
#pragma unroll
    for (int u = 0; u < RF_SC; u++) {
      const long long g = ((c0 + u) << 6) + lane;
      const uint32_t wb = __float_as_uint(o[u].w);
      const float thr2 = rf_certain_hit2(o[u].x, o[u].y, o[u].z, __uint_as_float(wb << 16));
      bool h1, h2;
      rf_list_neighbours(o[u].x, o[u].y, o[u].z, wb, thr2, h1, h2);
      const bool hit = (wb == Q.wq) && (h1 || h2) && rf_within(Q, o[u], g, px, py, pz);
      cnt += __popcll(__ballot(hit));
    }
  }
  return cnt;
}
// the INCOHERENT listed points of up to RF_UL ranges (taken from the ballot mask mu; lane k holds range r0 + k and its
// number of listed points nl) within the radius of the query.  Whole wave, uniform.
constexpr int RF_UL = 4;           // ranges / overflow-list loads in flight per lane
__device__ __forceinline__ int rf_count_slots(const RfQuery &Q, const RfScratch &W, unsigned long long &mu, long long r0, int nl,
                                              const double *px, const double *py, const double *pz) {
  const int lane = lane_id();
  float4 o[RF_UL];
  uint32_t op[RF_UL];
  int un[RF_UL];
  long long ur[RF_UL];
  bool big = false;                                  // (uniform) one of the ranges has more than 64 listed points
#pragma unroll
  for (int u = 0; u < RF_UL; u++) {
    un[u] = 0;
    ur[u] = 0;
    if (mu) {
      const int k = __ffsll((long long)mu) - 1;
      mu &= mu - 1ull;
      un[u] = __builtin_amdgcn_readlane(nl, k);
      ur[u] = r0 + k;
    }
    big |= un[u] > 64;
    const bool a = lane < un[u];
    o[u] = a ? W.uent[ur[u] * RF_WLIST + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    op[u] = a ? W.upos[ur[u] * RF_WLIST + lane] : 0u;
  }
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < RF_UL; u++) {
    const bool hit = (op[u] & RF_INCOH) && __float_as_uint(o[u].w) == Q.wq && rf_within(Q, o[u], (long long)(op[u] & ~RF_INCOH), px, py, pz);
    cnt += __popcll(__ballot(hit));
  }
  if (big) {                                         // the second halves (rare: sparse lists)
#pragma unroll
    for (int u = 0; u < RF_UL; u++) {
      const bool b = lane + 64 < un[u];
      o[u] = b ? W.uent[ur[u] * RF_WLIST + 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      op[u] = b ? W.upos[ur[u] * RF_WLIST + 64 + lane] : 0u;
    }
#pragma unroll
    for (int u = 0; u < RF_UL; u++) {
      const bool hit = (op[u] & RF_INCOH) && __float_as_uint(o[u].w) == Q.wq && rf_within(Q, o[u], (long long)(op[u] & ~RF_INCOH), px, py, pz);
      cnt += __popcll(__ballot(hit));
    }
  }
  return cnt;
}

// Phase B: one wave per queued point, counting from scratch.
// (register budgets for 6 / 8 waves per SIMD spilled and were slower: 55 / 69 us against 37)
__global__ __launch_bounds__(256, 5) void k_rf_resolve(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const float4 *__restrict__ pq, const long long *__restrict__ n_used_ptr, long long n_max,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, const double *__restrict__ radius,
    int nb, int S, RfScratch W) {
  long long n_used = n_max;
  if (n_used_ptr) { const long long u = *n_used_ptr; n_used = u < n_max ? u : n_max; }
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  const int lane = lane_id();
  // the parts of the queue: lane q holds the number of entries in the parts up to and including q
  const int q_incl = wave_incl_scan((int)min((long long)max(W.queue[16 * lane], 0), W.qcap));
  const int nq = __builtin_amdgcn_readlane(q_incl, 63);
  int *lq_tab = W.work + RF_WORK_HDR + 65536, *items = lq_tab + 4 * RF_LQ_CAP;
  for (int e = wave; e < nq; e += nwaves) {
    const int part = __popcll(__ballot(q_incl <= e));
    const int before = part ? __builtin_amdgcn_readlane(q_incl, part - 1) : 0;
    RfQuery Q;
    if (!rf_load_query(Q, W.queue[RF_QHDR + part * W.qcap + (e - before)], n_max, pq, px, py, pz, seg_base, seg_cnt, radius, S)) continue;
    const int n = (int)(Q.end - Q.base);
    int cnt = 0, n_items = 0, lq = -1;
    if (n <= RF_DIRECT) {
      // a short segment (the LiDAR lists): every point of it, straight from the shadow
      const long long c_lo = Q.base >> 6, c_hi = (Q.end - 1) >> 6;
      for (long long c0 = c_lo; c0 <= c_hi && cnt <= nb; c0 += RF_SC) {
        float4 o[RF_SC];
#pragma unroll
        for (int u = 0; u < RF_SC; u++) o[u] = pq[((c0 + u) << 6) + lane];      // (reads beyond the segment stay inside the scratch)
#pragma unroll
        for (int u = 0; u < RF_SC; u++) {
          const long long g = ((c0 + u) << 6) + lane;
          const bool hit = (g >= Q.base) && (g < Q.end) && rf_within(Q, o[u], g, px, py, pz);
          cnt += __popcll(__ballot(hit));
        }
      }
    } else {
      const long long r_lo = Q.base >> BOX_SHIFT, r_hi = (Q.end - 1) >> BOX_SHIFT, r_q = Q.i >> BOX_SHIFT;
      // (1) the listed points next door first: most queued points have their neighbours a few hundred positions away
      const long long n_lo = max(r_lo, r_q - RF_NEAR), n_hi = min(r_hi, r_q + RF_NEAR);
      {
        const long long rg = n_lo + lane;
        int nl = 0;
        if (rg <= n_hi) nl = min(max(__float_as_int(W.rrec[(size_t)rg * BOX_FLOATS + 3]), 0), RF_WLIST);
        unsigned long long mu = __ballot(nl > 0);
        while (mu && cnt <= nb) cnt += rf_count_slots(Q, W, mu, n_lo, nl, px, py, pz);
      }
      int n_inline = 0;
      for (long long r0 = r_lo; r0 <= r_hi && cnt <= nb; r0 += 64) {
        const long long rg = r0 + lane;
        bool cand = false, ucand = false;
        int nl = 0;
        if (rg <= r_hi) {
          const float4 *bp = (const float4 *)(W.rrec + (size_t)rg * BOX_FLOATS);
          const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
          cand = box_dist2(Q.qf.x, Q.qf.y, Q.qf.z, b0.x, b0.y, b0.z, b1.x, b1.y, b1.z) <= Q.hi2;
          nl = min(max(__float_as_int(b0.w), 0), RF_WLIST);
          ucand = nl > 0 && (rg < n_lo || rg > n_hi) &&
                  box_dist2(Q.qf.x, Q.qf.y, Q.qf.z, b2.x, b2.y, b2.z, b3.x, b3.y, b3.z) <= Q.hi2;
        }
        // (2) the incoherent points of the other ranges whose second box comes within the radius
        unsigned long long mu = __ballot(ucand);
        while (mu && cnt <= nb) cnt += rf_count_slots(Q, W, mu, r0, nl, px, py, pz);
        // (3) the coherent points of the ranges whose first box comes within the radius: the first RF_INLINE_CAND of
        // them are read here, the others become work items of k_rf_ranges
        unsigned long long m = __ballot(cand);
        while (m && cnt <= nb && n_inline < RF_INLINE_CAND) {            // uniform
          const int k = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          n_inline++;
          cnt += rf_scan_range(Q, r0 + k, pq, n_used, px, py, pz);
        }
        if (m && cnt <= nb) {                          // the rest of this batch: work items
          const int c = __popcll(m);
          int ib = 0;
          if (lane == 0) {
            if (lq < 0) lq = atomicAdd(&W.work[0], 1);
            ib = (lq < RF_LQ_CAP) ? atomicAdd(&W.work[16], c) : RF_ITEM_CAP;
          }
          lq = __builtin_amdgcn_readfirstlane(lq);
          ib = __builtin_amdgcn_readfirstlane(ib);
          if (lq < RF_LQ_CAP && ib + c <= RF_ITEM_CAP) {
            if ((m >> lane) & 1ull) {
              const int t = ib + __popcll(m & ((1ull << lane) - 1ull));
              items[2 * t] = lq;
              items[2 * t + 1] = (int)rg;
            }
            n_items += c;
          } else {                                     // tables full (never seen): read the ranges here after all
            // the slots this wave was given and does not fill must not be read as items: k_rf_ranges takes
            // min(work[16], RF_ITEM_CAP) slots, and what an earlier pass left in them would be counted for a live query
            if (lq < RF_LQ_CAP && ((m >> lane) & 1ull)) {
              const int t = ib + __popcll(m & ((1ull << lane) - 1ull));
              if (t < RF_ITEM_CAP) items[2 * t] = -1;
            }
            while (m && cnt <= nb) {
              const int k = __ffsll((long long)m) - 1;
              m &= m - 1ull;
              cnt += rf_scan_range(Q, r0 + k, pq, n_used, px, py, pz);
            }
          }
        }
      }
      // (4) incoherent points that found their range's slots full (pathological ranges): the segment's overflow list
      const int nov = (int)min((long long)max(W.ucount[Q.wq >> 16], 0), Q.end - Q.base);
      for (int k0 = 0; k0 < nov && cnt <= nb; k0 += 64 * RF_UL) {
        float4 o[RF_UL];
        bool in[RF_UL];
#pragma unroll
        for (int u = 0; u < RF_UL; u++) {
          const int k = k0 + u * 64 + lane;
          in[u] = k < nov;
          o[u] = in[u] ? W.ulist[Q.base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < RF_UL; u++) {
          const bool hit = in[u] && rf_within(Q, o[u], (long long)__float_as_uint(o[u].w), px, py, pz);
          cnt += __popcll(__ballot(hit));
        }
      }
    }
    if (lane == 0) {
      if (n_items && cnt <= nb) {                    // k_rf_ranges finishes the count
        lq_tab[4 * lq] = (int)Q.i;
        lq_tab[4 * lq + 1] = n_items;
        lq_tab[4 * lq + 2] = cnt;
        lq_tab[4 * lq + 3] = 0;
      } else {
        if (lq >= 0 && lq < RF_LQ_CAP) lq_tab[4 * lq + 1] = 0;   // (items of a query that got its answer meanwhile are skipped)
        W.flags[Q.i] = (cnt > nb) ? 1 : 0;
      }
    }
  }
}

// Phase B, the long queries: one wave per (query, candidate range); the hits meet in the query's record (count in the
// low word, finished items in the high word of ONE 64-bit atomic), whoever finishes the last item writes the flag.
__global__ __launch_bounds__(256) void k_rf_ranges(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
    const float4 *__restrict__ pq, const long long *__restrict__ n_used_ptr, long long n_max,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, const double *__restrict__ radius,
    int nb, int S, RfScratch W) {
  const int n_items = min(W.work[16], RF_ITEM_CAP);
  if (n_items <= 0) return;
  long long n_used = n_max;
  if (n_used_ptr) { const long long u = *n_used_ptr; n_used = u < n_max ? u : n_max; }
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  int *lq_tab = W.work + RF_WORK_HDR + 65536, *items = lq_tab + 4 * RF_LQ_CAP;
  for (int t = wave; t < n_items; t += nwaves) {
    const int lq = items[2 * t];
    const long long rg = items[2 * t + 1];
    if (lq < 0 || lq >= RF_LQ_CAP) continue;
    const int total = lq_tab[4 * lq + 1];
    if (total <= 0) continue;
    RfQuery Q;
    if (!rf_load_query(Q, lq_tab[4 * lq], n_max, pq, px, py, pz, seg_base, seg_cnt, radius, S)) continue;
    const int hits = rf_scan_range(Q, rg, pq, n_used, px, py, pz);
    if (lane_id() == 0) {
      const unsigned long long old = atomicAdd((unsigned long long *)(lq_tab + 4 * lq + 2), (unsigned long long)hits | (1ull << 32));
      if ((int)(old >> 32) + 1 == total) W.flags[Q.i] = ((int)(old & 0xFFFFFFFFull) + hits > nb) ? 1 : 0;
    }
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

// |a - b| in one instruction (the compiler expands __sad() into subtract, negate, max)
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t bh_hash(uint32_t ix, uint32_t iy, uint32_t iz) {
  // cell coordinates are below 2^12: 24-bit multiplies (three v_mul_lo_u32 per cell, eight cells per query, issue at a
  // quarter of the rate)
  uint32_t h = __umul24(ix, 0x9E3779u) ^ __umul24(iy, 0x85EBCBu) ^ __umul24(iz, 0xC2B2AFu);
  h ^= h >> 15;
  h ^= h >> 7;
  return h;
}

constexpr int BALL_TPW = 1;        // consecutive query tiles per workgroup (2 / 4 measured slower: the kernels' durations follow their longest workgroups)
template <int BT, int BH_MAX, int BH_HEADS>
__global__ __launch_bounds__(BT) void k_ball_flags(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ base_a,
    const int *__restrict__ cnt_a, const long long *__restrict__ base_b,
    const int *__restrict__ cnt_b, double T, double C, int S, const int *__restrict__ tile_off,
    uint8_t *__restrict__ flags, int masked) {
  __shared__ unsigned long long s_node[BH_MAX];
  __shared__ uint32_t s_head[BH_HEADS];
  __shared__ int s_pending, s_nkept;
  const int ntile = tile_off[S];
  int t = blockIdx.x * BALL_TPW;
  if (t >= ntile) return;
  const int t_end = min(t + BALL_TPW, ntile);
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
    const bool amask = masked == 2;      // the LiDAR list is not compacted either: flags[ba + i] says who is left of it
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
      if (threadIdx.x == 0) { s_pending = 0; s_nkept = 0; }
      __syncthreads();
      bool too_wide = false;
      int mine_kept = 0;
      for (int i = threadIdx.x; i < na; i += BT) {
        // (coordinates and flag are requested together: the table build is a dependent prologue of the workgroup)
        const double ax = px[ba + i], ay = py[ba + i], az = pz[ba + i];
        if (amask && !flags[ba + i]) continue;
        mine_kept++;
        const double fx = floor(ax * inv) + OFF, fy = floor(ay * inv) + OFF, fz = floor(az * inv) + OFF;
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
      if (mine_kept) s_nkept = 1;                   // (plain store, every writer the same value: atomics on one LDS word serialise lane by lane)
      __syncthreads();
      hash_ok = (s_pending == 0);
      hashed_s = s;
    }
    bool found = false;
    if (na <= BH_MAX && hash_ok) {
      if (s_nkept == 0) continue;        // the filter left no LiDAR point: fuse skipped (my_loader.py:602), the flags stand
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
          // one node of a chain (word `wv` of node `node`) against the query
          auto test_node = [&](uint32_t node, unsigned long long wv) {
            const uint32_t ax = (uint32_t)(wv & 0x1FFFFull), ay = (uint32_t)((wv >> 17) & 0x1FFFFull),
                           az = (uint32_t)((wv >> 34) & 0x1FFFFull);
            // each quantised difference is within 1 (+3e-11) unit of the true one, so the
            // true distance is at least |max(|d|-1, 0)| units, and C is 16/(1+1e-5) < 16 units:
            // 258 > 16.06^2 leaves room for the rounding of the quantisation itself.
            // |a - q| by v_sad_u32, the terms capped at 31 (one capped term alone is 961 > 257) so that the squares are
            // 24-bit multiplies: a 32-bit v_mul_lo_u32 issues at a quarter of the rate, and three of them were 40 % of a
            // node's 120 clocks -- the node test is what the query phase runs on (tools/ball_timing.py)
            const uint32_t sx_ = sad_u32(ax, (uint32_t)qx), sy_ = sad_u32(ay, (uint32_t)qy), sz_ = sad_u32(az, (uint32_t)qz);
            const uint32_t ex_ = min(sx_ ? sx_ - 1u : 0u, 31u), ey_ = min(sy_ ? sy_ - 1u : 0u, 31u),
                           ez_ = min(sz_ ? sz_ - 1u : 0u, 31u);
            if (!qin || __umul24(ex_, ex_) + __umul24(ey_, ey_) + __umul24(ez_, ez_) <= 257u) {
              const int j = (int)node - 1;
              const double ex = x - px[ba + j], ey = y - py[ba + j], ez = z - pz[ba + j];
              double d = ex * ex;
              d += ey * ey;
              d += ez * ez;
              if (d < T) found = true;                       // <=> sqrt(d) < C, see dfu3d_ballquery_fuse
            }
          };
          // walk a chain from `node`, whose word `wv` the caller has read already
          auto walk_from = [&](uint32_t node, unsigned long long wv) {
            while (true) {
              test_node(node, wv);
              if (found) return;
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
            // Four chains at a time.  A chain is a run of DEPENDENT LDS reads (the link sits in the node), a dense
            // instance has a hundred LiDAR points per cell, and a query with no neighbour walks all eight chains to
            // their ends: 200-300 reads in a row per lane were most of a workgroup's 22 us (tools/ball_timing.py) with
            // the vector pipes a third busy at eight waves per SIMD.  The four next-node reads of a step are independent.
#pragma unroll
            for (int g4 = 0; g4 < 8; g4 += 4) {
              if (found) break;
              uint32_t nd[4];
              unsigned long long wv[4];
#pragma unroll
              for (int c4 = 0; c4 < 4; c4++) { nd[c4] = heads[g4 + c4]; wv[c4] = nd[c4] ? s_node[nd[c4] - 1u] : 0ull; }
              while ((nd[0] | nd[1] | nd[2] | nd[3]) != 0u) {
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++)
                  if (nd[c4]) {
                    test_node(nd[c4], wv[c4]);
                    nd[c4] = (uint32_t)(wv[c4] >> 51);
                  }
                if (found) break;
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) wv[c4] = nd[c4] ? s_node[nd[c4] - 1u] : 0ull;
              }
            }
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
    int any_kept = 0;                            // (uniform after each barrier) LiDAR points the filter left, so far
    for (int j0 = 0; j0 < na; j0 += PTB) {
      const int m = min(PTB, na - j0);
      __syncthreads();
      if (threadIdx.x == 0) { s_pending = 0; s_nkept = 0; }
      __syncthreads();
      int mine_kept = 0;
      for (int i = threadIdx.x; i < m; i += BT) {
        const bool there = !amask || flags[ba + j0 + i];
        mine_kept += there ? 1 : 0;
        sx[i] = there ? px[ba + j0 + i] : (double)INFINITY;      // (a point the filter dropped is within C of nothing)
        sy[i] = py[ba + j0 + i];
        sz[i] = pz[ba + j0 + i];
      }
      if (mine_kept) s_nkept = 1;                   // (plain store, every writer the same value: atomics on one LDS word serialise lane by lane)
      __syncthreads();
      any_kept |= s_nkept;
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
      if (!s_pending && any_kept) break;         // (with nothing kept so far the rest of the list decides whether there is a fuse)
    }
    if (valid && any_kept) flags[bq + q] = found ? 1 : 0;      // (no LiDAR point left: fuse skipped, the flags stand)
  }
}


// ---------------------------------------------------------------- compaction
// In-order compaction of segment s by flags.  dst = src (in place) or, when
// dst_after_base != nullptr, directly behind another segment
// (dst_after_base[s] + dst_after_cnt[s]); base_out[s] is updated then.
constexpr int CPT = 1024;   // threads per compaction workgroup
constexpr int CPE = 2;      // consecutive elements per thread (4 needed 76 VGPRs: one 1024-thread workgroup per CU)
// (measured with tools/ab_builds.py, stage ballquery_fuse 0.78 ms: a second launch with 4 / 8 elements per thread for the lists
// above 16 384 positions -- fewer steps for the longest list -- 0.85 / 0.95: a lane's consecutive elements are 32 / 64 bytes apart per
// plane, the loads stop coalescing; coordinates requested together with the flag instead of behind it -- one round trip per
// step instead of two -- 0.86: two thirds of a pseudo list are dropped by the fuse, their coordinates were read for nothing)
// In-order compaction of segment s by flags.  dst = src (in place) or, when dst_after_base != nullptr, directly behind
// another segment (dst_after_base[s] + dst_after_cnt[s]); seg_base[s] is updated then.
// compact_front: that other segment carries flags as well (a joint filter pass without compaction).  The two lists are
// then ONE list to this kernel -- positions [0, front) the other segment (the short one: the per-instance LiDAR list),
// [front, front + n) this one -- compacted towards dst_after_base[s]: the survivors of the second list land directly behind
// the survivors of the first, no step and no barrier more than for the second list alone; both counts are updated.
// Needs base + front <= seg_base (the second list lies behind the first), so that a destination is never ahead of a source.
__global__ __launch_bounds__(CPT) void k_seg_compact(
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
    long long *__restrict__ seg_base, int *__restrict__ seg_cnt,
    const uint8_t *__restrict__ flags, const long long *__restrict__ dst_after_base,
    int *__restrict__ dst_after_cnt, int compact_front) {
  __shared__ int s_w[CPT / 64];
  // (NOT longest first as k_range_cluster_grid / k_fit_gather: a grid of size classes decides the class from seg_cnt, and this kernel
  // REWRITES seg_cnt -- a list that shrank across a class boundary was compacted a second time by the workgroup of its new
  // class whenever that one started late enough, i.e. in launches of thousands of segments only: 2 % more rows in bench.py's
  // parity block, nothing in the tests of the time.  tests/test_gpu_engine.py now compares one large launch with many small ones.)
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  const long long src = seg_base[s];
  const int front_all = dst_after_base ? dst_after_cnt[s] : 0;
  const int front = (compact_front && dst_after_base) ? front_all : 0;     // positions of the front list that are compacted here
  const long long src_f = dst_after_base ? dst_after_base[s] : 0;
  const long long dst = dst_after_base ? src_f + (front_all - front) : src;
  const int n_all = front + n;
  if (n_all == 0) {
    if (dst_after_base && threadIdx.x == 0) seg_base[s] = dst;
    return;
  }
  int running = 0, kept_front = 0;
  for (int t0 = 0; t0 < n_all; t0 += CPT * CPE) {
    const int i0 = t0 + threadIdx.x * CPE;
    bool f[CPE];
    double x[CPE], y[CPE], z[CPE];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < CPE; k++) {
      const int i = i0 + k;
      const long long g = (i < front) ? src_f + i : src + (i - front);
      f[k] = (i < n_all) && flags[g];
      x[k] = y[k] = z[k] = 0.0;
      if (f[k]) { x[k] = px[g]; y[k] = py[g]; z[k] = pz[g]; }
      mine += f[k] ? ((i < front) ? 0x10001 : 1) : 0;     // kept | kept of the front list, in one scan
    }
    int tot;
    int r = block_excl_scan<CPT / 64>(mine, s_w, tot) & 0xFFFF;   // barriers: loads above complete first  (a step holds 2048 positions)
#pragma unroll
    for (int k = 0; k < CPE; k++) {
      if (f[k]) {
        const long long d = dst + running + r;            // d <= source position: never ahead of the reads
        px[d] = x[k]; py[d] = y[k]; pz[d] = z[k];
        r++;
      }
    }
    running += tot & 0xFFFF;
    kept_front += tot >> 16;
  }
  if (threadIdx.x == 0) {
    seg_cnt[s] = running - kept_front;
    if (dst_after_base) seg_base[s] = dst + kept_front;
    if (front) dst_after_cnt[s] = kept_front;
  }
}

// ---------------------------------------------------------------- a11 voxel down-sample + statistical
// Open3D PointCloud::VoxelDownSample(voxel_size) (my_loader0.py:734; the leaf is not in the reference tree -- restated
// from Open3D's published source, `oracle/penet_oracle.py: voxel_down_sample` is the checker): with
//   voxel_min_bound = min over the points - voxel_size / 2,   index = (int)floor((p - voxel_min_bound) / voxel_size)
// every voxel's points are summed IN INPUT ORDER and the sum is divided by their number.  Open3D emits the centroids in
// the iteration order of an std::unordered_map, which is unspecified; this library DEFINES the order as first-seen:
// voxels come out in the order of their first point in the input list.
// One workgroup per segment.  A hash table over the voxel index (open addressing in global scratch, 2n .. 4n slots)
// gives every voxel the list position of its first point; the sums live at that position.  The points are walked in
// chunks of 1024 in list order: inside a chunk the first point of every voxel (no earlier point of the chunk in the same
// slot) takes the voxel's running sum, adds the chunk's points of that voxel in list order and puts it back -- so every
// sum is formed in exactly the reference's order, whatever the scheduling.  Quadratic in the chunk (LDS reads), linear
// in the segment; the stage is off by default and is built for exactness, not for speed.
constexpr int VD_T = 1024;
constexpr unsigned long long VD_EMPTY = ~0ull;
struct VdScratch {
  unsigned long long *hkey;          // 4 * pool_cap: segment s owns [4 * base, 4 * base + H), H = power of two in [2n, 4n)
  int *hfirst;                       // 4 * pool_cap: list position of the slot's first point
  double *ax, *ay, *az;              // pool_cap: running sums of the voxel whose first point sits at this pool position
  int *acnt, *vfirst;                // pool_cap: its number of points | for every point: the first point of its voxel
};
__host__ __device__ inline VdScratch vd_scratch(void *scratch, int64_t P) {
  VdScratch W;
  W.hkey = (unsigned long long *)scratch;
  W.ax = (double *)(W.hkey + 4 * P);
  W.ay = W.ax + P;
  W.az = W.ay + P;
  W.hfirst = (int *)(W.az + P);
  W.acnt = W.hfirst + 4 * P;
  W.vfirst = W.acnt + P;
  return W;
}
// L2-served loads of what other threads of the workgroup stored or updated with atomics (the vector L1 is not coherent)
__device__ __forceinline__ int ld_agent_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_d(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__global__ __launch_bounds__(VD_T) void k_voxel_down(
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
    const long long *__restrict__ seg_base, int *__restrict__ seg_cnt, const int *__restrict__ enable,
    double voxel, VdScratch W, uint32_t *__restrict__ status) {
  __shared__ int s_slot[VD_T];
  __shared__ double s_x[VD_T], s_y[VD_T], s_z[VD_T];
  __shared__ double s_red[3][VD_T / 64];
  __shared__ int s_w[VD_T / 64];
  const int s = blockIdx.x;
  if (!enable[s]) return;
  const int n = seg_cnt[s];
  if (n <= 0) return;
  const long long base = seg_base[s];
  int H = 2;
  while (H < 2 * n) H <<= 1;
  unsigned long long *hk = W.hkey + 4 * base;
  int *hf = W.hfirst + 4 * base;
  for (int i = threadIdx.x; i < H; i += VD_T) { hk[i] = VD_EMPTY; hf[i] = 0x7FFFFFFF; }
  double mx = INFINITY, my = INFINITY, mz = INFINITY;
  for (int i = threadIdx.x; i < n; i += VD_T) {
    W.ax[base + i] = 0.0; W.ay[base + i] = 0.0; W.az[base + i] = 0.0; W.acnt[base + i] = 0;
    mx = fmin(mx, px[base + i]); my = fmin(my, py[base + i]); mz = fmin(mz, pz[base + i]);
  }
  mx = wave_min_d(mx); my = wave_min_d(my); mz = wave_min_d(mz);
  if (lane_id() == 0) { s_red[0][threadIdx.x >> 6] = mx; s_red[1][threadIdx.x >> 6] = my; s_red[2][threadIdx.x >> 6] = mz; }
  __threadfence();
  __syncthreads();
  for (int w = 0; w < VD_T / 64; w++) { mx = fmin(mx, s_red[0][w]); my = fmin(my, s_red[1][w]); mz = fmin(mz, s_red[2][w]); }
  const double bx = mx - voxel * 0.5, by = my - voxel * 0.5, bz = mz - voxel * 0.5;     // voxel_min_bound
  bool range_err = false;
  for (int c0 = 0; c0 < n; c0 += VD_T) {
    const int i = c0 + (int)threadIdx.x;
    const bool valid = i < n;
    const int cn = min(VD_T, n - c0);
    int h = -1;
    if (valid) {
      const double x = px[base + i], y = py[base + i], z = pz[base + i];
      s_x[threadIdx.x] = x; s_y[threadIdx.x] = y; s_z[threadIdx.x] = z;
      const double fx = floor((x - bx) / voxel), fy = floor((y - by) / voxel), fz = floor((z - bz) / voxel);
      // (indices are >= 0 by construction; Open3D refuses a cloud wider than voxel_size * INT_MAX, this table one wider
      // than 2^21 voxels along an axis: 100 km at 0.05 m)
      if (!(fx >= 0.0 && fy >= 0.0 && fz >= 0.0 && fx < 2097152.0 && fy < 2097152.0 && fz < 2097152.0)) range_err = true;
      const unsigned long long key = (unsigned long long)(long long)fmin(fmax(fx, 0.0), 2097151.0) |
                                     ((unsigned long long)(long long)fmin(fmax(fy, 0.0), 2097151.0) << 21) |
                                     ((unsigned long long)(long long)fmin(fmax(fz, 0.0), 2097151.0) << 42);
      h = (int)(mix64(key) & (unsigned long long)(H - 1));
      while (true) {
        const unsigned long long old = atomicCAS(&hk[h], VD_EMPTY, key);
        if (old == VD_EMPTY || old == key) break;
        h = (h + 1) & (H - 1);
      }
      atomicMin(&hf[h], i);
    }
    s_slot[threadIdx.x] = h;
    __threadfence();
    __syncthreads();
    if (valid) {
      const int f = ld_agent_i(&hf[h]);               // final: every later point has a larger position
      W.vfirst[base + i] = f;
      bool leader = true;                              // no earlier point of the chunk in the same voxel
      for (int j = (int)threadIdx.x - 1; j >= 0 && leader; j--) leader = s_slot[j] != h;
      if (leader) {
        double sx = ld_agent_d(&W.ax[base + f]), sy = ld_agent_d(&W.ay[base + f]), sz = ld_agent_d(&W.az[base + f]);
        int c = ld_agent_i(&W.acnt[base + f]);
        for (int j = threadIdx.x; j < cn; j++)
          if (s_slot[j] == h) { sx += s_x[j]; sy += s_y[j]; sz += s_z[j]; c++; }     // AccumulatedPoint::AddPoint, in list order
        W.ax[base + f] = sx; W.ay[base + f] = sy; W.az[base + f] = sz; W.acnt[base + f] = c;
      }
    }
    __threadfence();
    __syncthreads();
  }
  if (range_err) atomicOr(status, DFU3D_ST_VOXEL_RANGE);
  // centroids in first-seen order, written over the list (every source is in the sums by now)
  int running = 0;
  for (int c0 = 0; c0 < n; c0 += VD_T) {
    const int i = c0 + (int)threadIdx.x;
    const bool first = (i < n) && (W.vfirst[base + i] == i);
    int tot;
    const int rank = running + block_excl_scan<VD_T / 64>(first ? 1 : 0, s_w, tot);
    if (first) {
      const double c = (double)ld_agent_i(&W.acnt[base + i]);
      px[base + rank] = ld_agent_d(&W.ax[base + i]) / c;          // AccumulatedPoint::GetAveragePoint
      py[base + rank] = ld_agent_d(&W.ay[base + i]) / c;
      pz[base + rank] = ld_agent_d(&W.az[base + i]) / c;
    }
    running += tot;
  }
  if (threadIdx.x == 0) seg_cnt[s] = running;
}

// Open3D remove_statistical_outlier, first half: mean distance to the knn nearest points (the point itself included).
// The candidate list of a query -- its knn smallest squared distances so far, ascending -- lives in LDS, query-minor
// (best[k][query]: consecutive lanes, consecutive words), and its largest entry in a register: a point farther than
// that is rejected without touching the list.  (Until round 4 the list was a private array of 64 doubles: 528 bytes of
// scratch per thread.)
constexpr int KMAX = 64;
constexpr int KQ = 128;     // queries per workgroup
__global__ __launch_bounds__(KQ) void k_knn_mean(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ seg_base,
    const int *__restrict__ seg_cnt, const int *__restrict__ enable, int knn, int S,
    const int *__restrict__ tile_off, double *__restrict__ mean_d) {
  __shared__ double sx[PT], sy[PT], sz[PT];
  extern __shared__ double s_best[];            // knn x KQ
  const int t = blockIdx.x;
  if (t >= tile_off[S]) return;
  const int s = find_segment(tile_off, S, t);
  if (!enable[s]) return;
  const int q0 = (t - tile_off[s]) * KQ;
  const int n = seg_cnt[s];
  const long long base = seg_base[s];
  const int q = q0 + threadIdx.x;
  const bool valid = q < n;
  double x = 0.0, y = 0.0, z = 0.0;
  if (valid) { x = px[base + q]; y = py[base + q]; z = pz[base + q]; }
  const int kk = min(knn, n);
  double *best = s_best + threadIdx.x;          // best[p * KQ]: ascending squared distances
  int nbest = 0;
  double worst = INFINITY;                      // best[kk - 1] once the list is full
  for (int j0 = 0; j0 < n; j0 += PT) {
    const int m = min(PT, n - j0);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += KQ) {
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
        if (nbest < kk || d < worst) {
          int p = (nbest < kk) ? nbest : kk - 1;
          while (p > 0 && best[(p - 1) * KQ] > d) { best[p * KQ] = best[(p - 1) * KQ]; p--; }
          best[p * KQ] = d;
          if (nbest < kk) nbest++;
          if (nbest == kk) worst = best[(kk - 1) * KQ];
        }
      }
    }
  }
  if (valid) {
    double sum = 0.0;
    for (int i = 0; i < nbest; i++) sum += sqrt(best[i * KQ]);
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
  return (int64_t)V * DFU3D_MAX_INST * ((int64_t)seg_ranges(a_cap) + seg_ranges(b_cap));
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
  const int nra = seg_ranges(a_cap), nrb = seg_ranges(b_cap);
  int32_t *rc_a = chunk_cnt, *rc_b = chunk_cnt + (size_t)V * nra * DFU3D_MAX_INST;
  const int gxa = std::min((nra + SEG_WPB - 1) / SEG_WPB, SEG_GX), gxb = std::min((nrb + SEG_WPB - 1) / SEG_WPB, SEG_GX);
  hipLaunchKernelGGL(k_seg_count, dim3(gxa, V), dim3(SEG_WPB * 64), 0, st, a_bits, a_n, a_cap, nra, rc_a);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_count, dim3(gxb, V), dim3(SEG_WPB * 64), 0, st, b_bits, b_n, b_cap, nrb, rc_b);
  DFU3D_LAUNCH_CHECK();
  const SegSide SA = {a_n, a_cap, nra, rc_a, cnt_a}, SB = {b_n, b_cap, nrb, rc_b, cnt_b};
  hipLaunchKernelGGL(k_seg_scan, dim3(V, 2), dim3(1024), 0, st, SA, SB, max_inst);
  DFU3D_LAUNCH_CHECK();
  const JointSegs J = {(long long *)base_ab, cnt_ab, rad_ab, rad_a, rad_b};
  hipLaunchKernelGGL(k_seg_alloc, dim3(1), dim3(1024), 0, st, S, cnt_a, cnt_b,
                     (long long *)base_a, (long long *)base_b, (long long)pool_cap,
                     (long long *)pool_cursor, status, J);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_write, dim3(gxa, (V + 7) / 8 * 8), dim3(SEG_WPB * 64), 0, st, a_bits, a_x, a_y, a_z, a_n, a_cap,
                     max_inst, nra, (const long long *)base_a, cnt_a, rc_a, px, py, pz, (float4 *)shadow, rad_a, 0, V);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_write, dim3(gxb, (V + 7) / 8 * 8), dim3(SEG_WPB * 64), 0, st, b_bits, b_x, b_y, b_z, b_n, b_cap,
                     max_inst, nrb, (const long long *)base_b, cnt_b, rc_b, px, py, pz, (float4 *)shadow, rad_b, S, V);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int64_t dfu3d_rf_shadow_bytes(int64_t pool_cap) { return pool_cap > 0 ? DFU3D_SHADOW_BYTES(pool_cap) : DFU3D_EINVAL; }
extern "C" int64_t dfu3d_rf_queue_ints(int64_t pool_cap) { return pool_cap > 0 ? DFU3D_RF_QUEUE_INTS(pool_cap) : DFU3D_EINVAL; }

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
  // scratch behind the shadow proper: one record (two boxes, a count) per 512 slots, the ranges' U slots, the
  // per-segment overflow lists (at the segments' own pool positions), counters, the tables of the long queries
  const size_t n_ranges = (size_t)((pool_cap + 511) / 512 + 1);
  RfScratch W;
  W.flags = flags;
  W.queue = queue;
  W.qcap = rf_queue_part_cap(pool_cap);
  W.rrec = (float *)(pq + pool_cap);
  W.uent = (float4 *)(W.rrec + BOX_FLOATS * n_ranges);
  W.upos = (uint32_t *)(W.uent + (size_t)RF_WLIST * n_ranges);
  W.ulist = (float4 *)(W.upos + (size_t)RF_WLIST * n_ranges);
  W.work = (int *)(W.ulist + pool_cap);
  W.ucount = W.work + RF_WORK_HDR;
  W.seg_base = (const long long *)seg_base;
  if (phases & DFU3D_RF_SHADOW) {
    // positions outside the given segments carry the "no segment" mark (all bits set)
    if (dfu3d_fill_async(pq, 0xFF, sizeof(float4) * (size_t)pool_cap, st) != hipSuccess) return DFU3D_ELAUNCH;
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, S, seg_cnt, tile_off, QT);
    DFU3D_LAUNCH_CHECK();
    const int g = tile_grid(pool_cap, S);
    hipLaunchKernelGGL(k_shadow_build, dim3(g < 4096 ? g : 4096), dim3(QT), 0, st, px, py, pz,
                       (const long long *)seg_base, seg_cnt, radius, S, tile_off, pq);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_FLAGS) {
    if (dfu3d_fill_small_async(queue, RF_QHDR * sizeof(int), W.work, sizeof(int) * (size_t)(RF_WORK_HDR + S), nullptr, 0, st) != hipSuccess)
      return DFU3D_ELAUNCH;
    const long long n_tiles = (pool_cap + RF_WG - 1) / RF_WG;
    const dim3 grid((unsigned)(n_tiles < RF_GRID ? n_tiles : RF_GRID));
    if (nb_points == 1)
      hipLaunchKernelGGL(k_rf_stream<true>, grid, dim3(RFB), 0, st, pq, (const long long *)n_used, (long long)pool_cap,
                         nb_points, S, W);
    else
      hipLaunchKernelGGL(k_rf_stream<false>, grid, dim3(RFB), 0, st, pq, (const long long *)n_used, (long long)pool_cap,
                         nb_points, S, W);
    DFU3D_LAUNCH_CHECK();
    const long long g_pair = ((long long)n_ranges + 3) / 4;                     // a wave per range
    hipLaunchKernelGGL(k_rf_pair, dim3((unsigned)(g_pair < 4096 ? g_pair : 4096)), dim3(256), 0, st, (const long long *)n_used,
                       (long long)pool_cap, nb_points, W);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_RESOLVE) {
    hipLaunchKernelGGL(k_rf_resolve, dim3(8192), dim3(256), 0, st, px, py, pz, pq, (const long long *)n_used,
                       (long long)pool_cap, (const long long *)seg_base, seg_cnt, radius, nb_points, S, W);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rf_ranges, dim3(512), dim3(256), 0, st, px, py, pz, pq, (const long long *)n_used,
                       (long long)pool_cap, (const long long *)seg_base, seg_cnt, radius, nb_points, S, W);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_COMPACT) {
    hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz,
                       (long long *)seg_base, seg_cnt, flags, (const long long *)nullptr,
                       (int *)nullptr, 0);
    DFU3D_LAUNCH_CHECK();
  }
  return DFU3D_OK;
}

extern "C" int64_t dfu3d_voxel_down_sample_scratch_bytes(int64_t pool_cap) {
  return pool_cap > 0 ? 80 * pool_cap : DFU3D_EINVAL;      // 4 P keys (8 B) + 3 P sums (8 B) + 4 P + P + P ints
}

extern "C" int dfu3d_voxel_down_sample(double *px, double *py, double *pz, const int64_t *seg_base,
                                       int32_t *seg_cnt, const int32_t *enable, double voxel_size,
                                       int32_t S, int64_t pool_cap, void *scratch, uint32_t *status,
                                       void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!px || !py || !pz || !seg_base || !seg_cnt || !enable || !scratch || !status) return DFU3D_EINVAL;
  if (S <= 0 || pool_cap <= 0 || !(voxel_size > 0.0) || !(voxel_size < 1e300)) return DFU3D_EINVAL;   // (Open3D: voxel_size <= 0 is an error)
  if (pool_cap >= (1ll << 29)) return DFU3D_ERANGE;            // hash regions are addressed with 32-bit positions
  if ((uintptr_t)scratch & 7u) return DFU3D_EINVAL;
  hipLaunchKernelGGL(k_voxel_down, dim3(S), dim3(VD_T), 0, (hipStream_t)stream, px, py, pz, (const long long *)seg_base,
                     seg_cnt, enable, voxel_size, vd_scratch(scratch, pool_cap), status);
  DFU3D_LAUNCH_CHECK();
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
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, S, seg_cnt, tile_off, KQ);
  DFU3D_LAUNCH_CHECK();
  if (hipFuncSetAttribute((const void *)k_knn_mean, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * KMAX * KQ)) != hipSuccess)
    return DFU3D_ELAUNCH;
  hipLaunchKernelGGL(k_knn_mean, dim3((unsigned)((pool_cap + KQ - 1) / KQ + S)), dim3(KQ), sizeof(double) * (size_t)nb_neighbors * KQ, st,
                     px, py, pz, (const long long *)seg_base, seg_cnt, enable, nb_neighbors, S, tile_off,
                     mean_d);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_stat_flags, dim3(S), dim3(256), 0, st, (const long long *)seg_base,
                     seg_cnt, enable, std_ratio, mean_d, flags);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz,
                     (long long *)seg_base, seg_cnt, flags, (const long long *)nullptr,
                     (int *)nullptr, 0);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

static int ballquery_fuse_impl(double *px, double *py, double *pz, const int64_t *base_a,
                                    int32_t *cnt_a, int64_t *base_b, int32_t *cnt_b,
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
  const int g_small = (int)(((pool_cap + BT_SMALL - 1) / BT_SMALL + S + BALL_TPW - 1) / BALL_TPW);
  hipLaunchKernelGGL((k_ball_flags<BT_SMALL, BH_MAX_SMALL, BH_HEADS_SMALL>), dim3(g_small), dim3(BT_SMALL), 0, st, px, py, pz,
                     (const long long *)base_a, cnt_a, (const long long *)base_b, cnt_b, T, C, S,
                     tile_small, flags, masked);
  DFU3D_LAUNCH_CHECK();
  const int g_big = (int)(((pool_cap + BT_BIG - 1) / BT_BIG + S + BALL_TPW - 1) / BALL_TPW);
  hipLaunchKernelGGL((k_ball_flags<BT_BIG, BH_MAX_BIG, BH_HEADS_BIG>), dim3(g_big), dim3(BT_BIG), 0, st, px, py, pz,
                     (const long long *)base_a, cnt_a, (const long long *)base_b, cnt_b, T, C, S,
                     tile_big, flags, masked);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz, (long long *)base_b,
                     cnt_b, flags, (const long long *)base_a, cnt_a, masked == 2 ? 1 : 0);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_ballquery_fuse(double *px, double *py, double *pz, const int64_t *base_a,
                                    const int32_t *cnt_a, int64_t *base_b, int32_t *cnt_b,
                                    double C, int32_t S, int64_t pool_cap, int32_t *tile_off,
                                    uint8_t *flags, void *stream) {
  return ballquery_fuse_impl(px, py, pz, base_a, (int32_t *)cnt_a, base_b, cnt_b, C, S, pool_cap, tile_off,
                             flags, 0, stream);
}

extern "C" int dfu3d_ballquery_fuse_masked(double *px, double *py, double *pz,
                                           const int64_t *base_a, const int32_t *cnt_a,
                                           int64_t *base_b, int32_t *cnt_b, double C, int32_t S,
                                           int64_t pool_cap, int32_t *tile_off, uint8_t *flags,
                                           void *stream) {
  return ballquery_fuse_impl(px, py, pz, base_a, (int32_t *)cnt_a, base_b, cnt_b, C, S, pool_cap, tile_off,
                             flags, 1, stream);
}

extern "C" int dfu3d_ballquery_fuse_joint(double *px, double *py, double *pz,
                                          const int64_t *base_a, int32_t *cnt_a,
                                          int64_t *base_b, int32_t *cnt_b, double C, int32_t S,
                                          int64_t pool_cap, int32_t *tile_off, uint8_t *flags,
                                          void *stream) {
  return ballquery_fuse_impl(px, py, pz, base_a, cnt_a, base_b, cnt_b, C, S, pool_cap, tile_off,
                             flags, 2, stream);
}
