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

// ---------------------------------------------------------------- build
__global__ __launch_bounds__(1024) void k_seg_count(const uint32_t *__restrict__ bits,
                                                   const int *__restrict__ n_item,
                                                   int cap_item, int max_inst,
                                                   int *__restrict__ cnt) {
  __shared__ int s_c[DFU3D_MAX_INST];
  const int v = blockIdx.x;
  if (threadIdx.x < DFU3D_MAX_INST) s_c[threadIdx.x] = 0;
  __syncthreads();
  const int n = min(n_item[v], cap_item);
  for (int t = threadIdx.x; t < n; t += 1024) {
    uint32_t b = bits[(size_t)v * cap_item + t];
    while (b) {
      const int j = __ffs((int)b) - 1;
      atomicAdd(&s_c[j], 1);
      b &= b - 1u;
    }
  }
  __syncthreads();
  if (threadIdx.x < max_inst) cnt[v * max_inst + threadIdx.x] = s_c[threadIdx.x];
}

__global__ __launch_bounds__(1024) void k_seg_alloc(int S, int *__restrict__ cnt_a,
                                                    int *__restrict__ cnt_b,
                                                    long long *__restrict__ base_a,
                                                    long long *__restrict__ base_b,
                                                    long long pool_cap,
                                                    long long *__restrict__ cursor,
                                                    uint32_t *__restrict__ status) {
  __shared__ int s_w[16];
  long long running = *cursor;
  bool over = false;
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
        over = over || (ca + cb > 0);
      } else {
        base_a[s] = b;
        base_b[s] = b + ca;
      }
    }
    running += tot;
  }
  if (over) atomicOr(status, DFU3D_ST_POOL_OVERFLOW);
  __syncthreads();
  if (threadIdx.x == 0) *cursor = running < pool_cap ? running : pool_cap;
}

// One workgroup per VIEW writes the ordered lists of all its instances in a single sweep over the view's
// items (a workgroup per instance read the 100k+ items of the view once per instance: L2-bound).  Per step of
// 1024 items: per-instance ballots inside each wave, wave totals through LDS, two barriers.
constexpr int SWT = 1024;
__global__ __launch_bounds__(SWT) void k_seg_write(
    const uint32_t *__restrict__ bits, const double *__restrict__ ix,
    const double *__restrict__ iy, const double *__restrict__ iz,
    const int *__restrict__ n_item, int cap_item, int max_inst,
    const long long *__restrict__ base, const int *__restrict__ cnt,
    double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz) {
  __shared__ int s_wc[SWT / 64][DFU3D_MAX_INST];      // per wave: items of instance j in this step
  __shared__ int s_run[DFU3D_MAX_INST];               // per instance: items written before this step
  const int v = blockIdx.x;
  const int n = min(n_item[v], cap_item);
  const int wave = threadIdx.x >> 6, lane = lane_id();
  uint32_t live = 0u;                                  // instances of this view with a non-empty list
  for (int j = 0; j < max_inst; j++)
    if (cnt[v * max_inst + j] > 0) live |= 1u << j;
  if (live == 0u || n == 0) return;
  if (threadIdx.x < DFU3D_MAX_INST) s_run[threadIdx.x] = 0;
  int mytot = 0;                                       // thread j < max_inst: instance j's total of the previous step
  for (int t0 = 0; t0 < n; t0 += SWT) {
    const int t = t0 + threadIdx.x;
    const size_t o = (size_t)v * cap_item + t;
    const uint32_t b = (t < n) ? (bits[o] & live) : 0u;
    __syncthreads();                                   // A: the previous step's s_wc / s_run have been read
    if ((int)threadIdx.x < max_inst) s_run[threadIdx.x] += mytot;
    for (uint32_t w = live; w; w &= w - 1u) {          // uniform: every live instance, every wave
      const int j = __ffs((int)w) - 1;
      const unsigned long long m = __ballot((b >> j) & 1u);
      if (lane == 0) s_wc[wave][j] = __popcll(m);
    }
    __syncthreads();                                   // B: counts of all waves, running totals up to date
    if ((int)threadIdx.x < max_inst) {
      int tsum = 0;
#pragma unroll
      for (int w = 0; w < SWT / 64; w++) tsum += s_wc[w][threadIdx.x];
      mytot = ((live >> threadIdx.x) & 1u) ? tsum : 0;
    }
    uint32_t wany = b;                                 // instances present in this wave (uniform after the OR)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wany |= (uint32_t)__shfl_xor((int)wany, m, 64);
    double x = 0.0, y = 0.0, z = 0.0;
    if (b) { x = ix[o]; y = iy[o]; z = iz[o]; }
    for (uint32_t w = wany; w; w &= w - 1u) {          // uniform per wave
      const int j = __ffs((int)w) - 1;
      const unsigned long long m = __ballot((b >> j) & 1u);
      int before = s_run[j];
      for (int ww = 0; ww < wave; ww++) before += s_wc[ww][j];
      if ((b >> j) & 1u) {
        const long long d = base[v * max_inst + j] + before + __popcll(m & ((1ull << lane) - 1ull));
        px[d] = x;
        py[d] = y;
        pz[d] = z;
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
// Phase A (streaming, no LDS): a query is first tested against itself and its
// two nearest list neighbours (lane^1, lane^2) -- the lists are in pixel / sweep
// order, so list neighbours are spatial neighbours and nb_points = 1 is settled
// right there for ~98 % of the points.  If a lane of the wave is still
// undecided, eight of the wave's points are broadcast one by one (v_readlane,
// scalar operands) until every lane has its nb_points+1 distinct hits.
// Whatever is still undecided is collected in a workgroup-local LDS list and
// appended with one global atomic to the queue of phase B, which sees the whole
// segment.
// Each workgroup walks RF_TPB consecutive query tiles: one binary search, then a
// linear step from segment to segment.
constexpr int RF_TPB = 8;
constexpr int RF_LONG = 2048;      // phase B hands segments longer than this to a whole workgroup
constexpr int RF_STRIDE = 8;       // phase A tries lanes 0, 8, 16, ... of the wave

__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(QT) void k_radius_flags(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ seg_base,
    const int *__restrict__ seg_cnt, const double *__restrict__ radius, int nb, int S,
    const int *__restrict__ tile_off, uint8_t *__restrict__ flags, int *__restrict__ queue) {
  __shared__ int2 s_list[RF_TPB * QT];
  __shared__ int s_n, s_base;
  const int ntile = tile_off[S];
  int t = blockIdx.x * RF_TPB;
  if (t >= ntile) return;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int t_end = min(t + RF_TPB, ntile);
  int s = find_segment(tile_off, S, t);
  const int lane = lane_id();
  const int w_in_tile = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  // segment facts stay in registers while consecutive tiles belong to the same segment
  int t_first = tile_off[s], t_next = tile_off[s + 1];
  int n = seg_cnt[s];
  long long base = seg_base[s];
  double r = radius[s];
  for (; t < t_end; t++) {
    if (t >= t_next) {
      do { s++; t_first = t_next; t_next = tile_off[s + 1]; } while (t >= t_next);
      n = seg_cnt[s];
      base = seg_base[s];
      r = radius[s];
    }
    const int w0 = (t - t_first) * QT + w_in_tile;         // first point of this wave (uniform)
    if (w0 >= n) continue;
    const int q = w0 + lane;
    const bool valid = q < n;
    if (!(r > 0.0)) {               // r == 0: no filter; r < 0 (or NaN): drop all
      if (valid) flags[base + q] = (r == 0.0) ? 1 : 0;
      continue;
    }
    const double r2 = r * r;
    double x = 0.0, y = 0.0, z = 0.0;
    if (valid) { x = px[base + q]; y = py[base + q]; z = pz[base + q]; }
    const int wn = min(64, n - w0);
    // self + the two nearest list neighbours (quad permutes, no LDS traffic)
    int cnt = 0;
    {
      const int l1 = lane ^ 1, l2 = lane ^ 2;
      const double x1 = shfl_xor_d(x, 1), y1 = shfl_xor_d(y, 1), z1 = shfl_xor_d(z, 1);
      const double x2 = shfl_xor_d(x, 2), y2 = shfl_xor_d(y, 2), z2 = shfl_xor_d(z, 2);
      double dx = x - x1, dy = y - y1, dz = z - z1;
      double d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      cnt = 1 + ((l1 < wn && d < r2) ? 1 : 0);                 // self: d == 0 < r2
      dx = x - x2; dy = y - y2; dz = z - z2;
      d = dx * dx;
      d += dy * dy;
      d += dz * dz;
      cnt += (l2 < wn && d < r2) ? 1 : 0;
    }
    // undecided lanes: a few of the wave's points, broadcast as scalar operands
    // (v_readlane), are tried by all of them at once; what is left after that is
    // almost surely isolated and goes to phase B
    if (__ballot(valid && cnt <= nb)) {
      const int l1 = lane ^ 1, l2 = lane ^ 2;
      for (int j = 0; j < wn; j += RF_STRIDE) {
        const double dx = x - readlane_d(x, j), dy = y - readlane_d(y, j), dz = z - readlane_d(z, j);
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        if (d < r2 && j != lane && j != l1 && j != l2) cnt++;   // never count a point twice
        if (__ballot(valid && cnt <= nb) == 0ull) break;
      }
    }
    const bool pending = valid && cnt <= nb;       // a lower bound that did not reach nb_points + 1
    if (valid && !pending) flags[base + q] = 1;
    const unsigned long long pm = __ballot(pending);
    if (pm) {                       // block-local list (LDS), one LDS atomic per wave
      int slot0 = 0;
      if (lane == 0) slot0 = atomicAdd(&s_n, __popcll(pm));
      slot0 = __builtin_amdgcn_readfirstlane(slot0);
      if (pending) {
        const int slot = slot0 + __popcll(pm & ((1ull << lane) - 1ull));
        s_list[slot] = make_int2(s, q);
      }
    }
  }
  __syncthreads();
  const int np = s_n;
  if (np == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&queue[0], np);        // one global atomic per workgroup
  __syncthreads();
  int2 *out = (int2 *)(queue + 2) + s_base;
  for (int i = threadIdx.x; i < np; i += QT) out[i] = s_list[i];
}

// Phase B: one wave per undecided query; the 64 lanes stride the whole segment
// (coalesced 8 B/lane loads) and leave together as soon as the count is reached.
__global__ __launch_bounds__(256) void k_radius_resolve(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ seg_base,
    const int *__restrict__ seg_cnt, const double *__restrict__ radius, int nb,
    uint8_t *__restrict__ flags, int *__restrict__ queue, long long pool_cap) {
  const int nq = queue[0];
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  const int lane = lane_id();
  for (int e = wave; e < nq; e += nwaves) {
    const int s = queue[2 + 2 * e], q = queue[3 + 2 * e];
    const int n = seg_cnt[s];
    const long long base = seg_base[s];
    const double r = radius[s], r2 = r * r;
    const double x = px[base + q], y = py[base + q], z = pz[base + q];
    // own 64-point chunk first (list neighbours are near), then the rest of the
    // segment RU chunks at a time so that the loads of one step overlap
    int cnt = 0;
    const int c0 = q >> 6;
    {
      const int j = (c0 << 6) + lane;
      bool hit = false;
      if (j < n) {
        const double dx = x - px[base + j], dy = y - py[base + j], dz = z - pz[base + j];
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        hit = d < r2;
      }
      cnt = __popcll(__ballot(hit));
    }
    constexpr int RU = 4;
    // then outwards from the own chunk, RU chunks per step (their loads overlap).  A long
    // segment that did not settle in the first step is swept by a whole workgroup
    // (k_radius_resolve_long); its entry moves to the top end of the queue
    bool handed = false;
    const int nch = (n + 63) >> 6;
    int lo = c0 - 1, hi = c0 + 1;
    for (int step = 0; (lo >= 0 || hi < nch) && cnt <= nb; step++) {
      if (step == 1 && n > RF_LONG) {
        int h = 0;
        if (lane == 0) h = atomicAdd(&queue[1], 1);
        h = __builtin_amdgcn_readfirstlane(h);
        if ((long long)nq + h + 1 <= pool_cap) {           // room left (always, in practice)
          if (lane == 0) {
            queue[2 + 2 * (pool_cap - 1 - h)] = s;
            queue[3 + 2 * (pool_cap - 1 - h)] = q;
          }
          handed = true;
          break;
        }
      }
      double cx[RU], cy[RU], cz[RU];
#pragma unroll
      for (int u = 0; u < RU; u++) {
        int c;                                            // uniform: next chunk above / below in turn
        if (u & 1) c = (lo >= 0) ? lo-- : ((hi < nch) ? hi++ : -1);
        else c = (hi < nch) ? hi++ : ((lo >= 0) ? lo-- : -1);
        const int j = (c << 6) + lane;
        const bool ok = (c >= 0) && (j < n);
        cx[u] = ok ? px[base + j] : __builtin_nan("");
        cy[u] = ok ? py[base + j] : 0.0;
        cz[u] = ok ? pz[base + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const double dx = x - cx[u], dy = y - cy[u], dz = z - cz[u];
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        cnt += __popcll(__ballot(d < r2));                 // NaN never compares below r2
      }
    }
    if (lane == 0 && !handed) flags[base + q] = (cnt > nb) ? 1 : 0;
  }
}

// Phase B': one workgroup per query that is (almost surely) isolated in a long
// segment: 1024 candidates per step, leaves (all waves together) as soon as the
// count is reached.
__global__ __launch_bounds__(256) void k_radius_resolve_long(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const long long *__restrict__ seg_base,
    const int *__restrict__ seg_cnt, const double *__restrict__ radius, int nb,
    uint8_t *__restrict__ flags, const int *__restrict__ queue, long long pool_cap) {
  __shared__ int s_cnt;
  const int nq = queue[0];
  long long nh = queue[1];
  if (nh > pool_cap - nq) nh = pool_cap - nq;             // entries beyond that were resolved inline
  const int lane = lane_id();
  for (long long e = blockIdx.x; e < nh; e += gridDim.x) {
    const int s = queue[2 + 2 * (pool_cap - 1 - e)], q = queue[3 + 2 * (pool_cap - 1 - e)];
    const int n = seg_cnt[s];
    const long long base = seg_base[s];
    const double r = radius[s], r2 = r * r;
    const double x = px[base + q], y = py[base + q], z = pz[base + q];
    if (threadIdx.x == 0) s_cnt = 1;                      // the query itself (d = 0 < r2)
    __syncthreads();
    // outwards from the query: per step 512 list positions above and 512 below it
    constexpr int RU = 4;
    int c = 1;
    const int reach = max(q, n - 1 - q);
    for (int k0 = 0; k0 < reach && c <= nb; k0 += 512) {
      double cx[RU], cy[RU], cz[RU];
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const int off = k0 + 1 + (u >> 1) * 256 + (int)threadIdx.x;     // 1 .. 512 beyond k0
        const int j = (u & 1) ? q - off : q + off;
        const bool ok = (j >= 0) && (j < n);
        cx[u] = ok ? px[base + j] : __builtin_nan("");
        cy[u] = ok ? py[base + j] : 0.0;
        cz[u] = ok ? pz[base + j] : 0.0;
      }
      int h = 0;
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const double dx = x - cx[u], dy = y - cy[u], dz = z - cz[u];
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        h += __popcll(__ballot(d < r2));
      }
      if (lane == 0 && h) atomicAdd(&s_cnt, h);
      __syncthreads();
      c = s_cnt;                                         // the same value for every thread ...
      __syncthreads();                                   // ... because nobody adds before all have read
    }
    if (threadIdx.x == 0) flags[base + q] = (c > nb) ? 1 : 0;
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
// A workgroup of 1024 threads walks BQ_TPB consecutive 1024-query tiles and
// rebuilds the table only when the instance changes (BQ_TPB = 1 measured best:
// more, smaller workgroups balance better than amortising the build).  Instances with more
// LiDAR points than the table holds, or beyond the quantised range, use the
// brute-force tile loop.
constexpr int BT = 1024;                       // queries per tile = threads per workgroup
constexpr int BH_HEADS = 8192;                 // 32 KB of LDS
constexpr int BH_MAX = 4096;                   // nodes: 32 KB of LDS; 12-bit index
constexpr int BQ_TPB = 1;

__device__ __forceinline__ uint32_t bh_hash(uint32_t ix, uint32_t iy, uint32_t iz) {
  uint32_t h = ix * 0x9E3779B1u ^ iy * 0x85EBCA77u ^ iz * 0xC2B2AE3Du;
  h ^= h >> 15;
  return h;
}

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
  int t = blockIdx.x * BQ_TPB;
  if (t >= ntile) return;
  const int t_end = min(t + BQ_TPB, ntile);
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
          for (int uz = z0; uz <= z1 && !found; uz++)
            for (int uy = y0; uy <= y1 && !found; uy++)
              for (int ux = x0; ux <= x1 && !found; ux++) {
                uint32_t node = s_head[bh_hash((uint32_t)ux, (uint32_t)uy, (uint32_t)uz) & mask];
                while (node) {
                  const unsigned long long wv = s_node[node - 1u];
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
                    if (d < T) { found = true; break; }      // <=> sqrt(d) < C, see dfu3d_ballquery_fuse
                  }
                  node = (uint32_t)(wv >> 51);
                }
              }
        }
        flags[bq + q] = found ? 1 : 0;
      }
      continue;
    }
    // brute force over LDS tiles (more LiDAR points than the table holds, or a huge extent)
    hashed_s = -1;                               // the tiles below overwrite the table
    double *sx = (double *)s_node, *sy = sx + PT, *sz = sy + PT;    // 24 KB of the nodes' LDS
    for (int j0 = 0; j0 < na; j0 += PT) {
      const int m = min(PT, na - j0);
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
constexpr int CPE = 4;      // consecutive elements per thread
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

extern "C" int dfu3d_segments_build(
    const uint32_t *a_bits, const double *a_x, const double *a_y, const double *a_z,
    const int32_t *a_n, int32_t a_cap, const uint32_t *b_bits, const double *b_x,
    const double *b_y, const double *b_z, const int32_t *b_n, int32_t b_cap, int32_t V,
    int32_t max_inst, int64_t pool_cap, int64_t *pool_cursor, double *px, double *py,
    double *pz, int64_t *base_a, int32_t *cnt_a, int64_t *base_b, int32_t *cnt_b,
    uint32_t *status, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!a_bits || !a_x || !a_y || !a_z || !a_n || !b_bits || !b_x || !b_y || !b_z || !b_n ||
      !pool_cursor || !px || !py || !pz || !base_a || !cnt_a || !base_b || !cnt_b || !status)
    return DFU3D_EINVAL;
  if (V <= 0 || max_inst <= 0 || a_cap <= 0 || b_cap <= 0 || pool_cap <= 0) return DFU3D_EINVAL;
  if (max_inst > DFU3D_MAX_INST) return DFU3D_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  const int S = V * max_inst;
  hipLaunchKernelGGL(k_seg_count, dim3(V), dim3(1024), 0, st, a_bits, a_n, a_cap, max_inst, cnt_a);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_count, dim3(V), dim3(1024), 0, st, b_bits, b_n, b_cap, max_inst, cnt_b);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_alloc, dim3(1), dim3(1024), 0, st, S, cnt_a, cnt_b,
                     (long long *)base_a, (long long *)base_b, (long long)pool_cap,
                     (long long *)pool_cursor, status);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_write, dim3(V), dim3(SWT), 0, st, a_bits, a_x, a_y, a_z, a_n, a_cap,
                     max_inst, (const long long *)base_a, cnt_a, px, py, pz);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_seg_write, dim3(V), dim3(SWT), 0, st, b_bits, b_x, b_y, b_z, b_n, b_cap,
                     max_inst, (const long long *)base_b, cnt_b, px, py, pz);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_radius_filter(double *px, double *py, double *pz, const int64_t *seg_base,
                                   int32_t *seg_cnt, const double *radius, int32_t nb_points,
                                   int32_t S, int64_t pool_cap, int32_t *tile_off,
                                   uint8_t *flags, int32_t *queue, int32_t phases,
                                   void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!px || !py || !pz || !seg_base || !seg_cnt || !radius || !tile_off || !flags || !queue)
    return DFU3D_EINVAL;
  if (S <= 0 || pool_cap <= 0 || nb_points < 0) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (phases & DFU3D_RF_TILES) {
    if (hipMemsetAsync(queue, 0, 2 * sizeof(int), st) != hipSuccess) return DFU3D_ELAUNCH;
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, S, seg_cnt, tile_off, QT);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_FLAGS) {
    hipLaunchKernelGGL(k_radius_flags, dim3((tile_grid(pool_cap, S) + RF_TPB - 1) / RF_TPB), dim3(QT), 0, st, px, py, pz,
                       (const long long *)seg_base, seg_cnt, radius, nb_points, S, tile_off, flags,
                       queue);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_RESOLVE) {
    hipLaunchKernelGGL(k_radius_resolve, dim3(2048), dim3(256), 0, st, px, py, pz,
                       (const long long *)seg_base, seg_cnt, radius, nb_points, flags, queue,
                       (long long)pool_cap);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_radius_resolve_long, dim3(2048), dim3(256), 0, st, px, py, pz,
                       (const long long *)seg_base, seg_cnt, radius, nb_points, flags, queue,
                       (long long)pool_cap);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_RF_COMPACT) {
    hipLaunchKernelGGL(k_seg_compact, dim3(S), dim3(CPT), 0, st, px, py, pz,
                       (long long *)seg_base, seg_cnt, flags, (const long long *)nullptr,
                       (const int *)nullptr);
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
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, S, cnt_b, tile_off, BT);
  DFU3D_LAUNCH_CHECK();
  const int ball_tiles = (int)((pool_cap + BT - 1) / BT + S);
  hipLaunchKernelGGL(k_ball_flags, dim3((ball_tiles + BQ_TPB - 1) / BQ_TPB), dim3(BT), 0, st, px, py, pz,
                     (const long long *)base_a, cnt_a, (const long long *)base_b, cnt_b, T, C, S,
                     tile_off, flags, masked);
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
