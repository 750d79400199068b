// fit_stage.hip -- per-instance clustering (adaptive range segmentation) and
// L-shape rectangle search + KITTI box assembly.
//
// a13  rectangle_fitting.py:161-191: the reference builds C_i = {j : d_ij <= R_i}
//      and merges intersecting sets until a fixed point.  i is in C_i and j in
//      C_i links i and j, so the result is the set of connected components of
//      the graph {i--j : d_ij <= R_i or d_ij <= R_j}; the merge loop leaves them
//      ordered by their smallest index.  Here: one workgroup per instance, a
//      lock-free union-find whose roots are always the smallest index of their
//      set (atomicMin link), parents / group summaries / group boxes in LDS.
// a14  rectangle_fitting.py:83-159: 89 candidate headings; one wave per heading,
//      lanes stride the cluster's points, three sweeps (extent, mean, variance)
//      with fp64 wave reductions; first strict maximum wins.
// a15  my_loader.py:633-702: box from the rectangle, fp64.
#include "common.hpp"

namespace {
// a register of lane l, l the same in every lane: v_readlane, not a shuffle through the LDS path (ds_bpermute)
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double rl_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

constexpr int CT = 512;            // threads per clustering workgroup
constexpr int GRP = 32;            // points per summary / bounding-box group
constexpr int BLK = 32;            // groups per block (1024 points): second pruning level
constexpr int PAIR_SMALL = 32;     // grid clustering: cell runs up to this length are paired by a single lane
constexpr int GU = 4;              // grid clustering: points per thread and step of a sweep over the instance
constexpr int SMALL_N = 4096;      // segments up to this size: 32-bit parents, 20 KB of LDS
constexpr int LARGE_N = 61440;     // up to this size: 16-bit parents in LDS (120 KB)
constexpr int LARGE_GRP = LARGE_N / GRP;
constexpr int GRID_NC_SMALL = 4096;   // grid path: cells kept in LDS (48 KB) ...
constexpr int GRID_NC_LARGE = 12288;  // ... or 144 KB for wide instances

// ---- parent-array accessors: 32-bit (LDS or global) and 16-bit (LDS only) ----
struct ParI {
  int *p;
  bool global;
  __device__ __forceinline__ int load(int i) const {
    return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void init(int i) const { p[i] = i; }
  __device__ __forceinline__ void flatten(int i, int r) const {
    if (global) atomicMin(p + i, r); else p[i] = r;
  }
  __device__ __forceinline__ int amin(int i, int v) const { return atomicMin(p + i, v); }
  // root shared by the 32 points of group g, or -1
  __device__ __forceinline__ int group_root(int g) const {
    const int r = load(g * 32);
    int diff = 0;
#pragma unroll 8
    for (int k = 1; k < 32; k++) diff |= load(g * 32 + k) ^ r;
    return diff ? -1 : r;
  }
};
struct ParH {
  unsigned short *p;
  __device__ __forceinline__ int load(int i) const {
    return (int)*(volatile unsigned short *)(p + i);
  }
  __device__ __forceinline__ void init(int i) const { p[i] = (unsigned short)i; }
  __device__ __forceinline__ void flatten(int i, int r) const { p[i] = (unsigned short)r; }
  // 16-bit atomic min through a 32-bit compare-and-swap on the containing word
  __device__ __forceinline__ int amin(int i, int v) const {
    unsigned int *w = (unsigned int *)p + (i >> 1);
    const int sh = (i & 1) * 16;
    unsigned int old = *(volatile unsigned int *)w;
    while (true) {
      const unsigned int cur = (old >> sh) & 0xFFFFu;
      if ((int)cur <= v) return (int)cur;
      const unsigned int nw = (old & ~(0xFFFFu << sh)) | ((unsigned int)v << sh);
      const unsigned int seen = atomicCAS(w, old, nw);
      if (seen == old) return (int)cur;
      old = seen;
    }
  }
  __device__ __forceinline__ int group_root(int g) const {     // 32 halfwords = 4 x 16 B
    const uint4 *w = (const uint4 *)(p + g * 32);
    const uint4 a = w[0], b = w[1], c = w[2], d = w[3];
    const unsigned int e = (a.x & 0xFFFFu) * 0x10001u;
    const unsigned int diff = (a.x ^ e) | (a.y ^ e) | (a.z ^ e) | (a.w ^ e) | (b.x ^ e) | (b.y ^ e) |
                              (b.z ^ e) | (b.w ^ e) | (c.x ^ e) | (c.y ^ e) | (c.z ^ e) | (c.w ^ e) |
                              (d.x ^ e) | (d.y ^ e) | (d.z ^ e) | (d.w ^ e);
    return diff ? -1 : (int)(a.x & 0xFFFFu);
  }
};

// find with path halving: every visited node is re-pointed at its grandparent.
// Any ancestor is a valid parent, so the plain store is benign next to the
// atomic links of uf_unite (a lost link is re-established by its retry loop).
template <class P>
__device__ __forceinline__ int uf_find(const P &par, int a) {
  int p = par.load(a);
  while (p != a) {
    const int gp = par.load(p);
    if (gp != p) par.flatten(a, gp);
    a = p;
    p = gp;
  }
  return a;
}
// returns the root of the merged set as seen by this thread
template <class P>
__device__ __forceinline__ int uf_unite(const P &par, int a, int b) {
  while (true) {
    a = uf_find(par, a);
    b = uf_find(par, b);
    if (a == b) return a;
    if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
    const int old = par.amin(a, b);
    if (old == a) return b;
    a = old;                                        // someone re-linked a: retry
  }
}

// One workgroup per instance.  Points are taken in chunks of CT queries (thread
// i owns point c0+tid and looks at every j < i).  Three exact prunes keep the
// O(n^2) pair loop cheap:
//  (1) before a chunk starts all earlier points are flattened (parent = root)
//      and every aligned group of 32 earlier points whose roots agree gets a
//      one-word summary; a query skips a group whose summary equals its own
//      current root;
//  (2) every group carries an outward-rounded bounding box; a group whose box
//      is farther than the largest radius of the instance cannot hold a
//      neighbour and is skipped;
//  (3) inside a group roots are compared before coordinates are touched, and
//      the squared distance s decides alone when s <= S_LO (sqrt(s) <= R0 <= R)
//      or s > S_HI (sqrt(s) > every R); only the thin band in between evaluates
//      the reference's sqrt expression (rectangle_fitting.py:167-170).
// None of the prunes changes a decision, so the labels equal the reference's.

template <class P, int NGRP>
__device__ __forceinline__ void cluster_body(const P par, int *s_summ, float4 *s_box,
                                             int *s_summ2, float4 *s_box2,
                                             double *s_red, const double *X, const double *Y,
                                             int n, double R0, double Rd, const int *perm,
                                             int *tmp, int *glabel) {
  constexpr int NBLK = (NGRP + BLK - 1) / BLK;
  const int ngrp_all = min((n + GRP - 1) / GRP, NGRP);
  double r2max = 0.0;
  for (int i = threadIdx.x; i < n; i += CT) {
    par.init(i);
    const double x = X[i], y = Y[i];
    r2max = fmax(r2max, x * x + y * y);
  }
  for (int g = threadIdx.x; g < ngrp_all; g += CT) {
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    const int je = min(g * GRP + GRP, n);
    for (int j = g * GRP; j < je; j++) {
      const double x = X[j], y = Y[j];
      x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
    }
    s_box[g] = make_float4(__double2float_rd(x0), __double2float_ru(x1),
                           __double2float_rd(y0), __double2float_ru(y1));
  }
  r2max = wave_max_d(r2max);
  if (lane_id() == 0) s_red[threadIdx.x >> 6] = r2max;
  __syncthreads();
  r2max = s_red[0];
#pragma unroll
  for (int w = 1; w < CT / 64; w++) r2max = fmax(r2max, s_red[w]);
  for (int B = threadIdx.x; B * BLK < ngrp_all; B += CT) {     // boxes of 1024-point blocks
    float4 b = s_box[B * BLK];
    const int ge = min(B * BLK + BLK, ngrp_all);
    for (int g = B * BLK + 1; g < ge; g++) {
      const float4 q = s_box[g];
      b.x = fminf(b.x, q.x); b.y = fmaxf(b.y, q.y); b.z = fminf(b.z, q.z); b.w = fmaxf(b.w, q.w);
    }
    s_box2[B] = b;
  }
  const double Rmax = (R0 + Rd * sqrt(r2max)) * (1.0 + 1e-9) + 1e-9;
  const double S_HI = Rmax * Rmax * (1.0 + 1e-9);      // s > S_HI  => sqrt(s) > every R_i
  const double S_LO = R0 * R0 * (1.0 - 1e-12);         // s <= S_LO => sqrt(s) <= R0 <= R_i

  for (int c0 = 0; c0 < n; c0 += CT) {
    // flatten + summarise everything before this chunk
    for (int k = threadIdx.x; k < c0; k += CT) par.flatten(k, uf_find(par, k));
    __syncthreads();
    const int ngrp = min(c0 / GRP, NGRP);
    for (int g = threadIdx.x; g < ngrp; g += CT) s_summ[g] = par.group_root(g);
    __syncthreads();
    for (int B = threadIdx.x; B < ngrp / BLK; B += CT) {        // fully flattened blocks only
      const int r = s_summ[B * BLK];
      int diff = 0;
#pragma unroll 8
      for (int k = 1; k < BLK; k++) diff |= s_summ[B * BLK + k] ^ r;
      s_summ2[B] = (r >= 0 && !diff) ? r : -1;
    }
    __syncthreads();
    const int nblk_sum = ngrp / BLK;
    const int i = c0 + threadIdx.x;
    const bool act = i < n;
    const int lane = lane_id();
    double xi = 0.0, yi = 0.0, Ri = 0.0;
    if (act) {
      xi = X[i];
      yi = Y[i];
      Ri = R0 + Rd * sqrt(xi * xi + yi * yi);                   // rectangle_fitting.py:167
    }
    int ri = act ? i : -2;
    // groups holding some j < i for any lane of this wave (wave-uniform bound)
    const int wave_hi = min(n, c0 + (int)(threadIdx.x | 63) + 1);
    const int gend = (wave_hi - 1 + GRP - 1) / GRP;
    // Order of the walk: first the flattened points (j < c0) backwards -- nearest
    // in input order first, so a point usually meets the (already merged) set it
    // belongs to within a few groups and can then skip whole 1024-point blocks --
    // and afterwards the points of its own chunk.
    const int gc = c0 / GRP;                                    // first group of this chunk
    auto visit_group = [&](int g, bool need_in) {
      const int jg = g * GRP;
      bool need = need_in && (jg < i);
      if (need && g < NGRP) {
        if (jg + GRP <= c0 && s_summ[g] == ri) need = false;    // (1) whole group is mine
        else {
          const float4 b = s_box[g];                            // (2) box too far
          const double ddx = fmax(fmax((double)b.x - xi, xi - (double)b.y), 0.0);
          const double ddy = fmax(fmax((double)b.z - yi, yi - (double)b.w), 0.0);
          if (ddx * ddx + ddy * ddy > S_HI) need = false;
        }
      }
      if (!__any(need)) return;                                 // wave-uniform
      // the wave fetches the group's 32 points once (one coalesced load each)
      const int jl = jg + (lane & (GRP - 1));
      const bool in = jl < n;
      const int pg = in ? par.load(jl) : -3;
      // cheap exit: every query of the wave that still needs this group sits in
      // one set and all 32 points already belong to it (own-chunk groups mostly)
      const int r_first = rl_i(ri, max(__ffsll((unsigned long long)__ballot(need)) - 1, 0));
      if (__all(!need || ri == r_first) && __all(!in || pg == r_first)) {
        return;
      }
      const double xg = in ? X[jl] : 0.0, yg = in ? Y[jl] : 0.0;
      for (int k = GRP - 1; k >= 0; k--) {
        if ((k & 7) == 7 && k != GRP - 1) {
          // every 8 points: once all queries that need this group have merged into
          // the set that owns all of its points, the rest of the group is moot
          const int rf = rl_i(ri, max(__ffsll((unsigned long long)__ballot(need)) - 1, 0));
          if (__all(!need || ri == rf) && __all(!in || pg == rf)) break;
        }
        const int pj = rl_i(pg, k);
        const double xj = rl_d(xg, k), yj = rl_d(yg, k);
        if (need && (jg + k < i) && pj != ri) {                 // (3) not in my set (yet)
          const double dx = xi - xj, dy = yi - yj;
          const double sq = dx * dx + dy * dy;
          bool adj;
          if (sq <= S_LO) adj = true;
          else if (sq > S_HI) adj = false;
          else {
            const double d = sqrt(sq);                          // rectangle_fitting.py:169
            adj = (d <= Ri) || (d <= R0 + Rd * sqrt(xj * xj + yj * yj));
          }
          if (adj) ri = uf_unite(par, ri, jg + k);
        }
      }
    };
    for (int B = (gc + BLK - 1) / BLK - 1; B >= 0; B--) {
      const int g_lo = B * BLK, g_hi = min(g_lo + BLK, gc);
      bool need2 = act;
      if (need2 && B < NBLK) {
        if (B < nblk_sum && s_summ2[B] == ri) need2 = false;    // (1) whole block is mine
        else {
          const float4 b = s_box2[B];                           // (2) block too far
          const double ddx = fmax(fmax((double)b.x - xi, xi - (double)b.y), 0.0);
          const double ddy = fmax(fmax((double)b.z - yi, yi - (double)b.w), 0.0);
          if (ddx * ddx + ddy * ddy > S_HI) need2 = false;
        }
      }
      if (!__any(need2)) continue;                              // wave-uniform
      if (g_lo >= NGRP) {                                       // beyond the summarised range
        for (int g = g_hi - 1; g >= g_lo; g--) visit_group(g, need2);
        continue;
      }
      // per lane: the groups of this block that are neither wholly mine nor too far
      unsigned int cand = 0u;
      if (need2) {
        unsigned int mm = 0u;
        const int ng = g_hi - g_lo;
#pragma unroll 8
        for (int q = 0; q < BLK; q++)
          mm |= (q < ng && s_summ[g_lo + q] != ri) ? (1u << q) : 0u;
        while (mm) {
          const int q = __ffs((int)mm) - 1;
          mm &= mm - 1u;
          const float4 b = s_box[g_lo + q];
          const double ddx = fmax(fmax((double)b.x - xi, xi - (double)b.y), 0.0);
          const double ddy = fmax(fmax((double)b.z - yi, yi - (double)b.w), 0.0);
          if (!(ddx * ddx + ddy * ddy > S_HI)) cand |= 1u << q;
        }
      }
      unsigned int U = cand;                                    // union over the wave
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) U |= (unsigned int)__shfl_xor((int)U, m, 64);
      while (U) {                                               // wave-uniform, backwards
        const int q = 31 - __clz((int)U);
        U &= ~(1u << q);
        visit_group(g_lo + q, need2 && ((cand >> q) & 1u));
      }
    }
    __syncthreads();     // every point of the chunk has met the earlier points: own-chunk groups mostly exit early
    for (int g = gc; g < gend; g++) visit_group(g, act);
    __syncthreads();
  }
  // Back to the reference's labelling: smallest ORIGINAL index of each cluster.
  for (int i = threadIdx.x; i < n; i += CT) {
    par.flatten(i, uf_find(par, i));
    tmp[i] = 0x7FFFFFFF;
  }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += CT) {       // wave-uniform trip count
    const int i = i0 + threadIdx.x;
    const int r = (i < n) ? par.load(i) : -1;
    const int pm = (i < n) ? perm[i] : 0x7FFFFFFF;
    // one atomic per wave when the whole wave sits in one cluster (the common case)
    const int r0 = __shfl(r, __ffsll((unsigned long long)__ballot(r >= 0)) - 1, 64);
    if (__all(r < 0 || r == r0)) {
      int m = pm;
#pragma unroll
      for (int k = 32; k >= 1; k >>= 1) m = min(m, __shfl_xor(m, k, 64));
      if (lane_id() == 0 && r0 >= 0) atomicMin(&tmp[r0], m);
    } else if (r >= 0) {
      atomicMin(&tmp[r], pm);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += CT)
    glabel[perm[i]] = __hip_atomic_load(&tmp[par.load(i)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Counting sort of one instance's points by spatial cell (row-major cells of
// side >= 1.5 m over the instance's bounding box).  In sorted order 32
// consecutive points are spatial neighbours, so the group summaries and boxes
// of cluster_body prune almost everything even when clusters interleave in
// input order (salt outliers along the same camera rays do exactly that).
// hist: LDS ints (aliases the parent array, dead until the sort is done).
template <int NCELL>
__device__ __forceinline__ void cell_sort(const double *X, const double *Y, int n, int *hist,
                                          double *s_red, int *s_w, double *SX, double *SY,
                                          int *perm) {
  double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
  for (int i = threadIdx.x; i < n; i += CT) {
    const double x = X[i], y = Y[i];
    x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
  }
  x0 = wave_min_d(x0); x1 = wave_max_d(x1); y0 = wave_min_d(y0); y1 = wave_max_d(y1);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0) { s_red[4 * w] = x0; s_red[4 * w + 1] = x1; s_red[4 * w + 2] = y0; s_red[4 * w + 3] = y1; }
  __syncthreads();
  for (int k = 0; k < CT / 64; k++) {
    x0 = fmin(x0, s_red[4 * k]); x1 = fmax(x1, s_red[4 * k + 1]);
    y0 = fmin(y0, s_red[4 * k + 2]); y1 = fmax(y1, s_red[4 * k + 3]);
  }
  __syncthreads();
  double ex = fmax(x1 - x0, 1e-6), ey = fmax(y1 - y0, 1e-6);
  if (!(ex < 1e12)) ex = 1e12;
  if (!(ey < 1e12)) ey = 1e12;
  double g = 1.5;
  if ((ex / g + 1.0) * (ey / g + 1.0) > (double)NCELL) g = sqrt(ex * ey / (double)NCELL) * 1.05 + 1e-9;
  int nx = (int)(ex / g) + 1, ny = (int)(ey / g) + 1;
  while ((long long)nx * ny > NCELL) { g *= 1.1; nx = (int)(ex / g) + 1; ny = (int)(ey / g) + 1; }
  const int ncell = nx * ny;
  const double inv = 1.0 / g;
  for (int c = threadIdx.x; c < ncell; c += CT) hist[c] = 0;
  __syncthreads();
  auto cell_of = [&](double x, double y) {
    const int cx = (int)fmin(fmax((x - x0) * inv, 0.0), (double)(nx - 1));
    const int cy = (int)fmin(fmax((y - y0) * inv, 0.0), (double)(ny - 1));
    return cy * nx + cx;
  };
  for (int i = threadIdx.x; i < n; i += CT) atomicAdd(&hist[cell_of(X[i], Y[i])], 1);
  __syncthreads();
  // exclusive scan of the cell counts (each thread owns a contiguous run of cells)
  const int per = (ncell + CT - 1) / CT;
  const int c_lo = min(threadIdx.x * per, ncell), c_hi = min(c_lo + per, ncell);
  int mine = 0;
  for (int c = c_lo; c < c_hi; c++) mine += hist[c];
  int tot;
  int run = block_excl_scan<CT / 64>(mine, s_w, tot);
  for (int c = c_lo; c < c_hi; c++) { const int h = hist[c]; hist[c] = run; run += h; }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += CT) {
    const double x = X[i], y = Y[i];
    const int pos = atomicAdd(&hist[cell_of(x, y)], 1);
    SX[pos] = x;
    SY[pos] = y;
    perm[pos] = i;
  }
  __syncthreads();
}

// ============================================================================
// Grid formulation of a13 (the fast path).  With cell side g = R_max/2 (+ margin)
//   * g*sqrt(2) < R0, so two points of one cell are ALWAYS adjacent (d <= R0 <= R_i):
//     a cell is one set from the start and union-find runs over cells, not points;
//   * cells more than 2 apart are farther than R_max: only the 5x5 neighbourhood
//     matters, 12 offsets per cell counting every unordered pair once.
// Two neighbouring cells are linked as soon as ONE adjacent point pair is found
// (the reference's predicate, evaluated exactly); pairs of cells that already
// share a root are never tested.  Touching cells go first so that dense
// instances collapse into one set before the gap-1 offsets are looked at.
// Labels = smallest original index of the cell-component.  Instances whose
// bounding box needs more than NC cells fall through to the point-level kernels.
template <int NC>
struct GridGeom { double x0, y0, inv, g, Rmax; int nx, ny, ncell; bool ok; };

template <int NC, int TCT>
__device__ __forceinline__ GridGeom<NC> grid_geometry(const double *X, const double *Y, int n,
                                                      double R0, double Rd, double *s_red) {
  double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY, r2 = 0.0;
  // GU points per thread and step, all loads requested before the first is used: the instance is walked by ONE
  // workgroup, and a step per memory round trip made the largest instance (50 000 points) the tail of the kernel
  for (int i0 = threadIdx.x; i0 < n; i0 += GU * TCT) {
    double xs[GU], ys[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      xs[u] = (i < n) ? X[i] : NAN; ys[u] = (i < n) ? Y[i] : NAN;
    }
#pragma unroll
    for (int u = 0; u < GU; u++) {
      if (i0 + u * TCT >= n) break;
      const double x = xs[u], y = ys[u];
      x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
      r2 = fmax(r2, x * x + y * y);
    }
  }
  x0 = wave_min_d(x0); x1 = wave_max_d(x1); y0 = wave_min_d(y0); y1 = wave_max_d(y1);
  r2 = wave_max_d(r2);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if (lane_id() == 0) { s_red[5 * w] = x0; s_red[5 * w + 1] = x1; s_red[5 * w + 2] = y0; s_red[5 * w + 3] = y1; s_red[5 * w + 4] = r2; }
  __syncthreads();
  for (int k = 0; k < TCT / 64; k++) {
    x0 = fmin(x0, s_red[5 * k]); x1 = fmax(x1, s_red[5 * k + 1]);
    y0 = fmin(y0, s_red[5 * k + 2]); y1 = fmax(y1, s_red[5 * k + 3]); r2 = fmax(r2, s_red[5 * k + 4]);
  }
  __syncthreads();
  GridGeom<NC> G;
  G.Rmax = (R0 + Rd * sqrt(r2)) * (1.0 + 1e-9) + 1e-12;
  G.g = 0.5 * G.Rmax * (1.0 + 1e-6) + 1e-12;          // cells 3 apart: gap 2g > Rmax
  G.x0 = x0; G.y0 = y0; G.inv = 1.0 / G.g;
  const double ex = x1 - x0, ey = y1 - y0;
  G.ok = (G.g * 1.4142135623730951 <= R0 * 0.999) && (ex >= 0.0) && (ey >= 0.0) &&
         (ex < 1e9) && (ey < 1e9);
  double fx = floor(ex * G.inv) + 1.0, fy = floor(ey * G.inv) + 1.0;
  if (!(G.ok && fx * fy <= (double)NC)) { G.ok = false; fx = 1.0; fy = 1.0; }
  G.nx = (int)fx; G.ny = (int)fy; G.ncell = G.nx * G.ny;
  return G;
}

// cell of a point, with its column and row (a caller that needs them must not divide the index again: a 32-bit
// division is ~25 instructions, the multiplies among them at a quarter of the rate).  nx, ny <= NC < 2^24.
template <int NC>
__device__ __forceinline__ int grid_cell(const GridGeom<NC> &G, double x, double y, int &cx, int &cy) {
  cx = (int)fmin(fmax((x - G.x0) * G.inv, 0.0), (double)(G.nx - 1));
  cy = (int)fmin(fmax((y - G.y0) * G.inv, 0.0), (double)(G.ny - 1));
  return (int)__umul24((uint32_t)cy, (uint32_t)G.nx) + cx;
}
template <int NC>
__device__ __forceinline__ int grid_cell(const GridGeom<NC> &G, double x, double y) {
  int cx, cy;
  return grid_cell(G, x, y, cx, cy);
}

__device__ __forceinline__ int cell_find(int *par_, int a) {
  volatile int *par = par_;            // other waves link cells concurrently
  int p = par[a];
  while (p != a) {
    const int gp = par[p];
    if (gp != p) par[a] = gp;
    a = p;
    p = gp;
  }
  return a;
}
__device__ __forceinline__ void cell_unite(int *par, int a, int b) {
  while (true) {
    a = cell_find(par, a);
    b = cell_find(par, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(par + a, b);
    if (old == a) return;
    a = old;
  }
}

constexpr int CL_CLASSES = 3, CL_TOP = 8192;   // size classes of the launch order (see the kernel)
// -1: instance not eligible for this variant (NC_MIN < ncell <= NC handled here)
// Launch bounds: the 48 KB variant is cut for 64 registers (eight waves per SIMD = TWO 1024-thread workgroups per compute
// unit; 74 registers meant one): the kernel is a chain of short sweeps between barriers, a second workgroup fills the
// waits -- 617 -> 501 us per 384 views (rocprofv3), with 44 B of scratch in the pairing loops.  (512-thread workgroups for the
// instances up to 2048 / 4096 points, three per compute unit: no different, tools/ab_builds.py.)
template <int NC, int NC_MIN, int TCT>
__global__ __launch_bounds__(TCT, (NC <= GRID_NC_SMALL ? 8 : 4)) void k_range_cluster_grid(
    const double *__restrict__ px, const double *__restrict__ py,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, double R0,
    double Rd, int *__restrict__ label, double *__restrict__ sx, double *__restrict__ sy,
    int *__restrict__ perm_all, long long pool_cap, int n_lo, int n_hi) {
  __shared__ int s_end[NC];          // cell -> end of its run in sorted order
  __shared__ int s_par[NC];          // union-find over cells (volatile use through pointers)
  __shared__ int s_min[NC];          // smallest original index per root cell
  __shared__ double s_red[5 * (TCT / 64)];
  __shared__ int s_w[TCT / 64];
  // Longest first: the grid is CL_CLASSES times the segments, and workgroup b takes segment b % S only if its size falls into
  // class b / S (above CL_TOP | above CL_TOP / 2 | the rest).  Workgroups start in the order of their index, so every long
  // instance is under way before the first short one starts -- an instance of 50 000 points runs for half of what this kernel
  // used to take, and in index order it could be among the last to start.  0.524 -> 0.448 ms (tools/ab_builds.py; classes above
  // 12 288 / 4 096: 0.450, above 24 576 / 8 192: 0.475; four / five / six classes from 16 384 / 16 384 / 32 768 down: 0.457 /
  // 0.461 / 0.478 -- every class costs a grid of workgroups that start only to leave).
  constexpr int NCLS = CL_CLASSES;
  const int S_ = (int)(gridDim.x / (unsigned)NCLS);
  const int cls = (int)(blockIdx.x / (unsigned)S_);
  const int s = (int)(blockIdx.x - (unsigned)cls * (unsigned)S_);
  const int n = seg_cnt[s];
  if (n <= n_lo || n > n_hi) return;                   // (n_lo >= 0) another launch of this kernel owns the instance
  {                                                    // class 0: above TOP, class k: above TOP >> k, the last: the rest
    int mine = NCLS - 1;
#pragma unroll
    for (int k = NCLS - 2; k >= 0; k--) if (n > (CL_TOP >> k)) mine = k;
    if (mine != cls) return;
  }
  const long long base = seg_base[s];
  const double *X = px + base, *Y = py + base;
  double *SX = sx + base, *SY = sy + base;
  int *perm = perm_all + base;
  // "handled" mark of the segment, in the first word of its third scratch plane (only the point-level
  // fallback uses that plane): the first variant writes it, the later kernels of the call read it and
  // leave at once instead of measuring the instance again
  int *mark = perm_all + 2 * pool_cap + base;
  if (NC_MIN > 0 && *mark) return;                     // the smaller variant did it
  const GridGeom<NC> G = grid_geometry<NC, TCT>(X, Y, n, R0, Rd, s_red);
  if (threadIdx.x == 0 && (NC_MIN == 0 || G.ok)) *mark = G.ok ? 1 : 0;
  if (!G.ok) return;                                   // too wide: a larger variant / the point-level kernels
  const int ncell = G.ncell;
  // ---- counting sort by cell ------------------------------------------------
  for (int c = threadIdx.x; c < ncell; c += TCT) { s_end[c] = 0; s_par[c] = c; s_min[c] = 0x7FFFFFFF; }
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < n; i0 += GU * TCT) {
    double xs[GU], ys[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      xs[u] = (i < n) ? X[i] : 0.0; ys[u] = (i < n) ? Y[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < GU; u++)
      if (i0 + u * TCT < n) atomicAdd(&s_end[grid_cell(G, xs[u], ys[u])], 1);
  }
  __syncthreads();
  {
    const int per = (ncell + TCT - 1) / TCT;
    const int c_lo = min((int)threadIdx.x * per, ncell), c_hi = min(c_lo + per, ncell);
    int mine = 0;
    for (int c = c_lo; c < c_hi; c++) mine += s_end[c];
    int tot;
    int run = block_excl_scan<TCT / 64>(mine, s_w, tot);
    for (int c = c_lo; c < c_hi; c++) { const int h = s_end[c]; s_end[c] = run; run += h; }
  }
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < n; i0 += GU * TCT) {
    double xs[GU], ys[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      xs[u] = (i < n) ? X[i] : 0.0; ys[u] = (i < n) ? Y[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      if (i >= n) break;
      const int pos = atomicAdd(&s_end[grid_cell(G, xs[u], ys[u])], 1);      // s_end[c] ends as the run's end
      SX[pos] = xs[u];
      SY[pos] = ys[u];
      perm[pos] = i;
    }
  }
  __syncthreads();
  // ---- link neighbouring cells ----------------------------------------------
  const double S_HI = G.Rmax * G.Rmax * (1.0 + 1e-9);
  const double S_LO = R0 * R0 * (1.0 - 1e-12);
  const int lane = lane_id();
  // Tight box of every cell's points, in 1/256 of the cell and rounded outward (one word per cell, kept in s_min
  // until the labels need it): two cells whose boxes are farther apart than R_max cannot hold an adjacent pair and
  // are never paired point by point -- those were the expensive pairs, every one of their point pairs had to fail.
  int *s_box = s_min;                // x0 | x1 << 8 | y0 << 16 | y1 << 24; lower bytes: floor(256 u) (the box starts at
                                     // byte/256 or before), upper bytes: ceil(256 u) - 1 (it ends at (byte+1)/256 or after)
  for (int c = threadIdx.x; c < ncell; c += TCT) s_box[c] = (int)0x00FF00FFu;   // empty box (never looked at for an empty cell)
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < n; i0 += GU * TCT) {
    double xs[GU], ys[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      xs[u] = (i < n) ? SX[i] : 0.0; ys[u] = (i < n) ? SY[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < GU; u++) {
      if (i0 + u * TCT >= n) break;
      int cx, cy;
      const int c = grid_cell(G, xs[u], ys[u], cx, cy);
      const double fu = (xs[u] - G.x0) * G.inv - (double)cx, fv = (ys[u] - G.y0) * G.inv - (double)cy;
      const unsigned ulo = (unsigned)(int)fmin(fmax(floor(fu * 256.0), 0.0), 255.0);
      const unsigned uhi = (unsigned)(int)fmin(fmax(ceil(fu * 256.0) - 1.0, 0.0), 255.0);
      const unsigned vlo = (unsigned)(int)fmin(fmax(floor(fv * 256.0), 0.0), 255.0);
      const unsigned vhi = (unsigned)(int)fmin(fmax(ceil(fv * 256.0) - 1.0, 0.0), 255.0);
      // widen the cell's box to this point; after the first few points of a cell hardly any point still does
      unsigned old = (unsigned)__hip_atomic_load(&s_box[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (true) {
        const unsigned nw = min(old & 255u, ulo) | (max((old >> 8) & 255u, uhi) << 8) |
                            (min((old >> 16) & 255u, vlo) << 16) | (max(old >> 24, vhi) << 24);
        if (nw == old) break;
        const unsigned seen = (unsigned)atomicCAS(&s_box[c], (int)old, (int)nw);
        if (seen == old) break;
        old = seen;
      }
    }
  }
  __syncthreads();
  // gap between the boxes of cells (cx, cy) and (qx, qy), in cells, minus two quanta of safety; squared against R_max
  const double box_lim = S_HI * (1.0 + 1e-6) * (G.inv * G.inv);
  auto boxes_apart = [&](int c, int cx, int cy, int nbc, int qx, int qy) {
    const unsigned a = (unsigned)s_box[c], b = (unsigned)s_box[nbc];
    const int ax0 = cx * 256 + (int)(a & 255u), ax1 = cx * 256 + (int)((a >> 8) & 255u) + 1;
    const int ay0 = cy * 256 + (int)((a >> 16) & 255u), ay1 = cy * 256 + (int)(a >> 24) + 1;
    const int bx0 = qx * 256 + (int)(b & 255u), bx1 = qx * 256 + (int)((b >> 8) & 255u) + 1;
    const int by0 = qy * 256 + (int)((b >> 16) & 255u), by1 = qy * 256 + (int)(b >> 24) + 1;
    const int gx = max(max(bx0 - ax1, ax0 - bx1) - 2, 0), gy = max(max(by0 - ay1, ay0 - by1) - 2, 0);
    const double g2 = (double)(gx * gx + gy * gy) * (1.0 / 65536.0);
    return g2 > box_lim;
  };
  // A point farther than R_max from the (outward rounded) box of the other cell has no neighbour there: only the
  // points of each cell that pass this test are paired.  Two dense groups separated by a little more than R --
  // where every pair used to be evaluated only to fail -- leave a thin strip on either side, usually nothing.
  struct CellBox { double x0, x1, y0, y1; };
  auto cell_box = [&](int c) {
    const unsigned a = (unsigned)s_box[c];
    const int cx = c % G.nx, cy = c / G.nx;
    CellBox B;
    B.x0 = G.x0 + ((double)cx + (double)((int)(a & 255u) - 1) * (1.0 / 256.0)) * G.g;       // one quantum of safety
    B.x1 = G.x0 + ((double)cx + (double)((int)((a >> 8) & 255u) + 2) * (1.0 / 256.0)) * G.g;
    B.y0 = G.y0 + ((double)cy + (double)((int)((a >> 16) & 255u) - 1) * (1.0 / 256.0)) * G.g;
    B.y1 = G.y0 + ((double)cy + (double)((int)(a >> 24) + 2) * (1.0 / 256.0)) * G.g;
    return B;
  };
  const double reach2 = S_HI * (1.0 + 1e-6);
  auto near_box = [&](const CellBox &B, double x, double y) {
    const double dx = fmax(fmax(B.x0 - x, x - B.x1), 0.0), dy = fmax(fmax(B.y0 - y, y - B.y1), 0.0);
    return dx * dx + dy * dy <= reach2;
  };
  // half neighbourhood: 4 touching offsets, then the 8 with a one-cell gap
  const int ODX[12] = {1, -1, 0, 1, 2, -2, -1, 0, 1, 2, -2, 2};
  const int ODY[12] = {0, 1, 1, 1, 0, 2, 2, 2, 2, 2, 1, 1};
  for (int round = 0; round < 2; round++) {
    const int o_lo = round ? 4 : 0, n_off = round ? 8 : 4;
    const int items = ncell * n_off;
    const int iters = (items + TCT - 1) / TCT;
    for (int it = 0; it < iters; it++) {                 // uniform trip count per block
      const int idx = it * TCT + threadIdx.x;
      int c = 0, nb = 0;
      bool need = false;
      if (idx < items) {
        c = idx / n_off;
        const int o = o_lo + (idx - c * n_off);
        const int cx = c % G.nx, cy = c / G.nx;
        const int qx = cx + ODX[o], qy = cy + ODY[o];
        if (qx >= 0 && qx < G.nx && qy < G.ny) {
          nb = qy * G.nx + qx;
          const int a0 = c ? s_end[c - 1] : 0, b0 = s_end[nb - 1];     // nb > c >= 0
          need = (s_end[c] > a0) && (s_end[nb] > b0) && !boxes_apart(c, cx, cy, nb, qx, qy) &&
                 (cell_find(s_par, c) != cell_find(s_par, nb));
        }
      }
      // Two short runs (the usual case: a cell holds a handful of points): the lane tests its pair by itself, at
      // most PAIR_SMALL^2 distance evaluations with an exit at the first adjacent pair -- 64 open pairs per wave
      // in flight instead of one (the wave-cooperative path below costs three dependent memory round trips and
      // eight fp64 wave reductions per pair, and one instance has thousands of open pairs).
      if (need) {
        const int a0 = c ? s_end[c - 1] : 0, a1 = s_end[c];
        const int b0 = s_end[nb - 1], b1 = s_end[nb];
        if (a1 - a0 <= PAIR_SMALL && b1 - b0 <= PAIR_SMALL) {
          need = false;
          bool found = false;
          const CellBox BB = cell_box(nb);
          for (int i = a0; i < a1 && !found; i++) {
            const double xa = SX[i], ya = SY[i];
            if (!near_box(BB, xa, ya)) continue;
            const double Ra = R0 + Rd * sqrt(xa * xa + ya * ya);
            for (int j0 = b0; j0 < b1 && !found; j0 += 4) {                // four points of the other run per round trip
              double xb4[4], yb4[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int j = min(j0 + u, b1 - 1);
                xb4[u] = SX[j]; yb4[u] = SY[j];
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if (j0 + u >= b1) break;
                const double xb = xb4[u], yb = yb4[u];
                const double dx = xa - xb, dy = ya - yb;
                const double sq = dx * dx + dy * dy;
                if (sq > S_HI) continue;
                if (sq <= S_LO) { found = true; break; }
                const double d = sqrt(sq);                                 // rectangle_fitting.py:169
                if (d <= Ra || d <= R0 + Rd * sqrt(xb * xb + yb * yb)) { found = true; break; }
              }
            }
          }
          if (found) cell_unite(s_par, c, nb);
        }
      }
      // the wave serves its lanes' remaining open pairs one at a time, all 64 lanes on the point pairs
      unsigned long long todo = __ballot(need);
      while (todo) {
        const int src = __ffsll(todo) - 1;
        todo &= todo - 1ull;
        const int cc = rl_i(c, src), nn = rl_i(nb, src);
        if (cell_find(s_par, cc) == cell_find(s_par, nn)) continue;   // merged meanwhile (uniform)
        const int a0 = cc ? s_end[cc - 1] : 0, a1 = s_end[cc];
        const int b0 = s_end[nn - 1], b1 = s_end[nn];
        // (the cells' boxes have been compared already: `need`)
        // 64 x 64 tiles of point pairs: each lane keeps one point of the second
        // run, the first run's points are broadcast lane by lane -- no memory traffic
        // inside a tile; leave at the first adjacent pair
        bool found = false;
        const CellBox BA = cell_box(cc), BB = cell_box(nn);
        for (int b_lo = b0; b_lo < b1 && !found; b_lo += 64) {
          const int ib = b_lo + lane;
          const double xb = (ib < b1) ? SX[ib] : 0.0, yb = (ib < b1) ? SY[ib] : 0.0;
          const bool vb = (ib < b1) && near_box(BA, xb, yb);
          if (!__any(vb)) continue;                                      // (uniform)
          for (int a_lo = a0; a_lo < a1 && !found; a_lo += 64) {
            const int ia = a_lo + lane;
            const double xal = (ia < a1) ? SX[ia] : 0.0, yal = (ia < a1) ? SY[ia] : 0.0;
            unsigned long long am = __ballot((ia < a1) && near_box(BB, xal, yal));
            bool adj = false;
            int steps = 0;
            while (am) {
              const int k = __ffsll((long long)am) - 1;
              am &= am - 1ull;
              const double xa = rl_d(xal, k), ya = rl_d(yal, k);
              const double dx = xa - xb, dy = ya - yb;
              const double sq = dx * dx + dy * dy;
              if (vb && sq <= S_HI) {
                if (sq <= S_LO) adj = true;
                else {
                  const double d = sqrt(sq);                               // rectangle_fitting.py:169
                  adj = adj || (d <= R0 + Rd * sqrt(xa * xa + ya * ya)) ||
                        (d <= R0 + Rd * sqrt(xb * xb + yb * yb));
                }
              }
              if ((++steps & 15) == 0 && __any(adj)) break;
            }
            found = __any(adj);
          }
        }
        if (found && lane == 0) cell_unite(s_par, cc, nn);
      }
    }
    __syncthreads();
  }
  // ---- labels: smallest original index of the component ---------------------
  for (int c = threadIdx.x; c < ncell; c += TCT) s_min[c] = 0x7FFFFFFF;        // (held the cell boxes until here)
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < n; i0 += GU * TCT) {
    double xs[GU], ys[GU];
    int pm[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      xs[u] = (i < n) ? SX[i] : 0.0; ys[u] = (i < n) ? SY[i] : 0.0; pm[u] = (i < n) ? perm[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < GU; u++)
      if (i0 + u * TCT < n) atomicMin(&s_min[cell_find(s_par, grid_cell(G, xs[u], ys[u]))], pm[u]);
  }
  __syncthreads();
  int *glabel = label + base;
  for (int i0 = threadIdx.x; i0 < n; i0 += GU * TCT) {
    double xs[GU], ys[GU];
    int pm[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = i0 + u * TCT;
      xs[u] = (i < n) ? SX[i] : 0.0; ys[u] = (i < n) ? SY[i] : 0.0; pm[u] = (i < n) ? perm[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < GU; u++)
      if (i0 + u * TCT < n) glabel[pm[u]] = s_min[cell_find(s_par, grid_cell(G, xs[u], ys[u]))];
  }
}

__global__ __launch_bounds__(CT) void k_range_cluster_small(
    const double *__restrict__ px, const double *__restrict__ py,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, double R0,
    double Rd, int *__restrict__ label, double *__restrict__ sx, double *__restrict__ sy,
    int *__restrict__ si, long long pool_cap) {
  __shared__ __attribute__((aligned(16))) int s_parent[SMALL_N];
  __shared__ int s_summ[SMALL_N / GRP];
  __shared__ float4 s_box[SMALL_N / GRP];
  __shared__ int s_summ2[SMALL_N / GRP / BLK];
  __shared__ float4 s_box2[SMALL_N / GRP / BLK];
  __shared__ double s_red[4 * (CT / 64)];
  __shared__ int s_w[CT / 64];
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  if (n == 0 || n > SMALL_N) return;
  const long long base = seg_base[s];
  if (si[2 * pool_cap + base]) return;                 // a grid variant did it (mark written there)
  cell_sort<SMALL_N>(px + base, py + base, n, s_parent, s_red, s_w, sx + base, sy + base, si + base);
  ParI par{s_parent, false};
  cluster_body<ParI, SMALL_N / GRP>(par, s_summ, s_box, s_summ2, s_box2, s_red, sx + base, sy + base,
                                    n, R0, Rd, si + base, si + pool_cap + base, label + base);
}

__global__ __launch_bounds__(CT) void k_range_cluster_large(
    const double *__restrict__ px, const double *__restrict__ py,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt, double R0,
    double Rd, int *__restrict__ label, double *__restrict__ sx, double *__restrict__ sy,
    int *__restrict__ si, long long pool_cap) {
  __shared__ __attribute__((aligned(16))) unsigned short s_parent[LARGE_N];
  __shared__ int s_summ[LARGE_GRP];
  __shared__ float4 s_box[LARGE_GRP];
  __shared__ int s_summ2[LARGE_GRP / BLK];
  __shared__ float4 s_box2[LARGE_GRP / BLK];
  __shared__ double s_red[4 * (CT / 64)];
  __shared__ int s_w[CT / 64];
  const int s = blockIdx.x;
  const int n = seg_cnt[s];
  if (n <= SMALL_N) return;
  const long long base = seg_base[s];
  if (si[2 * pool_cap + base]) return;                 // a grid variant did it (mark written there)
  cell_sort<LARGE_N / 2>(px + base, py + base, n, (int *)s_parent, s_red, s_w, sx + base, sy + base,
                         si + base);
  if (n <= LARGE_N) {
    ParH par{s_parent};
    cluster_body<ParH, LARGE_GRP>(par, s_summ, s_box, s_summ2, s_box2, s_red, sx + base, sy + base,
                                  n, R0, Rd, si + base, si + pool_cap + base, label + base);
  } else {          // parents in global memory: the third scratch plane is free until the epilogue
    ParI par{si + 2 * pool_cap + base, true};
    cluster_body<ParI, LARGE_GRP>(par, s_summ, s_box, s_summ2, s_box2, s_red, sx + base, sy + base,
                                  n, R0, Rd, si + base, si + pool_cap + base, label + base);
  }
}

// ---------------------------------------------------------------- a14/a15
constexpr int FT = 512;
constexpr int FW = FT / 64;
constexpr int MAXTH = 128;
constexpr int LDS_MEMBERS = 2048;   // clusters up to this size: members cached in LDS, one wave per heading

// block-wide reduction of K per-thread doubles (sum / min / max by OP): result in out[0..K)
struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };
template <class OP>
__device__ __forceinline__ double wave_red(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = OP::f(v, shfl_xor_d(v, m));
  return v;
}

struct Ext { double c1min, c1max, c2min, c2max; };

__device__ __forceinline__ double readlane_f64(double v, int l) {   // l uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void cross_point(double a0, double a1, double b0, double b1,
                                            double c0, double c1, double &x, double &y) {
  // my_loader.py:699-702
  x = (b0 * -c1 - b1 * -c0) / (a0 * b1 - a1 * b0);
  y = (a1 * -c0 - a0 * -c1) / (a0 * b1 - a1 * b0);
}

// rectangle (best heading + extents) -> KITTI box row (my_loader.py:633-702); one thread
__device__ void emit_box(double thb, double sin_s, double cos_s, double c1min, double c1max,
                         double c2min, double c2max, double zmax, int s, int v, int jinst, int kc,
                         int root, int m, const ViewCalib *calib, const int *inst_class,
                         const int *inst_is_car, const float *inst_box, const float *inst_score,
                         double car_aspect_max, int cap_rows, double *rows, int *n_rows,
                         uint32_t *status) {
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
      o[18] = thb; o[19] = c1min; o[20] = c2min; o[21] = c1max; o[22] = c2max;
      o[23] = (double)root;
    } else {
      atomicOr(status, DFU3D_ST_ROW_OVERFLOW);
    }
  }
}

// ---- workspace layout (doubles) ----------------------------------------------
//   [0]            int q_count | int big_count
//   [1]            int ticket counter of k_fit_big_cost (64 per item) | int ticket counter of k_fit_medium (64 per descriptor)
//   [2 ...)        cluster descriptors, 8 doubles each (cap_q of them):
//                  0 segment s, 1 cluster ordinal kc, 2 root (smallest point index), 3 members m,
//                  4 position of the members in gsx/gsy, 5 max z of the instance, 6 ordinal among
//                  the big clusters (m > LDS_MEMBERS), 7 spare
//   then           big_list: int32 descriptor index per big cluster (cap_big)
//   then           heading costs of the big clusters, MAXTH doubles each
struct FitWs {
  int *counters;
  double *dsc;
  int *big_list;
  double *big_cost;
  int cap_q, cap_big;
};
__host__ __device__ inline int fit_cap_q(int cap_rows) { return 2 * cap_rows + 64; }
__host__ __device__ inline FitWs fit_ws_view(double *ws, int cap_rows, int cap_big) {
  FitWs w;
  w.cap_q = fit_cap_q(cap_rows);
  w.cap_big = cap_big;
  w.counters = (int *)ws;
  w.dsc = ws + 2;
  w.big_list = (int *)(w.dsc + (size_t)8 * w.cap_q);
  w.big_cost = w.dsc + (size_t)8 * w.cap_q + (cap_big / 2 + 1);
  return w;
}

// ---- F1: per instance, members of every cluster made contiguous -----------------
// Clusters in ascending-root order (rectangle_fitting.py:188-191), members in index
// order.  One stable counting sort per GK clusters: every wave owns a contiguous
// range of the instance's points, counts its members per cluster, and after one
// scan writes them behind the members of the waves before it -- no barrier inside
// the sweeps, O(n) per GK clusters instead of O(n) per cluster.
constexpr int GK = 512;            // clusters per counting-sort pass (= FT: one thread per cluster in the hand-offs)

__device__ __forceinline__ int rank_in(const int *roots, int nk, int L) {   // roots ascending, L present
  int lo = 0, hi = nk - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (roots[mid] < L) lo = mid + 1; else hi = mid;
  }
  return (roots[lo] == L) ? lo : -1;
}

constexpr int GGU = 4;             // k_fit_gather: steps of 64 points whose loads are requested together (8 measured slower: 0.212
                                   // against 0.198 ms)
__global__ __launch_bounds__(FT) void k_fit_gather(
    const double *__restrict__ px, const double *__restrict__ py,
    const double *__restrict__ pz, const int *__restrict__ label,
    const long long *__restrict__ seg_base, const int *__restrict__ seg_cnt,
    double *__restrict__ gsx, double *__restrict__ gsy, int *__restrict__ sroot,
    uint32_t *__restrict__ status, double *__restrict__ fit_ws, int cap_rows, int cap_big) {
  __shared__ int s_roots[GK];
  __shared__ int s_cnt[FW][GK];
  __shared__ int s_tot[GK], s_off[GK];
  __shared__ double s_red[FW];
  __shared__ int s_w[FW];
  __shared__ int s_q0, s_total;
  // (longest first, as k_range_cluster_grid: the grid is three times the segments, classes above 16 384 / above 8 192 / the rest;
  // the stage 1.166 -> 1.117 ms, with 8 192 / 32 768 at the top 1.132 / 1.126)
  const int S_ = (int)(gridDim.x / 3u);
  const int cls = (int)(blockIdx.x / (unsigned)S_);
  const int s = (int)(blockIdx.x - (unsigned)cls * (unsigned)S_);
  const int n = seg_cnt[s];
  if (n == 0) return;
  if (cls != (n > 16384 ? 0 : (n > 8192 ? 1 : 2))) return;
  const long long base = seg_base[s];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const FitWs W = fit_ws_view(fit_ws, cap_rows, cap_big);

  // ordered list of cluster roots, and in the same sweep the max z over ALL instance points
  // (my_loader.py:647-648).  Every wave owns a contiguous range of the instance: it counts its roots, one hand-off
  // orders the waves, a second sweep writes them -- no barrier inside the sweeps (a workgroup-wide rank per 512 points
  // made the longest instance, 100 steps of two barriers, the duration of this kernel).
  const int R = (((n + FW - 1) / FW) + 63) & ~63;          // points per wave, a multiple of 64
  const int r0 = min(n, wave * R), r1 = min(n, r0 + R);
  double zm = -INFINITY;
  int mine = 0;
  for (int c = r0; c < r1; c += 64 * GGU) {
    int L[GGU];
    double zz[GGU];
#pragma unroll
    for (int u = 0; u < GGU; u++) {
      const int i = c + u * 64 + lane;
      L[u] = (i < r1) ? label[base + i] : -1;
      zz[u] = (i < r1) ? pz[base + i] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < GGU; u++) {
      const int i = c + u * 64 + lane;
      mine += __popcll(__ballot(i < r1 && L[u] == i));
      zm = fmax(zm, zz[u]);
    }
  }
  zm = wave_max_d(zm);
  if (lane == 0) { s_red[wave] = zm; s_w[wave] = mine; }
  __syncthreads();
  double zmax = s_red[0];
  int nroots = 0, my_off = 0;
#pragma unroll
  for (int w = 0; w < FW; w++) {
    zmax = fmax(zmax, s_red[w]);
    my_off += (w < wave) ? s_w[w] : 0;
    nroots += s_w[w];
  }
  for (int c = r0; c < r1; c += 64 * GGU) {         // GGU steps' labels requested before the first is used (this sweep
    int L[GGU];                                      // was one dependent load per 64 points: the longest instance's chain)
#pragma unroll
    for (int u = 0; u < GGU; u++) {
      const int i = c + u * 64 + lane;
      L[u] = (i < r1) ? label[base + i] : -1;
    }
#pragma unroll
    for (int u = 0; u < GGU; u++) {
      const int i = c + u * 64 + lane;
      const bool f = (i < r1) && (L[u] == i);
      const unsigned long long m = __ballot(f);
      if (f) sroot[base + my_off + __popcll(m & ((1ull << lane) - 1ull))] = i;
      my_off += __popcll(m);
    }
  }
  if (threadIdx.x == 0) s_q0 = atomicAdd(&W.counters[0], nroots);
  __syncthreads();
  const int q0 = s_q0;

  long long goff = 0;
  for (int k0 = 0; k0 < nroots; k0 += GK) {
    const int nk = min(GK, nroots - k0);
    __syncthreads();
    if (threadIdx.x < nk) s_roots[threadIdx.x] = sroot[base + k0 + threadIdx.x];
    for (int i = threadIdx.x; i < FW * GK; i += FT) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const int lo_root = s_roots[0], hi_root = s_roots[nk - 1];
    int memo_L = -2, memo_key = -1;                          // the lane's last successful search (labels are >= 0)
    // counts per (wave, cluster)
    for (int c0 = r0; c0 < r1; c0 += 64 * GGU) {              // GGU steps' labels requested before the first is used
      int Ls[GGU];
#pragma unroll
      for (int u = 0; u < GGU; u++) {
        const int i = c0 + u * 64 + lane;
        Ls[u] = (i < r1) ? label[base + i] : -1;
      }
#pragma unroll
      for (int u = 0; u < GGU; u++) {
        if (c0 + u * 64 >= r1) break;
        const int L = Ls[u];
        int key = -1;
        if (L == memo_L) key = memo_key;                       // (most points sit in one cluster: no search)
        else if (L >= lo_root && L <= hi_root) { key = rank_in(s_roots, nk, L); memo_L = L; memo_key = key; }
        unsigned long long rem = __ballot(key >= 0);
        while (rem) {
          const int src = __ffsll((long long)rem) - 1;
          const int k = rl_i(key, src);
          const unsigned long long m = __ballot(key == k);
          if (lane == src) s_cnt[wave][k] += __popcll(m);
          rem &= ~m;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < nk) {                                  // exclusive over the waves, per cluster
      int t = 0;
#pragma unroll
      for (int w = 0; w < FW; w++) { const int c = s_cnt[w][threadIdx.x]; s_cnt[w][threadIdx.x] = t; t += c; }
      s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    {                                                        // exclusive over the clusters
      int tot;
      const int ex = block_excl_scan<FW>(((int)threadIdx.x < nk) ? s_tot[threadIdx.x] : 0, s_w, tot);
      if ((int)threadIdx.x < nk) s_off[threadIdx.x] = ex;
      if (threadIdx.x == 0) s_total = tot;
    }
    __syncthreads();
    // stable scatter
    for (int c0 = r0; c0 < r1; c0 += 64 * GGU) {
      int keys[GGU];
      double xs[GGU], ys[GGU];
#pragma unroll
      for (int u = 0; u < GGU; u++) {
        const int i = c0 + u * 64 + lane;
        const int L = (i < r1) ? label[base + i] : -1;
        if (L == memo_L) keys[u] = memo_key;
        else if (L >= lo_root && L <= hi_root) { keys[u] = rank_in(s_roots, nk, L); memo_L = L; memo_key = keys[u]; }
        else keys[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < GGU; u++) {
        const int i = c0 + u * 64 + lane;
        xs[u] = 0.0; ys[u] = 0.0;
        if (keys[u] >= 0) { xs[u] = px[base + i]; ys[u] = py[base + i]; }
      }
#pragma unroll
      for (int u = 0; u < GGU; u++) {
        if (c0 + u * 64 >= r1) break;
        const int key = keys[u];
        unsigned long long rem = __ballot(key >= 0);
        while (rem) {
          const int src = __ffsll((long long)rem) - 1;
          const int k = rl_i(key, src);
          const unsigned long long m = __ballot(key == k);
          const int cur = s_cnt[wave][k];
          if (key == k) {
            const long long d = base + goff + s_off[k] + cur + __popcll(m & ((1ull << lane) - 1ull));
            gsx[d] = xs[u];
            gsy[d] = ys[u];
          }
          if (lane == src) s_cnt[wave][k] = cur + __popcll(m);
          rem &= ~m;
        }
      }
    }
    // descriptors
    if (threadIdx.x < nk) {
      const int k = threadIdx.x, e = q0 + k0 + k;
      if (e < W.cap_q) {
        double *d = W.dsc + (size_t)8 * e;
        const int m = s_tot[k];
        d[0] = (double)s; d[1] = (double)(k0 + k); d[2] = (double)s_roots[k]; d[3] = (double)m;
        d[4] = (double)(base + goff + s_off[k]); d[5] = zmax; d[6] = -1.0; d[7] = 0.0;
        if (m > LDS_MEMBERS) {
          const int bi = atomicAdd(&W.counters[1], 1);
          if (bi < cap_big) { W.big_list[bi] = e; d[6] = (double)bi; }
        }
      } else {
        atomicOr(status, DFU3D_ST_ROW_OVERFLOW);             // more clusters than 2 x cap_rows
      }
    }
    __syncthreads();
    goff += s_total;
  }
}

// ---- the heading search in two tiers ----------------------------------------------------------------------------
// rectangle_fitting.py:83-136 scores 89 headings by -var(E1) - var(E2) (np.var: mean, then squared deviations: three
// sweeps over the points with the extents) and keeps the first strict maximum.  Only the arg-max matters, so:
// tier 1 scores every heading in TWO sweeps (extents; then count, sum and sum of squares of D1 over E1 / D2 over E2,
// variance = q/n - (s/n)^2) -- the same fp64 quantities, a different summation: its cost differs from the reference's
// by at most ~3 n 2^-53 E[D^2] (n <= 10^6: 3e-10 relative to mag = E1[D1^2] + E2[D2^2] >= |cost|);
// tier 2 re-scores, with the reference's own three sweeps, only the headings whose tier-1 cost lies within
// FIT_TAU * (largest mag) of the best one -- one heading unless the point set has an exact or nearly exact symmetry.
// A heading outside that band cannot be the reference's arg-max, and inside it the reference's formula decides.
constexpr double FIT_TAU = 1e-8;
// v_min_f64 / v_max_f64 as they are: fmin() / fmax() on a loop-carried value make the compiler canonicalise it first
// (v_max_f64 x, x, x per use -- a quarter of the first sweep's instructions); no signalling NaN can reach these loops
__device__ __forceinline__ double min_raw_d(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_raw_d(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
struct FitAcc {                                   // tier 1, second sweep
  double s1, s2, q1, q2;
  int n1, n2;
  __device__ __forceinline__ void clear() { s1 = s2 = q1 = q2 = 0.0; n1 = n2 = 0; }
  __device__ __forceinline__ void add(double x, double y, double ct, double st, double a0, double a1, double b0, double b1) {
    const double c1 = x * ct + y * st;
    const double c2 = x * (-st) + y * ct;
    const double d1 = fmin(fabs(a1 - c1), fabs(c1 - a0));
    const double d2 = fmin(fabs(b1 - c2), fabs(c2 - b0));
    // the squares are summed with an FMA: the SPLIT is the reference's to the bit, the summation of the variance is
    // tier 1's own (the band covers it).  (An fp64 comparison issues at the rate of an addition -- tools/micro/
    // cmp_rates.hip -- so the sign of d1 - d2 instead of the comparison only added an instruction.)
    if (d1 < d2) { s1 += d1; q1 = fma(d1, d1, q1); n1++; } else { s2 += d2; q2 = fma(d2, d2, q2); n2++; }
  }
  __device__ __forceinline__ void wave_reduce() {
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); q1 = wave_sum_d(q1); q2 = wave_sum_d(q2);
    n1 = wave_sum_i(n1); n2 = wave_sum_i(n2);
  }
  __device__ __forceinline__ double cost(double &mag) const {
    double V1 = 0.0, V2 = 0.0, M1 = 0.0, M2 = 0.0;
    if (n1) { const double mu = s1 / (double)n1; M1 = q1 / (double)n1; V1 = -(M1 - mu * mu); }
    if (n2) { const double mu = s2 / (double)n2; M2 = q2 / (double)n2; V2 = -(M2 - mu * mu); }
    mag = M1 + M2;
    return V1 + V2;
  }
};
// the reference's cost of one heading, a wave over the points (lanes take points lane, lane + 64, ...): three sweeps
__device__ __forceinline__ double wave_exact_cost(const double *mx, const double *my, int m, double ct, double st) {
  const int lane = lane_id();
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
  return V1 + V2;
}
// From the tier-1 costs of all headings (cost[th], th < n_theta) and the largest mag: the arg-max of the reference.
// Whole wave, uniform; mx / my: the cluster's points (LDS or global); ctab / stab: cos / sin of the headings.
__device__ __forceinline__ int wave_pick_heading(const double *cost, int n_theta, double mag_max, const double *mx,
                                                 const double *my, int m, const double *ctab, const double *stab) {
  double top = -INFINITY;
  for (int th = 0; th < n_theta; th++) { const double c = cost[th]; if (top < c) top = c; }
  const double band = top - FIT_TAU * mag_max;
  int first = -1, ncand = 0;
  for (int th = 0; th < n_theta; th++)
    if (cost[th] >= band) { if (first < 0) first = th; ncand++; }
  if (ncand <= 1) return first < 0 ? 0 : first;          // (no heading with a cost that is a number: the reference keeps heading 0)
  int best = 0;
  double bc = -INFINITY;
  for (int th = first; th < n_theta; th++) {             // uniform: the candidates, with the reference's formula
    if (!(cost[th] >= band)) continue;
    const double c = wave_exact_cost(mx, my, m, ctab[th], stab[th]);
    if (bc < c) { bc = c; best = th; }
  }
  return best;
}

// ---- F2a: clusters of at most 64 points, one wave each, one lane per heading -----
// Members live in registers (lane i = member i) and are broadcast as scalar operands;
// no reductions at all.  rectangle_fitting.py:83-136.
__global__ __launch_bounds__(256) void k_fit_tiny(
    const double *__restrict__ gsx, const double *__restrict__ gsy, int max_inst,
    const ViewCalib *__restrict__ calib, const int *__restrict__ inst_class,
    const int *__restrict__ inst_is_car, const float *__restrict__ inst_box,
    const float *__restrict__ inst_score, int n_theta, double dtheta, double car_aspect_max,
    int cap_rows, double *__restrict__ rows, int *__restrict__ n_rows,
    uint32_t *__restrict__ status, double *__restrict__ fit_ws, int cap_big) {
  const FitWs W = fit_ws_view(fit_ws, cap_rows, cap_big);
  const int nq = min(W.counters[0], W.cap_q);
  const int lane = lane_id();
  const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  for (int e = gw; e < nq; e += nw) {
    const double *d = W.dsc + (size_t)8 * e;
    const int m = (int)d[3];
    if (m > 64) continue;
    const long long P = (long long)d[4];
    const double x = (lane < m) ? gsx[P + lane] : 0.0, y = (lane < m) ? gsy[P + lane] : 0.0;
    // tier 1: this lane's headings (lane, lane + 64), two sweeps each
    constexpr int TH_PER_LANE = MAXTH / 64;
    double c1t[TH_PER_LANE], magmax = 0.0, top = -INFINITY;
#pragma unroll
    for (int k = 0; k < TH_PER_LANE; k++) {
      const int th = lane + 64 * k;
      c1t[k] = -INFINITY;
      if (64 * k >= n_theta) continue;                 // uniform
      const double theta = (double)th * dtheta;
      const double ct = cos(theta), st = sin(theta);
      double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
      for (int j = 0; j < m; j++) {
        const double xj = readlane_f64(x, j), yj = readlane_f64(y, j);
        const double c1 = xj * ct + yj * st;
        const double c2 = xj * (-st) + yj * ct;
        a0 = min_raw_d(a0, c1); a1 = max_raw_d(a1, c1);
        b0 = min_raw_d(b0, c2); b1 = max_raw_d(b1, c2);
      }
      FitAcc A;
      A.clear();
      for (int j = 0; j < m; j++) A.add(readlane_f64(x, j), readlane_f64(y, j), ct, st, a0, a1, b0, b1);
      double mag;
      const double c = A.cost(mag);
      if (th < n_theta) {
        c1t[k] = c;
        if (top < c) top = c;
        if (mag > magmax) magmax = mag;                // (a NaN never wins)
      }
    }
    top = wave_max_d(top);
    magmax = wave_max_d(magmax);
    const double band = top - FIT_TAU * magmax;
    double bestc = -INFINITY;
    int bestth = 0x7FFFFFFF;
    {
      int ncand = 0;
#pragma unroll
      for (int k = 0; k < TH_PER_LANE; k++) ncand += __popcll(__ballot(c1t[k] >= band));
      if (ncand <= 1) {                                // the usual case: one heading in the band -- it is the arg-max
#pragma unroll
        for (int k = 0; k < TH_PER_LANE; k++)
          if (c1t[k] >= band) { bestc = c1t[k]; bestth = lane + 64 * k; }
      } else {
        // tier 2: the candidates with the reference's three sweeps, each on its own lane
#pragma unroll
        for (int k = 0; k < TH_PER_LANE; k++) {
          if (__ballot(c1t[k] >= band) == 0ull) continue;    // uniform
          const int th = lane + 64 * k;
          const double theta = (double)th * dtheta;
          const double ct = cos(theta), st = sin(theta);
          double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
          for (int j = 0; j < m; j++) {
            const double xj = readlane_f64(x, j), yj = readlane_f64(y, j);
            const double c1 = xj * ct + yj * st;
            const double c2 = xj * (-st) + yj * ct;
            a0 = fmin(a0, c1); a1 = fmax(a1, c1);
            b0 = fmin(b0, c2); b1 = fmax(b1, c2);
          }
          double s1 = 0.0, s2 = 0.0;
          int n1 = 0, n2 = 0;
          for (int j = 0; j < m; j++) {
            const double xj = readlane_f64(x, j), yj = readlane_f64(y, j);
            const double c1 = xj * ct + yj * st;
            const double c2 = xj * (-st) + yj * ct;
            const double d1 = fmin(fabs(a1 - c1), fabs(c1 - a0));
            const double d2 = fmin(fabs(b1 - c2), fabs(c2 - b0));
            if (d1 < d2) { s1 += d1; n1++; } else { s2 += d2; n2++; }
          }
          const double m1 = n1 ? s1 / (double)n1 : 0.0, m2 = n2 ? s2 / (double)n2 : 0.0;
          double q1 = 0.0, q2 = 0.0;
          for (int j = 0; j < m; j++) {
            const double xj = readlane_f64(x, j), yj = readlane_f64(y, j);
            const double c1 = xj * ct + yj * st;
            const double c2 = xj * (-st) + yj * ct;
            const double d1 = fmin(fabs(a1 - c1), fabs(c1 - a0));
            const double d2 = fmin(fabs(b1 - c2), fabs(c2 - b0));
            if (d1 < d2) { const double u = d1 - m1; q1 += u * u; }
            else { const double u = d2 - m2; q2 += u * u; }
          }
          double V1 = 0.0, V2 = 0.0;
          if (n1) V1 = -(q1 / (double)n1);
          if (n2) V2 = -(q2 / (double)n2);
          const double c = V1 + V2;
          // this lane's headings come in ascending order: keep the first strict maximum
          if (c1t[k] >= band && bestc < c) { bestc = c; bestth = th; }
        }
      }
    }
    // first strict maximum over the candidate headings (rectangle_fitting.py:135-136): the largest
    // cost, the smallest heading among equals; a lane without any has
    // bestc = -inf / bestth = INT_MAX and loses against everything, and if nobody has one
    // the loop of the reference never updates its initial choice, heading 0
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
      const double oc = shfl_xor_d(bestc, msk);
      const int ot = __shfl_xor(bestth, msk, 64);
      if (oc > bestc || (oc == bestc && ot < bestth)) { bestc = oc; bestth = ot; }
    }
    const int best = (bestth == 0x7FFFFFFF) ? 0 : bestth;
    const double thb = (double)best * dtheta;
    const double sin_s = sin(thb), cos_s = cos(thb);
    double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
    if (lane < m) {
      const double c1 = x * cos_s + y * sin_s;
      const double c2 = x * (-sin_s) + y * cos_s;
      a0 = c1; a1 = c1; b0 = c2; b1 = c2;
    }
    a0 = wave_min_d(a0); a1 = wave_max_d(a1);
    b0 = wave_min_d(b0); b1 = wave_max_d(b1);
    if (lane == 0) {
      const int s = (int)d[0], kc = (int)d[1], root = (int)d[2];
      const int v = s / max_inst, jinst = s - v * max_inst;
      emit_box(thb, sin_s, cos_s, a0, a1, b0, b1, d[5], s, v, jinst, kc, root, m, calib, inst_class,
               inst_is_car, inst_box, inst_score, car_aspect_max, cap_rows, rows, n_rows, status);
    }
  }
}

// ---- F2b: clusters of 65 .. LDS_MEMBERS points, one workgroup each ---------------
// (64 registers, no scratch: four workgroups per compute unit instead of three, 271 -> 256 us)
__global__ __launch_bounds__(FT, 8) void k_fit_medium(
    const double *__restrict__ gsx, const double *__restrict__ gsy, int max_inst,
    const ViewCalib *__restrict__ calib, const int *__restrict__ inst_class,
    const int *__restrict__ inst_is_car, const float *__restrict__ inst_box,
    const float *__restrict__ inst_score, int n_theta, double dtheta, double car_aspect_max,
    int cap_rows, double *__restrict__ rows, int *__restrict__ n_rows,
    uint32_t *__restrict__ status, double *__restrict__ fit_ws, int cap_big) {
  __shared__ double lx[LDS_MEMBERS], ly[LDS_MEMBERS];
  __shared__ double s_cost[MAXTH], s_mag[MAXTH], s_ct[MAXTH], s_st[MAXTH];
  __shared__ double s_ext[FW][4];
  const FitWs W = fit_ws_view(fit_ws, cap_rows, cap_big);
  const int nq = min(W.counters[0], W.cap_q);
  const int wave = threadIdx.x >> 6, lane = lane_id();
  if (threadIdx.x < MAXTH) {                   // heading table (rectangle_fitting.py:119-122)
    const double theta = (double)threadIdx.x * dtheta;
    s_ct[threadIdx.x] = cos(theta);
    s_st[threadIdx.x] = sin(theta);
  }
  __shared__ int s_item;
  while (true) {
    // descriptors are handed out through a counter (the spare word of the workspace header, zeroed with it): a cluster of
    // 2 000 members costs thirty times one of 65, and a fixed share of the queue per workgroup ended with a few workgroups
    // busy (the fit stage 1.120 -> 1.090 ms)
    __syncthreads();                             // the previous item is done with s_item and the members in LDS
    if (wave == 0) {
      const int ticket = atomicAdd(&W.counters[3], 1);        // every lane adds 1: the counter runs in units of 64 (see k_fit_big_cost)
      if (lane == 0) s_item = ticket >> 6;
    }
    __syncthreads();
    const int e = s_item;
    if (e >= nq) break;
    const double *d = W.dsc + (size_t)8 * e;
    const int m = (int)d[3];
    if (m <= 64 || m > LDS_MEMBERS) continue;
    __syncthreads();
    const long long P = (long long)d[4];
    for (int i = threadIdx.x; i < m; i += FT) { lx[i] = gsx[P + i]; ly[i] = gsy[P + i]; }
    __syncthreads();
    const double *mx = lx;
    const double *my = ly;
    // tier 1: 89 headings, two per wave and sweep (every LDS read serves both), two sweeps
    // (rectangle_fitting.py:119-136; see "the heading search in two tiers")
    for (int th0 = 2 * wave; th0 < n_theta; th0 += 2 * FW) {
      const int th1 = min(th0 + 1, n_theta - 1);       // (an odd heading count: the last one is scored twice)
      const double ct0 = s_ct[th0], st0 = s_st[th0], ct1 = s_ct[th1], st1 = s_st[th1];
      double a0 = INFINITY, a1 = -INFINITY, b0 = INFINITY, b1 = -INFINITY;
      double e0 = INFINITY, e1 = -INFINITY, f0 = INFINITY, f1 = -INFINITY;
      for (int i = lane; i < m; i += 64) {
        const double x = mx[i], y = my[i];
        double c1 = x * ct0 + y * st0;
        double c2 = x * (-st0) + y * ct0;
        a0 = min_raw_d(a0, c1); a1 = max_raw_d(a1, c1);
        b0 = min_raw_d(b0, c2); b1 = max_raw_d(b1, c2);
        c1 = x * ct1 + y * st1;
        c2 = x * (-st1) + y * ct1;
        e0 = min_raw_d(e0, c1); e1 = max_raw_d(e1, c1);
        f0 = min_raw_d(f0, c2); f1 = max_raw_d(f1, c2);
      }
      a0 = wave_min_d(a0); a1 = wave_max_d(a1); b0 = wave_min_d(b0); b1 = wave_max_d(b1);
      e0 = wave_min_d(e0); e1 = wave_max_d(e1); f0 = wave_min_d(f0); f1 = wave_max_d(f1);
      FitAcc A, B;
      A.clear(); B.clear();
      for (int i = lane; i < m; i += 64) {
        const double x = mx[i], y = my[i];
        A.add(x, y, ct0, st0, a0, a1, b0, b1);
        B.add(x, y, ct1, st1, e0, e1, f0, f1);
      }
      A.wave_reduce(); B.wave_reduce();
      double magA, magB;
      const double cA = A.cost(magA), cB = B.cost(magB);
      if (lane == 0) { s_cost[th0] = cA; s_mag[th0] = magA; s_cost[th1] = cB; s_mag[th1] = magB; }
    }
    __syncthreads();
    // the arg-max of the reference (tier 2 only for headings within the band): every wave computes the same answer
    double magmax = 0.0;
    for (int th = 0; th < n_theta; th++) { const double g = s_mag[th]; if (g > magmax) magmax = g; }
    const int best = wave_pick_heading(s_cost, n_theta, magmax, mx, my, m, s_ct, s_st);
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
      const int s = (int)d[0], kc = (int)d[1], root = (int)d[2];
      const int v = s / max_inst, jinst = s - v * max_inst;
      emit_box(thb, sin_s, cos_s, c1min, c1max, c2min, c2max, d[5], s, v, jinst, kc, root, m, calib,
               inst_class, inst_is_car, inst_box, inst_score, car_aspect_max, cap_rows, rows,
               n_rows, status);
    }
  }
}

// ---- large clusters: workgroup per (cluster, batch of BIGC_HB headings), items handed out through a counter ----
// The workgroup streams the cluster's members through LDS in chunks of BIGC_CH (three sweeps: extents; sums and counts
// of E1 / E2; squared deviations -- rectangle_fitting.py:83-111); every wave scores BIGC_HPW headings of the batch
// against each chunk with everything per heading in registers.  No LDS operand besides the two coordinates, no
// workgroup-wide reduction: the only barriers are the two around a chunk's load.
// (First formulation: eight headings per sweep, every thread other points -- 64 accumulators per lane, two waves
// per SIMD, 80 wave reductions + cross-wave sums and seven barriers per item.  A wave per heading reading the
// members straight from global memory was tried as well: every XCD fetched every cluster, 5.4 GB per launch.)
constexpr int BIGC_T = 512, BIGC_WAVES = BIGC_T / 64, BIGC_HPW = 2, BIGC_HB = BIGC_WAVES * BIGC_HPW;
constexpr int BIGC_CH = 4096;   // points staged per step: 64 KB of LDS, i.e. two workgroups per compute unit (four, which 64 registers would allow, measured 25 % slower)
constexpr int BIGC_LONG = 16384, BIGC_MID = 6144;   // size classes of the scheduling rounds
__global__ __launch_bounds__(BIGC_T) void k_fit_big_cost(const double *__restrict__ gsx,
                                                         const double *__restrict__ gsy, int n_theta,
                                                         double dtheta, double *__restrict__ fit_ws,
                                                         int cap_rows, int cap_big) {
  __shared__ double lx[BIGC_CH], ly[BIGC_CH];
  __shared__ double s_ct[MAXTH], s_st[MAXTH];
  __shared__ int s_item;
  const FitWs W = fit_ws_view(fit_ws, cap_rows, cap_big);
  const int nbig = min(W.counters[1], cap_big);
  if (nbig == 0) return;
  if (threadIdx.x < MAXTH) {                   // heading table (rectangle_fitting.py:119-122)
    const double theta = (double)threadIdx.x * dtheta;
    s_ct[threadIdx.x] = cos(theta);
    s_st[threadIdx.x] = sin(theta);
  }
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int nb = (n_theta + BIGC_HB - 1) / BIGC_HB;
  const int items = nbig * nb;
  while (true) {
    __syncthreads();                           // the previous item is done with s_item and the chunk
    if (wave == 0) {
      // every lane of wave 0 adds 1 (ONE atomic of +64): the counter runs in units of 64.  (An atomic under
      // `lane == 0` whose result steers the loop invites the compiler to split the loop by lane: it hangs.)
      const int ticket = atomicAdd(&W.counters[2], 1);
      if (lane == 0) s_item = ticket >> 6;
    }
    __syncthreads();
    const int ticket_item = s_item;
    if (ticket_item >= 3 * items) break;       // uniform; the counter only grows
    // Three rounds of tickets, longest clusters first: an item of a 50 000-point cluster runs for a good part of this
    // kernel's duration and must not be among the last to start (1.20 -> 0.94 ms; giving the long clusters finer items
    // as well -- one heading per wave -- gained nothing more).
    const int round = ticket_item / items;
    const int item = ticket_item - round * items;
    const int c = item / nb, tb = (item - c * nb) * BIGC_HB;
    const double *dsc = W.dsc + (size_t)8 * W.big_list[c];
    const int m = (int)dsc[3];
    if ((m > BIGC_LONG ? 0 : (m > BIGC_MID ? 1 : 2)) != round) continue;   // (uniform) not this round's
    const double *mx = gsx + (long long)dsc[4], *my = gsy + (long long)dsc[4];
    int th[BIGC_HPW];
    double ct[BIGC_HPW], st[BIGC_HPW], nst[BIGC_HPW];
#pragma unroll
    for (int h = 0; h < BIGC_HPW; h++) {
      th[h] = tb + wave * BIGC_HPW + h;
      const int tc = min(th[h], n_theta - 1);  // a heading past the end is scored and dropped
      ct[h] = s_ct[tc]; st[h] = s_st[tc]; nst[h] = -st[h];
    }
    double a0[BIGC_HPW], a1[BIGC_HPW], b0[BIGC_HPW], b1[BIGC_HPW];
#pragma unroll
    for (int h = 0; h < BIGC_HPW; h++) { a0[h] = INFINITY; a1[h] = -INFINITY; b0[h] = INFINITY; b1[h] = -INFINITY; }
    for (int c0 = 0; c0 < m; c0 += BIGC_CH) {
      const int cm = min(BIGC_CH, m - c0);
      __syncthreads();
      for (int i = threadIdx.x; i < cm; i += BIGC_T) { lx[i] = mx[c0 + i]; ly[i] = my[c0 + i]; }
      __syncthreads();
      for (int i = lane; i < cm; i += 64) {
        const double x = lx[i], y = ly[i];
#pragma unroll
        for (int h = 0; h < BIGC_HPW; h++) {
          const double c1 = x * ct[h] + y * st[h];
          const double c2 = x * nst[h] + y * ct[h];
          a0[h] = min_raw_d(a0[h], c1); a1[h] = max_raw_d(a1[h], c1);
          b0[h] = min_raw_d(b0[h], c2); b1[h] = max_raw_d(b1[h], c2);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < BIGC_HPW; h++) {
      a0[h] = wave_min_d(a0[h]); a1[h] = wave_max_d(a1[h]);
      b0[h] = wave_min_d(b0[h]); b1[h] = wave_max_d(b1[h]);
    }
    FitAcc A[BIGC_HPW];
#pragma unroll
    for (int h = 0; h < BIGC_HPW; h++) A[h].clear();
    for (int c0 = 0; c0 < m; c0 += BIGC_CH) {
      const int cm = min(BIGC_CH, m - c0);
      __syncthreads();
      for (int i = threadIdx.x; i < cm; i += BIGC_T) { lx[i] = mx[c0 + i]; ly[i] = my[c0 + i]; }
      __syncthreads();
      for (int i = lane; i < cm; i += 64) {
        const double x = lx[i], y = ly[i];
#pragma unroll
        for (int h = 0; h < BIGC_HPW; h++) A[h].add(x, y, ct[h], st[h], a0[h], a1[h], b0[h], b1[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < BIGC_HPW; h++) {
      A[h].wave_reduce();
      double mag;
      const double cst = A[h].cost(mag);
      if (lane == 0 && th[h] < n_theta) {
        W.big_cost[(size_t)c * MAXTH + th[h]] = cst;     // tier 1 (k_fit_big_box re-scores the headings in the band)
        if (mag > 0.0) atomicMax((unsigned long long *)(W.dsc + (size_t)8 * W.big_list[c] + 7), (unsigned long long)__double_as_longlong(mag));
      }
    }
  }
}

__global__ __launch_bounds__(FT) void k_fit_big_box(
    const double *__restrict__ gsx, const double *__restrict__ gsy, int max_inst,
    const ViewCalib *__restrict__ calib, const int *__restrict__ inst_class,
    const int *__restrict__ inst_is_car, const float *__restrict__ inst_box,
    const float *__restrict__ inst_score, int n_theta, double dtheta, double car_aspect_max,
    int cap_rows, double *__restrict__ rows, int *__restrict__ n_rows,
    uint32_t *__restrict__ status, double *__restrict__ fit_ws, int cap_big) {
  __shared__ double s_ext[FW][4];
  __shared__ double s_ct[MAXTH], s_st[MAXTH];
  const FitWs W = fit_ws_view(fit_ws, cap_rows, cap_big);
  const int nbig = min(W.counters[1], cap_big);
  if (threadIdx.x < MAXTH) {                   // heading table (rectangle_fitting.py:119-122)
    const double theta = (double)threadIdx.x * dtheta;
    s_ct[threadIdx.x] = cos(theta);
    s_st[threadIdx.x] = sin(theta);
  }
  for (int c = blockIdx.x; c < nbig; c += gridDim.x) {
  __syncthreads();
  const double *dsc = W.dsc + (size_t)8 * W.big_list[c];
  const int s = (int)dsc[0], kc = (int)dsc[1], root = (int)dsc[2], m = (int)dsc[3];
  const double zmax = dsc[5];
  const double *mx = gsx + (long long)dsc[4], *my = gsy + (long long)dsc[4];
  const double *cost = W.big_cost + (size_t)c * MAXTH;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  // the arg-max of the reference: tier-1 costs of k_fit_big_cost, the headings within the band re-scored with the
  // reference's three sweeps (one heading, unless the cluster has a symmetry); every wave computes the same answer
  const int best = wave_pick_heading(cost, n_theta, dsc[7], mx, my, m, s_ct, s_st);
  const double thb = (double)best * dtheta;
  const double sin_s = sin(thb), cos_s = cos(thb);
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
  __syncthreads();
  if (threadIdx.x == 0) {
    double c1min = s_ext[0][0], c1max = s_ext[0][1], c2min = s_ext[0][2], c2max = s_ext[0][3];
    for (int w = 1; w < FW; w++) {
      c1min = fmin(c1min, s_ext[w][0]); c1max = fmax(c1max, s_ext[w][1]);
      c2min = fmin(c2min, s_ext[w][2]); c2max = fmax(c2max, s_ext[w][3]);
    }
    const int v = s / max_inst, jinst = s - v * max_inst;
    emit_box(thb, sin_s, cos_s, c1min, c1max, c2min, c2max, zmax, s, v, jinst, kc, root, m, calib,
             inst_class, inst_is_car, inst_box, inst_score, car_aspect_max, cap_rows, rows, n_rows,
             status);
  }
  }
}

}  // namespace

extern "C" int dfu3d_range_cluster(const double *px, const double *py, const int64_t *seg_base,
                                   const int32_t *seg_cnt, int32_t S, double R0, double Rd,
                                   int32_t *label, double *sx, double *sy, int32_t *si,
                                   int64_t pool_cap, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!px || !py || !seg_base || !seg_cnt || !label || !sx || !sy || !si) return DFU3D_EINVAL;
  if (S <= 0 || pool_cap <= 0 || !(R0 > 0.0) || !(Rd >= 0.0)) return DFU3D_EINVAL;
  // fast path: union-find over spatial cells (two LDS footprints by bounding-box area)
  // (an instance is walked by ONE workgroup, and the longest instance is the tail of the stage: 1024 threads)
  hipLaunchKernelGGL((k_range_cluster_grid<GRID_NC_SMALL, 0, 1024>), dim3(CL_CLASSES * S), dim3(1024), 0,
                     (hipStream_t)stream, px, py, (const long long *)seg_base, seg_cnt, R0, Rd,
                     label, sx, sy, si, (long long)pool_cap, 0, 0x7FFFFFFF);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_range_cluster_grid<GRID_NC_LARGE, GRID_NC_SMALL, 1024>), dim3(CL_CLASSES * S), dim3(1024), 0,
                     (hipStream_t)stream, px, py, (const long long *)seg_base, seg_cnt, R0, Rd,
                     label, sx, sy, si, (long long)pool_cap, 0, 0x7FFFFFFF);
  DFU3D_LAUNCH_CHECK();
  // fallback for instances wider than the largest grid: point-level union-find
  // (two LDS footprints; each kernel returns at once for segments it does not own)
  hipLaunchKernelGGL(k_range_cluster_small, dim3(S), dim3(CT), 0, (hipStream_t)stream, px, py,
                     (const long long *)seg_base, seg_cnt, R0, Rd, label, sx, sy, si,
                     (long long)pool_cap);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_range_cluster_large, dim3(S), dim3(CT), 0, (hipStream_t)stream, px, py,
                     (const long long *)seg_base, seg_cnt, R0, Rd, label, sx, sy, si,
                     (long long)pool_cap);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}


extern "C" int64_t dfu3d_lshape_fit_ws_doubles(int64_t pool_cap, int32_t cap_rows) {
  if (pool_cap <= 0 || cap_rows <= 0) return DFU3D_EINVAL;
  const int64_t cap_big = pool_cap / LDS_MEMBERS + 1;
  return 2 + (int64_t)8 * fit_cap_q(cap_rows) + (cap_big / 2 + 1) + cap_big * MAXTH;
}

extern "C" int dfu3d_lshape_fit(const double *px, const double *py, const double *pz,
                                const int32_t *label, const int64_t *seg_base,
                                const int32_t *seg_cnt, int32_t S, int32_t max_inst,
                                const float *calib, const int32_t *inst_class,
                                const int32_t *inst_is_car, const float *inst_box,
                                const float *inst_score, int32_t n_theta, double dtheta,
                                double car_aspect_max, double *sx, double *sy, int32_t *sroot,
                                int32_t cap_rows, double *rows, int32_t *n_rows,
                                uint32_t *status, double *fit_ws, int64_t pool_cap,
                                void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!px || !py || !pz || !label || !seg_base || !seg_cnt || !calib || !inst_class ||
      !inst_is_car || !inst_box || !inst_score || !sx || !sy || !sroot || !rows || !n_rows ||
      !status || !fit_ws)
    return DFU3D_EINVAL;
  if (S <= 0 || max_inst <= 0 || cap_rows <= 0 || n_theta <= 0 || pool_cap <= 0) return DFU3D_EINVAL;
  if (n_theta > MAXTH) return DFU3D_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  const int cap_big = (int)(pool_cap / LDS_MEMBERS + 1);
  const ViewCalib *vc = (const ViewCalib *)calib;
  if (dfu3d_fill_small_async(fit_ws, 16, nullptr, 0, nullptr, 0, st) != hipSuccess) return DFU3D_ELAUNCH;
  // F1: members of every cluster contiguous, one descriptor per cluster
  hipLaunchKernelGGL(k_fit_gather, dim3(3 * S), dim3(FT), 0, st, px, py, pz, label,
                     (const long long *)seg_base, seg_cnt, sx, sy, sroot, status, fit_ws,
                     cap_rows, cap_big);
  DFU3D_LAUNCH_CHECK();
  // F2: heading search + box, by cluster size (each kernel loops over the descriptors it owns)
  const int cap_q = fit_cap_q(cap_rows);
  const int g_tiny = cap_q / 4 + 1 < 2048 ? cap_q / 4 + 1 : 2048;
  hipLaunchKernelGGL(k_fit_tiny, dim3(g_tiny), dim3(256), 0, st, sx, sy, max_inst, vc, inst_class,
                     inst_is_car, inst_box, inst_score, n_theta, dtheta, car_aspect_max, cap_rows,
                     rows, n_rows, status, fit_ws, cap_big);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fit_medium, dim3(cap_q < 2048 ? cap_q : 2048), dim3(FT), 0, st, sx, sy, max_inst,
                     vc, inst_class, inst_is_car, inst_box, inst_score, n_theta, dtheta,
                     car_aspect_max, cap_rows, rows, n_rows, status, fit_ws, cap_big);
  DFU3D_LAUNCH_CHECK();
  const long long bi = (long long)cap_big * ((n_theta + BIGC_HB - 1) / BIGC_HB);   // items at most
  const int g2 = (int)(bi < 2048 ? bi : 2048);                                    // persistent: a ticket counter hands out the items
  hipLaunchKernelGGL(k_fit_big_cost, dim3(g2), dim3(BIGC_T), 0, st, sx, sy, n_theta, dtheta,
                     fit_ws, cap_rows, cap_big);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fit_big_box, dim3(cap_big < 2048 ? cap_big : 2048), dim3(FT), 0, st, sx, sy, max_inst,
                     vc, inst_class, inst_is_car, inst_box, inst_score,
                     n_theta, dtheta, car_aspect_max, cap_rows, rows, n_rows, status, fit_ws,
                     cap_big);
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
