// pixel_stage.hip -- camera-side kernels: fp64 back-projection of the dense
// depth map and spherical voxel sampling (la_sampling2 / la_sampling20).
//
// Reference semantics (my_loader.py:507-509, 532-557, 166-180, 247-275 + the
// spconv point-to-voxel leaf): points are visited in row-major
// pixel order; a voxel keeps the first `max_points` points that fall in it; the
// representative is the first argmin of one coordinate among those; voxels come
// out in first-seen order, at most `max_voxels` of them.
//
// GPU formulation (order-free, deterministic): because input order == pixel
// index, "first seen" is the minimum pixel index of a bin and "the first 100"
// are the 100 smallest pixel indices.  Per view a direct-addressed table over
// the reachable (theta,phi) bins holds {min key, min (key',pixel), count, first
// pixel, rep pixel}:
//   P1  per pixel: back-project, bin; atomics: count++, first=min(pix),
//       kmin=min(key), combo=min(key' | pix) where key' is the order-preserving
//       key cut to its top 64-b bits and b = bits of a pixel index.  Nothing is
//       written per pixel: the depth map is the only O(pixels) stream of the stage.
//       The same update maintains a 1-bit-per-pixel map of the CURRENT first
//       pixels: whoever lowers first[bin] toggles the bit of its own pixel and the
//       bit of the pixel it displaced (the old value atomicMin returns).  XOR
//       commutes, every pixel wins at most once and is displaced at most once, so
//       when the pass is over exactly the final first pixels are set -- whatever
//       the order of the updates.  The map is stored tile by tile (128 B per
//       64x16 tile), a tile collects its own bits in LDS and flushes them with one
//       contiguous 128-byte wave atomic; workgroups are dispatched in raster order,
//       so displacements (scattered atomics) are rare.
//   O*  repair (no-op kernels when the queue is empty): the views concerned are
//       classified once more in fp64 with the bin id stored per pixel, the
//       queued bins' pixel lists are gathered, the max_points-th smallest pixel
//       index T is radix-selected, kmin and the representative recomputed over
//       pix <= T
//   P3  one workgroup per view scans the bit map in raster order: exclusive
//       popcount prefix per 64-pixel row piece -> the rank of any first-pixel = the
//       voxel's position in first-seen order
//   P4  walks the bit map in raster order (4096 pixels per workgroup): every set bit
//       is a voxel, its rank is the prefix plus its place in the workgroup's list, so
//       consecutive threads write consecutive voxels.  The first pixel is classified
//       once more to find its bin (6 % of the pixels), then: representative (the
//       pixel p* of combo is the smallest pixel among those whose CUT key is
//       minimal, a superset of the exact arg-mins, so it is the representative iff
//       its exact key equals kmin), xyz, instance bits (ONE gather from a bit-packed
//       mask plane, or max_inst byte gathers); table entry reset.  Bins where the
//       check fails (two keys differ only below the cut: practically never) and bins
//       that saw more than max_points pixels are queued for the exact repair.
#include "common.hpp"
#include "dbg.hpp"

namespace {

constexpr int PB = 256;            // threads per pixel block
constexpr int PPT = 4;             // pixels per thread (one float4)
constexpr int PBLK = PB * PPT;     // 1024 pixels per block
constexpr uint32_t NOBIN = 0xFFFFFFFFu;
constexpr uint32_t OVF_FLAG = 0x80000000u;

// Five planes (structure of arrays): the atomics of the binning pass walk consecutive bins, and with one plane per
// field a wave's 64 atomics of one instruction fall into 4-8 cache lines.  (One 32-byte record per bin was tried:
// the voxel pass gained nothing and the binning pass went from 3.1 to 5.3 ms -- the atomic units work per line.)
struct Table {
  unsigned long long *kmin, *combo;
  uint32_t *cnt, *first, *rep;
};
__host__ __device__ inline Table table_view(void *base, int64_t E) {
  Table t;
  t.kmin = (unsigned long long *)base;
  t.combo = t.kmin + E;
  t.cnt = (uint32_t *)(t.combo + E);
  t.first = t.cnt + E;
  t.rep = t.first + E;
  return t;
}
static_assert(DFU3D_TABLE_ENTRY_BYTES == 28, "table entry");

// order-preserving key cut to its top (64 - pix_bits) bits | pixel index
__device__ __forceinline__ unsigned long long combo_word(unsigned long long okey, uint32_t pix,
                                                         int pix_bits) {
  if (DBG_COMBO_KEYBITS < 64) okey &= ~0ull << (64 - DBG_COMBO_KEYBITS);      // (test builds: provoke the repair of k_bp_vox)
  return ((okey >> pix_bits) << pix_bits) | (unsigned long long)pix;
}

__global__ void k_table_init(unsigned long long *kmin, unsigned long long *combo, uint32_t *cnt,
                             uint32_t *first, uint32_t *rep, int64_t E) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E;
       i += (int64_t)gridDim.x * blockDim.x) {
    kmin[i] = ~0ull;
    combo[i] = ~0ull;
    cnt[i] = 0u;
    first[i] = NOBIN;
    rep[i] = NOBIN;
  }
}

// back-project pixel and classify; returns table index or NOBIN; key = chosen
// coordinate (canonical +0.0).
__device__ __forceinline__ uint32_t pixel_bin(const ViewCalib &c, const Recip &rc,
                                              const dfu3d_bin_geom &g,
                                              int W, int pix, float d, int key_axis,
                                              double &key, bool &range_err) {
  if (!(d >= (float)g.depth_min) || !(d > 0.0f)) return NOBIN;   // my_loader.py:507-509
  const int row = pix / W, col = pix - row * W;
  double x, y, z;
  pixel_to_lidar(c, rc, col, row, d, x, y, z);
  if (!(z < g.z_max)) return NOBIN;                              // my_loader.py:540
  double s = x * x;                                              // my_loader.py:167
  s = s + y * y;
  s = s + z * z;
  const double r = sqrt(s);
  const double theta = acos(z / r);                              // :168
  if (!(theta > g.theta_min)) return NOBIN;                      // :175
  const double phi = atan(y / x);                                // :169
  const double cr = floor((r - g.rmin_r) / g.vsize_r);
  const double ct = floor((theta - g.rmin_t) / g.vsize_t);
  const double cp = floor((phi - g.rmin_p) / g.vsize_p);
  if (!(cr >= 0.0 && cr < (double)g.grid_r)) return NOBIN;
  if (!(ct >= 0.0 && ct < (double)g.grid_t)) return NOBIN;
  if (!(cp >= 0.0 && cp < (double)g.grid_p)) return NOBIN;
  const int it = (int)ct - g.t_lo, ip = (int)cp - g.p_lo;
  if (it < 0 || it >= g.t_n || ip < 0 || ip >= g.p_n) { range_err = true; return NOBIN; }
  key = (key_axis == 2) ? z : y;
  key += 0.0;                                   // -0.0 -> +0.0 (round to nearest), every other value unchanged
  return (uint32_t)(it * g.p_n + ip);
}

// recompute only the key of a pixel already known to be kept
__device__ __forceinline__ double pixel_key(const ViewCalib &c, const Recip &rc, int W, int pix, float d,
                                            int key_axis) {
  const int row = pix / W, col = pix - row * W;
  double x, y, z;
  pixel_to_lidar(c, rc, col, row, d, x, y, z);
  double key = (key_axis == 2) ? z : y;
  key += 0.0;                                   // -0.0 -> +0.0 (round to nearest), every other value unchanged
  return key;
}

__device__ __forceinline__ void load4(const float *p, int base, int n, float d[PPT]) {
  if (base + PPT <= n) {
    const float4 q = *(const float4 *)(p + base);
    d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
  } else {
#pragma unroll
    for (int k = 0; k < PPT; k++) d[k] = (base + k < n) ? p[base + k] : 0.0f;
  }
}

// ---- P1 ---------------------------------------------------------------------
// first-pixel bit map, tile-major: tile (64 wide x 16 high, the tiles of k_bp_bin) t owns words [32t, 32t+32),
// pixel (row, col) is bit (row & 15) * 64 + (col & 63) of its tile
constexpr int TILE_W = 64, TILE_H = 16;            // 1024 pixels, 256 threads x 4
__device__ __forceinline__ void toggle_first_bit(uint32_t *bitmap_v, int W, int tiles_x, uint32_t pix) {
  const int row = (int)pix / W, col = (int)pix - row * W;
  const int tile = (row >> 4) * tiles_x + (col >> 6);
  const int local = (row & 15) * TILE_W + (col & 63);
  atomicXor(&bitmap_v[tile * 32 + (local >> 5)], 1u << (local & 31));
}

// Occupancy of a view's table: one BYTE per SEGMENT of 64 consecutive entries (a line of every plane).  Whoever commits
// to a bin writes a 1 for its segment -- a plain store, every writer the same value; the voxel pass walks the occupied
// segments and nothing else.  (One BIT per segment set with atomicOr was built first: every tile of an image region
// hit the same few words, and the binning pass went from 2.8 to 8.4 ms on those same-address atomics.)
constexpr int SEG_SHIFT = 6;
// (segments of 64 entries of the FLAT bin index, whatever the row length: every load of the voxel pass is one aligned
// line per plane; segments cut along the theta rows -- 25 per row, the last one partial -- measured 1.25 ms against 1.15;
// segments of 32 / 16 entries, two / four per wave: the voxel pass 0.95 / 0.96 ms against 0.96, the scan 7 / 19 us longer)
__device__ __forceinline__ void mark_segment(uint8_t *occ_v, uint32_t b) { occ_v[b >> SEG_SHIFT] = 1; }
// table update of the exact (tier-2) classification
__device__ __forceinline__ void commit_pixel(const Table &T, int64_t e, int pix, double key, int pix_bits,
                                             uint32_t *bitmap_v, int W, int tiles_x) {
  atomicAdd(&T.cnt[e], 1u);
  const uint32_t oldf = atomicMin(&T.first[e], (uint32_t)pix);
  atomicMin(&T.kmin[e], ordered_key(key));
  atomicMin(&T.combo[e], combo_word(ordered_key(key), (uint32_t)pix, pix_bits));
  if (oldf > (uint32_t)pix) {                      // this pixel is the bin's first pixel now ...
    toggle_first_bit(bitmap_v, W, tiles_x, (uint32_t)pix);
    if (oldf != NOBIN) toggle_first_bit(bitmap_v, W, tiles_x, oldf);   // ... and the one it displaced no longer is
  }
}

// ---- tier 1 of the classification: boundary tables instead of inverse trigonometry ------------------
// The reference bins theta = acos(z/r) and phi = atan(y/x) (my_loader.py:166-177).  Both functions are monotone, so
// "theta lies in bin k" is the same statement as "q = -z/r lies between -cos of the bin's two edges", and likewise
// for phi with q = t / (1 + |t|), t = y/x = tan(phi) -- computed as y sgn(x) / (|x| + |y|), stable for every
// direction, and with 1/2 <= dq/dphi <= 1.  Tier 1 therefore never evaluates acos / atan: per axis a table over q
// (uniform cells of width w, built on the device in fp64 by k_bp_tables, float32 entries) gives for the cell of q the
// ONE bin edge E nearest to the cell's centre (in q-space) and its index k: q is above or below E, hence in bin k or
// k-1, and the answer is taken only when |q - E| > delta, delta bounding |q - q of the fp64 reference| (see
// classify_fast).  That is enough because the builder keeps an entry only if every OTHER edge is farther than
// reach + TAB_DMAX from the cell's centre, reach = (1/2 + TAB_SLOP_W) w covering the cell and the rounding of the cell
// index (<= 0.02 cells: one FMA on |q| <= 1; make_fast_geom checks the bound): with delta < TAB_DMAX -- tested once
// per pixel -- no other edge is within delta of q.  Cells that fail (bins narrower than about 2.5e-4 in q-space: theta
// within ~0.1 rad of a pole for the product's 0.002 rad bins), and the two sentinel cells around the table (q outside
// it, NaN) hold E = NaN: every comparison fails, "undecided".  Edges outside the range of the angle are +-inf.
// (Round 2's cells held three consecutive edges and took four comparisons per axis in a 1 MB table; this form takes
// one, from ~80 KB.)
constexpr int TAB_T_MAX = 65534, TAB_P_MAX = 16382;    // cells per axis (8 B each, + 2 sentinels)
constexpr double TAB_SLOP_W = 1.0 / 16;
constexpr float TAB_DMAX = 1.0e-4f;
constexpr uint32_t AMBIG = 0xFFFFFFFEu;

struct FastGeom {
  float r_lo, r_hi, z_max, q_tmin;       // q_tmin = -cos(theta_min) (-inf / +inf outside (0, pi))
  float tinv, tc1, pinv, pc1;            // table index of q: (int)(q * inv + c1), c1 = 1 - q0 * inv (cell 0 is a sentinel)
  float dmax, pad0;                      // deltas at or above this are "undecided" (TAB_DMAX, or 0: no tier 1)
  int tJ, pJ;                            // cells per axis without the sentinels
  double tq0d, twd, pq0d, pwd;           // origins and cell widths in fp64, for the table builder
  double q_tmin_d;                       // -cos(theta_min) in fp64 (tier 1.5)
  int mid_ok, pad1;                      // the fp64 edge tables of tier 1.5 fit the scratch
};
inline FastGeom make_fast_geom(const dfu3d_bin_geom &g) {
  const double pi = 3.14159265358979323846;
  FastGeom f;
  // r certain only for the 1-cell grid and well inside it; otherwise an empty interval
  f.r_lo = (g.grid_r == 1) ? __builtin_fmaxf((float)(g.rmin_r + 2e-3), 1e-3f) : 1.0f;
  f.r_hi = (g.grid_r == 1) ? __builtin_fminf((float)((g.rmin_r + g.vsize_r) * 0.9998), 1e15f) : 0.0f;
  f.z_max = (float)g.z_max;
  f.q_tmin = g.theta_min <= 0.0 ? -INFINITY : (g.theta_min >= pi ? INFINITY : (float)-__builtin_cos(g.theta_min));
  f.q_tmin_d = g.theta_min <= 0.0 ? -(double)INFINITY : (g.theta_min >= pi ? (double)INFINITY : -__builtin_cos(g.theta_min));
  auto cells = [](double lo, double hi, double min_width, int cap, double &q0, double &w) {
    if (!(hi > lo) || !(min_width > 0.0)) { q0 = lo; w = 1.0; return 1; }
    double J = __builtin_ceil((hi - lo) / (0.5 * min_width));
    if (!(J >= 1.0)) J = 1.0;
    if (J > (double)cap) J = (double)cap;
    q0 = lo; w = (hi - lo) / J;
    return (int)J;
  };
  {   // theta: the window's edges, clipped to where tier 1 works at all (sin(theta)^2 > 1e-4)
    const double a_lo = __builtin_fmax(g.rmin_t + (double)g.t_lo * g.vsize_t, 0.0100002);
    const double a_hi = __builtin_fmin(g.rmin_t + (double)(g.t_lo + g.t_n) * g.vsize_t, pi - 0.0100002);
    // cells half as wide as a bin at sin(theta) = 0.2 or at the window's narrowest bin, whichever is wider (narrower
    // bins get cells that hold no decision)
    const double smin = __builtin_fmax(__builtin_fmin(__builtin_sin(a_lo), __builtin_sin(a_hi)), 0.2);
    f.tJ = cells(-__builtin_cos(a_lo), -__builtin_cos(a_hi), g.vsize_t * smin, TAB_T_MAX, f.tq0d, f.twd);
  }
  {   // phi: q = tan(phi) / (1 + |tan(phi)|) in (-1, 1); dq/dphi lies in [1/2, 1]
    const double b_lo = __builtin_fmax(g.rmin_p + (double)g.p_lo * g.vsize_p, -pi / 2);
    const double b_hi = __builtin_fmin(g.rmin_p + (double)(g.p_lo + g.p_n) * g.vsize_p, pi / 2);
    auto qphi = [](double P) { return __builtin_sin(P) / (__builtin_fabs(__builtin_cos(P)) + __builtin_fabs(__builtin_sin(P))); };
    f.pJ = cells(qphi(b_lo), qphi(b_hi), 0.5 * g.vsize_p, TAB_P_MAX, f.pq0d, f.pwd);
  }
  f.tinv = (float)(1.0 / f.twd); f.tc1 = (float)(1.0 - f.tq0d / f.twd);
  f.pinv = (float)(1.0 / f.pwd); f.pc1 = (float)(1.0 - f.pq0d / f.pwd);
  f.dmax = TAB_DMAX;
  // rounding of the cell index, in cells: the float32 constants and the FMA, each 2^-24 relative to a magnitude of at
  // most (1 + |q0|) / w + 1.  A geometry with bins so narrow that this eats the slop gets no tier 1 at all.
  const double slop_t = 1.8e-7 * ((1.0 + __builtin_fabs(f.tq0d)) / f.twd + 1.0), slop_p = 1.8e-7 * ((1.0 + __builtin_fabs(f.pq0d)) / f.pwd + 1.0);
  if (!(__builtin_fmax(slop_t, slop_p) <= TAB_SLOP_W)) f.dmax = 0.0f;
  if (g.t_n >= (1 << 16) || g.p_n >= (1 << 16)) f.dmax = 0.0f;          // k_bp_bin packs the window coordinates into 16 + 16 bits (classify_fast forms the bin index with a 24-bit multiply)
  // tier 1.5: one fp64 edge per bin boundary of the window behind the float32 tables, if the scratch holds them
  f.mid_ok = (f.dmax > 0.0f) && ((int64_t)f.tJ + f.pJ + 4 + (int64_t)g.t_n + g.p_n + 2 <= 2 * (int64_t)(TAB_T_MAX + TAB_P_MAX));
  f.pad1 = 0;
  if (DBG_NO_MID) f.mid_ok = 0;         // (test build: every pixel / voxel tier 1 leaves undecided takes the full fp64 path)
  f.pad0 = 0.0f;
  return f;
}

// one thread per table entry (tJ + 2 of theta, then pJ + 2 of phi): (E, k) = the edge nearest to the cell's centre and
// its index, or (NaN, 0)
__device__ __forceinline__ const double *edge_tab_t(const float2 *tab, const FastGeom &fg) { return (const double *)(tab + fg.tJ + fg.pJ + 4); }
__device__ __forceinline__ const double *edge_tab_p(const float2 *tab, const FastGeom &fg, const dfu3d_bin_geom &g) {
  return edge_tab_t(tab, fg) + g.t_n + 1;
}
inline int tables_threads(const FastGeom &fg, const dfu3d_bin_geom &g) {
  return fg.tJ + fg.pJ + 4 + (fg.mid_ok ? g.t_n + g.p_n + 2 : 0);
}
__global__ void k_bp_tables(dfu3d_bin_geom g, FastGeom fg, float2 *__restrict__ tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double pi = 3.14159265358979323846;
  if (i >= fg.tJ + fg.pJ + 4) {
    // tier 1.5 (pixel_bin_mid): edge k of the window in the space the fp64 test works in -- -cos(B) for theta,
    // tan(B) for phi (B = rmin + k * vsize as the reference's floor((angle - rmin) / vsize) implies it)
    const int e = i - (fg.tJ + fg.pJ + 4);
    if (!fg.mid_ok || e >= g.t_n + g.p_n + 2) return;
    double *et = (double *)(tab + fg.tJ + fg.pJ + 4);
    if (e <= g.t_n) {
      const double B = g.rmin_t + (double)(g.t_lo + e) * g.vsize_t;
      et[e] = B <= 0.0 ? -(double)INFINITY : (B >= pi ? (double)INFINITY : -cos(B));
    } else {
      const double B = g.rmin_p + (double)(g.p_lo + (e - g.t_n - 1)) * g.vsize_p;
      et[e] = B <= -pi / 2 ? -(double)INFINITY : (B >= pi / 2 ? (double)INFINITY : tan(B));
    }
    return;
  }
  const bool is_t = i < fg.tJ + 2;
  const int jj = is_t ? i : i - (fg.tJ + 2);
  const int J = is_t ? fg.tJ : fg.pJ;
  if (jj == 0 || jj == J + 1) { tab[i] = make_float2(__int_as_float(0x7FC00000), __int_as_float(0)); return; }
  const double w = is_t ? fg.twd : fg.pwd;
  const double c = (is_t ? fg.tq0d : fg.pq0d) + ((double)(jj - 1) + 0.5) * w;
  const double rmin = is_t ? g.rmin_t : g.rmin_p, vs = is_t ? g.vsize_t : g.vsize_p;
  auto edge = [&](long long k) -> double {          // edge k of the axis in q-space
    const double B = rmin + (double)k * vs;
    if (is_t) return B <= 0.0 ? -INFINITY : (B >= pi ? INFINITY : -cos(B));
    return B <= -pi / 2 ? -INFINITY : (B >= pi / 2 ? INFINITY : sin(B) / (fabs(cos(B)) + fabs(sin(B))));
  };
  // the angle at the centre: theta = acos(-q); phi = atan(q / (1 - |q|))
  const double a = is_t ? acos(fmin(fmax(-c, -1.0), 1.0)) : (fabs(c) < 1.0 ? atan(c / (1.0 - fabs(c))) : copysign(pi / 2, c));
  double kf = floor((a - rmin) / vs);
  kf = fmin(fmax(kf, -1.0e9), 1.0e9);
  long long kn = (long long)kf;                     // edge(kn) <= c < edge(kn + 1), up to rounding:
  for (int r = 0; r < 8 && edge(kn) > c; r++) kn--;
  for (int r = 0; r < 8 && edge(kn + 1) <= c; r++) kn++;
  const double below = c - edge(kn), above = edge(kn + 1) - c;        // >= 0 (inf for an edge outside the range)
  const bool up = above < below;
  const long long k1 = up ? kn + 1 : kn;
  const double others = fmin(up ? below : above, up ? edge(kn + 2) - c : c - edge(kn - 1));
  const bool ok = (edge(kn) <= c) && (c < edge(kn + 1)) && others >= (0.5 + TAB_SLOP_W) * w + 1.001 * (double)TAB_DMAX;
  tab[i] = ok ? make_float2((float)edge(k1), __int_as_float((int)k1)) : make_float2(__int_as_float(0x7FC00000), __int_as_float(0));
}

// bin of q from the axis' table; false = undecided (no branches: garbage in -- NaN, an index off the table -- meets
// a NaN entry and fails the comparison).  The index is clamped as an unsigned number: a negative one lands on the upper
// sentinel like one that is too large.
__device__ __forceinline__ bool tab_bin(const float2 *__restrict__ tab, float inv_w, float c1, int J, float q,
                                        float delta, uint32_t &k) {
  const uint32_t j = min((uint32_t)(int)__fmaf_rn(q, inv_w, c1), (uint32_t)(J + 1));
  const float2 e = *(const float2 *)((const char *)tab + (j << 3));      // 32-bit offset from a uniform base
  const float t = q - e.x;
  k = (uint32_t)(__float_as_int(e.y) + (__float_as_int(t) >> 31));       // below the edge: the bin under it
  return fabsf(t) > delta;
}

// ---- float32 back-projection estimate ------------------------------------------------------------
// calibration_kitti.py:134-144 + 89-102 collapse, per LiDAR coordinate j, to
//   p_j = d * (a_j*u + b_j*v + c_j) + e_j      a_j = M0j/fu, b_j = M1j/fv, c_j = M2j - cu*a_j - cv*b_j,
//                                             e_j = tx*M0j + ty*M1j + M3j          (M = rows of Minv)
// k_bp_prep forms the twelve constants per view in fp64 and rounds them to float32; a pixel then costs
// nine float32 FMAs.  ERR bounds |estimate - fp64 value| per coordinate: the constants carry 2^-24 relative
// rounding each, the three FMAs one rounding each, so |err| <= 2^-24 (3 d W_j + |e_j| + |p_j|) with
// W_j = |a_j| u + |b_j| v + |c_j|; the bound used is 2^-22 (d * wsum + esum + |p|_1), wsum / esum being the sums of
// the per-view maxima of W_j / |e_j| -- at least 1.3x that, and dfu3d_selftest_backproject measures
// the real ratio on the device (the GPU test requires < 0.5).
struct FastCal {
  float a[3], b[3], c[3], e[3];
  float wsum, esum, pad0, pad1;
  double rfu, rfv;                       // make_recip() of the view, formed once here instead of by every thread
};
static_assert(sizeof(FastCal) == 80, "FastCal");
__device__ __forceinline__ Recip recip_of(const FastCal &f) {
  Recip r;
  r.rfu = f.rfu; r.rfv = f.rfv;
  return r;
}

__global__ void k_bp_prep(const ViewCalib *__restrict__ calib, int V, int H, int W, FastCal *__restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const ViewCalib c = calib[v];
  FastCal f;
  double wsum = 0.0, esum = 0.0;
  for (int j = 0; j < 3; j++) {
    const double a = (double)c.Minv[0 + j] / (double)c.fu, b = (double)c.Minv[3 + j] / (double)c.fv;
    const double cc = (double)c.Minv[6 + j] - (double)c.cu * a - (double)c.cv * b;
    const double e = (double)c.tx * (double)c.Minv[0 + j] + (double)c.ty * (double)c.Minv[3 + j] + (double)c.Minv[9 + j];
    f.a[j] = (float)a; f.b[j] = (float)b; f.c[j] = (float)cc; f.e[j] = (float)e;
    // the terms of c_j cancel against a_j*u + b_j*v: bound W_j by the magnitudes of its parts
    wsum += fabs(a) * (double)(W > 1 ? W - 1 : 1) + fabs(b) * (double)(H > 1 ? H - 1 : 1) + fabs((double)c.Minv[6 + j]) +
            fabs((double)c.cu * a) + fabs((double)c.cv * b);
    esum += fabs((double)c.tx * (double)c.Minv[0 + j]) + fabs((double)c.ty * (double)c.Minv[3 + j]) + fabs((double)c.Minv[9 + j]);
  }
  f.wsum = (float)(wsum * 1.0000002);
  f.esum = (float)(esum * 1.0000002);
  f.pad0 = f.pad1 = 0.0f;
  const Recip r = make_recip(c);
  f.rfu = r.rfu; f.rfv = r.rfv;
  out[v] = f;
}

__device__ __forceinline__ void backproject_f32(const FastCal &fc, int col, int row, float d, float &xf, float &yf,
                                                float &zf, float &err) {
  const float u = (float)col, v = (float)row;
  const float w0 = __fmaf_rn(fc.a[0], u, __fmaf_rn(fc.b[0], v, fc.c[0]));
  const float w1 = __fmaf_rn(fc.a[1], u, __fmaf_rn(fc.b[1], v, fc.c[1]));
  const float w2 = __fmaf_rn(fc.a[2], u, __fmaf_rn(fc.b[2], v, fc.c[2]));
  xf = __fmaf_rn(d, w0, fc.e[0]);
  yf = __fmaf_rn(d, w1, fc.e[1]);
  zf = __fmaf_rn(d, w2, fc.e[2]);
  err = 2.3841858e-07f * (__fmaf_rn(fabsf(d), fc.wsum, fc.esum) + fabsf(xf) + fabsf(yf) + fabsf(zf));
}

// Tier 1 of the classification, all in float32.  From the back-projection estimate (coordinates within `err` of
// the reference's fp64 ones) it forms q_t = -zf/rf and q_p = yf/xf and looks both up in the edge tables; a decision
// is taken only when the estimate is farther from every boundary involved (z_max, theta_min, the r range, the bin
// edges) than a bound on |estimate - fp64 value|, in which case the fp64 path of pixel_bin() decides identically;
// everything else returns AMBIG and is classified by pixel_bin() (k_bp_bin_amb).  Error budget:
//   q_t   v_rsq at 1 ulp, the sum of squares and one product: <= 5.1e-7 relative, i.e. 5.1e-7 absolute
//         (|q_t| <= 1); the coordinate error turns the direction by at most turn = sqrt(3) err / r, which moves the
//         cosine by at most turn; table entries are rounded to float32 (6e-8).  delta_t = 1.5e-6 + 1.05 turn.
//   q_p   one sum, v_rcp, one product: <= 2.4e-7 (|q_p| < 1), table rounding 6e-8; the coordinate error moves phi by
//         at most turn / sin(theta) and q_p by no more than that (dq/dphi <= 1).
//         delta_p = 1.5e-6 + 1.1 turn / sin(theta).
// dfu3d_selftest_classify() runs this function against pixel_bin() on random pixels of the real geometry and
// calibration: the GPU tests require zero disagreements among the decided ones.
// The exact fp64 coordinate that serves as the voxel key is computed only for pixels that are kept (want_key).
// N pixels of one image row, straight-line code (the decisions of the N pixels are independent; written this way the
// compiler interleaves their transcendental and table latencies and pairs their float32 arithmetic): res[k] = NOBIN
// (certainly not binned), AMBIG (tier 2 decides) or the table index, with it / ip the window coordinates of the bin.
template <int N>
__device__ __forceinline__ void classify_fast(const FastCal &fc, const dfu3d_bin_geom &g, const FastGeom &fg,
                                              const float2 *__restrict__ tab, int row, const int (&col)[N],
                                              const float (&d)[N], uint32_t (&res)[N], int (&it)[N], int (&ip)[N]) {
  float xf[N], yf[N], zf[N], err[N];
  bool dead[N];
  bool any_live = false;
#pragma unroll
  for (int k = 0; k < N; k++) {
    backproject_f32(fc, col[k], row, d[k], xf[k], yf[k], zf[k], err[k]);
    const bool dok = (d[k] >= (float)g.depth_min) && (d[k] > 0.0f);      // my_loader.py:507-509
    dead[k] = !dok || (zf[k] > fg.z_max + err[k]);                       // certainly z >= z_max (my_loader.py:540)
    any_live = any_live || !dead[k];
    res[k] = NOBIN; it[k] = 0; ip[k] = 0;
  }
  if (!any_live) return;
#pragma unroll
  for (int k = 0; k < N; k++) {
    // `ok` collects the conditions of a decision with & (no short circuits: the masks are combined on the scalar
    // unit); a NaN anywhere makes one of them false
    bool ok = zf[k] < fg.z_max - err[k];
    const float r2 = xf[k] * xf[k] + yf[k] * yf[k] + zf[k] * zf[k];
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float rf = r2 * ir;                                            // (NaN for r2 = 0 or inf)
    // r bin: certain only well inside [rmin_r, rmin_r + vsize_r) and for the 1-cell grid; r_lo >= 1e-3 and
    // r_hi <= 1e15 (make_fast_geom), so `rin` also says that r is an ordinary number
    const float er = 2.5f * err[k];                                      // |rf - r| <= sqrt(3) err + rounding (3 ulp of r <= 0.75 err)
    const bool rin = (rf - er > fg.r_lo) & (rf + er < fg.r_hi);
    ok = ok & rin;
    const float cz = zf[k] * ir;
    const float s2 = __fmaf_rn(-cz, cz, 1.0f);
    ok = ok & (s2 > 1e-4f);                                              // near the poles
    const float turn = err[k] * ir * 1.7320510f;                         // direction error: |(dx,dy,dz)| <= sqrt(3) err
    const float dq = 1.5e-6f + 1.05f * turn;
    const float qt = -cz;
    const bool th_out = rin & (qt < fg.q_tmin - dq);                     // certainly theta <= theta_min (my_loader.py:175)
    ok = ok & (qt > fg.q_tmin + dq);
    uint32_t kt, kp;
    ok = ok & tab_bin(tab, fg.tinv, fg.tc1, fg.tJ, qt, dq, kt);
    ok = ok & (fabsf(xf[k]) > __fmaf_rn(8.0f, err[k], 1e-20f));          // the sign of x decides the branch of atan(y/x)
    const float qa = yf[k] * __builtin_amdgcn_rcpf(fabsf(xf[k]) + fabsf(yf[k]));
    const float qp = xf[k] < 0.0f ? -qa : qa;
    const float dp = 1.5e-6f + 1.1f * turn * __builtin_amdgcn_rsqf(s2);  // (s2 > 1e-4 or not ok); dp >= dq
    ok = ok & tab_bin(tab + fg.tJ + 2, fg.pinv, fg.pc1, fg.pJ, qp, dp, kp);
    ok = ok & (dp < fg.dmax);
    const uint32_t itk = kt - (uint32_t)g.t_lo, ipk = kp - (uint32_t)g.p_lo;      // (wraps for a bin below the window)
    ok = ok & (itk < (uint32_t)g.t_n) & (ipk < (uint32_t)g.p_n);
    it[k] = (int)itk;                                                    // (meaningful only for a decided, binned pixel)
    ip[k] = (int)ipk;
    // (a 24-bit multiply: v_mul_lo_u32 / v_mad_u64_u32 issue at a quarter of the rate; make_fast_geom switches tier 1 off
    // for a window of 2^24 or more bins along an axis)
    res[k] = (dead[k] | th_out) ? NOBIN : (ok ? __umul24(itk, (uint32_t)g.p_n) + ipk : AMBIG);
  }
}

__device__ __forceinline__ uint32_t pixel_bin_fast(const ViewCalib &c, const Recip &rc, const FastCal &fc,
                                                   const dfu3d_bin_geom &g, const FastGeom &fg,
                                                   const float2 *__restrict__ tab,
                                                   int row, int col, float d, const KeyCol &kc, bool want_key,
                                                   double &key, int &it_out, int &ip_out) {
  const int cols[1] = {col};
  const float ds[1] = {d};
  uint32_t res[1];
  int it[1], ip[1];
  classify_fast<1>(fc, g, fg, tab, row, cols, ds, res, it, ip);
  it_out = it[0];
  ip_out = ip[0];
  if (want_key && res[0] < AMBIG) {
    key = pixel_to_lidar_axis(c, rc, kc, col, row, d);
    key += 0.0;                                   // -0.0 -> +0.0 (round to nearest), every other value unchanged
  }
  return res[0];
}

// Tier 1.5, for the first pixels of P4 that tier 1 leaves undecided (0.7 %): the reference's own fp64 coordinates, r
// and z/r, y/x -- everything but acos and atan, which cost ten times the rest and 150 vector registers -- and the
// comparison moved to the other side of the monotone function: theta >= B <=> -z/r >= -cos(B), phi >= B <=> y/x >=
// tan(B), against one fp64 edge per bin boundary (k_bp_tables).  Which boundary: the float32 tables of tier 1, looked
// up with the float32 image of the fp64 q -- the pixel's bin is one of the two that meet at the edge the cell holds
// (the cell's validity rule with delta = 1e-7 << TAB_DMAX).  What can make the reference's floor((angle - rmin) /
// vsize) disagree with the comparison in q-space is rounding: of acos / atan (1 ulp), of the subtraction and the
// division (< 2e-15 rad for |angle - rmin| <= 10), of cos / tan in the table (1-2 ulp), of the quotient (1 ulp):
// < 1e-14 in all.  A decision is taken only when q is farther than 1e-12 from the edge (1e-12 (1 + t^2) for
// t = tan(phi): d phi = dt / (1 + t^2)), one hundred times that; the rest (1e-9 of the undecided pixels), NaN /
// infinite quotients, cells without a decision and bins outside the window are left to pixel_bin().  z_max, the r
// bin and depth_min are the reference's own comparisons on the reference's own numbers.
// dfu3d_selftest_classify counts a tier-1.5 decision that differs from pixel_bin() as a disagreement.
struct MidOut { uint32_t it, ip; double y, z; };    // window coordinates of the bin, the two coordinates a key can be
__device__ __forceinline__ uint32_t pixel_bin_mid(const ViewCalib &c, const Recip &rc, const dfu3d_bin_geom &g,
                                                  const FastGeom &fg, const float2 *__restrict__ tab, int W, int pix,
                                                  float d, bool &decided, MidOut *mo = nullptr) {
  // All the arithmetic first, then the two table cells together, then the two fp64 edges together: three dependent
  // memory round trips instead of five (one wave per workgroup of k_bp_bin runs this while three wait).  Indices are
  // clamped, so a pixel that turns out not to be binned at all reads harmlessly.
  const bool dok = (d >= (float)g.depth_min) && (d > 0.0f);      // my_loader.py:507-509
  const int row = pix / W, col = pix - row * W;
  double x, y, z;
  pixel_to_lidar(c, rc, col, row, d, x, y, z);
  double s = x * x;                                              // my_loader.py:167
  s = s + y * y;
  s = s + z * z;
  const double r = sqrt(s);
  const double qt = -(z / r);
  const double cr = floor((r - g.rmin_r) / g.vsize_r);
  const double t = y / x;
  const float tf = (float)t;
  const float qpf = tf * __builtin_amdgcn_rcpf(1.0f + fabsf(tf));        // (NaN for an infinite quotient: no decision)
  const uint32_t jt = min((uint32_t)(int)__fmaf_rn((float)qt, fg.tinv, fg.tc1), (uint32_t)(fg.tJ + 1));
  const uint32_t jp = min((uint32_t)(int)__fmaf_rn(qpf, fg.pinv, fg.pc1), (uint32_t)(fg.pJ + 1));
  const float2 et = tab[jt];
  const float2 ep = tab[fg.tJ + 2 + jp];
  const uint32_t i_t = (uint32_t)(__float_as_int(et.y) - g.t_lo);        // edge index inside the window: 0 .. t_n
  const uint32_t i_p = (uint32_t)(__float_as_int(ep.y) - g.p_lo);
  const double e_t = edge_tab_t(tab, fg)[min(i_t, (uint32_t)g.t_n)];
  const double e_p = edge_tab_p(tab, fg, g)[min(i_p, (uint32_t)g.p_n)];
  const double dt = qt - e_t, dp = t - e_p;
  const uint32_t it = i_t - (dt < 0.0 ? 1u : 0u), ip = i_p - (dp < 0.0 ? 1u : 0u);
  // the decisions, in the reference's order
  decided = true;
  if (!dok) return NOBIN;
  if (!(z < g.z_max)) return NOBIN;                              // my_loader.py:540
  if (qt < fg.q_tmin_d - 1e-12) return NOBIN;                    // theta <= theta_min for certain (:175)
  decided = qt > fg.q_tmin_d + 1e-12;                            // (false for NaN)
  if (decided && !(cr >= 0.0 && cr < (double)g.grid_r)) return NOBIN;
  decided = decided && (et.x == et.x) && (i_t <= (uint32_t)g.t_n) && (fabs(dt) > 1e-12);
  decided = decided && (ep.x == ep.x) && (i_p <= (uint32_t)g.p_n) && (fabs(dp) > 1e-12 * fma(t, t, 1.0));
  decided = decided && (it < (uint32_t)g.t_n) && (ip < (uint32_t)g.p_n);  // (a bin below / above the window: pixel_bin reports it)
  if (mo) { mo->it = it; mo->ip = ip; mo->y = y; mo->z = z; }
  return it * (uint32_t)g.p_n + ip;
}

// Pass 1 works on 2-D image tiles (TILE_W x TILE_H pixels, one float4 per
// thread): neighbouring pixels share spherical bins (~2.5 x 2.5 pixels per
// 0.002 rad bin), so the tile first aggregates count / first pixel / min key in
// an LDS window over the bins it touches and then issues one set of global
// atomics per touched bin instead of one per pixel (6x fewer for dense depth).
constexpr int WIN_T = 16, WIN_P = 48;              // LDS bin window (theta x phi)

constexpr int P1_AMB = 256;                        // undecided pixels a workgroup of k_bp_bin lists in LDS
constexpr int P1_OCC = 7;                          // workgroups per compute unit the register budget is cut for.  Round 3: 93 registers, five
                                                   // (budgets for 6 / 8 spilled: 3.98 / 6.11 ms against 3.23).  Round 4: a kept pixel carries 16 bits
                                                   // + 16 bits of window coordinates across the barrier and nothing else (its depth is read again, its
                                                   // key formed behind the barrier): 67 registers, seven waves per SIMD -- 2.85 -> 2.61 ms (six: 2.75;
                                                   // eight, with 12 B of scratch: 2.64)
constexpr int RPT = 2;                             // rows per thread: a workgroup's tile is TILE_W x (RPT * TILE_H) pixels --
                                                   // the window set-up, its flush and the reductions are paid once per 2048 pixels
__global__ __launch_bounds__(PB, P1_OCC) void k_bp_bin(
    const float *__restrict__ depth, const ViewCalib *__restrict__ calib,
    const FastCal *__restrict__ fastcal, const float2 *__restrict__ tab, dfu3d_bin_geom g, FastGeom fg, int W, int H,
    int tiles_x, int tiles_y, int key_axis,
    int64_t E_view, void *table, int64_t E_total, int *__restrict__ n_amb, uint32_t *__restrict__ amb_list,
    int pix_bits, uint32_t *__restrict__ bitmap, int BW, uint8_t *__restrict__ occ, int OW) {
  __shared__ uint32_t s_bits[32 * RPT];           // this workgroup's piece of the first-pixel bit map (RPT bit-map tiles)
  // the workgroup's undecided pixels: a short list (0.7 % of the pixels are undecided: 14 of a tile's 2048; a list for
  // all 2048 took 8 KB of the workgroup's LDS); what does not fit goes to the global list one pixel at a time
  __shared__ uint32_t s_amb[P1_AMB];
  __shared__ unsigned long long s_kmin[WIN_T * WIN_P], s_combo[WIN_T * WIN_P];
  __shared__ uint32_t s_cnt[WIN_T * WIN_P], s_first[WIN_T * WIN_P];
  __shared__ int s_namb, s_t0, s_p0;
  DBG_T_START();
  DBG_T_COUNT(8);
  const int v = blockIdx.y;
  const int HW = H * W;
  const ViewCalib c = calib[v];
  const FastCal fc = fastcal[v];
  const Recip rc = recip_of(fc);
  const KeyCol kcol = load_key_col(calib + v, key_axis);
  const Table T = table_view(table, E_total);
  const int64_t tb0 = (int64_t)v * E_view;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;     // ty counts RPT bit-map tile rows
  // (workgroup order, measured with tools/ab_builds.py: tiles column by column 2.60 ms against 2.57; the same tile of consecutive
  // views side by side -- no two workgroups in flight on one table -- 2.89: the table atomics live on lines their neighbours
  // have just brought to the memory side's cache)
  // row of the thread inside the tile's first TILE_H rows.  A wave holds four rows; FOUR APART (wave w: rows w, w + 4, w + 8,
  // w + 12), not adjacent: a bin is about three pixel rows tall, and lanes 16 apart that sit in adjacent rows hit the same
  // window slot in the same LDS atomic instruction, which then runs once per such lane (2.56 -> 2.53 ms, tools/ab_builds.py)
  const int trow = (int)(((threadIdx.x >> 4) & 3u) * 4u + (threadIdx.x >> 6));
  const int row0 = ty * (RPT * TILE_H) + trow;
  const int col = tx * TILE_W + (threadIdx.x & 15) * PPT;
  if (threadIdx.x == 0) { s_namb = 0; s_t0 = 0x7FFFFFFF; s_p0 = 0x7FFFFFFF; }
  if (threadIdx.x < 32 * RPT) s_bits[threadIdx.x] = 0u;
  uint32_t *bitmap_v = bitmap + (size_t)v * BW;
  uint8_t *occ_v = occ + (size_t)v * OW * 4;
  // The window keeps a bin's first pixel as its index INSIDE the workgroup's tile (row-major over the tile's
  // RPT * TILE_H rows of TILE_W pixels: the same order as the global index for pixels of one tile), which is also the
  // bit it has in s_bits; pixel `loc` of this tile became a bin's first pixel, oldf (global) is what it displaced
  auto new_first = [&](uint32_t loc, uint32_t oldf) {
    atomicOr(&s_bits[loc >> 5], 1u << (loc & 31));
    if (oldf != NOBIN) toggle_first_bit(bitmap_v, W, tiles_x, oldf);
  };
  const uint32_t pix00 = (uint32_t)(ty * (RPT * TILE_H) * W + tx * TILE_W);        // the tile's first pixel
  auto global_pix = [&](uint32_t loc) { return pix00 + __umul24(loc >> 6, (uint32_t)W) + (loc & 63u); };
  static_assert(TILE_W == 64, "tile-local pixel index");
  for (int i = threadIdx.x; i < WIN_T * WIN_P; i += PB) { s_kmin[i] = ~0ull; s_combo[i] = ~0ull; s_cnt[i] = 0u; s_first[i] = NOBIN; }
  __syncthreads();
  DBG_T(0);                                        // set-up: records, window reset, barrier
  // what a kept pixel carries across the origin's barrier: its window coordinates in ONE word (theta | phi << 16; NOBIN: not
  // kept) and its depth -- the exact fp64 key is formed where it is used, behind the barrier.  (Until round 4: the bin
  // index, the two coordinates and the key, 40 registers instead of 16, and the kernel sat at 93 of them, five waves
  // per SIMD.)
  uint32_t tp[RPT][PPT];
  bool inside[RPT];
  int tmin = 0x7FFFFFFF, pmin = 0x7FFFFFFF;
  uint32_t amb_mask = 0u;                         // bit r * PPT + k: pixel k of the thread's row r is undecided
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int row = row0 + r * TILE_H;
    inside[r] = (row < H) && (col < W);
#pragma unroll
    for (int k = 0; k < PPT; k++) tp[r][k] = NOBIN;
    if (!inside[r]) continue;
    const float *dv = depth + (size_t)v * HW;
    float d[PPT];
    load4(dv + (size_t)row * W, col, W, d);          // (columns past the image come back as depth 0: not binned)
    int cols[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) cols[k] = col + k;
    uint32_t res[PPT];
    int its[PPT], ips[PPT];
    classify_fast<PPT>(fc, g, fg, tab, row, cols, d, res, its, ips);
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const uint32_t b = res[k];
      if (b == AMBIG) {
        amb_mask |= 1u << (r * PPT + k);
      } else if (b != NOBIN) {
        tp[r][k] = (uint32_t)its[k] | ((uint32_t)ips[k] << 16);
        tmin = min(tmin, its[k]); pmin = min(pmin, ips[k]);
      }
    }
  }
  // the thread's undecided pixels (0.7 % of the pixels): one request for list slots per thread, not one per pixel
  if (amb_mask) {
    int slot = atomicAdd(&s_namb, (int)__popc(amb_mask));      // block-local list (LDS)
    while (amb_mask) {
      const int bit = __ffs((int)amb_mask) - 1;
      amb_mask &= amb_mask - 1u;
      const uint32_t pix = (uint32_t)((row0 + (bit / PPT) * TILE_H) * W + col + (bit % PPT));
      if (slot < P1_AMB) s_amb[slot] = pix;
      else amb_list[(size_t)v * HW + atomicAdd(&n_amb[v], 1)] = pix;
      slot++;
    }
  }
  DBG_T(1);                                        // depth loads, classification, exact keys (thread 0's wave)
  // window origin: wave minimum first (all lanes), then one LDS atomic per wave -- 256 lanes on two
  // addresses serialise
  tmin = wave_min_i_dpp(tmin);
  pmin = wave_min_i_dpp(pmin);
  if (lane_id() == 0 && tmin != 0x7FFFFFFF) { atomicMin(&s_t0, tmin); atomicMin(&s_p0, pmin); }
  __syncthreads();
  const int t0 = s_t0, p0 = s_p0;
  DBG_T(2);                                        // origin reduction + barrier (= waiting for the slowest wave's classification)
  // every kept pixel goes to the LDS window on its own.  (Round 2 merged the runs of equal bins among a thread's four
  // consecutive pixels first: fewer LDS atomics, but the bookkeeping of the runs took more vector instructions than the
  // atomics it saved -- the kernel is bound by vector issue, not by the LDS.)
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    if (!inside[r]) continue;
    const uint32_t base = (uint32_t)((row0 + r * TILE_H) * W + col);
    float dk[PPT];                                   // (the row's depths once more: a hit in the cache, eight registers less across the barrier)
    load4(depth + (size_t)v * HW + (size_t)(row0 + r * TILE_H) * W, col, W, dk);
    const uint32_t loc0 = (uint32_t)((trow + r * TILE_H) * TILE_W + (threadIdx.x & 15) * PPT);
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      if (tp[r][k] == NOBIN) continue;
      const uint32_t it_k = tp[r][k] & 0xFFFFu, ip_k = tp[r][k] >> 16;
      double key = pixel_to_lidar_axis(c, rc, kcol, col + k, row0 + r * TILE_H, dk[k]);
      key += 0.0;                                                 // -0.0 -> +0.0, every other value unchanged
      const unsigned long long ok = ordered_key(key);
      const unsigned long long cm = combo_word(ok, base + k, pix_bits);
      const uint32_t lt = it_k - (uint32_t)t0, lp = ip_k - (uint32_t)p0;
      if ((lt < (uint32_t)WIN_T) & (lp < (uint32_t)WIN_P)) {      // aggregate in the LDS window
        const uint32_t w = __umul24(lt, (uint32_t)WIN_P) + lp;     // (24-bit multiplies issue at four times the rate of v_mul_lo_u32)
        atomicAdd(&s_cnt[w], 1u);
        atomicMin(&s_first[w], loc0 + k);
        atomicMin(&s_kmin[w], ok);
        atomicMin(&s_combo[w], cm);
      } else {                                                    // outside the window: direct
        const uint32_t b = __umul24(it_k, (uint32_t)g.p_n) + ip_k;
        const int64_t e = tb0 + b;
        atomicAdd(&T.cnt[e], 1u);
        const uint32_t oldf = atomicMin(&T.first[e], base + k);
        atomicMin(&T.kmin[e], ok);
        atomicMin(&T.combo[e], cm);
        mark_segment(occ_v, b);
        if (oldf > base + k) new_first(loc0 + k, oldf);
      }
    }
  }
  // The tile's undecided pixels (14 of 2048 on average, listed in s_amb before the origin's barrier): the middle tier
  // decides nearly all of them (2 139 of 3.9 M are left on the bench workload), and a decided pixel joins the window
  // like any other.  As pixels of k_bp_bin_amb they cost five scattered global atomics each (20 M per launch: 0.43 ms);
  // here they cost this kernel 0.23 ms (one wave per workgroup works while three wait at the barrier below; doing it
  // at the very end of the workgroup with direct commits instead measured 0.34 ms).  What is left stays listed.
  {
    const int na_l = min(s_namb, P1_AMB);
    for (int i = threadIdx.x; i < na_l; i += PB) {
      const uint32_t pix = s_amb[i];
      bool decided = false;
      MidOut mo;
      uint32_t b = NOBIN;
      if (fg.mid_ok) b = pixel_bin_mid(c, rc, g, fg, tab, W, (int)pix, depth[(size_t)v * HW + pix], decided, &mo);
      if (!decided) continue;
      s_amb[i] = NOBIN;                                           // done
      if (b == NOBIN) continue;                                   // certainly not binned
      double key = (key_axis == 2) ? mo.z : mo.y;
      key += 0.0;                                                 // -0.0 -> +0.0, every other value unchanged
      const unsigned long long ok = ordered_key(key);
      const unsigned long long cm = combo_word(ok, pix, pix_bits);
      const int prow = (int)pix / W, pcol = (int)pix - prow * W;
      const uint32_t loc = (uint32_t)((prow - ty * (RPT * TILE_H)) * TILE_W + (pcol - tx * TILE_W));
      const uint32_t lt = mo.it - (uint32_t)t0, lp = mo.ip - (uint32_t)p0;
      if ((lt < (uint32_t)WIN_T) & (lp < (uint32_t)WIN_P)) {
        const uint32_t w = __umul24(lt, (uint32_t)WIN_P) + lp;     // (24-bit multiplies issue at four times the rate of v_mul_lo_u32)
        atomicAdd(&s_cnt[w], 1u);
        atomicMin(&s_first[w], loc);
        atomicMin(&s_kmin[w], ok);
        atomicMin(&s_combo[w], cm);
      } else {
        const int64_t e = tb0 + b;
        atomicAdd(&T.cnt[e], 1u);
        const uint32_t oldf = atomicMin(&T.first[e], pix);
        atomicMin(&T.kmin[e], ok);
        atomicMin(&T.combo[e], cm);
        mark_segment(occ_v, b);
        if (oldf > pix) new_first(loc, oldf);
      }
    }
  }
  DBG_T(3);                                        // LDS window atomics
  __syncthreads();
  DBG_T(4);                                        // barrier
  // flush the window: one set of global atomics per touched bin.  The atomic on `first` returns the value it replaced
  // (the bit-map update below needs it): all of a thread's atomics are issued before the first returned value is looked
  // at -- one memory round trip per workgroup instead of one per slot of the thread.
  constexpr int FL = (WIN_T * WIN_P + PB - 1) / PB;
  uint32_t f_loc[FL], f_new[FL], f_old[FL], f_seg[FL];
#pragma unroll
  for (int i = 0; i < FL; i++) {
    const int w = threadIdx.x + i * PB;
    f_loc[i] = 0u; f_new[i] = NOBIN; f_old[i] = 0u; f_seg[i] = NOBIN;
    const uint32_t cw = (w < WIN_T * WIN_P) ? s_cnt[w] : 0u;
    if (cw == 0u) continue;
    static_assert(WIN_T * WIN_P <= 768 && WIN_P == 48, "w / 48 as (w * 1366) >> 16 is exact below 768 * 48 / 18");
    const uint32_t wq = __umul24((uint32_t)w, 1366u) >> 16, wr = (uint32_t)w - __umul24(wq, (uint32_t)WIN_P);   // w / 48, w % 48
    const uint32_t b = __umul24((uint32_t)t0 + wq, (uint32_t)g.p_n) + ((uint32_t)p0 + wr);
    const int64_t e = tb0 + b;
    f_seg[i] = b >> SEG_SHIFT;
    f_loc[i] = s_first[w];
    f_new[i] = global_pix(f_loc[i]);
    atomicAdd(&T.cnt[e], cw);
    f_old[i] = atomicMin(&T.first[e], f_new[i]);
    atomicMin(&T.kmin[e], s_kmin[w]);
    atomicMin(&T.combo[e], s_combo[w]);
  }
#pragma unroll
  for (int i = 0; i < FL; i++)
    if (f_seg[i] != NOBIN) occ_v[f_seg[i]] = 1;                          // (consecutive lanes flush consecutive bins: 1-3 addresses per wave)
#pragma unroll
  for (int i = 0; i < FL; i++)
    if (f_old[i] > f_new[i]) new_first(f_loc[i], f_old[i]);      // (f_new = NOBIN, the largest value, for an idle slot)
  DBG_T(5);                                        // flush loop (issue of the global atomics)
  __syncthreads();
  DBG_T(6);                                        // barrier
  // the workgroup's own bits: contiguous 128-byte wave atomics, one per bit-map tile (XOR: other tiles may already
  // have toggled here)
  if (threadIdx.x < 32 * RPT && s_bits[threadIdx.x]) {
    const int sub = threadIdx.x >> 5;                             // which of the RPT bit-map tiles
    if (ty * RPT + sub < tiles_y)
      atomicXor(&bitmap_v[((size_t)(ty * RPT + sub) * tiles_x + tx) * 32 + (threadIdx.x & 31)], s_bits[threadIdx.x]);
  }
  const int na = min(s_namb, P1_AMB);
  DBG_T(7);                                        // bit-map flush
  for (int i = threadIdx.x; i < na; i += PB)      // (what the middle tier left: in practice nothing)
    if (s_amb[i] != NOBIN) amb_list[(size_t)v * HW + atomicAdd(&n_amb[v], 1)] = s_amb[i];
}


// Tier 2: the undecided pixels, full fp64 classification (pixel_bin).
__global__ __launch_bounds__(256) void k_bp_bin_amb(
    const float *__restrict__ depth, const ViewCalib *__restrict__ calib, dfu3d_bin_geom g, int W,
    int HW, int key_axis, int64_t E_view, void *table, int64_t E_total, const int *__restrict__ n_amb,
    const uint32_t *__restrict__ amb_list, uint32_t *__restrict__ status, int pix_bits,
    uint32_t *__restrict__ bitmap, int BW, int tiles_x, uint8_t *__restrict__ occ, int OW) {
  const int v = blockIdx.y;
  const int na = n_amb[v];
  const ViewCalib c = calib[v];
  const Recip rc = make_recip(c);
  const Table T = table_view(table, E_total);
  const int64_t tb0 = (int64_t)v * E_view;
  bool rerr = false;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < na; e += gridDim.x * 256) {
    const int pix = (int)amb_list[(size_t)v * HW + e];
    double key;
    const uint32_t b = pixel_bin(c, rc, g, W, pix, depth[(size_t)v * HW + pix], key_axis, key, rerr);
    if (b != NOBIN) {
      commit_pixel(T, tb0 + b, pix, key, pix_bits, bitmap + (size_t)v * BW, W, tiles_x);
      mark_segment(occ + (size_t)v * OW * 4, b);
    }
  }
  if (rerr) atomicOr(status, DFU3D_ST_BIN_RANGE);
}

// ---- P3: exclusive popcount prefix of the bit map in raster order (one workgroup per view) ------
// item j = y * tiles_x + tx: the 64 pixels of image row y inside tile column tx (two words of the tile-major map)
constexpr int SCB = 1024;
__device__ __forceinline__ unsigned long long row_piece(const uint32_t *bitmap_v, int tiles_x, int y, int tx) {
  return *(const unsigned long long *)(bitmap_v + ((size_t)(y >> 4) * tiles_x + tx) * 32 + (y & 15) * 2);
}
__global__ __launch_bounds__(SCB) void k_bp_scan(int BW, int NJ, int tiles_x, const uint32_t *__restrict__ bitmap,
                                                 uint32_t *__restrict__ wpre, int *__restrict__ n_vox, int cap_vox,
                                                 uint32_t *__restrict__ status, const uint32_t *__restrict__ occ, int OW,
                                                 int NSEG, int *__restrict__ seg_list, int *__restrict__ n_occ) {
  __shared__ int s_w[SCB / 64];
  const int v = blockIdx.x;
  // The occupied segments of the view's table in ascending order -- theta row by theta row, along phi inside a row: the
  // work list of the voxel pass (OW words of four occupancy bytes).  Consecutive entries are neighbours along phi, i.e.
  // along an image row: their voxels have consecutive ranks, and the four waves of a workgroup of the voxel pass, which
  // take four of them at the same time, write neighbouring pieces of the outputs and read the same bit-map lines.
  // (The list ordered column by column -- vertical neighbours, which share depth-map and mask lines -- was measured:
  // the voxel pass 1.33 ms instead of 1.15; a run of four vertical neighbours per wave: 1.40.)
  {
    const uint32_t *ov = occ + (size_t)v * OW;
    int run = 0;
    for (int w0 = 0; w0 < OW; w0 += SCB) {
      const int w = w0 + threadIdx.x;
      const uint32_t word = (w < OW) ? ov[w] : 0u;
      int tot;
      int at = run + block_excl_scan<SCB / 64>(__popc(word & 0x01010101u), s_w, tot);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if ((word >> (8 * q)) & 1u) seg_list[(size_t)v * NSEG + at++] = w * 4 + q;
      run += tot;
    }
    if (threadIdx.x == 0) n_occ[v] = run;
  }
  const uint32_t *bv = bitmap + (size_t)v * BW;
  int running = 0;
  for (int j0 = 0; j0 < NJ; j0 += SCB) {
    const int j = j0 + threadIdx.x;
    int c = 0;
    if (j < NJ) {
      const int y = j / tiles_x, tx = j - y * tiles_x;
      c = __popcll(row_piece(bv, tiles_x, y, tx));
    }
    int tot;
    const int ex = block_excl_scan<SCB / 64>(c, s_w, tot);
    if (j < NJ) wpre[(size_t)v * NJ + j] = (uint32_t)(running + ex);
    running += tot;
  }
  if (threadIdx.x == 0) {
    n_vox[v] = running;                                   // voxels = first pixels (clamped by k_bp_finalize)
    if (running > cap_vox) atomicOr(status, DFU3D_ST_VOX_OVERFLOW);
  }
}

struct VoxOut {
  uint32_t *vox_pix, *it_bits;
  double *it_x, *it_y, *it_z;
};

__device__ __forceinline__ void emit_voxel(const VoxOut &o, size_t at, const ViewCalib &c, const Recip &rc,
                                           const float *depth_v, int W, uint32_t pix, const void *masks,
                                           int mask_format, int m, int max_inst, int HW, int v) {
  const int row = (int)pix / W, col = (int)pix - row * W;
  double x, y, z;
  pixel_to_lidar(c, rc, col, row, depth_v[pix], x, y, z);
  o.vox_pix[at] = pix;
  o.it_bits[at] = masks ? mask_bits_at(masks, mask_format, v, max_inst, m, HW, (int)pix) : 0u;
  o.it_x[at] = x;
  o.it_y[at] = y;
  o.it_z[at] = z;
}

// ---- P4: walk over the occupied table segments: rank, representative, outputs, table reset -------------------
// A wave takes one segment of 64 consecutive table entries at a time: the four planes of the segment are four coalesced
// loads, every entry with a count is a voxel, and its place in first-seen order is the rank of its FIRST pixel in the
// bit map (prefix of the pixel's row piece + set bits before it).  What a voxel needs besides its entry -- the bit-map
// word and prefix at its first pixel, depth and mask word at its representative pixel -- is requested together: TWO
// dependent round trips per segment.  The representative is the pixel p* of the packed word: the smallest pixel among
// those whose CUT key is minimal, a superset of the exact arg-mins, so it is the representative iff its exact key
// equals kmin; bins where that fails (two keys differ only below the cut: practically never) and bins that saw more
// than max_points pixels go to the exact repair.  The entries are reset with four coalesced stores.
// (Until round 4 the pass walked the BIT MAP: a voxel was found as a set bit, its first pixel's depth was gathered and
// the pixel classified a second time -- float32 tier, middle tier, a parked rest for a full-fp64 kernel -- only to learn
// its bin; the table entry was three scattered reads behind that: four dependent round trips, 86 registers, 1.42 ms.)
constexpr int VXB = 256;
constexpr int VOX_GX = 512;         // workgroups per view at most (2 048 waves; a wave walks its segments with the grid's stride).
                                    // A camera of the bench touches ~1 800 of its table's 20 252 segments: a grid over the whole
                                    // table would be workgroups that start only to find nothing.  Measured: 128 / 256 / 512 /
                                    // 1 024 workgroups 1.13 / 1.17 / 1.07 / 1.07 ms; register budgets for 7 / 8 waves per SIMD
                                    // (the code needs 76 registers: 6 waves) spill: 1.23 / 1.58 ms
struct VoxWalk {
  const int *seg_list, *n_occ;
  const uint32_t *bitmap, *wpre;
  int NSEG, BW, NJ, tiles_x;
};
__global__ __launch_bounds__(VXB) void k_bp_vox(
    const float *__restrict__ depth, const ViewCalib *__restrict__ calib, const void *__restrict__ masks, int mask_format,
    const int *__restrict__ n_inst, int max_inst, int W, int HW, int64_t E_view, void *table, int64_t E_total, int cap_vox,
    VoxWalk Wk, VoxOut out, int key_axis, int pix_bits, int max_points, int max_voxels, int cap_q,
    uint32_t *__restrict__ q_bins, int *__restrict__ q_rank, int *__restrict__ n_q, uint32_t *__restrict__ status, int V) {
  // XCD-aware placement: the hardware deals consecutive workgroups (x fastest) round the eight XCDs, each with an L2 of its
  // own.  All workgroups of ONE view go to ONE XCD -- view = 8 * (turn / gridDim.x) + XCD: its gathers (depth and mask word
  // at the representative pixels: neighbouring bins, neighbouring pixels, the same lines) are fetched into one L2 instead of
  // up to eight, and the scattered outputs (rank order: 4 / 8 bytes per voxel and plane) fill their lines in one L2 before
  // they are written back, instead of leaving as partial lines from several.  1.21 -> 1.00 ms (tools/ab_builds.py, builds
  // alternating in one process).  The same placement for k_bp_bin, whose traffic is streamed depth and memory-side atomics,
  // measured 2.57 -> 2.68 ms: not taken there.  (Only a placement: any dealing of workgroups gives the same result.)
  const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, turn = lin >> 3;
  const int v = (int)((turn / gridDim.x) * 8u + (lin & 7u)), bx = (int)(turn % gridDim.x);
  if (v >= V) return;                               // (the grid's y is V rounded up to a multiple of 8)
  const int n_occ = Wk.n_occ[v];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(bx * (VXB / 64) + (threadIdx.x >> 6)));
  if (wave >= n_occ) return;
  const ViewCalib c = calib[v];
  const Recip rc = make_recip(c);
  const Table T = table_view(table, E_total);
  const int64_t tb0 = (int64_t)v * E_view;
  const float *dv = depth + (size_t)v * HW;
  const uint32_t *bv = Wk.bitmap + (size_t)v * Wk.BW;
  const int m_inst = masks ? min(max(n_inst[v], 0), max_inst) : 0;
  const unsigned long long pix_mask = (1ull << pix_bits) - 1ull;
  // consecutive list entries are neighbours along an image row (k_bp_scan): the four waves of a workgroup take four of
  // them at the same time
  for (int i = wave; i < n_occ; i += gridDim.x * (VXB / 64)) {                  // (uniform per wave)
    const int sg = Wk.seg_list[(size_t)v * Wk.NSEG + i];
    const int64_t b = ((int64_t)sg << SEG_SHIFT) + lane;                      // bin of the view
    const bool in = b < E_view;
    const int64_t e = tb0 + (in ? b : 0);
    const uint32_t cw = in ? T.cnt[e] : 0u;
    const uint32_t f = T.first[e];
    const unsigned long long e_kmin = T.kmin[e], e_combo = T.combo[e];
    const bool vox = cw != 0u && cw < OVF_FLAG;                                // (a flagged count belongs to a repair in flight: never here)
    if (__ballot(vox) == 0ull) continue;
    int k = 0;
    uint32_t pix = 0u, m_bits = 0u;
    float d_pix = 0.0f;
    if (vox) {
      const int fr = (int)f / W, fc = (int)f - fr * W;
      const unsigned long long word = row_piece(bv, Wk.tiles_x, fr, fc >> 6);
      const uint32_t pre = Wk.wpre[(size_t)v * Wk.NJ + (size_t)fr * Wk.tiles_x + (fc >> 6)];
      pix = (uint32_t)(e_combo & pix_mask);
      d_pix = dv[pix];
      m_bits = masks ? mask_bits_at(masks, mask_format, v, max_inst, m_inst, HW, (int)pix) : 0u;
      k = (int)pre + __popcll(word & ((1ull << (fc & 63)) - 1ull));
    }
    // an entry is left as it is when a repair in flight owns it, when its voxel is beyond cap_vox (DFU3D_ST_VOX_OVERFLOW,
    // raised by the scan: the table stays dirty) and when it is queued for the repair below; every other entry of the
    // segment -- the empty ones too: whole lines -- is written back clean
    bool keep = !in || cw >= OVF_FLAG || (vox && k >= cap_vox);
    if (vox && !keep) {
      const int row = (int)pix / W, col = (int)pix - row * W;
      double x, yy, z;
      pixel_to_lidar(c, rc, col, row, d_pix, x, yy, z);
      double key = (key_axis == 2) ? z : yy;
      key += 0.0;                                   // -0.0 -> +0.0 (round to nearest), every other value unchanged
      // over the cap ("the first max_points pixels" must be found), or two keys that agree in their top bits only
      if (cw > (uint32_t)max_points || ordered_key(key) != e_kmin) {
        const int slot = atomicAdd(&n_q[v], 1);        // exact repair (k_ovf_*, k_bp_fix); the entry stays as it is
        if (slot < cap_q) { q_bins[(size_t)v * cap_q + slot] = (uint32_t)b; q_rank[(size_t)v * cap_q + slot] = k; }
        else atomicOr(status, DFU3D_ST_VOX_PTS_OVERFLOW);
        keep = true;
      } else if (k < max_voxels) {
        const size_t at = (size_t)v * cap_vox + k;
        out.vox_pix[at] = pix;
        out.it_bits[at] = m_bits;
        out.it_x[at] = x;
        out.it_y[at] = yy;
        out.it_z[at] = z;
      }
    }
    if (!keep) {                                    // (rep is only ever written by the repair)
      T.kmin[e] = ~0ull;
      T.combo[e] = ~0ull;
      T.cnt[e] = 0u;
      T.first[e] = NOBIN;
    }
  }
}

// ---- O1: bin id per pixel, exact classification, only for views with a repair queue ----
__global__ __launch_bounds__(PB) void k_bp_rebin(
    const float *__restrict__ depth, const ViewCalib *__restrict__ calib, dfu3d_bin_geom g, int W, int HW,
    int key_axis, const int *__restrict__ n_q, uint32_t *__restrict__ pix_bin) {
  const int v = blockIdx.y;
  if (n_q[v] == 0) return;                           // the usual case
  const ViewCalib c = calib[v];
  const Recip rc = make_recip(c);
  bool rerr = false;
  for (int pix = blockIdx.x * PB + threadIdx.x; pix < HW; pix += gridDim.x * PB) {
    double key;
    pix_bin[(size_t)v * HW + pix] = pixel_bin(c, rc, g, W, pix, depth[(size_t)v * HW + pix], key_axis, key, rerr);
  }
}

// ---- O2: allocate a pixel list per queued bin ----------------------------------
// rep[e] <- list base, cnt[e] <- OVF_FLAG | 0 (fill cursor), q_cnt <- count
__global__ void k_ovf_alloc(void *table, int64_t E_total, int64_t E_view, int cap_q,
                            const uint32_t *__restrict__ q_bins,
                            const int *__restrict__ n_q, int *__restrict__ q_cnt,
                            int *__restrict__ q_cursor, int HW,
                            uint32_t *__restrict__ status) {
  const int v = blockIdx.y;
  const int no = min(n_q[v], cap_q);
  const Table T = table_view(table, E_total);
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < no; s += gridDim.x * blockDim.x) {
    const int64_t e = (int64_t)v * E_view + q_bins[(size_t)v * cap_q + s];
    const int c = (int)T.cnt[e];
    const int off = atomicAdd(&q_cursor[v], c);
    if (off + c > HW) atomicOr(status, DFU3D_ST_VOX_PTS_OVERFLOW);   // cannot happen
    q_cnt[(size_t)v * cap_q + s] = c;
    T.rep[e] = (uint32_t)off;
    T.cnt[e] = OVF_FLAG;
  }
}

// ---- O3: gather the pixel indices of queued bins -------------------------------
__global__ __launch_bounds__(PB) void k_ovf_gather(
    const uint32_t *__restrict__ pix_bin, void *table, int64_t E_total, int64_t E_view,
    int HW, const int *__restrict__ n_q, uint32_t *__restrict__ q_list) {
  const int v = blockIdx.y;
  if (n_q[v] == 0) return;                         // the usual case: a few workgroups per view leave at once
  const Table T = table_view(table, E_total);
  for (int base = blockIdx.x * PBLK + threadIdx.x * PPT; base < HW; base += gridDim.x * PBLK) {
    for (int k = 0; k < PPT; k++) {
      const int pix = base + k;
      if (pix >= HW) break;
      const uint32_t b = pix_bin[(size_t)v * HW + pix];
      if (b == NOBIN) continue;
      const int64_t e = (int64_t)v * E_view + b;
      if (T.cnt[e] & OVF_FLAG) {
        const uint32_t pos = atomicAdd(&T.cnt[e], 1u) & ~OVF_FLAG;
        q_list[(size_t)v * HW + T.rep[e] + pos] = (uint32_t)pix;
      }
    }
  }
}

// ---- O4: per overflow bin, T = max_points-th smallest pixel, kmin over pix<=T -
__global__ __launch_bounds__(256) void k_ovf_select(
    const float *__restrict__ depth, const ViewCalib *__restrict__ calib, int W, int HW,
    int key_axis, int max_points, void *table, int64_t E_total, int64_t E_view,
    int cap_ovf, const uint32_t *__restrict__ ovf_bins, const int *__restrict__ n_ovf,
    const int *__restrict__ ovf_cnt, const uint32_t *__restrict__ ovf_list) {
  __shared__ int hist[256];
  __shared__ uint32_t s_sel[2];
  __shared__ unsigned long long s_min[4];
  const int v = blockIdx.y;
  const int no = min(n_ovf[v], cap_ovf);
  const Table T = table_view(table, E_total);
  const ViewCalib c = calib[v];
  const Recip rc = make_recip(c);
  for (int s = blockIdx.x; s < no; s += gridDim.x) {     // uniform per block
  const int64_t e = (int64_t)v * E_view + ovf_bins[(size_t)v * cap_ovf + s];
  const int n = ovf_cnt[(size_t)v * cap_ovf + s];
  const uint32_t *lst = ovf_list + (size_t)v * HW + T.rep[e];
  __syncthreads();
  // radix select (24 bits, 3 x 8) of the value with 0-based rank max_points-1
  uint32_t prefix = 0u, mask = 0u;
  int kk = max_points - 1;
  for (int shift = 16; shift >= 0; shift -= 8) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const uint32_t key = lst[i];
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFFu], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0, b = 0;
      for (; b < 256; b++) {
        const int c = hist[b];
        if (acc + c > kk) break;
        acc += c;
      }
      if (b > 255) b = 255;
      s_sel[0] = prefix | ((uint32_t)b << shift);
      s_sel[1] = (uint32_t)(kk - acc);
    }
    __syncthreads();
    prefix = s_sel[0];
    kk = (int)s_sel[1];
    mask |= (0xFFu << shift);
    __syncthreads();
  }
  const uint32_t thr = prefix;
  unsigned long long m = ~0ull;
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t pix = lst[i];
    if (pix <= thr) {
      const unsigned long long k = ordered_key(
          pixel_key(c, rc, W, (int)pix, depth[(size_t)v * HW + pix], key_axis));
      m = k < m ? k : m;
    }
  }
#pragma unroll
  for (int x = 32; x >= 1; x >>= 1) {
    const unsigned long long o = __shfl_xor(m, x, 64);
    m = o < m ? o : m;
  }
  if (lane_id() == 0) s_min[threadIdx.x >> 6] = m;
  __syncthreads();
  for (int w = 0; w < 4; w++) m = s_min[w] < m ? s_min[w] : m;     // the same value in every thread
  // representative: the smallest pixel <= T whose key is that minimum
  uint32_t rp = NOBIN;
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t pix = lst[i];
    if (pix <= thr && pix < rp) {
      const unsigned long long k = ordered_key(
          pixel_key(c, rc, W, (int)pix, depth[(size_t)v * HW + pix], key_axis));
      if (k == m) rp = pix;
    }
  }
#pragma unroll
  for (int x = 32; x >= 1; x >>= 1) rp = min(rp, (uint32_t)__shfl_xor((int)rp, x, 64));
  __syncthreads();
  if (lane_id() == 0) s_min[threadIdx.x >> 6] = (unsigned long long)rp;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 0; w < 4; w++) rp = min(rp, (uint32_t)s_min[w]);
    T.kmin[e] = m;
    T.cnt[e] = OVF_FLAG | thr;       // P4 takes the representative from T.rep for these bins
    T.rep[e] = rp;
  }
  }
}

// ---- O5: outputs of the repaired bins ---------------------------------------------
__global__ __launch_bounds__(256) void k_bp_fix(
    const float *__restrict__ depth, const ViewCalib *__restrict__ calib,
    const void *__restrict__ masks, int mask_format, const int *__restrict__ n_inst, int max_inst, int W,
    int HW, int max_voxels, int64_t E_view, void *table, int64_t E_total, int cap_vox, VoxOut out, int cap_q,
    const uint32_t *__restrict__ q_bins, const int *__restrict__ q_rank, const int *__restrict__ n_q) {
  const int v = blockIdx.y;
  const int no = min(n_q[v], cap_q);
  if (blockIdx.x * 256 >= no) return;
  const Table T = table_view(table, E_total);
  const ViewCalib c = calib[v];
  const Recip rc = make_recip(c);
  const int m = masks ? min(max(n_inst[v], 0), max_inst) : 0;
  for (int s = blockIdx.x * 256 + threadIdx.x; s < no; s += gridDim.x * 256) {
    const int64_t e = (int64_t)v * E_view + q_bins[(size_t)v * cap_q + s];
    const int k = q_rank[(size_t)v * cap_q + s];
    if (k < max_voxels)
      emit_voxel(out, (size_t)v * cap_vox + k, c, rc, depth + (size_t)v * HW, W, T.rep[e], masks, mask_format, m,
                 max_inst, HW, v);
    T.kmin[e] = ~0ull;
    T.combo[e] = ~0ull;
    T.cnt[e] = 0u;
    T.first[e] = NOBIN;
    T.rep[e] = NOBIN;
  }
}

__global__ void k_bp_finalize(int V, int max_voxels, int cap_vox, int *__restrict__ n_vox) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  n_vox[v] = min(min(n_vox[v], cap_vox), max_voxels);
}

// tier 1 (pixel_bin_fast) against the fp64 classification (pixel_bin) on n pseudo-random pixels of view 0:
// out[0] pixels tried, out[1] undecided by tier 1, out[2] DISAGREEMENTS among the decided (bin, or the key of a kept
// pixel), out[3] pixels tier 1 kept.  Depths in [d_lo, d_hi), every fourth one 50x closer.
__global__ void k_selftest_classify(const ViewCalib *__restrict__ calib, const FastCal *__restrict__ fastcal,
                                    const float2 *__restrict__ tab, dfu3d_bin_geom g, FastGeom fg, int H, int W,
                                    int key_axis, long long n, unsigned long long seed, double d_lo, double d_hi,
                                    unsigned long long *out) {
  const ViewCalib c = calib[0];
  const FastCal fc = fastcal[0];
  const Recip rc = make_recip(c);
  const KeyCol kcol = load_key_col(calib, key_axis);
  unsigned long long tried = 0, undecided = 0, wrong = 0, kept = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long a = mix64(seed + 3ull * (unsigned long long)i),
                             b = mix64(seed + 3ull * (unsigned long long)i + 1ull),
                             e = mix64(seed + 3ull * (unsigned long long)i + 2ull);
    const int col = (int)(a % (unsigned long long)W), row = (int)(b % (unsigned long long)H);
    const double ud = (double)(e >> 11) * (1.0 / 9007199254740992.0);
    const float d = (float)((d_lo + (d_hi - d_lo) * ud) * ((i & 3) == 0 ? 0.02 : 1.0));
    double k1 = 0.0, k2 = 0.0;
    int it_, ip_;
    bool rerr = false;
    const uint32_t b1 = pixel_bin_fast(c, rc, fc, g, fg, tab, row, col, d, kcol, true, k1, it_, ip_);
    const uint32_t b2 = pixel_bin(c, rc, g, W, row * W + col, d, key_axis, k2, rerr);
    tried++;
    if (b1 == AMBIG) {                              // tier 1.5 takes it in P4: a decision of its own must be pixel_bin's
      undecided++;
      if (fg.mid_ok) {
        bool dec = false;
        const uint32_t b3 = pixel_bin_mid(c, rc, g, fg, tab, W, row * W + col, d, dec);
        if (dec && b3 != b2) wrong++;
      }
      continue;
    }
    if (b1 != b2 || (b1 != NOBIN && __double_as_longlong(k1) != __double_as_longlong(k2))) wrong++;
    if (b1 != NOBIN) kept++;
  }
  tried = wave_sum_u64(tried); undecided = wave_sum_u64(undecided);
  wrong = wave_sum_u64(wrong); kept = wave_sum_u64(kept);
  if (lane_id() == 0) {
    atomicAdd(&out[0], tried); atomicAdd(&out[1], undecided); atomicAdd(&out[2], wrong); atomicAdd(&out[3], kept);
  }
}

// max over n pseudo-random pixels of |float32 back-projection - fp64 back-projection| / bound, over the three
// coordinates (view 0 of `calib`, depths in [d_lo, d_hi), every fourth one 50x closer)
__global__ void k_selftest_backproject(const ViewCalib *__restrict__ calib, const FastCal *__restrict__ fastcal,
                                       int H, int W, long long n, unsigned long long seed, double d_lo,
                                       double d_hi, unsigned long long *out) {
  const ViewCalib c = calib[0];
  const FastCal fc = fastcal[0];
  const Recip rc = make_recip(c);
  double worst = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long a = mix64(seed + 3ull * (unsigned long long)i),
                             b = mix64(seed + 3ull * (unsigned long long)i + 1ull),
                             e = mix64(seed + 3ull * (unsigned long long)i + 2ull);
    const int col = (int)(a % (unsigned long long)W), row = (int)(b % (unsigned long long)H);
    const double ud = (double)(e >> 11) * (1.0 / 9007199254740992.0);
    const float d = (float)((d_lo + (d_hi - d_lo) * ud) * ((i & 3) == 0 ? 0.02 : 1.0));
    if (!(d > 0.0f)) continue;
    double x, y, z;
    pixel_to_lidar(c, rc, col, row, d, x, y, z);
    float xf, yf, zf, err;
    backproject_f32(fc, col, row, d, xf, yf, zf, err);
    const double m = fmax(fmax(fabs((double)xf - x), fabs((double)yf - y)), fabs((double)zf - z));
    worst = fmax(worst, m / (double)err);
  }
  worst = wave_max_d(worst);
  if (lane_id() == 0) atomicMax(&out[0], (unsigned long long)__double_as_longlong(worst));
}

}  // namespace

DBG_T_READER(dfu3d_debug_timing_pixel)

extern "C" int dfu3d_selftest_classify(const float *calib, int32_t H, int32_t W, const dfu3d_bin_geom *geom,
                                       int32_t key_axis, int64_t n, uint64_t seed, double d_lo, double d_hi,
                                       void *scratch, uint64_t *out4, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!calib || !geom || !scratch || !out4 || n <= 0 || H <= 0 || W <= 0 || !(d_hi > d_lo) || !(d_lo >= 0.0))
    return DFU3D_EINVAL;
  if (key_axis != 1 && key_axis != 2) return DFU3D_EINVAL;
  if ((uintptr_t)scratch & 15u) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const FastGeom fg = make_fast_geom(*geom);
  FastCal *fastcal = (FastCal *)scratch;
  float2 *tab = (float2 *)((char *)scratch + 128);
  if (hipMemsetAsync(out4, 0, 32, st) != hipSuccess) return DFU3D_ELAUNCH;
  hipLaunchKernelGGL(k_bp_prep, dim3(1), dim3(64), 0, st, (const ViewCalib *)calib, 1, H, W, fastcal);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bp_tables, dim3((tables_threads(fg, *geom) + 255) / 256), dim3(256), 0, st, *geom, fg, tab);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_selftest_classify, dim3(2048), dim3(256), 0, st, (const ViewCalib *)calib, fastcal, tab,
                     *geom, fg, H, W, key_axis, (long long)n, (unsigned long long)seed, d_lo, d_hi,
                     (unsigned long long *)out4);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_selftest_backproject(const float *calib, int32_t H, int32_t W, int64_t n, uint64_t seed,
                                          double d_lo, double d_hi, void *scratch64, double *out1, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!calib || !scratch64 || !out1 || n <= 0 || H <= 0 || W <= 0 || !(d_hi > d_lo) || !(d_lo >= 0.0)) return DFU3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out1, 0, 8, st) != hipSuccess) return DFU3D_ELAUNCH;
  hipLaunchKernelGGL(k_bp_prep, dim3(1), dim3(64), 0, st, (const ViewCalib *)calib, 1, H, W, (FastCal *)scratch64);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_selftest_backproject, dim3(2048), dim3(256), 0, st, (const ViewCalib *)calib,
                     (const FastCal *)scratch64, H, W, (long long)n, (unsigned long long)seed, d_lo, d_hi,
                     (unsigned long long *)out1);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int64_t dfu3d_bin_table_geometry(dfu3d_bin_geom *g) {
  if (!g) return DFU3D_EINVAL;
  const double pi = 3.14159265358979323846;
  // bins reachable with theta in (theta_min, pi], phi in [-pi/2, pi/2]
  auto binf = [](double v, double mn, double vs) { return (int64_t)__builtin_floor((v - mn) / vs); };
  int64_t t0 = binf(g->theta_min, g->rmin_t, g->vsize_t) - 1;
  int64_t t1 = binf(pi, g->rmin_t, g->vsize_t) + 2;
  int64_t p0 = binf(-pi / 2, g->rmin_p, g->vsize_p) - 1;
  int64_t p1 = binf(pi / 2, g->rmin_p, g->vsize_p) + 2;
  if (t0 < 0) t0 = 0;
  if (p0 < 0) p0 = 0;
  if (t1 > g->grid_t) t1 = g->grid_t;
  if (p1 > g->grid_p) p1 = g->grid_p;
  if (t1 <= t0 || p1 <= p0) return DFU3D_EINVAL;
  g->t_lo = (int32_t)t0;
  g->t_n = (int32_t)(t1 - t0);
  g->p_lo = (int32_t)p0;
  g->p_n = (int32_t)(p1 - p0);
  return (int64_t)g->t_n * g->p_n;
}

extern "C" int dfu3d_bin_table_init(void *table, int64_t E, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!table || E <= 0) return DFU3D_EINVAL;
  const Table T = table_view(table, E);
  hipLaunchKernelGGL(k_table_init, dim3(2048), dim3(256), 0, (hipStream_t)stream, T.kmin,
                     T.combo, T.cnt, T.first, T.rep, E);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

// Scratch carve-up.
// blk_cnt (int32 words): n_amb[V], n_q[V], q_cursor[V], n_occ[V], bitmap[V*BW], occ[V*OW] -- everything up to here is
//   zeroed at the start of a pass --, wpre[V*NJ], seg_list[V*NSEG], q_cnt[V*cap_q], q_bins[V*cap_q], q_rank[V*cap_q], the float32
//   calibration constants (80 B per view) and the edge tables of tier 1 (8 B x (TAB_T_MAX + TAB_P_MAX + 4) at most; the carve-up keeps round 2's 16 B)
//   (BW = 32 words per 64x16 tile, NJ = H * tiles_x, NSEG = 64-entry segments of a view's table, OW = NSEG / 4 words of
//   occupancy bytes, cap_q: queue_cap)
// pix_bin (uint32 words): [0, V*HW) bin id per pixel (written only for views under repair),
//   [V*HW, 2*V*HW) undecided-pixel lists, later the pixel lists of the repair.
static inline int queue_cap(int64_t HW, int max_points, int cap_vox) {
  if (DBG_COMBO_KEYBITS < 64) return cap_vox;             // test builds provoke the repair for a large share of the bins
  const int64_t q = HW / (max_points + 1) + 1 + 4096;     // bins over the cap + room for key collisions
  return (int)(q < cap_vox ? q : cap_vox);
}

// segments (64 entries) and occupancy words of a view's table; an upper bound when the geometry is not known yet
static inline int64_t table_segments(int64_t E_view) { return (E_view + 63) >> SEG_SHIFT; }

extern "C" int64_t dfu3d_backproject_scratch_words(int32_t V, int32_t H, int32_t W,
                                                    int32_t cap_vox, int32_t max_points, int64_t table_entries,
                                                    int64_t *pix_words, int64_t *blk_words) {
  if (V <= 0 || H <= 0 || W <= 0 || cap_vox <= 0 || max_points < 1 || table_entries <= 0) return DFU3D_EINVAL;
  const int64_t HW = (int64_t)H * W;
  const int64_t tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  const int64_t BW = tiles_x * tiles_y * 32, NJ = (int64_t)H * tiles_x;
  const int64_t cap_q = queue_cap(HW, max_points, cap_vox);
  if (pix_words) *pix_words = 2 * V * HW;
  const int64_t NSEG = table_segments(table_entries), OW = (NSEG + 3) / 4;
  if (blk_words) *blk_words = 4 * (int64_t)V + V * BW + V * OW + V * NJ + V * NSEG + 3 * V * cap_q + 20 * (int64_t)V + 16 +
                              4 * (int64_t)(TAB_T_MAX + TAB_P_MAX) + 16;
  return 0;
}

extern "C" int dfu3d_backproject_bin(
    const float *depth, const float *calib, const void *masks, int32_t mask_format, const int32_t *n_inst,
    int32_t V, int32_t max_inst, int32_t H, int32_t W, const dfu3d_bin_geom *geom,
    int32_t key_axis, void *table, uint32_t *pix_bin, int32_t *blk_cnt, int32_t cap_vox,
    int32_t *n_vox, uint32_t *vox_pix, uint32_t *it_bits, double *it_x, double *it_y,
    double *it_z, uint32_t *status, int32_t phases, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!depth || !calib || !geom || !table || !pix_bin || !blk_cnt || !n_vox || !vox_pix ||
      !it_bits || !it_x || !it_y || !it_z || !status)
    return DFU3D_EINVAL;
  if (masks && !n_inst) return DFU3D_EINVAL;
  if (V <= 0 || H <= 0 || W <= 0 || cap_vox <= 0) return DFU3D_EINVAL;
  if (key_axis != 1 && key_axis != 2) return DFU3D_EINVAL;
  if (max_inst > DFU3D_MAX_INST) return DFU3D_ERANGE;
  if (masks && !mask_format_ok(mask_format, max_inst)) return DFU3D_EINVAL;
  const int64_t HW64 = (int64_t)H * W;
  if (HW64 >= (1ll << 24)) return DFU3D_ERANGE;     // 24-bit pixel radix select
  if (W % 4) return DFU3D_EINVAL;                    // float4 row loads
  if (geom->max_points_per_voxel < 1) return DFU3D_EINVAL;
  if (((uintptr_t)blk_cnt & 7u) != 0) return DFU3D_EINVAL;   // 64-bit counters in front
  const int HW = (int)HW64;
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  const int BW = tiles_x * tiles_y * 32, NJ = H * tiles_x;
  const int cap_q = queue_cap(HW64, geom->max_points_per_voxel, cap_vox);
  const int64_t E_view = (int64_t)geom->t_n * geom->p_n;
  const int64_t E_total = E_view * V;
  hipStream_t st = (hipStream_t)stream;
  const int NSEG = (int)table_segments(E_view), OW = (NSEG + 3) / 4;       // OW: words of the occupancy bytes
  int *n_amb = blk_cnt;
  int *n_q = n_amb + V;
  int *q_cursor = n_q + V;
  int *n_occ = q_cursor + V;
  uint32_t *bitmap = (uint32_t *)(n_occ + V);
  uint32_t *occ = bitmap + (size_t)V * BW;
  uint32_t *wpre = occ + (size_t)V * OW;
  int *seg_list = (int *)(wpre + (size_t)V * NJ);
  int *q_cnt = seg_list + (size_t)V * NSEG;
  uint32_t *q_bins = (uint32_t *)(q_cnt + (size_t)V * cap_q);
  int *q_rank = (int *)(q_bins + (size_t)V * cap_q);
  FastCal *fastcal = (FastCal *)(((uintptr_t)(q_rank + (size_t)V * cap_q) + 15) & ~(uintptr_t)15);   // 80 B per view
  const float2 *tab = (const float2 *)(fastcal + V);                  // edge tables of tier 1: (tJ + pJ + 4) x 8 B
  const FastGeom fg = make_fast_geom(*geom);
  int pix_bits = 1;
  while ((1ll << pix_bits) < HW64) pix_bits++;
  uint32_t *q_list = pix_bin + (size_t)V * HW;       // undecided pixels first, repair lists later
  const ViewCalib *cal = (const ViewCalib *)calib;
  const VoxOut out = {vox_pix, it_bits, it_x, it_y, it_z};
  const VoxWalk Wk = {seg_list, n_occ, bitmap, wpre, NSEG, BW, NJ, tiles_x};

  if (phases & DFU3D_BP_BIN) {
    if (dfu3d_fill_async(blk_cnt, 0, sizeof(int) * (4 * (size_t)V + (size_t)V * BW + (size_t)V * OW), st) != hipSuccess) return DFU3D_ELAUNCH;
    hipLaunchKernelGGL(k_bp_prep, dim3((V + 63) / 64), dim3(64), 0, st, cal, V, H, W, fastcal);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bp_tables, dim3((tables_threads(fg, *geom) + 255) / 256), dim3(256), 0, st, *geom, fg, (float2 *)tab);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bp_bin, dim3(tiles_x * ((tiles_y + RPT - 1) / RPT), V), dim3(PB), 0, st, depth, cal, fastcal, tab, *geom,
                       fg, W, H, tiles_x, tiles_y, key_axis, E_view, table, E_total, n_amb, q_list, pix_bits,
                       bitmap, BW, (uint8_t *)occ, OW);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_BP_AMB) {
    hipLaunchKernelGGL(k_bp_bin_amb, dim3(64, V), dim3(256), 0, st, depth, cal, *geom, W, HW,
                       key_axis, E_view, table, E_total, n_amb, q_list, status, pix_bits, bitmap, BW, tiles_x, (uint8_t *)occ, OW);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_BP_MARK) {
    hipLaunchKernelGGL(k_bp_scan, dim3(V), dim3(SCB), 0, st, BW, NJ, tiles_x, bitmap, wpre, n_vox, cap_vox, status, occ, OW,
                       NSEG, seg_list, n_occ);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_BP_VOX) {
    // a wave per occupied segment (more of them per wave only when a view occupies more than 2 048)
    const int gx = std::min((NSEG + (VXB / 64) - 1) / (VXB / 64), VOX_GX);
    hipLaunchKernelGGL(k_bp_vox, dim3(gx, (V + 7) / 8 * 8), dim3(VXB), 0, st, depth, cal, masks, mask_format, n_inst, max_inst, W, HW, E_view,
                       table, E_total, cap_vox, Wk, out, key_axis, pix_bits, geom->max_points_per_voxel, geom->max_voxels,
                       cap_q, q_bins, q_rank, n_q, status, V);
    DFU3D_LAUNCH_CHECK();
  }
  if (phases & DFU3D_BP_REPAIR) {
    // exact repair of the queued bins (more than max_points pixels, or a key collision below the cut of the
    // packed word); every kernel leaves at once for a view whose queue is empty
    const int nblk = (HW + PB - 1) / PB;
    // (grids of a few workgroups per view: the queues are empty in all but pathological passes, and 75 000 workgroups that
    // start only to find that out cost 25 us per pass; a view under repair walks its pixels with the grid's stride)
    hipLaunchKernelGGL(k_bp_rebin, dim3(nblk < 16 ? nblk : 16, V), dim3(PB), 0, st, depth, cal, *geom, W, HW,
                       key_axis, n_q, pix_bin);
    DFU3D_LAUNCH_CHECK();
    const int ga = (cap_q + 255) / 256;
    hipLaunchKernelGGL(k_ovf_alloc, dim3(ga < 16 ? ga : 16, V), dim3(256), 0, st, table, E_total, E_view, cap_q,
                       q_bins, n_q, q_cnt, q_cursor, HW, status);
    DFU3D_LAUNCH_CHECK();
    const int nblk4 = (HW + PBLK - 1) / PBLK;
    hipLaunchKernelGGL(k_ovf_gather, dim3(nblk4 < 16 ? nblk4 : 16, V), dim3(PB), 0, st, pix_bin, table, E_total,
                       E_view, HW, n_q, q_list);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ovf_select, dim3(cap_q < 32 ? cap_q : 32, V), dim3(256), 0, st, depth, cal, W, HW,
                       key_axis, geom->max_points_per_voxel, table, E_total, E_view, cap_q,
                       q_bins, n_q, q_cnt, q_list);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bp_fix, dim3(ga < 16 ? ga : 16, V), dim3(256), 0, st, depth, cal, masks, mask_format, n_inst,
                       max_inst, W, HW, geom->max_voxels, E_view, table, E_total, cap_vox, out, cap_q, q_bins, q_rank,
                       n_q);
    DFU3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bp_finalize, dim3((V + 255) / 256), dim3(256), 0, st, V, geom->max_voxels, cap_vox, n_vox);
    DFU3D_LAUNCH_CHECK();
  }
  return DFU3D_OK;
}
