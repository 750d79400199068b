// lidar_stage.hip -- LiDAR-side kernels: FOV filter, seeded plane RANSAC,
// above-plane + point->pixel label inheritance.
//
// One workgroup (1024 threads = 16 waves) per camera view walks the frame's
// (N,4) float32 points with float4 (16 B/lane) loads and keeps the reference's
// ORDER by ballot + popcount ranks inside each wave and an LDS hand-off of the
// 16 wave totals (ordered compaction, hazard H3).
#include "common.hpp"

namespace {

// Threads per workgroup of the one-workgroup-per-view kernels (k_fov_filter, k_plane_ransac, k_project_rows).  A step of 64
// frames is 384 views on 256 compute units: with 1024 threads -- and k_plane_ransac's 128 registers -- a compute unit held ONE
// workgroup and the kernels ran as two rounds; with 512 two fit and all views are in flight at once.  Measured in one
// process (tools/ab_builds.py), 1024 / 512 / 256 threads: k_plane_ransac 0.199 / 0.148 / 0.185 ms, k_fov_filter 0.118 /
// 0.103 / 0.141, k_project_rows + k_label_rows 0.048 / 0.041 / 0.046.
constexpr int NT = 512;
constexpr int NW = NT / 64;

// ---------------------------------------------------------------- a4 FOV
// vis_utils.py:108-123 / 152-154
// Every wave owns a contiguous range of the frame's points (FOV_PW per pass of the workgroup): it evaluates its
// points, keeps the FOV flags as ballot words in registers, counts them, and after ONE hand-off of the 16 wave totals
// writes its indices behind those of the waves before it -- the order of the reference, two barriers per 16 384
// points instead of two per 1024.
constexpr int FOV_PW = 1024;                       // points per wave and pass
__global__ __launch_bounds__(NT) void k_fov_filter(
    const float4 *__restrict__ pts, const int *__restrict__ pt_off,
    const int *__restrict__ view_frame, const ViewCalib *__restrict__ calib,
    float fovH, float fovW, int capN, int *__restrict__ fov_idx,
    int *__restrict__ n_fov) {
  __shared__ int s_w[NW];
  const int v = blockIdx.x;
  const int f = view_frame[v];
  const int p0 = pt_off[f];
  const int n = pt_off[f + 1] - p0;
  const ViewCalib c = calib[v];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  int running = 0;
  for (int base = 0; base < n; base += NW * FOV_PW) {          // uniform
    const int w0 = base + wave * FOV_PW;
    unsigned long long flags[FOV_PW / 64];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < FOV_PW / 64; k++) {
      const int i = w0 + k * 64 + lane;
      bool ok = false;
      if (i < n) {
        const float4 p = pts[p0 + i];
        float r[3], u, w, d;
        lidar_to_rect_f32(c.M43, p.x, p.y, p.z, r);
        rect_to_img_f32(c.P2, r, u, w, d);
        ok = (u >= 0.0f) && (u < fovW) && (w >= 0.0f) && (w < fovH) && (d >= 0.0f);
      }
      flags[k] = __ballot(ok);
      mine += __popcll(flags[k]);
    }
    __syncthreads();                                           // the previous pass has read s_w
    if (lane == 0) s_w[wave] = mine;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ww++) {
      const int cw = s_w[ww];
      before += (ww < wave) ? cw : 0;
      tot += cw;
    }
    int pos = running + before;
#pragma unroll
    for (int k = 0; k < FOV_PW / 64; k++) {
      const unsigned long long m = flags[k];
      if ((m >> lane) & 1ull) {
        const int q = pos + __popcll(m & ((1ull << lane) - 1ull));
        if (q < capN) fov_idx[(size_t)v * capN + q] = w0 + k * 64 + lane;
      }
      pos += __popcll(m);
    }
    running += tot;
  }
  if (threadIdx.x == 0) n_fov[v] = running < capN ? running : capN;
}

// ---------------------------------------------------------------- a5 RANSAC
// Seeded 3-point RANSAC with sklearn's inlier rule (|res| <= MAD(z)), see
// oracle/penet_oracle.py:plane_ransac for the bit-level specification.

__device__ __forceinline__ unsigned int f32_key(float v) {
  const unsigned int b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// One histogram increment per lane with active set, for a wave whose keys crowd into few buckets (the high bytes
// of a view's heights do: thousands of equal sign/exponent bytes): up to four rounds in which the first remaining
// lane's bucket is counted by ballot and added once, plain LDS atomics for what is left.  Same-address LDS
// atomics serialise lane by lane; this was most of the kernel's time.
__device__ __forceinline__ void hist_add_wave(int *hist, bool active, int bucket) {
  unsigned long long rem = __ballot(active);
  const int lane = lane_id();
  for (int r = 0; r < 4 && rem; r++) {                      // uniform
    const int leader = __ffsll((long long)rem) - 1;
    const int lb = __shfl(bucket, leader, 64);
    const bool same = active && bucket == lb;
    const unsigned long long sm = __ballot(same);
    if (lane == leader) atomicAdd(&hist[lb], __popcll(sm));
    rem &= ~sm;
    if (same) active = false;
  }
  if (active) atomicAdd(&hist[bucket], 1);
}

// Radix select of the value with 0-based rank k among n keys produced by
// keyfn(i) (64-bit ordered keys).  All threads of the block call it; returns
// the key to every thread.  hist: 256 ints of LDS, s_sel: 3 x u64 of LDS.
// Leaves as soon as the bucket that holds the rank has a single key in it (one more sweep fetches that key).
template <typename KeyFn>
__device__ unsigned long long block_select(int n, int k, KeyFn keyfn, int *hist,
                                           unsigned long long *s_sel) {
  unsigned long long prefix = 0ull, mask = 0ull;
  int kk = k;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int b = threadIdx.x; b < 256; b += NT) hist[b] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += NT) {              // uniform trip count: ballots inside
      const int i = base + threadIdx.x;
      unsigned long long key = 0ull;
      bool act = false;
      if (i < n) { key = keyfn(i); act = (key & mask) == prefix; }
      hist_add_wave(hist, act, (int)((key >> shift) & 0xFFull));
    }
    __syncthreads();
    if (threadIdx.x < 64) {       // the bucket that holds rank kk: one wave, four buckets per lane
      const int l = threadIdx.x;
      const int c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
      const int inc = wave_incl_scan(c0 + c1 + c2 + c3);
      const unsigned long long past = __ballot(inc > kk);     // lanes whose running total passes kk
      const int tl = past ? (__ffsll((long long)past) - 1) : 63;   // (past == 0 cannot happen for k < n)
      if (l == tl) {
        int acc = inc - (c0 + c1 + c2 + c3), b = 4 * l, cb = c0;
        if (acc + c0 <= kk) { acc += c0; b++; cb = c1;
          if (acc + c1 <= kk) { acc += c1; b++; cb = c2;
            if (acc + c2 <= kk) { acc += c2; b++; cb = c3; } } }
        s_sel[0] = prefix | ((unsigned long long)b << shift);
        s_sel[1] = (unsigned long long)(kk - acc);
        s_sel[2] = (unsigned long long)cb;
      }
    }
    __syncthreads();
    prefix = s_sel[0];
    kk = (int)s_sel[1];
    const int in_bucket = (int)s_sel[2];
    mask |= (0xFFull << shift);
    __syncthreads();
    if (in_bucket == 1 && shift > 0) {                      // (uniform) exactly one key carries this prefix
      for (int i = threadIdx.x; i < n; i += NT) {
        const unsigned long long key = keyfn(i);
        if ((key & mask) == prefix) s_sel[0] = key;
      }
      __syncthreads();
      prefix = s_sel[0];
      __syncthreads();
      return prefix;
    }
  }
  return prefix;
}

__device__ __forceinline__ double key_to_double(unsigned long long k) {
  const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// median of n fp64 values valfn(i) exactly as np.median: mean of the two
// middle order statistics for even n.  The lower one needs no second selection: it is the upper one again when
// enough keys equal it, otherwise the largest key below it -- one sweep.
template <typename ValFn>
__device__ double block_median(int n, ValFn valfn, int *hist, unsigned long long *s_sel) {
  auto keyfn = [&](int i) { return ordered_key(valfn(i)); };
  const int k2 = n / 2;
  const unsigned long long hk = block_select(n, k2, keyfn, hist, s_sel);
  const double hi = key_to_double(hk);
  if (n & 1) return hi;
  if (threadIdx.x == 0) { s_sel[0] = 0ull; hist[0] = 0; }
  __syncthreads();
  int less = 0;
  unsigned long long below = 0ull;                          // ordered keys of real numbers are > 0
  for (int i = threadIdx.x; i < n; i += NT) {
    const unsigned long long key = keyfn(i);
    if (key < hk) { less++; below = key > below ? key : below; }
  }
  less = wave_sum_i(less);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = ((unsigned long long)(unsigned int)__shfl_xor((int)(below >> 32), m, 64) << 32) |
                                 (unsigned long long)(unsigned int)__shfl_xor((int)(below & 0xFFFFFFFFull), m, 64);
    below = o > below ? o : below;
  }
  if (lane_id() == 0) {
    if (less) atomicAdd(&hist[0], less);
    if (below) atomicMax(&s_sel[0], below);
  }
  __syncthreads();
  const int n_less = hist[0];
  const unsigned long long lk = (n_less <= k2 - 1) ? hk : s_sel[0];
  __syncthreads();
  return (key_to_double(lk) + hi) / 2.0;
}

__device__ __forceinline__ double block_sum_d(double v, double *s_red) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if (lane_id() == 0) s_red[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NW; i++) t += s_red[i];
  return t;
}
__device__ __forceinline__ int block_sum_i(int v, int *s_red) {
  v = wave_sum_i(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if (lane_id() == 0) s_red[w] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) t += s_red[i];
  return t;
}

constexpr int RB = 8;   // RANSAC trials scored per sweep over the candidates
constexpr int RS_LDS = 3072;   // candidates kept in LDS (48 KB); a view of the bench has ~2500

__global__ __launch_bounds__(NT) void k_plane_ransac(
    const float4 *__restrict__ pts, const int *__restrict__ pt_off,
    const int *__restrict__ view_frame, const int *__restrict__ fov_idx,
    const int *__restrict__ n_fov, int capN, float max_hs, float xy_range,
    int trials, unsigned long long seed, const long long *__restrict__ key,
    int *__restrict__ cand_idx, double *__restrict__ plane) {
  __shared__ int s_w[NW];
  __shared__ int s_hist[256];
  __shared__ unsigned long long s_sel[3];
  __shared__ double s_red[NW];
  __shared__ int s_redi[NW];
  __shared__ double s_model[RB][3];
  __shared__ int s_valid[RB], s_tot[RB], s_cnt[NW][RB];
  __shared__ float4 s_pt[RS_LDS];
  const int v = blockIdx.x;
  const int p0 = pt_off[view_frame[v]];
  const int nf = n_fov[v];
  const int *fidx = fov_idx + (size_t)v * capN;
  int *cidx = cand_idx + (size_t)v * capN;
  // candidate set: my_loader.py:449-453 (float32 compares)
  int n = 0;
  for (int base = 0; base < nf; base += NT) {
    const int t = base + threadIdx.x;
    bool ok = false;
    int pi = 0;
    if (t < nf) {
      pi = fidx[t];
      const float4 p = pts[p0 + pi];
      ok = (p.z < max_hs) && (p.x > -xy_range) && (p.x < xy_range) &&
           (p.y > -xy_range) && (p.y < xy_range);
    }
    int tot;
    const int r = block_rank<NW>(ok, s_w, tot);
    if (ok) cidx[n + r] = pi;
    n += tot;
  }
  __syncthreads();
  // the candidates' coordinates in LDS (the first RS_LDS of them): every sweep below -- two medians of eight radix passes,
  // thirteen batches of trials, the refit -- read a candidate through its index (two dependent global round trips);
  // from LDS it is one
  for (int i = threadIdx.x; i < min(n, RS_LDS); i += NT) s_pt[i] = pts[p0 + cidx[i]];
  __syncthreads();
  auto cand = [&](int i) -> float4 { return i < RS_LDS ? s_pt[i] : pts[p0 + cidx[i]]; };
  double *out = plane + (size_t)v * 4;
  if (n < 3) {
    if (threadIdx.x == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; out[3] = 1e30; }
    return;
  }
  auto zval = [&](int i) { return (double)cand(i).z; };
  const double med = block_median(n, zval, s_hist, s_sel);
  auto dev = [&](int i) { return fabs((double)cand(i).z - med); };
  const double thr = block_median(n, dev, s_hist, s_sel);

  // trials in batches of RB: thread t < RB draws the samples of trial t0+t and fits its
  // plane, then every thread scores its points against the RB planes at once and the RB
  // inlier counts are reduced together (3 barriers per batch instead of 2 per trial);
  // the winner is still the FIRST trial with the largest count
  int best_cnt = -1;
  double ba = 0.0, bb = 0.0, bc = 0.0;
  const unsigned long long vkey = (unsigned long long)key[v];
  for (int t0 = 0; t0 < trials; t0 += RB) {
    if (threadIdx.x < RB) {
      const int t = t0 + (int)threadIdx.x;
      bool valid = false;
      double ca = 0.0, cb = 0.0, cc = 0.0;
      if (t < trials) {
        int idx[3];
        int got = 0;
        for (int a = 0; a < 64 && got < 3; a++) {
          const unsigned long long r = mix64(seed ^ mix64((vkey << 20) ^ ((unsigned long long)t << 8) ^ (unsigned long long)a));
          const int i = (int)__umul64hi(r, (unsigned long long)n);
          bool dup = false;
          for (int q = 0; q < got; q++) dup |= (idx[q] == i);
          if (!dup) idx[got++] = i;
        }
        if (got == 3) {
          const float4 q0 = cand(idx[0]), q1 = cand(idx[1]), q2 = cand(idx[2]);
          const double x0 = q0.x, y0 = q0.y, z0 = q0.z;
          const double dx1 = (double)q1.x - x0, dy1 = (double)q1.y - y0, dz1 = (double)q1.z - z0;
          const double dx2 = (double)q2.x - x0, dy2 = (double)q2.y - y0, dz2 = (double)q2.z - z0;
          const double det = dx1 * dy2 - dx2 * dy1;
          if (fabs(det) > 1e-12) {
            ca = (dz1 * dy2 - dz2 * dy1) / det;
            cb = (dx1 * dz2 - dx2 * dz1) / det;
            cc = (z0 - ca * x0) - cb * y0;
            valid = true;
          }
        }
      }
      s_model[threadIdx.x][0] = ca; s_model[threadIdx.x][1] = cb; s_model[threadIdx.x][2] = cc;
      s_valid[threadIdx.x] = valid ? 1 : 0;
    }
    __syncthreads();
    int cnt[RB];
#pragma unroll
    for (int k = 0; k < RB; k++) cnt[k] = 0;
    for (int i = threadIdx.x; i < n; i += NT) {
      const float4 p = cand(i);
      const double px_ = (double)p.x, py_ = (double)p.y, pz_ = (double)p.z;
#pragma unroll
      for (int k = 0; k < RB; k++) {
        const double res = fabs(pz_ - ((s_model[k][0] * px_ + s_model[k][1] * py_) + s_model[k][2]));
        cnt[k] += (res <= thr) ? 1 : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < RB; k++) {
      const int c = wave_sum_i(cnt[k]);
      if (lane_id() == 0) s_cnt[threadIdx.x >> 6][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < RB) {
      int c = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) c += s_cnt[w][threadIdx.x];
      s_tot[threadIdx.x] = c;
    }
    __syncthreads();
    for (int k = 0; k < RB; k++) {                 // in trial order, in every thread alike
      if (!s_valid[k]) continue;
      const int c = s_tot[k];
      if (c > best_cnt) { best_cnt = c; ba = s_model[k][0]; bb = s_model[k][1]; bc = s_model[k][2]; }
    }
    __syncthreads();
  }
  if (best_cnt < 0) {
    if (threadIdx.x == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; out[3] = 1e30; }
    return;
  }
  // least-squares refit on the inliers (centred normal equations)
  double sx = 0.0, sy = 0.0, sz = 0.0;
  int k = 0;
  for (int i = threadIdx.x; i < n; i += NT) {
    const float4 p = cand(i);
    const double res = fabs((double)p.z - ((ba * (double)p.x + bb * (double)p.y) + bc));
    if (res <= thr) { sx += p.x; sy += p.y; sz += p.z; k++; }
  }
  k = block_sum_i(k, s_redi);
  sx = block_sum_d(sx, s_red);
  sy = block_sum_d(sy, s_red);
  sz = block_sum_d(sz, s_red);
  const double mx = sx / k, my = sy / k, mz = sz / k;
  double sxx = 0.0, sxy = 0.0, syy = 0.0, sxz = 0.0, syz = 0.0;
  for (int i = threadIdx.x; i < n; i += NT) {
    const float4 p = cand(i);
    const double res = fabs((double)p.z - ((ba * (double)p.x + bb * (double)p.y) + bc));
    if (res <= thr) {
      const double ux = (double)p.x - mx, uy = (double)p.y - my, uz = (double)p.z - mz;
      sxx += ux * ux; sxy += ux * uy; syy += uy * uy; sxz += ux * uz; syz += uy * uz;
    }
  }
  sxx = block_sum_d(sxx, s_red);
  sxy = block_sum_d(sxy, s_red);
  syy = block_sum_d(syy, s_red);
  sxz = block_sum_d(sxz, s_red);
  syz = block_sum_d(syz, s_red);
  if (threadIdx.x == 0) {
    double ca = ba, cb = bb, cc = bc;
    const double det2 = sxx * syy - sxy * sxy;
    const double ref = fmax(sxx * syy, 1e-300);
    if (det2 > 1e-12 * ref) {
      ca = (sxz * syy - syz * sxy) / det2;
      cb = (syz * sxx - sxz * sxy) / det2;
      cc = (mz - ca * mx) - cb * my;
    }
    const double norm = sqrt((ca * ca + cb * cb) + 1.0);   // my_loader.py:457-466
    out[0] = -(ca / norm);
    out[1] = -(cb / norm);
    out[2] = 1.0 / norm;
    out[3] = -(cc / norm);
  }
}

// ---------------------------------------------------------------- a5/a6
// my_loader.py:471-477 (above_plane), 517-530 (round + in-bounds compaction)
__global__ __launch_bounds__(NT) void k_project_rows(
    const float4 *__restrict__ pts, const int *__restrict__ pt_off,
    const int *__restrict__ view_frame, const ViewCalib *__restrict__ calib,
    const double *__restrict__ plane, const int *__restrict__ fov_idx,
    const int *__restrict__ n_fov, int capN, float boundH, float boundW, int W,
    double plane_offset, float xy_range, int *__restrict__ ag_pt,
    int *__restrict__ ib_pix, int *__restrict__ n_ag, int *__restrict__ Kout) {
  __shared__ int s_w[NW];
  const int v = blockIdx.x;
  const int p0 = pt_off[view_frame[v]];
  const int nf = n_fov[v];
  const ViewCalib c = calib[v];
  const double pl0 = plane[v * 4 + 0], pl1 = plane[v * 4 + 1], pl2 = plane[v * 4 + 2],
               pl3 = plane[v * 4 + 3];
  const double pn = sqrt((pl0 * pl0 + pl1 * pl1) + pl2 * pl2);
  const int *fidx = fov_idx + (size_t)v * capN;
  int nag = 0, nib = 0;
  for (int base = 0; base < nf; base += NT) {
    const int t = base + threadIdx.x;
    bool ag = false, ib = false;
    int pi = 0, pix = 0;
    if (t < nf) {
      pi = fidx[t];
      const float4 p = pts[p0 + pi];
      double d = (((double)p.x * pl0 + (double)p.y * pl1) + (double)p.z * pl2) + pl3;
      d = d / pn;
      const bool below = d < plane_offset;
      const bool inr = (p.x < xy_range) && (p.x > -xy_range) && (p.y < xy_range) && (p.y > -xy_range);
      ag = !(below && inr);
      if (ag) {
        float r[3], u, w, dep;
        lidar_to_rect_f32(c.M43, p.x, p.y, p.z, r);
        rect_to_img_f32(c.P2, r, u, w, dep);
        const float ru = rintf(u), rv = rintf(w);        // np.round: half to even
        ib = (0.0f <= ru) && (ru < boundW) && (0.0f <= rv) && (rv < boundH);
        if (ib) pix = (int)rv * W + (int)ru;
      }
    }
    int tot_ag, tot_ib;
    const int r_ag = block_rank<NW>(ag, s_w, tot_ag);
    const int r_ib = block_rank<NW>(ib, s_w, tot_ib);
    if (ag) ag_pt[(size_t)v * capN + nag + r_ag] = pi;
    if (ib) ib_pix[(size_t)v * capN + nib + r_ib] = pix;
    nag += tot_ag;
    nib += tot_ib;
  }
  if (threadIdx.x == 0) {
    n_ag[v] = nag;
    Kout[v] = nib < nag ? nib : nag;   // my_loader.py:527
  }
}

// row t < K: point = ag_pt[t] (positional), pixel = ib_pix[t]; bits from masks
__global__ __launch_bounds__(256) void k_label_rows(
    const float4 *__restrict__ pts, const int *__restrict__ pt_off,
    const int *__restrict__ view_frame, const int *__restrict__ ag_pt,
    const int *__restrict__ ib_pix, const int *__restrict__ Kin,
    const void *__restrict__ masks, int mask_format, const int *__restrict__ n_inst, int max_inst,
    int HW, int capN, uint32_t *__restrict__ it_bits, double *__restrict__ it_x,
    double *__restrict__ it_y, double *__restrict__ it_z) {
  const int v = blockIdx.y;
  const int K = Kin[v];
  const int p0 = pt_off[view_frame[v]];
  const int m = min(max(n_inst[v], 0), max_inst);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < K; t += gridDim.x * 256) {   // K is a few hundred rows
    const size_t o = (size_t)v * capN + t;
    const float4 p = pts[p0 + ag_pt[o]];
    it_bits[o] = mask_bits_at(masks, mask_format, v, max_inst, m, HW, ib_pix[o]);
    it_x[o] = (double)p.x;
    it_y[o] = (double)p.y;
    it_z[o] = (double)p.z;
  }
}

// uint8 planes (V, max_inst, H, W) -> one word per pixel (V, H, W), bit j = plane j > 0 for j < n_inst[v]
template <typename WordT>
__global__ __launch_bounds__(256) void k_pack_masks(const uint8_t *__restrict__ masks,
                                                    const int *__restrict__ n_inst, int max_inst,
                                                    int HW, WordT *__restrict__ out) {
  const int v = blockIdx.y;
  const int m = min(max(n_inst[v], 0), max_inst);
  const uint8_t *mb = masks + (size_t)v * max_inst * HW;
  // 4 pixels per thread: one 32-bit load per plane
  for (int q = blockIdx.x * 256 + threadIdx.x; q < (HW + 3) / 4; q += gridDim.x * 256) {
    const int pix = q * 4;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (pix + 4 <= HW && (HW & 3) == 0) {
      for (int j = 0; j < m; j++) {
        const uint32_t b4 = *(const uint32_t *)(mb + (size_t)j * HW + pix);
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] |= (((b4 >> (8 * k)) & 0xFFu) != 0u) ? (1u << j) : 0u;
      }
    } else {
      for (int j = 0; j < m; j++)
        for (int k = 0; k < 4 && pix + k < HW; k++) w[k] |= (mb[(size_t)j * HW + pix + k] > 0) ? (1u << j) : 0u;
    }
    for (int k = 0; k < 4 && pix + k < HW; k++) out[(size_t)v * HW + pix + k] = (WordT)w[k];
  }
}

}  // namespace

extern "C" int dfu3d_fov_filter(const float *points, const int32_t *pt_off,
                                const int32_t *view_frame, const float *calib,
                                int32_t V, int32_t fov_h, int32_t fov_w,
                                int32_t cap_n, int32_t *fov_idx, int32_t *n_fov,
                                void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!points || !pt_off || !view_frame || !calib || !fov_idx || !n_fov) return DFU3D_EINVAL;
  if (V <= 0 || cap_n <= 0 || fov_h <= 0 || fov_w <= 0) return DFU3D_EINVAL;
  hipLaunchKernelGGL(k_fov_filter, dim3(V), dim3(NT), 0, (hipStream_t)stream,
                     (const float4 *)points, pt_off, view_frame,
                     (const ViewCalib *)calib, (float)fov_h, (float)fov_w, cap_n,
                     fov_idx, n_fov);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_plane_ransac(const float *points, const int32_t *pt_off,
                                  const int32_t *view_frame, const int32_t *fov_idx,
                                  const int32_t *n_fov, int32_t V, int32_t cap_n,
                                  double max_hs, double xy_range, int32_t trials,
                                  uint64_t seed, const int64_t *key,
                                  int32_t *cand_idx, double *plane, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!points || !pt_off || !view_frame || !fov_idx || !n_fov || !key || !cand_idx || !plane)
    return DFU3D_EINVAL;
  if (V <= 0 || cap_n <= 0 || trials < 0) return DFU3D_EINVAL;
  hipLaunchKernelGGL(k_plane_ransac, dim3(V), dim3(NT), 0, (hipStream_t)stream,
                     (const float4 *)points, pt_off, view_frame, fov_idx, n_fov,
                     cap_n, (float)max_hs, (float)xy_range, trials,
                     (unsigned long long)seed, (const long long *)key, cand_idx, plane);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_project_label(
    const float *points, const int32_t *pt_off, const int32_t *view_frame,
    const float *calib, const double *plane, const int32_t *fov_idx,
    const int32_t *n_fov, const void *masks, int32_t mask_format, const int32_t *n_inst, int32_t V,
    int32_t max_inst, int32_t H, int32_t W, int32_t bounds_h, int32_t bounds_w, int32_t cap_n,
    double plane_offset, double xy_range, int32_t *ag_pt, int32_t *ib_pix, int32_t *n_ag, int32_t *K,
    uint32_t *it_bits, double *it_x, double *it_y, double *it_z, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!points || !pt_off || !view_frame || !calib || !plane || !fov_idx || !n_fov ||
      !masks || !n_inst || !ag_pt || !ib_pix || !n_ag || !K || !it_bits || !it_x ||
      !it_y || !it_z)
    return DFU3D_EINVAL;
  if (V <= 0 || cap_n <= 0 || H <= 0 || W <= 0 || max_inst <= 0) return DFU3D_EINVAL;
  if (bounds_h <= 0 || bounds_w <= 0 || bounds_h > H || bounds_w > W) return DFU3D_EINVAL;
  if (max_inst > DFU3D_MAX_INST) return DFU3D_ERANGE;
  if (!mask_format_ok(mask_format, max_inst)) return DFU3D_EINVAL;
  hipLaunchKernelGGL(k_project_rows, dim3(V), dim3(NT), 0, (hipStream_t)stream,
                     (const float4 *)points, pt_off, view_frame,
                     (const ViewCalib *)calib, plane, fov_idx, n_fov, cap_n,
                     (float)bounds_h, (float)bounds_w, W, plane_offset, (float)xy_range, ag_pt,
                     ib_pix, n_ag, K);
  DFU3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_label_rows, dim3((cap_n + 255) / 256 < 8 ? (cap_n + 255) / 256 : 8, V), dim3(256), 0,
                     (hipStream_t)stream, (const float4 *)points, pt_off, view_frame,
                     ag_pt, ib_pix, K, masks, mask_format, n_inst, max_inst, H * W, cap_n, it_bits,
                     it_x, it_y, it_z);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

extern "C" int dfu3d_pack_masks(const uint8_t *masks, const int32_t *n_inst, int32_t V, int32_t max_inst,
                                int32_t H, int32_t W, void *out, int32_t word_bytes, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!masks || !n_inst || !out) return DFU3D_EINVAL;
  if (V <= 0 || H <= 0 || W <= 0 || max_inst <= 0) return DFU3D_EINVAL;
  if (max_inst > DFU3D_MAX_INST) return DFU3D_ERANGE;
  if (word_bytes == 0 || !mask_format_ok(word_bytes, max_inst)) return DFU3D_EINVAL;
  const int HW = H * W;
  const int gx = ((HW + 3) / 4 + 255) / 256;
  const dim3 grid(gx < 2048 ? gx : 2048, V);
  hipStream_t st = (hipStream_t)stream;
  if (word_bytes == 1)
    hipLaunchKernelGGL(k_pack_masks<uint8_t>, grid, dim3(256), 0, st, masks, n_inst, max_inst, HW, (uint8_t *)out);
  else if (word_bytes == 2)
    hipLaunchKernelGGL(k_pack_masks<uint16_t>, grid, dim3(256), 0, st, masks, n_inst, max_inst, HW, (uint16_t *)out);
  else
    hipLaunchKernelGGL(k_pack_masks<uint32_t>, grid, dim3(256), 0, st, masks, n_inst, max_inst, HW, (uint32_t *)out);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}
