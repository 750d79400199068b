// chain_stage.hip -- the whole pseudo-box path for V views behind ONE C call
// (vis_utils.py:136-166 -> my_loader.py:502-702 -> label rows), for hosts that do not want
// to sequence the stage entry points themselves.  It only sequences them: every kernel is
// the one the per-stage entry points launch, the workspace is carved from one caller-owned
// buffer, nothing is allocated, nothing synchronises.
#include "common.hpp"

namespace {

struct ChainWs {
  int32_t *fov_idx, *cand_idx, *ag_pt, *ib_pix, *n_fov, *n_ag, *K;
  double *plane;
  uint32_t *a_bits;
  double *a_x, *a_y, *a_z;
  int32_t *n_vox;
  uint32_t *vox_pix, *b_bits;
  double *b_x, *b_y, *b_z;
  void *table;
  uint32_t *pix_bin;
  int32_t *blk_cnt;
  double *px, *py, *pz, *sx, *sy;
  int32_t *label, *sroot, *si3;
  double *fit_ws;
  uint8_t *flags;
  int64_t *base_a, *base_b, *base_ab;
  int32_t *cnt_a, *cnt_b, *cnt_all, *cnt_ab, *tile_off, *queue, *stat_enable;
  double *rad_ab, *mean_d;
  void *vd_scratch;
  void *shadow;
  int32_t *chunk_cnt;
  int64_t *pool_cursor;
  int64_t table_entries;
};

// carve the workspace; with base == nullptr only the size is computed
int64_t carve(const dfu3d_chain_cfg *c, char *base, ChainWs *w) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) -> char * {
    char *p = base ? base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return p;
  };
  const int64_t V = c->V, S = (int64_t)c->V * c->max_inst, N = V * c->cap_n, X = V * c->cap_vox, P = c->pool_cap;
  ChainWs t;
  t.fov_idx = (int32_t *)take(4 * N); t.cand_idx = (int32_t *)take(4 * N);
  t.ag_pt = (int32_t *)take(4 * N); t.ib_pix = (int32_t *)take(4 * N);
  t.n_fov = (int32_t *)take(4 * V); t.n_ag = (int32_t *)take(4 * V); t.K = (int32_t *)take(4 * V);
  t.plane = (double *)take(8 * V * 4);
  t.a_bits = (uint32_t *)take(4 * N);
  t.a_x = (double *)take(8 * N); t.a_y = (double *)take(8 * N); t.a_z = (double *)take(8 * N);
  t.n_vox = (int32_t *)take(4 * V);
  t.vox_pix = (uint32_t *)take(4 * X); t.b_bits = (uint32_t *)take(4 * X);
  t.b_x = (double *)take(8 * X); t.b_y = (double *)take(8 * X); t.b_z = (double *)take(8 * X);
  t.table_entries = (int64_t)c->geom.t_n * c->geom.p_n;
  int64_t pw = 0, bw = 0;
  if (c->dense) {
    dfu3d_backproject_scratch_words(c->V, c->H, c->W, c->cap_vox, c->geom.max_points_per_voxel, t.table_entries, &pw, &bw);
    t.table = take(V * t.table_entries * DFU3D_TABLE_ENTRY_BYTES);
    t.pix_bin = (uint32_t *)take(4 * pw);
    t.blk_cnt = (int32_t *)take(4 * bw);
  } else {
    t.table = nullptr; t.pix_bin = nullptr; t.blk_cnt = nullptr;
  }
  t.px = (double *)take(8 * P); t.py = (double *)take(8 * P); t.pz = (double *)take(8 * P);
  t.sx = (double *)take(8 * P); t.sy = (double *)take(8 * P);
  t.label = (int32_t *)take(4 * P); t.sroot = (int32_t *)take(4 * P); t.si3 = (int32_t *)take(4 * 3 * P);
  t.fit_ws = (double *)take(8 * dfu3d_lshape_fit_ws_doubles(P, c->cap_rows));
  t.flags = (uint8_t *)take(P);
  t.base_a = (int64_t *)take(8 * S); t.base_b = (int64_t *)take(8 * S); t.base_ab = (int64_t *)take(8 * 2 * S);
  t.cnt_a = (int32_t *)take(4 * S); t.cnt_b = (int32_t *)take(4 * S); t.cnt_all = (int32_t *)take(4 * S);
  t.cnt_ab = (int32_t *)take(4 * 2 * S); t.tile_off = (int32_t *)take(4 * (2 * S + 2));
  t.queue = (int32_t *)take(4 * DFU3D_RF_QUEUE_INTS(P)); t.stat_enable = (int32_t *)take(4 * S);
  t.shadow = take(DFU3D_SHADOW_BYTES(P));
  t.chunk_cnt = (int32_t *)take(4 * dfu3d_segments_scratch_words(c->V, c->cap_n, c->cap_vox));
  t.rad_ab = (double *)take(8 * 2 * S);
  t.mean_d = c->stat_filter ? (double *)take(8 * P) : nullptr;
  t.vd_scratch = c->stat_filter ? (void *)take(dfu3d_voxel_down_sample_scratch_bytes(P)) : nullptr;
  t.pool_cursor = (int64_t *)take(8);
  if (w) *w = t;
  return off;
}

bool cfg_ok(const dfu3d_chain_cfg *c) {
  return c && c->V > 0 && c->H > 0 && c->W > 0 && c->max_inst > 0 && c->max_inst <= DFU3D_MAX_INST &&
         c->cap_n > 0 && c->cap_vox > 0 && c->cap_rows > 0 && c->pool_cap > 0 && c->n_theta > 0 &&
         c->bounds_h > 0 && c->bounds_w > 0 && c->bounds_h <= c->H && c->bounds_w <= c->W &&
         mask_format_ok(c->mask_format, c->max_inst) &&
         (!c->dense || (c->geom.t_n > 0 && c->geom.p_n > 0)) &&
         (!c->stat_filter || (c->stat_voxel > 0.0 && c->stat_nb_neighbors >= 1));
}

// apply_fov == 0: the caller's points are already the FOV points (vis_utils.py:152-154 done upstream)
__global__ void k_all_points(const int *__restrict__ pt_off, const int *__restrict__ view_frame,
                             int cap_n, int *__restrict__ fov_idx, int *__restrict__ n_fov) {
  const int v = blockIdx.y;
  const int f = view_frame[v];
  const int n = min(pt_off[f + 1] - pt_off[f], cap_n);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap_n; i += gridDim.x * blockDim.x)
    fov_idx[(size_t)v * cap_n + i] = i;
  if (blockIdx.x == 0 && threadIdx.x == 0) n_fov[v] = n;
}

__global__ void k_sum_counts(int S, const int *__restrict__ a, const int *__restrict__ b, int *__restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S) out[s] = a[s] + b[s];
}

__global__ void k_fill_i32(int n, int v, int *__restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

extern "C" int64_t dfu3d_chain_workspace_bytes(const dfu3d_chain_cfg *cfg) {
  if (!cfg_ok(cfg)) return DFU3D_EINVAL;
  return carve(cfg, nullptr, nullptr);
}

extern "C" int64_t dfu3d_workspace_bytes(int32_t stage, const dfu3d_sizes *z) {
  if (!z || z->V <= 0 || z->max_inst <= 0 || z->max_inst > DFU3D_MAX_INST) return DFU3D_EINVAL;
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  const int64_t V = z->V, S = V * z->max_inst, P = z->pool_cap, N = V * (int64_t)z->cap_n;
  switch (stage) {
    case DFU3D_STAGE_FOV_FILTER:
      return 0;
    case DFU3D_STAGE_SEGMENTS_BUILD:                     /* chunk_cnt */
      return (z->cap_n > 0 && z->cap_vox > 0) ? up(4 * dfu3d_segments_scratch_words(z->V, z->cap_n, z->cap_vox)) : DFU3D_EINVAL;
    case DFU3D_STAGE_PLANE_RANSAC:                       /* cand_idx */
      return z->cap_n > 0 ? up(4 * N) : DFU3D_EINVAL;
    case DFU3D_STAGE_PROJECT_LABEL:                      /* ag_pt, ib_pix */
      return z->cap_n > 0 ? 2 * up(4 * N) : DFU3D_EINVAL;
    case DFU3D_STAGE_BACKPROJECT_BIN: {                  /* table, pix_bin, blk_cnt */
      int64_t pw = 0, bw = 0;
      if (z->table_entries <= 0 ||
          dfu3d_backproject_scratch_words(z->V, z->H, z->W, z->cap_vox, z->max_points_per_voxel, z->table_entries, &pw, &bw))
        return DFU3D_EINVAL;
      return up(V * z->table_entries * DFU3D_TABLE_ENTRY_BYTES) + up(4 * pw) + up(4 * bw);
    }
    case DFU3D_STAGE_RADIUS_FILTER:                      /* shadow, tile_off, flags, queue (2S joint segments) */
      return P > 0 ? up(DFU3D_SHADOW_BYTES(P)) + up(4 * (2 * S + 1)) + up(P) + up(4 * DFU3D_RF_QUEUE_INTS(P)) : DFU3D_EINVAL;
    case DFU3D_STAGE_STAT_FILTER:                        /* tile_off, flags, mean_d */
      return P > 0 ? up(4 * (S + 1)) + up(P) + up(8 * P) : DFU3D_EINVAL;
    case DFU3D_STAGE_VOXEL_DOWN_SAMPLE:                  /* scratch */
      return P > 0 ? up(dfu3d_voxel_down_sample_scratch_bytes(P)) : DFU3D_EINVAL;
    case DFU3D_STAGE_BALLQUERY_FUSE:                     /* tile_off, flags */
      return P > 0 ? up(4 * (2 * S + 2)) + up(P) : DFU3D_EINVAL;
    case DFU3D_STAGE_RANGE_CLUSTER:                      /* sx, sy, si */
      return P > 0 ? 2 * up(8 * P) + up(12 * P) : DFU3D_EINVAL;
    case DFU3D_STAGE_LSHAPE_FIT:                         /* sx, sy, sroot, fit_ws */
      return (P > 0 && z->cap_rows > 0)
                 ? 2 * up(8 * P) + up(4 * P) + up(8 * dfu3d_lshape_fit_ws_doubles(P, z->cap_rows)) : DFU3D_EINVAL;
    case DFU3D_STAGE_PSEUDO_BOXES: {
      dfu3d_chain_cfg c = {};
      c.V = z->V; c.H = z->H; c.W = z->W; c.max_inst = z->max_inst; c.cap_n = z->cap_n; c.cap_vox = z->cap_vox;
      c.cap_rows = z->cap_rows; c.dense = z->dense; c.stat_filter = z->stat_filter; c.pool_cap = z->pool_cap;
      c.bounds_h = z->H; c.bounds_w = z->W; c.n_theta = 1;
      c.geom.max_points_per_voxel = z->max_points_per_voxel;
      c.geom.t_n = 1; c.geom.p_n = (int32_t)z->table_entries;       /* only the product enters the size */
      if (z->dense && (z->table_entries <= 0 || z->table_entries > 0x7FFFFFFF)) return DFU3D_EINVAL;
      return dfu3d_chain_workspace_bytes(&c);
    }
    default:
      return DFU3D_EINVAL;
  }
}

extern "C" int dfu3d_chain_workspace_init(const dfu3d_chain_cfg *cfg, void *workspace, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!cfg_ok(cfg) || !workspace) return DFU3D_EINVAL;
  ChainWs w;
  carve(cfg, (char *)workspace, &w);
  if (cfg->dense) {
    const int rc = dfu3d_bin_table_init(w.table, (int64_t)cfg->V * w.table_entries, stream);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_fill_i32, dim3((cfg->V * cfg->max_inst + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, cfg->V * cfg->max_inst, 1, w.stat_enable);
  DFU3D_LAUNCH_CHECK();
  return DFU3D_OK;
}

#define CHAIN_TRY(call)      \
  do {                       \
    const int rc_ = (call);  \
    if (rc_) return rc_;     \
  } while (0)

extern "C" int dfu3d_pseudo_boxes(
    const dfu3d_chain_cfg *cfg, const float *points, const int32_t *pt_off,
    const int32_t *view_frame, const float *calib, const void *masks, const int32_t *n_inst,
    const float *depth, const int64_t *view_key, const double *plane_in, const int32_t *inst_class,
    const int32_t *inst_is_car, const double *inst_r_lidar, const double *inst_r_pseudo,
    const float *inst_box, const float *inst_score, void *workspace, double *rows, int32_t *n_rows,
    uint32_t *status, void *stream) {
  DFU3D_CLEAR_STALE_ERROR();
  if (!cfg_ok(cfg) || !points || !pt_off || !view_frame || !calib || !masks || !n_inst || !inst_class ||
      !inst_is_car || !inst_r_lidar || !inst_r_pseudo || !inst_box || !inst_score || !workspace || !rows ||
      !n_rows || !status)
    return DFU3D_EINVAL;
  if (!plane_in && !view_key) return DFU3D_EINVAL;
  if (cfg->dense && !depth) return DFU3D_EINVAL;
  ChainWs w;
  carve(cfg, (char *)workspace, &w);
  hipStream_t st = (hipStream_t)stream;
  const int V = cfg->V, M = cfg->max_inst, S = V * M, cap_n = cfg->cap_n;
  // (one small kernel, not three memsets: common.hpp, k_fill_words)
  if (dfu3d_fill_small_async(n_rows, sizeof(int32_t), status, sizeof(uint32_t), w.pool_cursor, sizeof(int64_t), st) != hipSuccess)
    return DFU3D_ELAUNCH;
  // a4
  if (cfg->apply_fov) {
    CHAIN_TRY(dfu3d_fov_filter(points, pt_off, view_frame, calib, V, cfg->fov_h, cfg->fov_w, cap_n,
                               w.fov_idx, w.n_fov, stream));
  } else {
    hipLaunchKernelGGL(k_all_points, dim3((cap_n + 255) / 256 < 64 ? (cap_n + 255) / 256 : 64, V), dim3(256), 0, st,
                       pt_off, view_frame, cap_n, w.fov_idx, w.n_fov);
    DFU3D_LAUNCH_CHECK();
  }
  // a5
  const double *plane = plane_in;
  if (!plane) {
    CHAIN_TRY(dfu3d_plane_ransac(points, pt_off, view_frame, w.fov_idx, w.n_fov, V, cap_n, cfg->plane_max_hs,
                                 cfg->plane_range, cfg->ransac_trials, cfg->ransac_seed, view_key,
                                 w.cand_idx, w.plane, stream));
    plane = w.plane;
  }
  // a5/a6
  CHAIN_TRY(dfu3d_project_label(points, pt_off, view_frame, calib, plane, w.fov_idx, w.n_fov, masks,
                                cfg->mask_format, n_inst, V, M, cfg->H, cfg->W, cfg->bounds_h, cfg->bounds_w,
                                cap_n, cfg->plane_offset, cfg->plane_range, w.ag_pt,
                                w.ib_pix, w.n_ag, w.K, w.a_bits, w.a_x, w.a_y, w.a_z, stream));
  // a7-a9
  if (cfg->dense) {
    CHAIN_TRY(dfu3d_backproject_bin(depth, calib, masks, cfg->mask_format, n_inst, V, M, cfg->H, cfg->W, &cfg->geom, 1, w.table,
                                    w.pix_bin, w.blk_cnt, cfg->cap_vox, w.n_vox, w.vox_pix, w.b_bits, w.b_x,
                                    w.b_y, w.b_z, status, DFU3D_BP_ALL, stream));
  } else {
    if (dfu3d_fill_async(w.n_vox, 0, sizeof(int32_t) * V, st) != hipSuccess) return DFU3D_ELAUNCH;
  }
  const bool joint = !cfg->stat_filter;   // one radius-filter pass over the LiDAR and pseudo lists together
  CHAIN_TRY(dfu3d_segments_build(w.a_bits, w.a_x, w.a_y, w.a_z, w.K, cap_n, w.b_bits, w.b_x, w.b_y, w.b_z,
                                 w.n_vox, cfg->cap_vox, V, M, cfg->pool_cap, w.pool_cursor, w.px, w.py, w.pz,
                                 w.base_a, w.cnt_a, w.base_b, w.cnt_b, status, inst_r_lidar, inst_r_pseudo,
                                 joint ? w.shadow : nullptr, joint ? w.base_ab : nullptr, w.cnt_ab, w.rad_ab,
                                 w.chunk_cnt, stream));
  // a10 (+a11) + a12
  if (joint) {
    // the shadow and the joint segment table come from the segment build; both lists keep their flags for the
    // joint fuse (one compaction launch for both lists and both filters)
    CHAIN_TRY(dfu3d_radius_filter(w.px, w.py, w.pz, w.base_ab, w.cnt_ab, w.rad_ab, cfg->nb_points, 2 * S,
                                  cfg->pool_cap, w.pool_cursor, w.shadow, w.tile_off, w.flags, w.queue,
                                  DFU3D_RF_FLAGS | DFU3D_RF_RESOLVE, stream));
    CHAIN_TRY(dfu3d_ballquery_fuse_joint(w.px, w.py, w.pz, w.base_a, w.cnt_a, w.base_b, w.cnt_b, cfg->fuse_C,
                                          S, cfg->pool_cap, w.tile_off, w.flags, stream));
  } else {
    CHAIN_TRY(dfu3d_radius_filter(w.px, w.py, w.pz, w.base_a, w.cnt_a, inst_r_lidar, cfg->nb_points, S,
                                  cfg->pool_cap, w.pool_cursor, w.shadow, w.tile_off, w.flags, w.queue,
                                  DFU3D_RF_ALL, stream));
    CHAIN_TRY(dfu3d_radius_filter(w.px, w.py, w.pz, w.base_b, w.cnt_b, inst_r_pseudo, cfg->nb_points, S,
                                  cfg->pool_cap, w.pool_cursor, w.shadow, w.tile_off, w.flags, w.queue,
                                  DFU3D_RF_ALL, stream));
    CHAIN_TRY(dfu3d_voxel_down_sample(w.px, w.py, w.pz, w.base_b, w.cnt_b, w.stat_enable, cfg->stat_voxel, S,
                                      cfg->pool_cap, w.vd_scratch, status, stream));
    CHAIN_TRY(dfu3d_stat_filter(w.px, w.py, w.pz, w.base_b, w.cnt_b, w.stat_enable, cfg->stat_nb_neighbors,
                                cfg->stat_std_ratio, S, cfg->pool_cap, w.tile_off, w.flags, w.mean_d, nullptr,
                                stream));
    CHAIN_TRY(dfu3d_ballquery_fuse(w.px, w.py, w.pz, w.base_a, w.cnt_a, w.base_b, w.cnt_b, cfg->fuse_C, S,
                                   cfg->pool_cap, w.tile_off, w.flags, stream));
  }
  hipLaunchKernelGGL(k_sum_counts, dim3((S + 255) / 256), dim3(256), 0, st, S, w.cnt_a, w.cnt_b, w.cnt_all);
  DFU3D_LAUNCH_CHECK();
  // a13-a15
  CHAIN_TRY(dfu3d_range_cluster(w.px, w.py, w.base_a, w.cnt_all, S, cfg->R0, cfg->Rd, w.label, w.sx, w.sy,
                                w.si3, cfg->pool_cap, stream));
  CHAIN_TRY(dfu3d_lshape_fit(w.px, w.py, w.pz, w.label, w.base_a, w.cnt_all, S, M, calib, inst_class,
                             inst_is_car, inst_box, inst_score, cfg->n_theta, cfg->dtheta, cfg->car_aspect_max,
                             w.sx, w.sy, w.sroot, cfg->cap_rows, rows, n_rows, status, w.fit_ws, cfg->pool_cap,
                             stream));
  return DFU3D_OK;
}
