/*
 * dfu3d.h -- C ABI of libdfu3d_hip.so: the MI355X (gfx950) implementation of
 * DFU3D's pseudo-box generation hot path.
 *
 * The reference has no FFI layer on this path: it is pure Python/NumPy calling
 * third-party natives.  Each entry point below therefore
 * replaces one reference *function body* (cited as file:line under
 * tools/PENet/ of the reference) and is what the reference's Python would bind
 * with ctypes (binding stubs: INTEGRATION.md).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller; the library never
 *     allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - returns DFU3D_OK or a negative DFU3D_E* code for host-detectable argument
 *     errors; device-detected conditions (capacity overflow) set bits in the
 *     caller's `status` word(s) -- see DFU3D_ST_*;
 *   - re-entrant across streams as long as the buffers differ; no global state.
 *
 * Data model.  A launch covers V camera *views*; view v looks at LiDAR frame
 * view_frame[v] whose points are rows pt_off[f] .. pt_off[f+1] of the packed
 * (N,4) float32 tensor `points`.  Per-view calibration is 48 floats
 * (DFU3D_CALIB_FLOATS): M43[12] = (V2C^T @ R0^T) row-major (4,3); P2[12] row-major
 * (3,4); cu,cv,fu,fv,tx,ty; Minv[12] = rows 0..3 x cols 0..2 of
 * inv((R0_ext @ V2C_ext)^T); 6 pad  (calibration_kitti.py:62-102).
 * Instance *segments* are indexed s = v*max_inst + j.  Per-instance point sets
 * live in a structure-of-arrays fp64 pool (px,py,pz) at [seg_base[s],
 * seg_base[s]+seg_cnt[s]).
 */
#ifndef DFU3D_H
#define DFU3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFU3D_VERSION 150          /* 0.1.5: dfu3d_ballquery_fuse_joint, dfu3d_voxel_down_sample (+ dfu3d_chain_cfg.stat_voxel), no DFU3D_RF_SHORT_LISTS; 141: 128-byte scratch of dfu3d_selftest_backproject */
#define DFU3D_CALIB_FLOATS 48
#define DFU3D_ROW_DOUBLES 24       /* see dfu3d_lshape_fit */
#define DFU3D_TABLE_ENTRY_BYTES 28 /* see dfu3d_backproject_bin */
#define DFU3D_MAX_INST 32

#define DFU3D_OK 0
#define DFU3D_EINVAL (-1)          /* bad argument (null pointer, size <= 0 ...) */
#define DFU3D_ELAUNCH (-2)         /* hipLaunch failed (see hipGetLastError)    */
#define DFU3D_ERANGE (-3)          /* size exceeds a compiled-in limit          */

/* bits OR-ed into *status by kernels */
#define DFU3D_ST_POOL_OVERFLOW 1u   /* instance pool too small (seg_alloc)      */
#define DFU3D_ST_VOX_OVERFLOW 2u    /* more voxels than cap_vox in a view       */
#define DFU3D_ST_ROW_OVERFLOW 4u    /* more box rows than cap_rows              */
#define DFU3D_ST_BIN_RANGE 8u       /* a spherical bin fell outside the table   */
#define DFU3D_ST_VOX_PTS_OVERFLOW 16u /* overflow-bin list too small            */
#define DFU3D_ST_VOXEL_RANGE 32u    /* dfu3d_voxel_down_sample: a segment wider than 2^21 voxels along an axis */

int dfu3d_version(void);
const char *dfu3d_strerror(int code);

/* ---- scratch sizes, uniform over the stages -------------------------------
 * The library never allocates: every entry point takes its scratch from the caller.
 * dfu3d_workspace_bytes(stage, sizes) = the bytes of scratch that stage's entry point needs, i.e. the
 * sum of its scratch arguments (listed with each entry point below), each rounded up to 256 bytes --
 * one allocation carved in argument order serves the call.  Inputs and outputs are not included.
 * DFU3D_STAGE_PSEUDO_BOXES = the single workspace of dfu3d_pseudo_boxes (== dfu3d_chain_workspace_bytes).
 * Returns a negative DFU3D_E* code for an unknown stage or impossible sizes. */
typedef struct dfu3d_sizes {
  int32_t V, H, W, max_inst;         /* views per call, mask / depth canvas, instance slots per view */
  int32_t cap_n, cap_vox, cap_rows;  /* points per frame, voxels per view, box rows per call          */
  int32_t max_points_per_voxel;      /* 100 (my_loader.py:73)                                        */
  int64_t pool_cap;                  /* instance pool slots per call                                 */
  int64_t table_entries;             /* per view, from dfu3d_bin_table_geometry                      */
  int32_t dense, stat_filter;
} dfu3d_sizes;
#define DFU3D_STAGE_FOV_FILTER 0
#define DFU3D_STAGE_PLANE_RANSAC 1
#define DFU3D_STAGE_PROJECT_LABEL 2
#define DFU3D_STAGE_BACKPROJECT_BIN 3
#define DFU3D_STAGE_SEGMENTS_BUILD 4
#define DFU3D_STAGE_RADIUS_FILTER 5
#define DFU3D_STAGE_STAT_FILTER 6
#define DFU3D_STAGE_BALLQUERY_FUSE 7
#define DFU3D_STAGE_RANGE_CLUSTER 8
#define DFU3D_STAGE_LSHAPE_FIT 9
#define DFU3D_STAGE_PSEUDO_BOXES 10
#define DFU3D_STAGE_VOXEL_DOWN_SAMPLE 11
int64_t dfu3d_workspace_bytes(int32_t stage, const dfu3d_sizes *sizes);

/* Geometry of the spherical-bin table used by dfu3d_backproject_bin.  Filled by
 * dfu3d_bin_table_geometry from the voxel parameters (my_loader.py:69-83). */
typedef struct dfu3d_bin_geom {
  double vsize_r, vsize_t, vsize_p;   /* voxel size (r, theta, phi)           */
  double rmin_r, rmin_t, rmin_p;      /* range minimum                        */
  int32_t grid_r, grid_t, grid_p;     /* grid size (1, 5000, 5000)            */
  int32_t t_lo, t_n;                  /* table rows cover theta bins [t_lo, t_lo+t_n) */
  int32_t p_lo, p_n;                  /* table cols cover phi bins            */
  int32_t max_points_per_voxel;       /* 100                                   */
  int32_t max_voxels;                 /* 1000000                               */
  double theta_min;                   /* 1.5  (my_loader.py:175)               */
  double z_max;                       /* 1.0  (my_loader.py:540)               */
  double depth_min;                   /* 0.001 (my_loader.py:507)              */
} dfu3d_bin_geom;

/* Host helper (no GPU work): fills t_lo/t_n/p_lo/p_n so that every bin
 * reachable with theta in (theta_min, pi], phi in [-pi/2, pi/2] is inside the
 * table; returns the number of table entries per view (t_n * p_n). */
int64_t dfu3d_bin_table_geometry(dfu3d_bin_geom *g);

/* ---- a4: get_fov_flag (vis_utils.py:108-123, 152-154) ----------------------
 * fov_idx[v*cap_n + k] = frame-local index of the k-th point of view v's frame
 * with 0<=u<fov_w, 0<=v<fov_h, depth>=0 (unrounded float32 u,v); n_fov[v] = count. */
int dfu3d_fov_filter(const float *points, const int32_t *pt_off,
                     const int32_t *view_frame, const float *calib, int32_t V,
                     int32_t fov_h, int32_t fov_w, int32_t cap_n,
                     int32_t *fov_idx, int32_t *n_fov, void *stream);

/* ---- a5: estimate_plane (my_loader.py:448-469), seeded RANSAC (hazard H1) ---
 * plane[v][4] (fp64, unit normal up, offset).  cand_idx: int32 scratch
 * (V*cap_n).  key[v] seeds the per-view sample stream. */
int dfu3d_plane_ransac(const float *points, const int32_t *pt_off,
                       const int32_t *view_frame, const int32_t *fov_idx,
                       const int32_t *n_fov, int32_t V, int32_t cap_n,
                       double max_hs, double xy_range, int32_t trials,
                       uint64_t seed, const int64_t *key, int32_t *cand_idx,
                       double *plane, void *stream);

/* Instance masks come in one of two layouts (`mask_format`):
 *   DFU3D_MASK_BYTES (0): uint8 planes (V, max_inst, H, W) -- np.uint8(mask_image) of
 *     my_loader.py:523-525, one byte gather per instance and looked-up pixel;
 *   1 / 2 / 4: ONE word of that many bytes per pixel, (V, H, W), bit j = instance j
 *     (max_inst <= 8 / 16 / 32) -- one gather per looked-up pixel and 1/8 of the bytes in
 *     HBM and over PCIe; dfu3d_pack_masks converts, a mask provider can deliver it directly. */
#define DFU3D_MASK_BYTES 0
int dfu3d_pack_masks(const uint8_t *masks, const int32_t *n_inst, int32_t V, int32_t max_inst,
                     int32_t H, int32_t W, void *out, int32_t word_bytes, void *stream);

/* ---- a5/a6: above_plane + point->pixel label inheritance
 * (my_loader.py:471-477, 517-530; hazard H3) ---------------------------------
 * For view v: rows t < K[v] are the first K above-plane FOV points; row t gets
 * the instance bits of the t-th IN-BOUNDS rounded pixel.  Outputs per row:
 * it_bits (bit j = uint8 mask_j > 0), it_x/y/z (fp64 coordinates of the row's
 * point), all at [v*cap_n + t]; n_ag[v], K[v].
 * masks: (H, W) images per view in `mask_format`; n_inst[v] <= max_inst <= 32.
 * bounds_h / bounds_w: the in-bounds test of my_loader.py:526 (hard-coded 900 / 1600 there;
 * <= H, W -- hazard H11: the mask canvas may be larger than the bounds).
 * ag_pt / ib_pix: int32 scratch (V*cap_n each). */
int dfu3d_project_label(const float *points, const int32_t *pt_off,
                        const int32_t *view_frame, const float *calib,
                        const double *plane, const int32_t *fov_idx,
                        const int32_t *n_fov, const void *masks, int32_t mask_format,
                        const int32_t *n_inst, int32_t V, int32_t max_inst,
                        int32_t H, int32_t W, int32_t bounds_h, int32_t bounds_w,
                        int32_t cap_n, double plane_offset, double xy_range,
                        int32_t *ag_pt, int32_t *ib_pix, int32_t *n_ag,
                        int32_t *K, uint32_t *it_bits, double *it_x,
                        double *it_y, double *it_z, void *stream);

/* ---- a7/a8/a9: back-projection + spherical voxel sampling
 * (my_loader.py:507-509, 532-557, 166-180, 247-275) --------------------------
 * depth: float32 (V,H,W).  For every view: back-project all pixels with
 * depth >= depth_min (fp64), keep z_lidar < z_max and theta > theta_min, bin
 * (r,theta,phi), keep at most max_points_per_voxel points per bin in pixel
 * order, choose the first argmin of KEY among them (key_axis 1 = y:
 * la_sampling20, 2 = z: la_sampling2), emit voxels in first-seen order.
 * Outputs per voxel k < n_vox[v] at [v*cap_vox + k]: vox_pix (pixel index of
 * the representative), it_x/y/z (its fp64 LiDAR coordinates), it_bits
 * (instance bits at that pixel; masks may be NULL -> 0).
 * `geom` is a HOST pointer (read at call time).
 * Scratch: pix_bin (uint32 words) and blk_cnt (int32 words, 8-byte aligned) sized by
 * dfu3d_backproject_scratch_words; table: V*table_entries entries of
 * DFU3D_TABLE_ENTRY_BYTES (uint64 min-key and min-(key|pixel) planes, then
 * uint32 count / first-pixel / representative planes), initialised once with dfu3d_bin_table_init and left clean by every
 * call that returns without DFU3D_ST_VOX_OVERFLOW.  The depth map is the only
 * per-pixel stream of the stage: nothing is written per pixel. */
int dfu3d_bin_table_init(void *table, int64_t table_entries_total, void *stream);
int64_t dfu3d_backproject_scratch_words(int32_t V, int32_t H, int32_t W,
                                        int32_t cap_vox, int32_t max_points, int64_t table_entries,
                                        int64_t *pix_words, int64_t *blk_words);
int dfu3d_backproject_bin(const float *depth, const float *calib,
                          const void *masks, int32_t mask_format, const int32_t *n_inst,
                          int32_t V, int32_t max_inst, int32_t H, int32_t W,
                          const dfu3d_bin_geom *geom, int32_t key_axis,
                          void *table, uint32_t *pix_bin, int32_t *blk_cnt,
                          int32_t cap_vox, int32_t *n_vox, uint32_t *vox_pix,
                          uint32_t *it_bits, double *it_x, double *it_y,
                          double *it_z, uint32_t *status, int32_t phases,
                          void *stream);
/* `phases` selects which kernels of the stage a call enqueues (DFU3D_BP_ALL in
 * production; single phases let a caller bracket one kernel group with HIP events
 * on its stream).  The phases of one pass must be issued in this order, each once:
 * BIN starts a pass (it zeroes the per-view counters and the first-pixel bit map and
 * fills the table); AMB / MARK / VOX consume what BIN left and VOX resets every table
 * entry it finishes, so a phase repeated without BIN in front of it finds nothing to do
 * -- it must not be relied on to reproduce outputs. */
#define DFU3D_BP_BIN 1     /* k_bp_bin: back-project, bin, table atomics, touched-bin list          */
#define DFU3D_BP_AMB 2     /* k_bp_bin_amb: the pixels float32 could not classify, in fp64           */
#define DFU3D_BP_MARK 4    /* k_bp_scan: popcount prefix of the first-pixel bit map, list of occupied table segments */
#define DFU3D_BP_VOX 8     /* k_bp_vox: walk over the occupied table segments: rank, representative, outputs, reset */
#define DFU3D_BP_REPAIR 16 /* exact repair of bins over the cap / key collisions (no-ops when none)  */
#define DFU3D_BP_ALL 31

/* ---- per-instance point sets (my_loader.py:547-565) ------------------------
 * Builds, for every segment s = v*max_inst + j, the ordered list of LiDAR rows
 * (items A) and voxel representatives (items B) whose bit j is set, in the
 * fp64 pool: [base_a[s], +cnt_a[s]) followed directly by [base_b[s], +cnt_b[s]).
 * pool_cursor: device int64 (in/out) next free pool slot.
 * Optional (NULL to skip), for the one-pass radius filter below:
 *   rad_a / rad_b fp64 (S): the filter radii of the LiDAR / pseudo lists;
 *   shadow: DFU3D_SHADOW_BYTES(pool_cap), the float32 shadow the filter streams (x, y, z, list | radius);
 *   base_ab / cnt_ab / rad_ab (2S each): the joint segment table, s < S = LiDAR lists, S+s = pseudo lists.
 * Scratch: chunk_cnt int32 (dfu3d_segments_scratch_words): member counts per (view, 256-item range, instance),
 * turned into their running sums along the view by the call. */
int64_t dfu3d_segments_scratch_words(int32_t V, int32_t a_cap, int32_t b_cap);
int dfu3d_segments_build(const uint32_t *a_bits, const double *a_x,
                         const double *a_y, const double *a_z,
                         const int32_t *a_n, int32_t a_cap,
                         const uint32_t *b_bits, const double *b_x,
                         const double *b_y, const double *b_z,
                         const int32_t *b_n, int32_t b_cap, int32_t V,
                         int32_t max_inst, int64_t pool_cap,
                         int64_t *pool_cursor, double *px, double *py,
                         double *pz, int64_t *base_a, int32_t *cnt_a,
                         int64_t *base_b, int32_t *cnt_b, uint32_t *status,
                         const double *rad_a, const double *rad_b, void *shadow,
                         int64_t *base_ab, int32_t *cnt_ab, double *rad_ab,
                         int32_t *chunk_cnt, void *stream);

/* ---- a10: Open3D remove_radius_outlier (my_loader.py:581-599) --------------
 * In-place, order-preserving: keeps point i of segment s iff
 * #{j in s : |p_i-p_j|^2 < radius[s]^2, j == i included} > nb_points.
 * radius[s] < 0 drops the whole segment (hazard H4), radius[s] == 0 keeps it.
 * The filter streams a float32 SHADOW of the pool (16 B per slot: x, y, z, segment | radius) and
 * takes only decisions that float32 rounding cannot change; everything else is decided in fp64
 * from the pool.  Segments must not overlap; S < 65535.
 * n_used: device int64 = number of pool slots in use (NULL: pool_cap).
 * Scratch: shadow (DFU3D_SHADOW_BYTES(pool_cap) bytes, 16-byte aligned: the 16 B per slot the filter streams,
 * then what the later phases read -- two bounding boxes and 128 slots for points without a list neighbour
 * per 512 slots, per-segment overflow lists (16 B per slot at most), their lengths and the work items of
 * queries near many boxes), tile_off int32 (S+1), flags uint8 (pool_cap), queue int32 (DFU3D_RF_QUEUE_INTS(pool_cap), 16-byte
 * aligned: undecided pool positions between the kernels). */
int dfu3d_radius_filter(double *px, double *py, double *pz,
                        const int64_t *seg_base, int32_t *seg_cnt,
                        const double *radius, int32_t nb_points, int32_t S,
                        int64_t pool_cap, const int64_t *n_used, void *shadow,
                        int32_t *tile_off, uint8_t *flags,
                        int32_t *queue, int32_t phases, void *stream);
#define DFU3D_SHADOW_BYTES(pool_cap) (32 * (int64_t)(pool_cap) + (64 + 2048 + 512) * (((int64_t)(pool_cap) + 511) / 512 + 1) + 9699456)
/* int32 elements of `queue`: 64 parts (one per 64th of the 2048-slot workgroups) behind their 64 counters */
#define DFU3D_RF_QUEUE_INTS(pool_cap) (1024 + 64 * ((((int64_t)(pool_cap) + 2047) / 2048 + 63) / 64) * 2048)
/* the two macros as functions, for hosts that cannot evaluate C macros (ctypes, cgo, JNI) */
int64_t dfu3d_rf_shadow_bytes(int64_t pool_cap);
int64_t dfu3d_rf_queue_ints(int64_t pool_cap);
#define DFU3D_RF_SHADOW 1   /* shadow of the given segments (not needed behind dfu3d_segments_build(..., shadow)) */
#define DFU3D_RF_FLAGS 2    /* k_rf_stream: list neighbours, then the point's whole 512-slot range, float32 */
#define DFU3D_RF_RESOLVE 4  /* k_rf_resolve: the undecided against their whole segment                       */
#define DFU3D_RF_COMPACT 8  /* ordered in-place compaction of the given segments                            */
#define DFU3D_RF_ALL 15

/* ---- a11: Open3D voxel_down_sample(voxel_size) (my_loader0.py:734; dormant) ----
 * The first half of the reference's (commented) pair `voxel_down_sample(0.05)` ->
 * `remove_statistical_outlier(30, 0.3)`.  Every enabled segment is replaced IN PLACE by the
 * centroids of its occupied voxels: voxel_min_bound = min over the segment - voxel_size / 2,
 * voxel index = floor((p - voxel_min_bound) / voxel_size), centroid = sum of the voxel's
 * points in list order / their number (Open3D's AccumulatedPoint).  Open3D emits them in
 * the iteration order of an unordered_map (unspecified); this library DEFINES the order:
 * first-seen -- a voxel's place is that of its first point in the list.  seg_cnt[s] becomes
 * the number of voxels.  A segment wider than 2^21 voxels along an axis raises
 * DFU3D_ST_VOXEL_RANGE.  Scratch: dfu3d_voxel_down_sample_scratch_bytes(pool_cap) bytes,
 * 8-byte aligned (hash table 4 x pool_cap slots, running sums, first-point positions). */
int64_t dfu3d_voxel_down_sample_scratch_bytes(int64_t pool_cap);
int dfu3d_voxel_down_sample(double *px, double *py, double *pz,
                            const int64_t *seg_base, int32_t *seg_cnt,
                            const int32_t *enable, double voxel_size, int32_t S,
                            int64_t pool_cap, void *scratch, uint32_t *status,
                            void *stream);

/* ---- a11: Open3D remove_statistical_outlier (my_loader0.py:735; dormant) ---
 * keep i iff 0 < mean_knn_dist_i < mu + std_ratio * sigma (self included in
 * the k nearest; Bessel sigma).  Scratch: mean_d fp64 (pool_cap), stats fp64
 * (S*4), tile_off, flags as above.  Segments with enable[s]==0 are untouched. */
int dfu3d_stat_filter(double *px, double *py, double *pz,
                      const int64_t *seg_base, int32_t *seg_cnt,
                      const int32_t *enable, int32_t nb_neighbors,
                      double std_ratio, int32_t S, int64_t pool_cap,
                      int32_t *tile_off, uint8_t *flags, double *mean_d,
                      double *stats, void *stream);

/* ---- a12: BallQuery fuse (my_loader.py:489-494, 601-605) -------------------
 * Keeps query point i of segment B iff min_j |q_i - a_j| < C over segment A
 * (strict; skipped -- all kept -- when either set is empty), then moves the
 * survivors to directly follow segment A: on return the instance's points are
 * [base_a[s], base_a[s] + cnt_a[s] + cnt_b[s]) and base_b[s] is updated.
 * Scratch: tile_off int32 (2S+2: the query-tile lists of the two builds of the kernel), flags uint8 (pool_cap). */
int dfu3d_ballquery_fuse(double *px, double *py, double *pz,
                         const int64_t *base_a, const int32_t *cnt_a,
                         int64_t *base_b, int32_t *cnt_b, double C, int32_t S,
                         int64_t pool_cap, int32_t *tile_off, uint8_t *flags,
                         void *stream);
/* Same, but flags[base_b[s] + i] holds on entry the keep mask of a filter that
 * was run without its compaction phase (dfu3d_radius_filter with
 * phases = DFU3D_RF_ALL & ~DFU3D_RF_COMPACT): masked-out points are neither
 * queries nor survivors, so my_loader.py:587-605 (filter, then fuse) costs one
 * compaction of the pseudo points instead of two. */
int dfu3d_ballquery_fuse_masked(double *px, double *py, double *pz,
                                const int64_t *base_a, const int32_t *cnt_a,
                                int64_t *base_b, int32_t *cnt_b, double C,
                                int32_t S, int64_t pool_cap, int32_t *tile_off,
                                uint8_t *flags, void *stream);
/* Same, behind a JOINT filter pass: flags holds the keep mask of segment A as well
 * (dfu3d_radius_filter over the 2S lists A | B of dfu3d_segments_build's joint table with
 * phases = DFU3D_RF_FLAGS | DFU3D_RF_RESOLVE).  Points of A the filter dropped are not
 * candidates; if it left none the fuse is skipped as for an empty A (my_loader.py:602).
 * A is compacted in place, cnt_a[s] updated, and the survivors of B follow it: one
 * compaction launch for both lists and both filters. */
int dfu3d_ballquery_fuse_joint(double *px, double *py, double *pz,
                               const int64_t *base_a, int32_t *cnt_a,
                               int64_t *base_b, int32_t *cnt_b, double C,
                               int32_t S, int64_t pool_cap, int32_t *tile_off,
                               uint8_t *flags, void *stream);

/* ---- f-2: points in boxes / ground-truth database --------------------------
 * (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-171 points_in_boxes_cpu;
 *  pcdet/datasets/kitti/kitti_dataset.py:284-331 create_groundtruth_database)
 * boxes: float64 (B,7) [x y z dx dy dz heading] in the LiDAR frame (converted to
 * float32 for the test exactly as the reference's `.float()` does).
 * dfu3d_points_in_boxes_mask: one frame, pts float32 with row stride pt_stride
 * (3 or 4 floats), out int32 (B,n) = the reference's return value.
 * dfu3d_gt_database: many frames at once.  points (N,4) float32 packed by frame
 * (pt_off), box b belongs to frame box_frame[b].  Outputs: box_cnt[b] points
 * inside box b; box_off (Bt+1) exclusive scan; for box b, at
 * [box_off[b], box_off[b+1]): idx_out = in-frame point indices in ascending
 * order, gt_pts (.,4) float32 = (float32)(double(xyz) - box centre), intensity
 * -- the bytes kitti_dataset.py:317-321 writes to gt_database/<frame>_<class>_<i>.bin.
 * More than cap_out points in total raise DFU3D_ST_POOL_OVERFLOW (counts stay
 * exact, lists are cut). */
int dfu3d_points_in_boxes_mask(const float *pts, int32_t n, int32_t pt_stride,
                               const double *boxes, int32_t B, int32_t *out,
                               void *stream);
int dfu3d_gt_database(const float *points, const int32_t *pt_off,
                      const int32_t *box_frame, const double *boxes, int32_t Bt,
                      int32_t *box_cnt, int64_t *box_off, int64_t cap_out,
                      int32_t *idx_out, float *gt_pts, uint32_t *status,
                      void *stream);

/* ---- f-3: overlap of rotated boxes in the ground plane, IoU criteria, rotated NMS -----------
 * Stands in for pcdet/ops/iou3d_nms (boxes_overlap_bev_gpu / boxes_iou_bev_gpu / nms_gpu / nms_normal_gpu,
 * iou3d_nms.cpp:120-177) and for the AP evaluator's rotate_iou_gpu_eval
 * (pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py:262-330, numba-CUDA).  The overlap is the EXACT area of
 * the intersection polygon (clipping in the frame of one box, float32); the reference's kernels over-estimate it by a
 * thin sliver when a corner lies within their 1e-2 margin outside the other box.
 * boxes: float32 (.,7) [x y z dx dy dz heading].
 * dfu3d_boxes_bev: out float32 (n,m); mode 0 = overlap area, mode 1 = BEV IoU.  _paired: row i with row i, out (n).
 * dfu3d_rotate_iou_eval: boxes (n,5) / query_boxes (k,5) [cx cy w h angle], out (n,k); criterion -1 = IoU,
 *   0 = overlap / area(box), 1 = overlap / area(query box), other = overlap   (rotate_iou.py:247-255).
 * dfu3d_nms_bev / dfu3d_nms_normal_bev (axis-aligned IoU, headings ignored): boxes already sorted by descending
 * score; keep[0..num_keep) = kept positions in ascending order, box i suppressed iff an earlier kept box j has
 * IoU(j,i) > thresh.  Scratch: mask uint64 (n * ceil(n/64)); one wave computes one mask word (ballot), the walk over
 * the mask runs on the device (the reference copies the mask to the host).  n <= 32768. */
int dfu3d_boxes_bev(const float *boxes_a, int32_t n, const float *boxes_b,
                    int32_t m, float *out, int32_t mode, void *stream);
int dfu3d_boxes_bev_paired(const float *boxes_a, const float *boxes_b, int32_t n, float *out,
                           int32_t mode, void *stream);
int dfu3d_rotate_iou_eval(const float *boxes, int32_t n, const float *query_boxes, int32_t k,
                          float *out, int32_t criterion, void *stream);
int dfu3d_nms_bev(const float *boxes, int32_t n, float thresh, uint64_t *mask,
                  int64_t *keep, int32_t *num_keep, void *stream);
int dfu3d_nms_normal_bev(const float *boxes, int32_t n, float thresh, uint64_t *mask,
                         int64_t *keep, int32_t *num_keep, void *stream);

/* ---- f-3, AP evaluation: the KITTI evaluator of pcdet/datasets/kitti/kitti_object_eval_python/eval.py -----------
 * (reached from kitti_dataset.py:421-431 `evaluation`).  All frames of a split are ONE batch: ground truths and
 * detections are flat arrays with per-frame offsets gt_off / dt_off (F+1, int64).  Per box: bbox float64 (.,4)
 * [x1 y1 x2 y2], cam float64 (.,7) [x y z l h w ry] (location, dimensions, rotation_y), alpha, score float64;
 * code int32 = index of the lower-cased name in the evaluator's class list (eval.py:32), 1000 + k for a name that is
 * neutral for class k ('van' for car, 'person_sitting' for pedestrian, :47-51), -1 otherwise; gt_dontcare int32 = 1
 * for 'DontCare' rows (:72-73); gt_occluded int32, gt_truncated float64.
 * dfu3d_eval_overlaps: calculate_iou_partly (:336-417) restricted to pairs of the same frame.  ov float64
 *   [ov_off[f] + i * D_f + j] for ground truth i, detection j of frame f (ov_off = prefix sums of G_f * D_f, F+1);
 *   metric 0: 2-D boxes (image_box_overlap :95-124), 1: BEV IoU of [x z l w ry] (float32 exact polygon overlap, as
 *   dfu3d_rotate_iou_eval), 2: volume IoU (d3_box_overlap_kernel :126-151, result rounded to float32 as there).
 * dfu3d_eval_match_scores: compute_statistics_jit with compute_fp False, thresh 0 (:497-511) for every frame and every
 *   cell (class, difficulty, min_overlap) of `combos`: matched float64 (n_combo, n_gt) receives, per frame, the
 *   scores of the matched detections at the head of the frame's ground-truth slots and NaN behind them; n_valid int32
 *   (n_combo, F) the number of counted ground truths (clean_data's num_valid_gt).
 * dfu3d_eval_match_stats: fused_compute_statistics (:304-333): thresholds float64 (n_combo, t_stride) with n_thresh
 *   (n_combo) valid entries each; pr int64 (n_combo, t_stride, 3) = tp, fp, fn summed over the frames (zeroed by the
 *   call); sim float64 (n_combo, F, t_stride) = per-frame orientation similarity (0 where the reference returns -1),
 *   required iff compute_aos.  One lane runs the reference's assignment for one threshold; a wave = the thresholds of
 *   one frame and cell.  max_dt = largest detection count of a frame, <= 2048 (DFU3D_ERANGE beyond). */
typedef struct dfu3d_eval_combo {
  int32_t cls;          /* class index k (see `code`) */
  int32_t difficulty;   /* 0 easy, 1 moderate, 2 hard; >= 3: no level rules (get_range_eval_result) */
  double min_overlap;
} dfu3d_eval_combo;
#define DFU3D_EVAL_MAX_DET 2048
int dfu3d_eval_overlaps(int32_t metric, int32_t F, const int64_t *gt_off, const int64_t *dt_off,
                        const int64_t *ov_off, const double *gt_bbox, const double *dt_bbox,
                        const double *gt_cam, const double *dt_cam, double *ov, int64_t n_pairs,
                        void *stream);
int dfu3d_eval_match_scores(int32_t metric, int32_t F, int32_t max_dt, const int64_t *gt_off,
                            const int64_t *dt_off, const int64_t *ov_off, const double *ov,
                            const int32_t *gt_code, const int32_t *gt_dontcare, const double *gt_bbox,
                            const double *gt_alpha, const int32_t *gt_occluded, const double *gt_truncated,
                            const int32_t *dt_code, const double *dt_bbox, const double *dt_alpha,
                            const double *dt_score, const dfu3d_eval_combo *combos, int32_t n_combo,
                            int64_t n_gt, double *matched, int32_t *n_valid, void *stream);
int dfu3d_eval_match_stats(int32_t metric, int32_t F, int32_t max_dt, const int64_t *gt_off,
                           const int64_t *dt_off, const int64_t *ov_off, const double *ov,
                           const int32_t *gt_code, const int32_t *gt_dontcare, const double *gt_bbox,
                           const double *gt_alpha, const int32_t *gt_occluded, const double *gt_truncated,
                           const int32_t *dt_code, const double *dt_bbox, const double *dt_alpha,
                           const double *dt_score, const dfu3d_eval_combo *combos, int32_t n_combo,
                           const double *thresholds, const int32_t *n_thresh, int32_t t_stride,
                           int32_t compute_aos, int64_t *pr, double *sim, void *stream);

/* ---- f-4: la_sampling of the ground-truth sampling augmentor (pcdet/datasets/augmentor/database_sampler_virtual.py:319-351)
 * A batch of B objects: float32 rows `points` (n_points, n_cols >= 3; x y z first), object b owns rows
 * [obj_off[b], obj_off[b+1]).  Every object is binned by (theta // vert_res, fan // hor_res) of its points' spherical
 * coordinates (float32, NumPy's operation order and its float floor division), every bin keeps the point with the
 * smallest theta (first on ties), bins in first-seen order; an object that would keep fewer than 5 rows is copied
 * unchanged.  out: same shape as points (not the same buffer); object b's rows start at obj_off[b], out_cnt[b] of them.
 * scratch: 24 bytes per point, 8-byte aligned. */
int dfu3d_la_sampling(const float *points, int32_t n_cols, const int64_t *obj_off, int32_t B,
                      float vert_res, float hor_res, float *out, int32_t *out_cnt, void *scratch,
                      int64_t n_points, void *stream);

/* ---- self test of the two-tier bin classification ---------------------------
 * dfu3d_backproject_bin decides a pixel's spherical bin in float32 when every float32 estimate is farther from
 * every boundary involved than a bound on its error, and in fp64 otherwise.  This entry point runs both
 * classifications over n pseudo-random pixels of an H x W image with depths in [d_lo, d_hi) (every fourth one
 * 50x closer) under the ONE calibration record `calib` and the bin geometry `geom` (host pointer, after
 * dfu3d_bin_table_geometry): out4 (device) = { pixels tried, undecided in float32, DISAGREEMENTS among the decided
 * (bin or voxel key), kept by float32 }.  The pixels float32 leaves undecided also go through the middle tier the
 * voxel pass uses for them (fp64 coordinates against fp64 bin edges in cos / tan space, no acos / atan); a decision
 * of that tier that differs from the full fp64 classification counts as a disagreement too.  out4[2] must be 0.  scratch: DFU3D_SELFTEST_SCRATCH_BYTES, 16-byte
 * aligned. */
#define DFU3D_SELFTEST_SCRATCH_BYTES (128 + 16 * (65536 + 16384))
int dfu3d_selftest_classify(const float *calib, int32_t H, int32_t W, const dfu3d_bin_geom *geom,
                            int32_t key_axis, int64_t n, uint64_t seed, double d_lo, double d_hi,
                            void *scratch, uint64_t *out4, void *stream);
/* The same classification starts from a float32 back-projection of the pixel (nine FMAs) with a bound on its
 * error.  This entry point measures, over n pseudo-random pixels of an H x W image with depths in [d_lo, d_hi)
 * (every fourth one 50x closer) under the ONE calibration record `calib`, out1[0] = max over pixels and
 * coordinates of |float32 estimate - fp64 back-projection| / bound.  scratch64: 128 bytes of device scratch
 * (64 up to version 140), 16-byte aligned. */
int dfu3d_selftest_backproject(const float *calib, int32_t H, int32_t W, int64_t n, uint64_t seed,
                               double d_lo, double d_hi, void *scratch64, double *out1, void *stream);

/* ---- a13: _adoptive_range_segmentation (rectangle_fitting.py:161-191) ------
 * label[seg_base[s] + i] = smallest in-segment index of the cluster that
 * contains point i (clusters = connected components of d_ij <= R_i or
 * d_ij <= R_j, R = R0 + Rd*|p|_xy).  No wall-clock abort (hazard H2).
 * Scratch: sx, sy fp64 (pool_cap each), si int32 (3*pool_cap): the points of
 * each instance are counting-sorted by spatial cell before they are merged. */
int dfu3d_range_cluster(const double *px, const double *py,
                        const int64_t *seg_base, const int32_t *seg_cnt,
                        int32_t S, double R0, double Rd, int32_t *label,
                        double *sx, double *sy, int32_t *si, int64_t pool_cap,
                        void *stream);

/* ---- a14/a15: _rectangle_search + GenerateAnns
 * (rectangle_fitting.py:83-159; my_loader.py:633-702) ------------------------
 * One row per (segment, cluster) in rows[DFU3D_ROW_DOUBLES]:
 *   0 view, 1 inst j, 2 cluster k, 3 class index, 4 alpha, 5..8 bbox x1 y1 x2 y2,
 *   9 h, 10 w, 11 l, 12 x, 13 y, 14 z (rect camera), 15 ry, 16 score,
 *   17 number of cluster points, 18 best heading theta*, 19..22 rectangle
 *   offsets c (min c1, min c2, max c1, max c2: RectangleData.c,
 *   rectangle_fitting.py:145-157), 23 smallest point index of the cluster.
 * Rows are appended in arbitrary order (sort by cols 0..2); n_rows is a device
 * counter.  inst_class/inst_is_car: int32 (S); inst_box: float32 (S,4);
 * inst_score float32 (S).  Scratch: sx, sy fp64 (pool_cap), sroot int32
 * (pool_cap), fit_ws fp64 (dfu3d_lshape_fit_ws_doubles(pool_cap, cap_rows): one
 * descriptor per cluster -- at most 2*cap_rows+64 clusters per call, more raise
 * DFU3D_ST_ROW_OVERFLOW -- and the heading costs of the clusters that are too
 * large for one workgroup and are spread over the chip as (cluster, 8-heading
 * batch) workgroups). */
int64_t dfu3d_lshape_fit_ws_doubles(int64_t pool_cap, int32_t cap_rows);
int dfu3d_lshape_fit(const double *px, const double *py, const double *pz,
                     const int32_t *label, const int64_t *seg_base,
                     const int32_t *seg_cnt, int32_t S, int32_t max_inst,
                     const float *calib, const int32_t *inst_class,
                     const int32_t *inst_is_car, const float *inst_box,
                     const float *inst_score, int32_t n_theta, double dtheta,
                     double car_aspect_max, double *sx, double *sy,
                     int32_t *sroot, int32_t cap_rows, double *rows,
                     int32_t *n_rows, uint32_t *status, double *fit_ws,
                     int64_t pool_cap, void *stream);

/* ---- the whole path behind one call ------------------------------------------
 * vis_utils.py:136-166 -> my_loader.py:502-702 for V views: FOV filter, plane,
 * label inheritance, back-projection + voxel sampling, per-instance lists,
 * radius filter (+ statistical filter), BallQuery fuse, clustering, L-shape fit,
 * box rows.  It sequences the stage entry points above on `stream`, with their
 * scratch carved from ONE caller-owned workspace (dfu3d_chain_workspace_bytes;
 * dfu3d_chain_workspace_init once, it initialises the spherical-bin table).
 * Segment arrays (inst_*) are (V*max_inst); inst_r_lidar / inst_r_pseudo are the
 * radius-filter radii per instance (0 = no filter, < 0 = drop all: hazard H4).
 * plane_in: fp64 (V,4) planes to use, or NULL = seeded RANSAC keyed by view_key.
 * depth: float32 (V,H,W), required when cfg->dense.  rows / n_rows / status as
 * in dfu3d_lshape_fit; n_rows and status are reset by the call. */
typedef struct dfu3d_chain_cfg {
  int32_t V, H, W, max_inst, cap_n, cap_vox, cap_rows;
  int32_t dense, apply_fov, fov_h, fov_w, stat_filter;
  int32_t bounds_h, bounds_w;       /* my_loader.py:526 (<= H, W)           */
  int32_t mask_format, reserved0;   /* DFU3D_MASK_BYTES or 1 / 2 / 4        */
  double stat_voxel;                /* voxel size of the down-sample in front of the statistical filter (my_loader0.py:734: 0.05) */
  int64_t pool_cap;
  double plane_max_hs, plane_range, plane_offset;
  int32_t ransac_trials, nb_points;
  uint64_t ransac_seed;
  double fuse_C, R0, Rd;
  int32_t n_theta, stat_nb_neighbors;
  double dtheta, car_aspect_max, stat_std_ratio;
  dfu3d_bin_geom geom;              /* after dfu3d_bin_table_geometry */
} dfu3d_chain_cfg;
int64_t dfu3d_chain_workspace_bytes(const dfu3d_chain_cfg *cfg);
int dfu3d_chain_workspace_init(const dfu3d_chain_cfg *cfg, void *workspace, void *stream);
int dfu3d_pseudo_boxes(const dfu3d_chain_cfg *cfg, const float *points,
                       const int32_t *pt_off, const int32_t *view_frame,
                       const float *calib, const void *masks,
                       const int32_t *n_inst, const float *depth,
                       const int64_t *view_key, const double *plane_in,
                       const int32_t *inst_class, const int32_t *inst_is_car,
                       const double *inst_r_lidar, const double *inst_r_pseudo,
                       const float *inst_box, const float *inst_score,
                       void *workspace, double *rows, int32_t *n_rows,
                       uint32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFU3D_H */
