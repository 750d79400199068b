"""Batched pseudo-box generation on one MI355X.

Drives the C-ABI stages (include/dfu3d.h) over a batch of camera views that is
resident in HBM.  Mirrors the stage order of the reference's
depth2pointsrgbpm (tools/PENet/dataloaders/my_loader.py:502-617, SURVEY.md
A.3):  FOV filter -> plane -> above-plane + label inheritance ->
[back-projection + spherical voxel sampling] -> per-instance radius filters ->
BallQuery fuse -> range clustering -> rectangle search -> box rows.

Views are processed in chunks (`views_per_chunk`) so that the workspace is
bounded; no host synchronisation happens inside a chunk -- counts stay on the
device -- and one sync at the end of run() collects the row counters.
"""
from dataclasses import dataclass, field
from typing import Optional

import ctypes

import numpy as np
import torch

from . import stages as st
from ._lib import Dfu3dError
from .params import Params


@dataclass
class ViewBatch:
    """Inputs for V camera views over F LiDAR frames (device tensors)."""
    points: torch.Tensor        # (N,4) f32, frames packed back to back
    pt_off: torch.Tensor        # (F+1,) i32
    view_frame: torch.Tensor    # (V,) i32
    calib: torch.Tensor         # (V,48) f32  (Calibration.record())
    masks: torch.Tensor         # (V,M,H,W) u8 planes, or (V,H,W) packed words when mask_format is 1/2/4
    n_inst: torch.Tensor        # (V,) i32
    inst_class: torch.Tensor    # (V,M) i32  index into the 10 nuScenes names
    inst_is_car: torch.Tensor   # (V,M) i32  SEEM name == "Car" (my_loader.py:651)
    inst_r_lidar: torch.Tensor  # (V,M) f64  Params.instance_radii()[0]
    inst_r_pseudo: torch.Tensor  # (V,M) f64
    inst_box: torch.Tensor      # (V,M,4) f32 xyxy
    inst_score: torch.Tensor    # (V,M) f32
    view_key: torch.Tensor      # (V,) i64   RANSAC stream key
    host_pt_off: np.ndarray     # CPU copies for argument validation
    host_view_frame: np.ndarray
    depth: Optional[torch.Tensor] = None   # (V,H,W) f32; None = sparse mode (H20)
    plane: Optional[torch.Tensor] = None   # (V,4) f64; None = fit with RANSAC
    mask_format: int = 0                   # stages.MASK_BYTES, or the packed word size in bytes (1 / 2 / 4)
    # identity of every view for the gathered rows: global frame index and camera;
    # None = frame index inside this batch, camera = view position inside its frame
    frame_ids: Optional[np.ndarray] = None     # (F,) int64 global index of the batch's frames
    view_cam: Optional[np.ndarray] = None      # (V,) int   camera of every view

    def pack_masks(self, word_bytes=None):
        """Replace the uint8 mask planes by ONE packed word per pixel (dfu3d_pack_masks): 1/8 of the
        bytes, one gather per looked-up pixel.  No-op when already packed."""
        if self.mask_format != st.MASK_BYTES:
            return self
        V, M, H, W = self.masks.shape
        self.masks = st.pack_masks(self.masks.contiguous(), self.n_inst.contiguous(), V, M, H, W, word_bytes=word_bytes)
        self.mask_format = self.masks.element_size()
        return self


class PseudoBoxEngine:
    def __init__(self, params: Params, H: int, W: int, max_inst: int, cap_n: int,
                 views_per_chunk: int, dense: bool = True, cap_vox: int = 1 << 18,
                 pool_per_view: int = 1 << 16, rows_per_view: int = 64,
                 device="cuda:0", apply_fov: bool = True, lanes: int = 1, graphs: bool = False,
                 chain: bool = False, free_lanes: bool = False):
        if not torch.cuda.is_available():
            raise Dfu3dError("PseudoBoxEngine needs a GPU (no CPU fallback)")
        # H, W: the canvas of the masks and the depth maps.  The in-bounds test of my_loader.py:526 is
        # hard-coded to 1600x900 in the reference whatever the image size (hazard H11); here it is
        # params.bounds_hw and only has to fit on the canvas.
        if params.bounds_hw[0] > int(H) or params.bounds_hw[1] > int(W) or min(params.bounds_hw) < 1:
            raise Dfu3dError("params.bounds_hw %s does not fit on the %dx%d mask / depth canvas"
                             % (tuple(params.bounds_hw), H, W))
        self.p = params
        self.apply_fov = bool(apply_fov)   # False: the caller already FOV-filtered (vis_utils.py:152-154)
        self.H, self.W, self.M = int(H), int(W), int(max_inst)
        self.cap_n = int(cap_n)
        self.Vc = int(views_per_chunk)
        self.dense = bool(dense)
        self.cap_vox = int(cap_vox) if dense else 1
        self.pool_cap = int(pool_per_view) * self.Vc
        self.cap_rows = int(rows_per_view) * self.Vc
        self.dev = torch.device(device)
        self.n_theta, self.dtheta = params.thetas()
        # `lanes` independent workspaces, each with its own HIP stream: chunks are
        # dealt round-robin so that the long single-workgroup tails of one chunk
        # (clustering / L-shape of a huge instance) overlap the streaming kernels
        # of the others.
        # graphs=True: the launch chain of every chunk is captured into a hipGraph the first time a
        # batch (identified by the addresses of its tensors) is seen and replayed afterwards --
        # no per-launch host work, no gaps between the ~50 kernels of a chunk.
        # chain=True: every chunk is ONE C call (dfu3d_pseudo_boxes) that sequences the same stage entry
        # points over a single workspace -- what a non-Python host would bind.  Per-kernel timing needs the
        # stage-by-stage path, so `timing` overrides it.
        # free_lanes=True (with lanes > 1, without graphs): every lane on a stream of its own and NO join between the lanes at the
        # end of a pass -- a lane's next chunk waits for that lane's previous chunk only, the pass's handle carries one event per
        # lane and collect() waits for those.  Lanes can then run out of step (one in the instruction-bound binning pass while the
        # other is in the memory- and latency-bound stages) instead of being re-aligned at every pass.
        # (run(sync=True) still orders the calling stream behind the pass; after launch() the lanes' work is ordered behind nothing
        # but collect(): call the other methods of the engine only when every pass in flight has been collected.)
        self.chain = bool(chain)
        self.graphs = bool(graphs)
        self.free_lanes = bool(free_lanes) and int(lanes) > 1 and not self.graphs
        self._graph_key = None
        self._graph_state = None
        self.lanes = [self._make_lane(i, cap_n) for i in range(max(1, int(lanes)))]
        self._bind(self.lanes[0])
        # optional per-kernel timing with HIP events on the launch stream
        self.timing = False
        self._events = []
        self._counts = {}


    class _Lane:
        pass

    def _make_lane(self, index, cap_n):
        L = PseudoBoxEngine._Lane()
        L.stream = (torch.cuda.current_stream(self.dev) if index == 0 and not self.graphs and not self.free_lanes
                    else torch.cuda.Stream(self.dev))
        with torch.cuda.stream(L.stream):
            V, S = self.Vc, self.Vc * self.M
            d = self.dev
            i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=d)
            f64 = lambda *s: torch.empty(s, dtype=torch.float64, device=d)
            L.fov_idx, L.cand_idx = i32(V * cap_n), i32(V * cap_n)
            L.ag_pt, L.ib_pix = i32(V * cap_n), i32(V * cap_n)
            L.n_fov, L.n_ag, L.K = i32(V), i32(V), i32(V)
            L.plane = f64(V * 4)
            L.a_bits = i32(V * cap_n)
            L.a_x, L.a_y, L.a_z = f64(V * cap_n), f64(V * cap_n), f64(V * cap_n)
            cv = self.cap_vox
            L.n_vox = torch.zeros(V, dtype=torch.int32, device=d)
            L.vox_pix, L.b_bits = i32(V * cv), i32(V * cv)
            L.b_x, L.b_y, L.b_z = f64(V * cv), f64(V * cv), f64(V * cv)
            if self.dense:
                L.geom, L.E = st.make_geom(
                    self.p.depth_min, self.p.z_max, self.p.theta_min, self.p.vsize,
                    self.p.vrange_min, self.p.vgrid, self.p.max_points_per_voxel, self.p.max_voxels)
                L.table = torch.empty(V * L.E * st.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=d)
                st.bin_table_init(L.table, V * L.E)
                pw, bw = st.backproject_scratch_words(V, self.H, self.W, cv, self.p.max_points_per_voxel, L.geom)
                L.pix_bin, L.blk_cnt = i32(pw), i32(bw)
            pc = self.pool_cap
            L.px, L.py, L.pz = f64(pc), f64(pc), f64(pc)
            L.sx, L.sy = f64(pc), f64(pc)
            L.label, L.sroot = i32(pc), i32(pc)
            L.si3 = i32(3 * pc)
            L.fit_ws = f64(int(st._lib.lib().dfu3d_lshape_fit_ws_doubles(pc, self.cap_rows)))
            L.flags = torch.empty(pc, dtype=torch.uint8, device=d)
            L.mean_d = f64(pc) if self.p.stat_filter else None
            L.vd_scratch = (torch.empty(st.voxel_down_sample_scratch_bytes(pc), dtype=torch.uint8, device=d)
                            if self.p.stat_filter else None)
            L.base_a = torch.empty(S, dtype=torch.int64, device=d)
            L.base_b = torch.empty(S, dtype=torch.int64, device=d)
            L.cnt_a, L.cnt_b, L.cnt_all = i32(S), i32(S), i32(S)
            L.tile_off = i32(2 * S + 2)
            L.base_ab = torch.empty(2 * S, dtype=torch.int64, device=d)
            L.cnt_ab = i32(2 * S)
            L.rad_ab = f64(2 * S)
            L.queue = i32(st.rf_queue_ints(pc))
            L.shadow = torch.empty(st.shadow_floats(pc), dtype=torch.float32, device=d)
            L.chunk_cnt = i32(int(st._lib.lib().dfu3d_segments_scratch_words(V, cap_n, cv)))
            L.pool_cursor = torch.zeros(1, dtype=torch.int64, device=d)
            L.stat_enable = torch.ones(S, dtype=torch.int32, device=d)
            L.rf_points = torch.zeros(1, dtype=torch.int64, device=d)   # timing mode only
            if self.chain:
                from ._lib import ChainCfg
                p = self.p
                c = ChainCfg()
                c.V, c.H, c.W, c.max_inst, c.cap_n, c.cap_vox, c.cap_rows = (V, self.H, self.W, self.M, cap_n,
                                                                            self.cap_vox, self.cap_rows)
                c.dense, c.apply_fov, c.fov_h, c.fov_w = int(self.dense), int(self.apply_fov), int(p.fov_hw[0]), int(p.fov_hw[1])
                c.bounds_h, c.bounds_w, c.mask_format = int(p.bounds_hw[0]), int(p.bounds_hw[1]), st.MASK_BYTES
                c.stat_filter, c.pool_cap = int(p.stat_filter), self.pool_cap
                c.plane_max_hs, c.plane_range, c.plane_offset = p.plane_max_hs, p.plane_range, p.plane_offset
                c.ransac_trials, c.nb_points = int(p.ransac_trials), int(p.nb_points)
                c.ransac_seed = int(p.ransac_seed) & ((1 << 64) - 1)
                c.fuse_C, c.R0, c.Rd = p.fuse_C, p.R0, p.Rd
                c.n_theta, c.stat_nb_neighbors = int(self.n_theta), int(p.stat_nb_neighbors)
                c.dtheta, c.car_aspect_max, c.stat_std_ratio = self.dtheta, p.car_aspect_max, p.stat_std_ratio
                c.stat_voxel = p.stat_voxel
                if self.dense:
                    c.geom = L.geom
                lib = st._lib.lib()
                nbytes = int(lib.dfu3d_chain_workspace_bytes(c))
                if nbytes <= 0:
                    raise Dfu3dError("dfu3d_chain_workspace_bytes: invalid configuration")
                L.chain_cfg = c
                L.chain_ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
                st._lib.check(lib.dfu3d_chain_workspace_init(c, L.chain_ws.data_ptr(), st._stream()),
                              "dfu3d_chain_workspace_init")
        return L

    def _bind(self, L):
        """Make lane L's buffers the ones the stage calls use."""
        for k, v in L.__dict__.items():
            if k != "stream":
                setattr(self, k, v)

    # ------------------------------------------------------------------
    def _run(self, name, fn, *a, **k):
        if not self.timing:
            return fn(*a, **k)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a, **k)
        e1.record()
        self._events.append((name, e0, e1))

    def _phased(self, prefix, fn, phases, *a, **kw):
        """Issue a multi-kernel stage phase by phase so each kernel is bracketed."""
        if not self.timing:
            allbits = 0
            for _, bit in phases:
                allbits |= bit
            return fn(*a, phases=allbits, **kw)
        for tag, bit in phases:
            self._run(prefix + tag, fn, *a, phases=bit, **kw)

    def timing_summary(self):
        """-> {name: (total_ms, launches)}; synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self._events:
            t, n = out.get(name, (0.0, 0))
            out[name] = (t + e0.elapsed_time(e1), n + 1)
        return out

    def reset_timing(self):
        self._events = []
        self._counts = {}
        for L in self.lanes:
            L.rf_points.zero_()

    def rf_points_total(self):
        return int(sum(int(L.rf_points.item()) for L in self.lanes))

    def _count(self, name, t):
        """timing mode: accumulate a device-side count (units of the algorithmic byte figures)."""
        if self.timing:
            v = t.sum().to(torch.int64) if isinstance(t, torch.Tensor) else torch.tensor(int(t), device=self.dev)
            self._counts[name] = self._counts.get(name, 0) + v

    def counters(self):
        """-> {name: total over the bracketed passes}; synchronises."""
        return {k: int(v.item()) if isinstance(v, torch.Tensor) else int(v) for k, v in getattr(self, "_counts", {}).items()}

    # ------------------------------------------------------------------
    def _chunk_chain(self, b: ViewBatch, v0: int, v1: int, rows, n_rows, status):
        """One dfu3d_pseudo_boxes call for the chunk [v0, v1).  Every operand goes through the same
        dtype / device / size check as on the stage-by-stage path (a kernel must never see a foreign buffer)."""
        V, M = v1 - v0, self.M
        S = V * M
        st._check_frames(b.points, b.pt_off, b.view_frame[v0:v1], V, self.cap_n, b.host_pt_off,
                         b.host_view_frame[v0:v1])
        c = lambda t: t.contiguous()
        f = lambda t: c(t[v0:v1].reshape(-1))
        dense = self.dense and b.depth is not None
        if self.dense and b.depth is None:
            raise Dfu3dError("chain=True with a dense engine needs depth (build the engine with dense=False)")
        F = b.pt_off.numel() - 1
        T = torch
        args = [
            st._chk(c(b.points), "points", T.float32), st._chk(c(b.pt_off), "pt_off", T.int32, numel=F + 1),
            st._chk(c(b.view_frame[v0:v1]), "view_frame", T.int32, numel=V),
            st._chk(c(b.calib[v0:v1]), "calib", T.float32, numel=V * st.CALIB_FLOATS),
            st._chk_masks(c(b.masks[v0:v1]), b.mask_format, V, M, self.H, self.W),
            st._chk(c(b.n_inst[v0:v1]), "n_inst", T.int32, numel=V),
            st._chk(c(b.depth[v0:v1]), "depth", T.float32, numel=V * self.H * self.W) if dense else None,
            st._chk(c(b.view_key[v0:v1]), "view_key", T.int64, numel=V),
            None if b.plane is None else st._chk(c(b.plane[v0:v1]), "plane", T.float64, numel=V * 4),
            st._chk(f(b.inst_class), "inst_class", T.int32, numel=S),
            st._chk(f(b.inst_is_car), "inst_is_car", T.int32, numel=S),
            st._chk(f(b.inst_r_lidar), "inst_r_lidar", T.float64, numel=S),
            st._chk(f(b.inst_r_pseudo), "inst_r_pseudo", T.float64, numel=S),
            st._chk(f(b.inst_box), "inst_box", T.float32, numel=S * 4),
            st._chk(f(b.inst_score), "inst_score", T.float32, numel=S),
            st._chk(self.chain_ws, "workspace", T.uint8),
            st._chk(rows, "rows", T.float64, numel=self.cap_rows * st.ROW_DOUBLES),
            st._chk(n_rows, "n_rows", T.int32, numel=1), st._chk(status, "status", T.int32, numel=1)]
        self.chain_cfg.mask_format = int(b.mask_format)
        rc = st._lib.lib().dfu3d_pseudo_boxes(self.chain_cfg, *args, st._stream())
        st._lib.check(rc, "dfu3d_pseudo_boxes")

    def _chunk(self, b: ViewBatch, v0: int, v1: int, rows, n_rows, status):
        if self.chain and not self.timing:
            return self._chunk_chain(b, v0, v1, rows, n_rows, status)
        p, V, M, H, W = self.p, v1 - v0, self.M, self.H, self.W
        if V != self.Vc:
            raise Dfu3dError("internal: chunk size mismatch")
        S, cap_n = V * M, self.cap_n
        vf = b.view_frame[v0:v1]
        calib = b.calib[v0:v1]
        masks = b.masks[v0:v1]
        n_inst = b.n_inst[v0:v1]
        R = self._run
        if self.apply_fov:
            R("fov_filter", st.fov_filter, b.points, b.pt_off, vf, calib, V, p.fov_hw, cap_n,
              self.fov_idx, self.n_fov, b.host_pt_off, b.host_view_frame[v0:v1])
        else:                       # every point of the frame, in order
            st._check_frames(b.points, b.pt_off, vf, V, cap_n, b.host_pt_off, b.host_view_frame[v0:v1])
            sizes = (b.pt_off[1:] - b.pt_off[:-1])[vf.long()]
            self.n_fov.copy_(sizes)
            self.fov_idx.view(V, cap_n).copy_(
                torch.arange(cap_n, dtype=torch.int32, device=self.dev).expand(V, cap_n))
        if b.plane is None:
            R("plane_ransac", st.plane_ransac, b.points, b.pt_off, vf, self.fov_idx, self.n_fov, V, cap_n,
                            p.plane_max_hs, p.plane_range, p.ransac_trials, p.ransac_seed,
                            b.view_key[v0:v1], self.cand_idx, self.plane)
            plane = self.plane
        else:
            plane = b.plane[v0:v1].reshape(-1)
        R("project_label", st.project_label, b.points, b.pt_off, vf, calib, plane, self.fov_idx, self.n_fov, masks,
                         n_inst, V, M, H, W, cap_n, p.plane_offset,
                         p.plane_range, self.ag_pt, self.ib_pix, self.n_ag, self.K, self.a_bits,
                         self.a_x, self.a_y, self.a_z, mask_format=b.mask_format, bounds_hw=p.bounds_hw)
        self._count("fov_points", self.n_fov)
        self._count("label_rows", self.K)
        if self.dense and b.depth is not None:
            self._phased("bp_", st.backproject_bin,
                         (("bin", st.BP_BIN), ("amb", st.BP_AMB), ("mark", st.BP_MARK), ("vox", st.BP_VOX),
                          ("repair", st.BP_REPAIR)),
                         b.depth[v0:v1], calib, masks, n_inst, V, M, H, W, self.geom,
                         self.E, 1, self.table, self.pix_bin, self.blk_cnt, self.cap_vox,
                         self.n_vox, self.vox_pix, self.b_bits, self.b_x, self.b_y,
                         self.b_z, status, mask_format=b.mask_format)
            self._count("voxels", self.n_vox)
            self._count("amb_pixels", self.blk_cnt[:V])
        else:
            self.n_vox.zero_()
        self.pool_cursor.zero_()
        rl = b.inst_r_lidar[v0:v1].reshape(-1)
        rp = b.inst_r_pseudo[v0:v1].reshape(-1)
        fused = not p.stat_filter
        R("segments_build", st.segments_build, self.a_bits, self.a_x, self.a_y, self.a_z, self.K, cap_n, self.b_bits,
                          self.b_x, self.b_y, self.b_z, self.n_vox, self.cap_vox, V, M,
                          self.pool_cap, self.pool_cursor, self.px, self.py, self.pz,
                          self.base_a, self.cnt_a, self.base_b, self.cnt_b, status,
                          rad_a=rl.contiguous(), rad_b=rp.contiguous(), shadow=self.shadow if fused else None,
                          base_ab=self.base_ab if fused else None, cnt_ab=self.cnt_ab, rad_ab=self.rad_ab,
                          chunk_cnt=self.chunk_cnt)
        if self.timing:
            self.rf_points += self.cnt_a.sum() + self.cnt_b.sum()
        self._count("pool_points", self.pool_cursor)
        if fused:
            # ONE pass over the LiDAR and the pseudo lists of all instances (2S segments) on the float32
            # shadow written by the segment build; both keep their flags -- the fuse below reads them and
            # compacts the lists once for both filters
            self._phased("rf_", st.radius_filter, (("flags", st.RF_FLAGS), ("resolve", st.RF_RESOLVE)),
                         self.px, self.py, self.pz, self.base_ab, self.cnt_ab, self.rad_ab, p.nb_points, 2 * S,
                         self.pool_cap, self.tile_off, self.flags, self.queue, shadow=self.shadow,
                         n_used=self.pool_cursor)
        else:           # the (dormant) statistical filter sits in between and needs the filtered lists
            rf_ph = (("shadow", st.RF_SHADOW), ("flags", st.RF_FLAGS), ("resolve", st.RF_RESOLVE),
                     ("compact", st.RF_COMPACT))
            self._phased("rf_", st.radius_filter, rf_ph, self.px, self.py, self.pz, self.base_a,
                         self.cnt_a, rl, p.nb_points, S, self.pool_cap, self.tile_off, self.flags,
                         self.queue, shadow=self.shadow, n_used=self.pool_cursor)
            self._phased("rf_", st.radius_filter, rf_ph, self.px, self.py, self.pz, self.base_b,
                         self.cnt_b, rp, p.nb_points, S, self.pool_cap, self.tile_off, self.flags,
                         self.queue, shadow=self.shadow, n_used=self.pool_cursor)
        if p.stat_filter:       # the reference's dormant pair (my_loader0.py:734-735): centroids per 0.05 m voxel, then the k-NN rule
            R("voxel_down_sample", st.voxel_down_sample, self.px, self.py, self.pz, self.base_b, self.cnt_b,
              self.stat_enable, p.stat_voxel, S, self.pool_cap, self.vd_scratch, status)
            R("stat_filter", st.stat_filter, self.px, self.py, self.pz, self.base_b, self.cnt_b, self.stat_enable,
                           p.stat_nb_neighbors, p.stat_std_ratio, S, self.pool_cap,
                           self.tile_off, self.flags, self.mean_d)
        self._count("rf_undecided", self.queue[0:1024:16])
        self._count("ball_points", self.cnt_a)
        self._count("ball_points", self.cnt_b)
        R("ballquery_fuse", st.ballquery_fuse, self.px, self.py, self.pz, self.base_a, self.cnt_a, self.base_b,
                          self.cnt_b, p.fuse_C, S, self.pool_cap, self.tile_off, self.flags,
                          masked="joint" if fused else False)
        torch.add(self.cnt_a, self.cnt_b, out=self.cnt_all)     # cat(lidar, pseudo)
        self._count("instance_points", self.cnt_all)
        R("range_cluster", st.range_cluster, self.px, self.py, self.base_a, self.cnt_all, S, p.R0,
          p.Rd, self.label, self.pool_cap, self.sx, self.sy, self.si3)
        R("lshape_fit", st.lshape_fit, self.px, self.py, self.pz, self.label, self.base_a, self.cnt_all, S, M,
                      calib, b.inst_class[v0:v1].reshape(-1), b.inst_is_car[v0:v1].reshape(-1),
                      b.inst_box[v0:v1].reshape(-1), b.inst_score[v0:v1].reshape(-1),
                      self.n_theta, self.dtheta, p.car_aspect_max, self.sx, self.sy, self.sroot,
                      self.cap_rows, rows, n_rows, status, self.pool_cap, self.fit_ws)

    # ------------------------------------------------------------------
    def run(self, b: ViewBatch, sync: bool = True):
        """-> (rows (n,18) f64 device tensor sorted by (view, inst, cluster), status word)."""
        V = b.view_frame.numel()
        if V % self.Vc:
            raise Dfu3dError("number of views (%d) must be a multiple of views_per_chunk (%d)"
                             % (V, self.Vc))
        want = (V, self.M, self.H, self.W) if b.mask_format == st.MASK_BYTES else (V, self.H, self.W)
        if tuple(b.masks.shape) != want:
            raise Dfu3dError("masks: expected shape %s for mask_format %d" % (want, b.mask_format))
        if b.depth is not None and tuple(b.depth.shape) != (V, self.H, self.W):
            raise Dfu3dError("depth: expected shape %s" % ((V, self.H, self.W),))
        nch = V // self.Vc
        d = self.dev
        if self.graphs and not self.timing:
            return self._run_graphed(b, nch, sync)
        rows = torch.empty((nch, self.cap_rows * st.ROW_DOUBLES), dtype=torch.float64, device=d)
        n_rows = torch.zeros((nch,), dtype=torch.int32, device=d)
        status = torch.zeros((nch,), dtype=torch.int32, device=d)
        main = torch.cuda.current_stream(self.dev)
        free = self.free_lanes and not self.timing
        for L in (self.lanes if free else self.lanes[1:]):
            L.stream.wait_stream(main)              # inputs / row buffers are ready
            for t in (rows, n_rows, status):
                t.record_stream(L.stream)
        for c in range(nch):
            L = self.lanes[c % len(self.lanes)]
            self._bind(L)
            with torch.cuda.stream(L.stream if len(self.lanes) > 1 else main):
                self._chunk(b, c * self.Vc, (c + 1) * self.Vc, rows[c], n_rows[c:c + 1],
                            status[c:c + 1])
        if free:
            done = []
            for L in self.lanes[:min(nch, len(self.lanes))]:
                e = torch.cuda.Event()
                e.record(L.stream)                  # this lane's chunks of the pass are complete here
                done.append(e)
            if sync:
                for e in done:
                    main.wait_event(e)
        else:
            for L in self.lanes[1:]:
                main.wait_stream(L.stream)
            done = torch.cuda.Event()
            done.record(main)                       # the pass is complete here, whatever is enqueued behind it
        self._last = (rows, n_rows, status, done)
        if not sync:
            return None, None
        return self.collect()

    @staticmethod
    def _batch_key(b: ViewBatch):
        return tuple((k, t.data_ptr(), tuple(t.shape)) for k, t in sorted(b.__dict__.items())
                     if isinstance(t, torch.Tensor))

    def _run_graphed(self, b: ViewBatch, nch: int, sync: bool):
        d = self.dev
        main = torch.cuda.current_stream(d)
        key = self._batch_key(b)
        if key != self._graph_key:
            # static outputs + one captured graph per chunk (on the chunk's lane)
            rows = torch.empty((nch, self.cap_rows * st.ROW_DOUBLES), dtype=torch.float64, device=d)
            n_rows = torch.zeros((nch,), dtype=torch.int32, device=d)
            status = torch.zeros((nch,), dtype=torch.int32, device=d)
            torch.cuda.synchronize(d)
            graphs = []
            for c in range(nch):
                L = self.lanes[c % len(self.lanes)]
                self._bind(L)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=L.stream):
                    n_rows[c:c + 1].zero_()
                    status[c:c + 1].zero_()
                    self._chunk(b, c * self.Vc, (c + 1) * self.Vc, rows[c], n_rows[c:c + 1],
                                status[c:c + 1])
                graphs.append(g)
            torch.cuda.synchronize(d)
            self._graph_key = key
            self._graph_state = (graphs, rows, n_rows, status, b)     # b: keeps the captured addresses alive
        graphs, rows, n_rows, status, _ = self._graph_state
        for L in self.lanes:
            L.stream.wait_stream(main)
        for c, g in enumerate(graphs):
            with torch.cuda.stream(self.lanes[c % len(self.lanes)].stream):
                g.replay()
        for L in self.lanes:
            main.wait_stream(L.stream)
        self._last = (rows, n_rows, status, None)
        if not sync:
            return None, None
        return self.collect()

    def virtual_points(self, b: ViewBatch, v0: int = 0):
        """Voxel representatives under la_sampling2's rule (argmin z, my_loader.py:247-260)
        for the chunk of views starting at v0 -> (n_vox (Vc,), pixel (Vc,cap_vox), xyz (Vc,cap_vox,3))."""
        if not self.dense or b.depth is None:
            raise Dfu3dError("virtual_points needs the dense path")
        V = self.Vc
        status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        st.backproject_bin(b.depth[v0:v0 + V], b.calib[v0:v0 + V], None, None, V, self.M, self.H,
                           self.W, self.geom, self.E, 2, self.table, self.pix_bin, self.blk_cnt,
                           self.cap_vox, self.n_vox, self.vox_pix, self.b_bits, self.b_x, self.b_y,
                           self.b_z, status)
        xyz = torch.stack([self.b_x, self.b_y, self.b_z], 1).view(V, self.cap_vox, 3)
        return self.n_vox.clone(), self.vox_pix.view(V, self.cap_vox).clone(), xyz, int(status.item())

    def gather_layout(self, rows, b: ViewBatch):
        """Engine rows (n,24) -> (n,18) rows carrying the GLOBAL frame index and the camera (dist.rows_for_gather).
        The per-view lookup tables go to the device once per batch object."""
        from .dist import rows_for_gather
        maps = getattr(b, "_gather_maps", None)
        if maps is None:
            V = b.view_frame.numel()
            fid = b.frame_ids if b.frame_ids is not None else np.arange(b.pt_off.numel() - 1)
            if b.view_cam is not None:
                cam = b.view_cam
            else:                                   # position of the view among the views of its frame
                vf = np.asarray(b.host_view_frame)
                cam = np.array([int((vf[:i] == vf[i]).sum()) for i in range(V)], np.int64)
            d = rows.device
            maps = (torch.as_tensor(np.asarray(b.host_view_frame), device=d), torch.as_tensor(np.asarray(fid), device=d),
                    torch.as_tensor(np.asarray(cam), device=d))
            b._gather_maps = maps
        return rows_for_gather(rows, *maps)

    def launch(self, b: ViewBatch):
        """Enqueue a pass and return a handle for collect(): the host can enqueue the NEXT pass before it waits for
        this one (the row buffers are per pass; the workspace is reused in stream order).
        Limits of the pipelining: with graphs=True every pass writes the SAME captured row buffers, so a
        handle would be overwritten by the next launch -- refused; and a pass that raises DFU3D_ST_VOX_OVERFLOW leaves
        its bin table dirty, which collect() repairs only after the NEXT pass may already have been enqueued on it:
        collect() then reports the status, and the caller must not use the rows of the pass launched in between
        (bench.py aborts on any status)."""
        if self.graphs:
            raise Dfu3dError("launch() / collect() pipelining is not available with graphs=True (one set of captured "
                             "row buffers): use run()")
        self.run(b, sync=False)
        return self._last

    def collect(self, handle=None):
        """Rows of the pass behind `handle` (default: the last one).  May be called under another stream than the one the
        pass was launched on (`with torch.cuda.stream(side): eng.collect(h)`): the reads then wait for THAT pass only, not
        for whatever was enqueued behind it -- a caller that launches pass i+1 before it collects pass i gets the rows of
        pass i while pass i+1 runs, and can enqueue pass i+2 long before the GPU runs dry (bench.py).  The returned
        tensors belong to the calling stream."""
        h = handle if handle is not None else self._last
        rows, n_rows, status = h[:3]
        if len(h) > 3 and h[3] is not None:
            cur = torch.cuda.current_stream(self.dev)
            for e in (h[3] if isinstance(h[3], (list, tuple)) else (h[3],)):
                cur.wait_event(e)
            for t in (rows, n_rows, status):
                t.record_stream(cur)
        both = torch.stack((n_rows, status)).cpu().numpy()        # the one host sync
        counts = both[0]
        stw = int(np.bitwise_or.reduce(both[1].astype(np.int64)))
        if self.dense and (stw & st.ST_VOX_OVERFLOW):   # a pass that overflowed left its bin table dirty
            for L in self.lanes:                         # (the chain keeps its own table inside its workspace)
                with torch.cuda.stream(L.stream):
                    st.bin_table_init(L.table, self.Vc * L.E)
                    if self.chain:
                        st._lib.check(st._lib.lib().dfu3d_chain_workspace_init(
                            L.chain_cfg, L.chain_ws.data_ptr(), st._stream()), "dfu3d_chain_workspace_init")
            torch.cuda.synchronize()
        parts = []
        for c, n in enumerate(counts):
            n = int(min(n, self.cap_rows))
            if n:
                r = rows[c].view(self.cap_rows, st.ROW_DOUBLES)[:n].clone()
                r[:, 0] += c * self.Vc
                parts.append(r)
        if parts:
            out = torch.cat(parts, 0)
            key = (out[:, 0].long() * self.M + out[:, 1].long()) * (1 << 31) + out[:, 2].long()
            out = out[torch.argsort(key)]
        else:
            out = torch.empty((0, st.ROW_DOUBLES), dtype=torch.float64, device=self.dev)
        return out, stw
