"""Shape-checked Python wrappers over the C ABI (one per entry point of
include/dfu3d.h).  PyTorch is used for device memory and streams only; every
operand's dtype, device, contiguity and size is verified on the host before a
hand-written kernel is launched (a faulting kernel can reset the whole node).
"""
import ctypes
import math

import numpy as np
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import BinGeom, Dfu3dError

CALIB_FLOATS = 48
ROW_DOUBLES = 24
MAX_INST = 32

BP_BIN, BP_AMB, BP_MARK, BP_VOX, BP_REPAIR, BP_ALL = 1, 2, 4, 8, 16, 31
MASK_BYTES = 0
RF_SHADOW, RF_FLAGS, RF_RESOLVE, RF_COMPACT, RF_ALL = 1, 2, 4, 8, 15

ST_POOL_OVERFLOW = 1
ST_VOX_OVERFLOW = 2
ST_ROW_OVERFLOW = 4
ST_BIN_RANGE = 8
ST_VOX_PTS_OVERFLOW = 16
ST_VOXEL_RANGE = 32
STATUS_TEXT = {
    ST_POOL_OVERFLOW: "instance point pool too small (raise pool_cap)",
    ST_VOX_OVERFLOW: "more voxels than cap_vox in a view (raise cap_vox; bin table must be re-initialised)",
    ST_ROW_OVERFLOW: "more box rows than cap_rows",
    ST_BIN_RANGE: "a spherical bin fell outside the bin table",
    ST_VOX_PTS_OVERFLOW: "overflow-bin pixel list too small",
    ST_VOXEL_RANGE: "voxel_down_sample: a segment is wider than 2^21 voxels along an axis",
}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype, numel=None, min_numel=None):
    if not isinstance(t, torch.Tensor):
        raise Dfu3dError("%s: expected a torch.Tensor" % name)
    if not t.is_cuda:
        raise Dfu3dError("%s: must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != dtype:
        raise Dfu3dError("%s: dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise Dfu3dError("%s: must be contiguous" % name)
    if numel is not None and t.numel() != numel:
        raise Dfu3dError("%s: %d elements, expected %d" % (name, t.numel(), numel))
    if min_numel is not None and t.numel() < min_numel:
        raise Dfu3dError("%s: %d elements, need at least %d" % (name, t.numel(), min_numel))
    return ctypes.c_void_p(t.data_ptr())


def make_geom(depth_min=0.001, z_max=1.0, theta_min=1.5,
              vsize=(200.0, 0.0020000000949949026, 0.0020000000949949026),
              vrange_min=(-100.0, -5.0, -5.0), vgrid=(1, 5000, 5000),
              max_points_per_voxel=100, max_voxels=1000000):
    g = BinGeom()
    g.vsize_r, g.vsize_t, g.vsize_p = vsize
    g.rmin_r, g.rmin_t, g.rmin_p = vrange_min
    g.grid_r, g.grid_t, g.grid_p = vgrid
    g.max_points_per_voxel = int(max_points_per_voxel)
    g.max_voxels = int(max_voxels)
    g.theta_min, g.z_max, g.depth_min = theta_min, z_max, depth_min
    n = _lib.lib().dfu3d_bin_table_geometry(ctypes.byref(g))
    if n <= 0:
        raise Dfu3dError("invalid voxel geometry")
    return g, int(n)


def backproject_scratch_words(V, H, W, cap_vox, max_points, geom):
    """geom: the dfu3d_bin_geom of make_geom (its table decides the size of the segment lists)."""
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    rc = _lib.lib().dfu3d_backproject_scratch_words(V, H, W, cap_vox, max_points, int(geom.t_n) * int(geom.p_n),
                                                    ctypes.byref(a), ctypes.byref(b))
    if rc != 0:
        raise Dfu3dError("dfu3d_backproject_scratch_words: invalid sizes")
    return a.value, b.value


TABLE_ENTRY_BYTES = 28


def bin_table_init(table, entries_total):
    p = _chk(table, "table", torch.uint8, min_numel=entries_total * TABLE_ENTRY_BYTES)
    if table.data_ptr() % 8:
        raise Dfu3dError("table: must be 8-byte aligned")
    _lib.check(_lib.lib().dfu3d_bin_table_init(p, entries_total, _stream()), "bin_table_init")


def _check_frames(points, pt_off, view_frame, V, cap_n, host_pt_off=None, host_view_frame=None):
    """host_* are CPU copies used for bounds validation (no device sync)."""
    if points.dim() != 2 or points.shape[1] != 4:
        raise Dfu3dError("points: expected (N,4)")
    if host_pt_off is not None:
        off = [int(x) for x in host_pt_off]
        if off[0] < 0 or off[-1] > points.shape[0] or any(b < a for a, b in zip(off, off[1:])):
            raise Dfu3dError("pt_off: not a monotone partition of points")
        if max(b - a for a, b in zip(off, off[1:])) > cap_n:
            raise Dfu3dError("cap_n smaller than the largest frame")
        if host_view_frame is not None:
            vf = [int(x) for x in host_view_frame]
            if len(vf) != V or min(vf) < 0 or max(vf) >= len(off) - 1:
                raise Dfu3dError("view_frame: frame index out of range")


def fov_filter(points, pt_off, view_frame, calib, V, fov_hw, cap_n, fov_idx, n_fov,
               host_pt_off=None, host_view_frame=None):
    _check_frames(points, pt_off, view_frame, V, cap_n, host_pt_off, host_view_frame)
    rc = _lib.lib().dfu3d_fov_filter(
        _chk(points, "points", torch.float32), _chk(pt_off, "pt_off", torch.int32, min_numel=2),
        _chk(view_frame, "view_frame", torch.int32, numel=V),
        _chk(calib, "calib", torch.float32, numel=V * CALIB_FLOATS), V, int(fov_hw[0]),
        int(fov_hw[1]), cap_n, _chk(fov_idx, "fov_idx", torch.int32, numel=V * cap_n),
        _chk(n_fov, "n_fov", torch.int32, numel=V), _stream())
    _lib.check(rc, "dfu3d_fov_filter")


def plane_ransac(points, pt_off, view_frame, fov_idx, n_fov, V, cap_n, max_hs, xy_range,
                 trials, seed, key, cand_idx, plane):
    rc = _lib.lib().dfu3d_plane_ransac(
        _chk(points, "points", torch.float32), _chk(pt_off, "pt_off", torch.int32, min_numel=2),
        _chk(view_frame, "view_frame", torch.int32, numel=V),
        _chk(fov_idx, "fov_idx", torch.int32, numel=V * cap_n),
        _chk(n_fov, "n_fov", torch.int32, numel=V), V, cap_n, float(max_hs), float(xy_range),
        int(trials), ctypes.c_uint64(int(seed) & ((1 << 64) - 1)),
        _chk(key, "key", torch.int64, numel=V),
        _chk(cand_idx, "cand_idx", torch.int32, numel=V * cap_n),
        _chk(plane, "plane", torch.float64, numel=V * 4), _stream())
    _lib.check(rc, "dfu3d_plane_ransac")


def mask_word_bytes(max_inst):
    """Smallest packed mask word (bytes) that holds max_inst instance bits."""
    return 1 if max_inst <= 8 else (2 if max_inst <= 16 else 4)


_WORD_DTYPE = {1: torch.uint8, 2: torch.int16, 4: torch.int32}


def _chk_masks(masks, mask_format, V, max_inst, H, W):
    """masks: uint8 planes (V,max_inst,H,W) for MASK_BYTES, else packed words (V,H,W) of mask_format bytes."""
    if max_inst < 1 or max_inst > MAX_INST:
        raise Dfu3dError("max_inst must be in [1, %d]" % MAX_INST)
    if mask_format == MASK_BYTES:
        return _chk(masks, "masks", torch.uint8, numel=V * max_inst * H * W)
    if mask_format not in _WORD_DTYPE or mask_word_bytes(max_inst) > mask_format:
        raise Dfu3dError("mask_format %r cannot hold %d instance bits" % (mask_format, max_inst))
    if masks.data_ptr() % mask_format:
        raise Dfu3dError("masks: misaligned for %d-byte words" % mask_format)
    return _chk(masks, "masks (packed)", _WORD_DTYPE[mask_format], numel=V * H * W)


def pack_masks(masks, n_inst, V, max_inst, H, W, out=None, word_bytes=None):
    """uint8 planes (V,max_inst,H,W) -> packed words (V,H,W), bit j = plane j > 0 (dfu3d_pack_masks)."""
    wb = word_bytes or mask_word_bytes(max_inst)
    if out is None:
        out = torch.empty((V, H, W), dtype=_WORD_DTYPE[wb], device=masks.device)
    rc = _lib.lib().dfu3d_pack_masks(_chk(masks, "masks", torch.uint8, numel=V * max_inst * H * W),
                                     _chk(n_inst, "n_inst", torch.int32, numel=V), V, max_inst, H, W,
                                     _chk(out, "out", _WORD_DTYPE[wb], numel=V * H * W), wb, _stream())
    _lib.check(rc, "dfu3d_pack_masks")
    return out


def project_label(points, pt_off, view_frame, calib, plane, fov_idx, n_fov, masks, n_inst, V,
                  max_inst, H, W, cap_n, plane_offset, xy_range, ag_pt, ib_pix, n_ag, K,
                  it_bits, it_x, it_y, it_z, mask_format=MASK_BYTES, bounds_hw=None):
    """H, W: mask canvas; bounds_hw: the in-bounds test of my_loader.py:526 (default: the canvas)."""
    bh, bw = (H, W) if bounds_hw is None else (int(bounds_hw[0]), int(bounds_hw[1]))
    if bh > H or bw > W or bh < 1 or bw < 1:
        raise Dfu3dError("bounds %s must lie inside the mask canvas (%d,%d)" % ((bh, bw), H, W))
    rc = _lib.lib().dfu3d_project_label(
        _chk(points, "points", torch.float32), _chk(pt_off, "pt_off", torch.int32, min_numel=2),
        _chk(view_frame, "view_frame", torch.int32, numel=V),
        _chk(calib, "calib", torch.float32, numel=V * CALIB_FLOATS),
        _chk(plane, "plane", torch.float64, numel=V * 4),
        _chk(fov_idx, "fov_idx", torch.int32, numel=V * cap_n),
        _chk(n_fov, "n_fov", torch.int32, numel=V),
        _chk_masks(masks, mask_format, V, max_inst, H, W), int(mask_format),
        _chk(n_inst, "n_inst", torch.int32, numel=V), V, max_inst, H, W, bh, bw, cap_n,
        float(plane_offset), float(xy_range),
        _chk(ag_pt, "ag_pt", torch.int32, numel=V * cap_n),
        _chk(ib_pix, "ib_pix", torch.int32, numel=V * cap_n),
        _chk(n_ag, "n_ag", torch.int32, numel=V), _chk(K, "K", torch.int32, numel=V),
        _chk(it_bits, "it_bits", torch.int32, numel=V * cap_n),
        _chk(it_x, "it_x", torch.float64, numel=V * cap_n),
        _chk(it_y, "it_y", torch.float64, numel=V * cap_n),
        _chk(it_z, "it_z", torch.float64, numel=V * cap_n), _stream())
    _lib.check(rc, "dfu3d_project_label")


def backproject_bin(depth, calib, masks, n_inst, V, max_inst, H, W, geom, table_entries,
                    key_axis, table, pix_bin, blk_cnt, cap_vox, n_vox, vox_pix, it_bits, it_x,
                    it_y, it_z, status, phases=BP_ALL, mask_format=MASK_BYTES):
    pw, bw = backproject_scratch_words(V, H, W, cap_vox, geom.max_points_per_voxel, geom)
    if table.data_ptr() % 8 or blk_cnt.data_ptr() % 8:
        raise Dfu3dError("table / blk_cnt: must be 8-byte aligned")
    if depth.data_ptr() % 16:
        raise Dfu3dError("depth: must be 16-byte aligned")
    if W % 4:
        raise Dfu3dError("W must be a multiple of 4 (float4 depth loads)")
    mp = _chk_masks(masks, mask_format, V, max_inst, H, W) if masks is not None else None
    ni = _chk(n_inst, "n_inst", torch.int32, numel=V) if n_inst is not None else None
    rc = _lib.lib().dfu3d_backproject_bin(
        _chk(depth, "depth", torch.float32, numel=V * H * W),
        _chk(calib, "calib", torch.float32, numel=V * CALIB_FLOATS), mp, int(mask_format), ni, V, max_inst, H, W,
        ctypes.byref(geom), int(key_axis),
        _chk(table, "table", torch.uint8, min_numel=V * table_entries * TABLE_ENTRY_BYTES),
        _chk(pix_bin, "pix_bin", torch.int32, min_numel=pw),
        _chk(blk_cnt, "blk_cnt", torch.int32, min_numel=bw), cap_vox,
        _chk(n_vox, "n_vox", torch.int32, numel=V),
        _chk(vox_pix, "vox_pix", torch.int32, numel=V * cap_vox),
        _chk(it_bits, "it_bits", torch.int32, numel=V * cap_vox),
        _chk(it_x, "it_x", torch.float64, numel=V * cap_vox),
        _chk(it_y, "it_y", torch.float64, numel=V * cap_vox),
        _chk(it_z, "it_z", torch.float64, numel=V * cap_vox),
        _chk(status, "status", torch.int32, min_numel=1), int(phases), _stream())
    _lib.check(rc, "dfu3d_backproject_bin")


def shadow_floats(pool_cap):
    """float32 elements of the radius filter's shadow scratch (DFU3D_SHADOW_BYTES / 4)."""
    return 8 * pool_cap + (16 + 512 + 128) * ((pool_cap + 511) // 512 + 1) + 9699456 // 4


def rf_queue_ints(pool_cap):
    """int32 elements of the radius filter's queue scratch (DFU3D_RF_QUEUE_INTS)."""
    return 1024 + 64 * (((pool_cap + 2047) // 2048 + 63) // 64) * 2048


def segments_build(a_bits, a_x, a_y, a_z, a_n, a_cap, b_bits, b_x, b_y, b_z, b_n, b_cap, V,
                   max_inst, pool_cap, pool_cursor, px, py, pz, base_a, cnt_a, base_b, cnt_b,
                   status, rad_a=None, rad_b=None, shadow=None, base_ab=None, cnt_ab=None, rad_ab=None, chunk_cnt=None):
    """rad_a / rad_b + shadow (+ the joint 2S segment table): everything the one-pass radius filter needs."""
    S = V * max_inst
    nw = int(_lib.lib().dfu3d_segments_scratch_words(V, a_cap, b_cap))
    if chunk_cnt is None:
        chunk_cnt = torch.empty(nw, dtype=torch.int32, device=px.device)
    opt = lambda t, name, dt, n: None if t is None else _chk(t, name, dt, numel=n)
    if shadow is not None and shadow.data_ptr() % 16:
        raise Dfu3dError("shadow: must be 16-byte aligned")
    rc = _lib.lib().dfu3d_segments_build(
        _chk(a_bits, "a_bits", torch.int32, numel=V * a_cap),
        _chk(a_x, "a_x", torch.float64, numel=V * a_cap),
        _chk(a_y, "a_y", torch.float64, numel=V * a_cap),
        _chk(a_z, "a_z", torch.float64, numel=V * a_cap), _chk(a_n, "a_n", torch.int32, numel=V),
        a_cap, _chk(b_bits, "b_bits", torch.int32, numel=V * b_cap),
        _chk(b_x, "b_x", torch.float64, numel=V * b_cap),
        _chk(b_y, "b_y", torch.float64, numel=V * b_cap),
        _chk(b_z, "b_z", torch.float64, numel=V * b_cap), _chk(b_n, "b_n", torch.int32, numel=V),
        b_cap, V, max_inst, pool_cap, _chk(pool_cursor, "pool_cursor", torch.int64, numel=1),
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(pz, "pz", torch.float64, numel=pool_cap),
        _chk(base_a, "base_a", torch.int64, numel=S), _chk(cnt_a, "cnt_a", torch.int32, numel=S),
        _chk(base_b, "base_b", torch.int64, numel=S), _chk(cnt_b, "cnt_b", torch.int32, numel=S),
        _chk(status, "status", torch.int32, min_numel=1),
        opt(rad_a, "rad_a", torch.float64, S), opt(rad_b, "rad_b", torch.float64, S),
        opt(shadow, "shadow", torch.float32, shadow_floats(pool_cap)), opt(base_ab, "base_ab", torch.int64, 2 * S),
        opt(cnt_ab, "cnt_ab", torch.int32, 2 * S), opt(rad_ab, "rad_ab", torch.float64, 2 * S),
        _chk(chunk_cnt, "chunk_cnt", torch.int32, min_numel=nw), _stream())
    _lib.check(rc, "dfu3d_segments_build")


def radius_filter(px, py, pz, seg_base, seg_cnt, radius, nb_points, S, pool_cap, tile_off, flags,
                  queue, phases=RF_ALL, shadow=None, n_used=None):
    """shadow: float32 (4*pool_cap) scratch (allocated here when omitted); n_used: device int64 (1,) or None."""
    if shadow is None:
        shadow = torch.empty(shadow_floats(pool_cap), dtype=torch.float32, device=px.device)
    if shadow.data_ptr() % 16:
        raise Dfu3dError("shadow: must be 16-byte aligned")
    rc = _lib.lib().dfu3d_radius_filter(
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(pz, "pz", torch.float64, numel=pool_cap),
        _chk(seg_base, "seg_base", torch.int64, numel=S),
        _chk(seg_cnt, "seg_cnt", torch.int32, numel=S),
        _chk(radius, "radius", torch.float64, numel=S), int(nb_points), S, pool_cap,
        None if n_used is None else _chk(n_used, "n_used", torch.int64, numel=1),
        _chk(shadow, "shadow", torch.float32, numel=shadow_floats(pool_cap)),
        _chk(tile_off, "tile_off", torch.int32, min_numel=S + 1),
        _chk(flags, "flags", torch.uint8, numel=pool_cap),
        _chk(queue, "queue", torch.int32, min_numel=rf_queue_ints(pool_cap)), int(phases), _stream())
    _lib.check(rc, "dfu3d_radius_filter")


def voxel_down_sample_scratch_bytes(pool_cap):
    return int(_lib.lib().dfu3d_voxel_down_sample_scratch_bytes(int(pool_cap)))


def voxel_down_sample(px, py, pz, seg_base, seg_cnt, enable, voxel_size, S, pool_cap, scratch, status):
    """Open3D voxel_down_sample per enabled segment, in place, centroids in first-seen voxel order (include/dfu3d.h)."""
    rc = _lib.lib().dfu3d_voxel_down_sample(
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(pz, "pz", torch.float64, numel=pool_cap), _chk(seg_base, "seg_base", torch.int64, numel=S),
        _chk(seg_cnt, "seg_cnt", torch.int32, numel=S), _chk(enable, "enable", torch.int32, numel=S),
        float(voxel_size), S, pool_cap,
        _chk(scratch, "scratch", torch.uint8, min_numel=voxel_down_sample_scratch_bytes(pool_cap)),
        _chk(status, "status", torch.int32, min_numel=1), _stream())
    _lib.check(rc, "dfu3d_voxel_down_sample")


def stat_filter(px, py, pz, seg_base, seg_cnt, enable, nb_neighbors, std_ratio, S, pool_cap,
                tile_off, flags, mean_d):
    rc = _lib.lib().dfu3d_stat_filter(
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(pz, "pz", torch.float64, numel=pool_cap),
        _chk(seg_base, "seg_base", torch.int64, numel=S),
        _chk(seg_cnt, "seg_cnt", torch.int32, numel=S),
        _chk(enable, "enable", torch.int32, numel=S), int(nb_neighbors), float(std_ratio), S,
        pool_cap, _chk(tile_off, "tile_off", torch.int32, min_numel=S + 1),
        _chk(flags, "flags", torch.uint8, numel=pool_cap),
        _chk(mean_d, "mean_d", torch.float64, numel=pool_cap), None, _stream())
    _lib.check(rc, "dfu3d_stat_filter")


def ballquery_fuse(px, py, pz, base_a, cnt_a, base_b, cnt_b, C, S, pool_cap, tile_off, flags, masked=False):
    """masked=True: `flags` holds the keep mask of a radius_filter call over the B lists made without RF_COMPACT;
    masked="joint": of a joint call over the A and B lists (cnt_a is updated: A is compacted here too)."""
    L = _lib.lib()
    fn = L.dfu3d_ballquery_fuse_joint if masked == "joint" else (L.dfu3d_ballquery_fuse_masked if masked else L.dfu3d_ballquery_fuse)
    rc = fn(
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(pz, "pz", torch.float64, numel=pool_cap),
        _chk(base_a, "base_a", torch.int64, numel=S), _chk(cnt_a, "cnt_a", torch.int32, numel=S),
        _chk(base_b, "base_b", torch.int64, numel=S), _chk(cnt_b, "cnt_b", torch.int32, numel=S),
        float(C), S, pool_cap, _chk(tile_off, "tile_off", torch.int32, min_numel=2 * S + 2),
        _chk(flags, "flags", torch.uint8, numel=pool_cap), _stream())
    _lib.check(rc, "dfu3d_ballquery_fuse")


def points_in_boxes_mask(points, boxes):
    """points (n,3|4) float32 CUDA, boxes (B,7) float64 CUDA -> int32 (B,n)
    (roiaware_pool3d_utils.points_in_boxes_cpu, pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-25)."""
    if points.dim() != 2 or points.shape[1] not in (3, 4) or boxes.dim() != 2 or boxes.shape[1] != 7:
        raise Dfu3dError("points must be (n,3|4), boxes (B,7)")
    n, B = int(points.shape[0]), int(boxes.shape[0])
    out = torch.zeros((B, n), dtype=torch.int32, device=points.device)
    rc = _lib.lib().dfu3d_points_in_boxes_mask(
        _chk(points, "points", torch.float32, numel=n * points.shape[1]), n, int(points.shape[1]),
        _chk(boxes, "boxes", torch.float64, numel=B * 7), B, _chk(out, "out", torch.int32, numel=B * n),
        _stream())
    _lib.check(rc, "dfu3d_points_in_boxes_mask")
    return out


def gt_database(points, pt_off, box_frame, boxes, cap_out, host_pt_off=None):
    """Many frames at once -> (box_cnt (Bt,), box_off (Bt+1,), idx (cap_out,), gt_pts (cap_out,4), status)."""
    Bt = int(boxes.shape[0])
    dev = points.device
    if Bt == 0:
        raise Dfu3dError("no boxes")
    if points.dim() != 2 or points.shape[1] != 4:
        raise Dfu3dError("points must be (N,4) float32")
    if host_pt_off is not None:
        hp = np.asarray(host_pt_off)
        if hp[0] != 0 or np.any(np.diff(hp) < 0) or hp[-1] > points.shape[0]:
            raise Dfu3dError("pt_off must be non-decreasing within the point tensor")
    cnt = torch.zeros(Bt, dtype=torch.int32, device=dev)
    off = torch.zeros(Bt + 1, dtype=torch.int64, device=dev)
    idx = torch.empty(cap_out, dtype=torch.int32, device=dev)
    gt = torch.empty((cap_out, 4), dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().dfu3d_gt_database(
        _chk(points, "points", torch.float32, numel=points.shape[0] * 4),
        _chk(pt_off, "pt_off", torch.int32, min_numel=2),
        _chk(box_frame, "box_frame", torch.int32, numel=Bt),
        _chk(boxes, "boxes", torch.float64, numel=Bt * 7), Bt,
        _chk(cnt, "cnt", torch.int32, numel=Bt), _chk(off, "off", torch.int64, numel=Bt + 1), int(cap_out),
        _chk(idx, "idx", torch.int32, numel=cap_out), _chk(gt, "gt", torch.float32, numel=cap_out * 4),
        _chk(status, "status", torch.int32, numel=1), _stream())
    _lib.check(rc, "dfu3d_gt_database")
    return cnt, off, idx, gt, status


def boxes_bev(boxes_a, boxes_b, iou=True):
    """(N,7),(M,7) float32 CUDA -> (N,M) float32 BEV IoU (iou=True) or overlap area."""
    n, m = int(boxes_a.shape[0]), int(boxes_b.shape[0])
    out = torch.zeros((n, m), dtype=torch.float32, device=boxes_a.device)
    if n == 0 or m == 0:
        return out
    rc = _lib.lib().dfu3d_boxes_bev(_chk(boxes_a, "boxes_a", torch.float32, numel=n * 7), n,
                                    _chk(boxes_b, "boxes_b", torch.float32, numel=m * 7), m,
                                    _chk(out, "out", torch.float32, numel=n * m), 1 if iou else 0, _stream())
    _lib.check(rc, "dfu3d_boxes_bev")
    return out


def boxes_bev_paired(boxes_a, boxes_b, iou=True):
    """(N,7),(N,7) -> (N,) overlap area / BEV IoU of row i with row i."""
    n = int(boxes_a.shape[0])
    if int(boxes_b.shape[0]) != n:
        raise Dfu3dError("paired overlap needs as many boxes on both sides")
    out = torch.zeros((n,), dtype=torch.float32, device=boxes_a.device)
    if n:
        rc = _lib.lib().dfu3d_boxes_bev_paired(_chk(boxes_a, "boxes_a", torch.float32, numel=n * 7),
                                               _chk(boxes_b, "boxes_b", torch.float32, numel=n * 7), n,
                                               _chk(out, "out", torch.float32, numel=n), 1 if iou else 0, _stream())
        _lib.check(rc, "dfu3d_boxes_bev_paired")
    return out


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """(N,5),(K,5) float32 CUDA [cx cy w h angle] -> (N,K): the AP evaluator's rotated IoU (dfu3d_rotate_iou_eval)."""
    n, k = int(boxes.shape[0]), int(query_boxes.shape[0])
    out = torch.zeros((n, k), dtype=torch.float32, device=boxes.device)
    if n and k:
        rc = _lib.lib().dfu3d_rotate_iou_eval(_chk(boxes, "boxes", torch.float32, numel=n * 5), n,
                                              _chk(query_boxes, "query_boxes", torch.float32, numel=k * 5), k,
                                              _chk(out, "out", torch.float32, numel=n * k), int(criterion), _stream())
        _lib.check(rc, "dfu3d_rotate_iou_eval")
    return out


EVAL_MAX_DET = 2048          # DFU3D_EVAL_MAX_DET: detections per frame the evaluator takes


class EvalFrames:
    """All frames of an evaluation as flat CUDA tensors (see include/dfu3d.h, "f-3, AP evaluation"): offsets int64
    (F+1), per-box float64 bbox (.,4), cam (.,7), alpha, score, truncated and int32 code, dontcare, occluded."""

    FIELDS_GT = ("gt_code", "gt_dontcare", "gt_bbox", "gt_alpha", "gt_occluded", "gt_truncated")
    FIELDS_DT = ("dt_code", "dt_bbox", "dt_alpha", "dt_score")

    def __init__(self, **t):
        self.__dict__.update(t)
        self.F = int(self.gt_off.numel()) - 1
        self.n_gt, self.n_dt = int(self.gt_code.numel()), int(self.dt_code.numel())

    def _box_args(self):
        G, D = self.n_gt, self.n_dt
        f64, i32 = torch.float64, torch.int32
        return [_chk(self.gt_code, "gt_code", i32, numel=G), _chk(self.gt_dontcare, "gt_dontcare", i32, numel=G),
                _chk(self.gt_bbox, "gt_bbox", f64, numel=4 * G), _chk(self.gt_alpha, "gt_alpha", f64, numel=G),
                _chk(self.gt_occluded, "gt_occluded", i32, numel=G), _chk(self.gt_truncated, "gt_truncated", f64, numel=G),
                _chk(self.dt_code, "dt_code", i32, numel=D), _chk(self.dt_bbox, "dt_bbox", f64, numel=4 * D),
                _chk(self.dt_alpha, "dt_alpha", f64, numel=D), _chk(self.dt_score, "dt_score", f64, numel=D)]

    def _off_args(self):
        F = self.F
        return [_chk(self.gt_off, "gt_off", torch.int64, numel=F + 1), _chk(self.dt_off, "dt_off", torch.int64, numel=F + 1),
                _chk(self.ov_off, "ov_off", torch.int64, numel=F + 1)]


def eval_overlaps(fr: EvalFrames, metric):
    """-> float64 (n_pairs): overlap of detection j with ground truth i of frame f at ov_off[f] + i * D_f + j."""
    ov = torch.zeros(max(fr.n_pairs, 1), dtype=torch.float64, device=fr.gt_off.device)
    if fr.n_pairs:
        rc = _lib.lib().dfu3d_eval_overlaps(
            int(metric), fr.F, *fr._off_args(), _chk(fr.gt_bbox, "gt_bbox", torch.float64, numel=4 * fr.n_gt),
            _chk(fr.dt_bbox, "dt_bbox", torch.float64, numel=4 * fr.n_dt),
            _chk(fr.gt_cam, "gt_cam", torch.float64, numel=7 * fr.n_gt),
            _chk(fr.dt_cam, "dt_cam", torch.float64, numel=7 * fr.n_dt), _chk(ov, "ov", torch.float64, numel=fr.n_pairs),
            fr.n_pairs, _stream())
        _lib.check(rc, "dfu3d_eval_overlaps")
    return ov


def _combo_tensor(combos, device):
    """[(class index, difficulty, min_overlap)] -> the dfu3d_eval_combo array as a CUDA byte tensor."""
    arr = (_lib.EvalCombo * len(combos))()
    for k, (c, d, mo) in enumerate(combos):
        if int(d) < 0:
            raise Dfu3dError("difficulty must be >= 0")
        arr[k].cls, arr[k].difficulty, arr[k].min_overlap = int(c), int(d), float(mo)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


def eval_match_scores(fr: EvalFrames, ov, metric, combos):
    """First pass of eval_class for every cell of `combos` -> (matched float64 (n_combo, n_gt) NaN-padded per frame,
    n_valid int32 (n_combo, F))."""
    dev = fr.gt_off.device
    C = len(combos)
    matched = torch.full((C, max(fr.n_gt, 1)), float("nan"), dtype=torch.float64, device=dev)
    n_valid = torch.zeros((C, fr.F), dtype=torch.int32, device=dev)
    cb = _combo_tensor(combos, dev)
    rc = _lib.lib().dfu3d_eval_match_scores(
        int(metric), fr.F, fr.max_dt, *fr._off_args(), _chk(ov, "ov", torch.float64, min_numel=max(fr.n_pairs, 1)),
        *fr._box_args(), _chk(cb, "combos", torch.uint8, numel=16 * C), C, fr.n_gt,
        _chk(matched, "matched", torch.float64, min_numel=C * fr.n_gt), _chk(n_valid, "n_valid", torch.int32, numel=C * fr.F),
        _stream())
    _lib.check(rc, "dfu3d_eval_match_scores")
    return matched[:, :fr.n_gt], n_valid


def eval_match_stats(fr: EvalFrames, ov, metric, combos, thresholds, n_thresh, compute_aos=False):
    """Second pass: thresholds float64 (n_combo, T), n_thresh int32 (n_combo) -> (pr int64 (n_combo, T, 3) = tp, fp, fn
    over all frames, sim float64 (n_combo, F, T) or None)."""
    dev = fr.gt_off.device
    C, T = int(thresholds.shape[0]), int(thresholds.shape[1])
    if C != len(combos) or T < 1:
        raise Dfu3dError("thresholds must be (n_combo, T >= 1)")
    pr = torch.zeros((C, T, 3), dtype=torch.int64, device=dev)
    sim = torch.zeros((C, fr.F, T), dtype=torch.float64, device=dev) if compute_aos else None
    cb = _combo_tensor(combos, dev)
    rc = _lib.lib().dfu3d_eval_match_stats(
        int(metric), fr.F, fr.max_dt, *fr._off_args(), _chk(ov, "ov", torch.float64, min_numel=max(fr.n_pairs, 1)),
        *fr._box_args(), _chk(cb, "combos", torch.uint8, numel=16 * C), C,
        _chk(thresholds, "thresholds", torch.float64, numel=C * T), _chk(n_thresh, "n_thresh", torch.int32, numel=C), T,
        1 if compute_aos else 0, _chk(pr, "pr", torch.int64, numel=C * T * 3),
        _chk(sim, "sim", torch.float64, numel=C * fr.F * T) if compute_aos else None, _stream())
    _lib.check(rc, "dfu3d_eval_match_stats")
    return pr, sim


def nms_bev(boxes, thresh, normal=False):
    """boxes (N,7) float32 CUDA sorted by descending score -> (keep int64 (N,), number kept)."""
    n = int(boxes.shape[0])
    dev = boxes.device
    keep = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    if n == 0:
        return keep[:0], 0
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    mask = torch.empty(max(n * ((n + 63) // 64), 1), dtype=torch.int64, device=dev)
    fn = _lib.lib().dfu3d_nms_normal_bev if normal else _lib.lib().dfu3d_nms_bev
    rc = fn(_chk(boxes, "boxes", torch.float32, numel=n * 7), n, ctypes.c_float(thresh),
            _chk(mask, "mask", torch.int64, min_numel=1), _chk(keep, "keep", torch.int64, min_numel=1),
            _chk(num, "num_keep", torch.int32, numel=1), _stream())
    _lib.check(rc, "dfu3d_nms_bev")
    return keep, int(num.item())


SELFTEST_SCRATCH_BYTES = 128 + 16 * (65536 + 16384)


def selftest_classify(calib_record, H, W, geom, key_axis, n, seed=1, d_lo=0.5, d_hi=120.0, device="cuda:0"):
    """Float32 tier of the bin classification against the fp64 one on n random pixels under one calibration record
    and the BinGeom `geom` -> dict(tried, undecided, wrong, kept); `wrong` must be 0 (dfu3d_selftest_classify)."""
    cal = torch.as_tensor(np.asarray(calib_record, np.float32).reshape(CALIB_FLOATS)).to(device).contiguous()
    scratch = torch.zeros(SELFTEST_SCRATCH_BYTES // 4, dtype=torch.float32, device=device)
    out = torch.zeros(4, dtype=torch.int64, device=device)
    rc = _lib.lib().dfu3d_selftest_classify(_chk(cal, "calib", torch.float32, numel=CALIB_FLOATS), int(H), int(W),
                                            ctypes.byref(geom), int(key_axis), int(n), int(seed), float(d_lo),
                                            float(d_hi), _chk(scratch, "scratch", torch.float32),
                                            _chk(out, "out", torch.int64, numel=4), _stream())
    _lib.check(rc, "dfu3d_selftest_classify")
    t, u, w, k = out.cpu().tolist()
    return {"tried": t, "undecided": u, "wrong": w, "kept": k}


def selftest_backproject(calib_record, H, W, n, seed=1, d_lo=0.5, d_hi=120.0, device="cuda:0"):
    """-> max |float32 back-projection - fp64| / bound over n random pixels under one 48-float calibration record."""
    cal = torch.as_tensor(np.asarray(calib_record, np.float32).reshape(CALIB_FLOATS)).to(device).contiguous()
    scratch = torch.zeros(32, dtype=torch.float32, device=device)
    out = torch.zeros(1, dtype=torch.float64, device=device)
    rc = _lib.lib().dfu3d_selftest_backproject(_chk(cal, "calib", torch.float32, numel=CALIB_FLOATS), int(H), int(W),
                                               int(n), int(seed), float(d_lo), float(d_hi),
                                               _chk(scratch, "scratch", torch.float32, numel=32),
                                               _chk(out, "out", torch.float64, numel=1), _stream())
    _lib.check(rc, "dfu3d_selftest_backproject")
    return float(out.item())


def range_cluster(px, py, seg_base, seg_cnt, S, R0, Rd, label, pool_cap, sx=None, sy=None, si=None):
    dev = px.device
    sx = sx if sx is not None else torch.empty(pool_cap, dtype=torch.float64, device=dev)
    sy = sy if sy is not None else torch.empty(pool_cap, dtype=torch.float64, device=dev)
    si = si if si is not None else torch.empty(3 * pool_cap, dtype=torch.int32, device=dev)
    rc = _lib.lib().dfu3d_range_cluster(
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(seg_base, "seg_base", torch.int64, numel=S),
        _chk(seg_cnt, "seg_cnt", torch.int32, numel=S), S, float(R0), float(Rd),
        _chk(label, "label", torch.int32, numel=pool_cap),
        _chk(sx, "sx", torch.float64, numel=pool_cap), _chk(sy, "sy", torch.float64, numel=pool_cap),
        _chk(si, "si", torch.int32, numel=3 * pool_cap), pool_cap, _stream())
    _lib.check(rc, "dfu3d_range_cluster")


def lshape_fit(px, py, pz, label, seg_base, seg_cnt, S, max_inst, calib, inst_class,
               inst_is_car, inst_box, inst_score, n_theta, dtheta, car_aspect_max, sx, sy,
               sroot, cap_rows, rows, n_rows, status, pool_cap, fit_ws=None):
    nws = int(_lib.lib().dfu3d_lshape_fit_ws_doubles(pool_cap, cap_rows))
    if fit_ws is None:
        fit_ws = torch.empty(nws, dtype=torch.float64, device=px.device)
    if S % max_inst:
        raise Dfu3dError("S must be V*max_inst")
    V = S // max_inst
    rc = _lib.lib().dfu3d_lshape_fit(
        _chk(px, "px", torch.float64, numel=pool_cap), _chk(py, "py", torch.float64, numel=pool_cap),
        _chk(pz, "pz", torch.float64, numel=pool_cap),
        _chk(label, "label", torch.int32, numel=pool_cap),
        _chk(seg_base, "seg_base", torch.int64, numel=S),
        _chk(seg_cnt, "seg_cnt", torch.int32, numel=S), S, max_inst,
        _chk(calib, "calib", torch.float32, numel=V * CALIB_FLOATS),
        _chk(inst_class, "inst_class", torch.int32, numel=S),
        _chk(inst_is_car, "inst_is_car", torch.int32, numel=S),
        _chk(inst_box, "inst_box", torch.float32, numel=S * 4),
        _chk(inst_score, "inst_score", torch.float32, numel=S), int(n_theta), float(dtheta),
        float(car_aspect_max), _chk(sx, "sx", torch.float64, numel=pool_cap),
        _chk(sy, "sy", torch.float64, numel=pool_cap),
        _chk(sroot, "sroot", torch.int32, numel=pool_cap), cap_rows,
        _chk(rows, "rows", torch.float64, numel=cap_rows * ROW_DOUBLES),
        _chk(n_rows, "n_rows", torch.int32, numel=1),
        _chk(status, "status", torch.int32, min_numel=1),
        _chk(fit_ws, "fit_ws", torch.float64, min_numel=nws), pool_cap, _stream())
    _lib.check(rc, "dfu3d_lshape_fit")


def status_message(word):
    return "; ".join(t for b, t in STATUS_TEXT.items() if word & b) or "ok"
