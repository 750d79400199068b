"""CPU-only: the C-ABI library builds/loads and exports exactly what
include/dfu3d.h declares (no compute calls without a GPU)."""
import ctypes

import numpy as np

from dfu3d_amd import _lib


def test_library_exports_every_header_symbol():
    L = _lib.lib()
    declared = _lib.header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), "libdfu3d_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert set(_lib.SIGNATURES) == set(declared)


def test_version_and_strerror():
    L = _lib.lib()
    assert L.dfu3d_version() == 100
    assert L.dfu3d_strerror(0) == b"ok"
    assert b"invalid" in L.dfu3d_strerror(-1)


def test_argument_validation_without_gpu():
    """Null pointers / bad sizes are rejected on the host before any launch."""
    L = _lib.lib()
    assert L.dfu3d_fov_filter(None, None, None, None, 1, 900, 1600, 10, None, None, None) == -1
    assert L.dfu3d_range_cluster(None, None, None, None, 0, 3.0, 0.001, None, None, None, None, 8, None) == -1


def test_bin_table_geometry_covers_reachable_bins():
    from dfu3d_amd.params import Params
    p = Params()
    g = _lib.BinGeom()
    g.vsize_r, g.vsize_t, g.vsize_p = p.vsize
    g.rmin_r, g.rmin_t, g.rmin_p = p.vrange_min
    g.grid_r, g.grid_t, g.grid_p = p.vgrid
    g.max_points_per_voxel, g.max_voxels = 100, 1000000
    g.theta_min, g.z_max, g.depth_min = p.theta_min, p.z_max, p.depth_min
    n = _lib.lib().dfu3d_bin_table_geometry(ctypes.byref(g))
    assert n == g.t_n * g.p_n and 1_000_000 < n < 1_500_000
    th = np.array([np.nextafter(1.5, 2), np.pi])
    ph = np.array([-np.pi / 2, np.pi / 2])
    tb = np.floor((th + 5.0) / p.vsize[1])
    pb = np.floor((ph + 5.0) / p.vsize[2])
    assert g.t_lo <= tb.min() and tb.max() < g.t_lo + g.t_n
    assert g.p_lo <= pb.min() and pb.max() < g.p_lo + g.p_n
