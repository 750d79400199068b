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
    assert L.dfu3d_version() == 150 == _lib.header_version()
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


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of dfu3d_bin_geom and dfu3d_chain_cfg as a C compiler sees include/dfu3d.h == the ctypes
    mirrors in dfu3d_amd/_lib.py (a silent drift would corrupt every dfu3d_pseudo_boxes call)."""
    import ctypes
    import os
    import subprocess
    from dfu3d_amd import _build, _lib
    fields_geom = [f[0] for f in _lib.BinGeom._fields_]
    fields_cfg = [f[0] for f in _lib.ChainCfg._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "dfu3d.h"', 'int main(void) {',
           'printf("%zu\\n", sizeof(dfu3d_bin_geom));', 'printf("%zu\\n", sizeof(dfu3d_chain_cfg));']
    src += ['printf("%%zu\\n", offsetof(dfu3d_bin_geom, %s));' % f for f in fields_geom]
    src += ['printf("%%zu\\n", offsetof(dfu3d_chain_cfg, %s));' % f for f in fields_cfg]
    src += ['return 0; }']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", _build.INCLUDE, str(c), "-o", exe])
    vals = [int(v) for v in subprocess.check_output([exe]).decode().split()]
    assert vals[0] == ctypes.sizeof(_lib.BinGeom) and vals[1] == ctypes.sizeof(_lib.ChainCfg)
    k = 2
    for f in fields_geom:
        assert vals[k] == getattr(_lib.BinGeom, f).offset, f
        k += 1
    for f in fields_cfg:
        assert vals[k] == getattr(_lib.ChainCfg, f).offset, f
        k += 1


def test_sizes_struct_layout_and_workspace_bytes(tmp_path):
    """dfu3d_sizes mirrors the header; dfu3d_workspace_bytes answers for every stage and its PSEUDO_BOXES figure is
    the chain workspace of the same configuration."""
    import subprocess
    from dfu3d_amd import _build
    fields = [f[0] for f in _lib.Sizes._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "dfu3d.h"', 'int main(void) {',
           'printf("%zu\\n", sizeof(dfu3d_sizes));']
    src += ['printf("%%zu\\n", offsetof(dfu3d_sizes, %s));' % f for f in fields] + ['return 0; }']
    c = tmp_path / "sz.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "sz")
    subprocess.check_call(["gcc", "-I", _build.INCLUDE, str(c), "-o", exe])
    vals = [int(v) for v in subprocess.check_output([exe]).decode().split()]
    assert vals[0] == ctypes.sizeof(_lib.Sizes)
    for k, f in enumerate(fields):
        assert vals[1 + k] == getattr(_lib.Sizes, f).offset, f
    L = _lib.lib()
    z = _lib.Sizes()
    z.V, z.H, z.W, z.max_inst, z.cap_n, z.cap_vox, z.cap_rows, z.max_points_per_voxel = 96, 900, 1600, 8, 34720, 1 << 18, 6144, 100
    z.pool_cap, z.table_entries, z.dense, z.stat_filter = 96 << 17, 824 * 1573, 1, 0
    got = [L.dfu3d_workspace_bytes(s, ctypes.byref(z)) for s in range(12)]
    assert got[0] == 0 and all(g > 0 for i, g in enumerate(got) if i != 0)
    assert all(g % 256 == 0 for g in got)
    assert got[5] >= 16 * z.pool_cap + z.pool_cap + 4 * z.pool_cap            # shadow (+ boxes) + flags + queue
    assert got[11] == L.dfu3d_voxel_down_sample_scratch_bytes(z.pool_cap) == 80 * z.pool_cap      # DFU3D_STAGE_VOXEL_DOWN_SAMPLE
    assert L.dfu3d_workspace_bytes(12, ctypes.byref(z)) == -1 and L.dfu3d_workspace_bytes(5, None) == -1
    # == dfu3d_chain_workspace_bytes of the same configuration
    c = _lib.ChainCfg()
    c.V, c.H, c.W, c.max_inst, c.cap_n, c.cap_vox, c.cap_rows = 96, 900, 1600, 8, 34720, 1 << 18, 6144
    c.dense, c.pool_cap, c.bounds_h, c.bounds_w, c.n_theta = 1, 96 << 17, 900, 1600, 89
    c.geom.max_points_per_voxel, c.geom.t_n, c.geom.p_n = 100, 824, 1573
    assert got[10] == L.dfu3d_chain_workspace_bytes(ctypes.byref(c))
    assert got[10] > got[3]                               # the chain holds the back-projection scratch and more
