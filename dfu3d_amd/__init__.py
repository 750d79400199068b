"""dfu3d_amd -- MI355X-native pseudo-box generation for DFU3D.

The geometry lives in libdfu3d_hip.so (include/dfu3d.h); this package is the host side: the ctypes binding
(`_lib`, `stages`), the batched engine (`engine`), the directory pipeline (`pipeline`, `reader_pool`) and the
mirrors of the reference's Python surface (`penet/`, `pcdet_kitti/`).
"""
__version__ = "0.1.0"
