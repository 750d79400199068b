"""Drop-in mirror of the reference's pseudo-label generator surface
(tools/PENet/): same module names, function names and argument meaning, with
the geometry running in libdfu3d_hip.so on the GPU."""
