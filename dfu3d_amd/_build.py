"""Build libdfu3d_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m dfu3d_amd._build` or __graft_entry__.build().  hipcc cross-compiles
without a GPU; the built .so is git-ignored but travels with the repo snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(CSRC, "libdfu3d_hip.so")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # numerics contract: no implicit FMA contraction; IEEE fp32 div/sqrt
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    return _stale(OUT)


def _deps():
    return (sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc"))
            + [os.path.join(INCLUDE, "dfu3d.h")])


def _stale(out):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in _deps())


def _compile(out, extra, verbose):
    """hipcc into a temporary file next to `out`, then an atomic rename, all under an exclusive lock: every rank of
    a multi-rank launch may find the library stale at the same moment (fresh checkout, copy without time stamps);
    one of them compiles, the others wait on the lock and find the finished file -- nobody ever dlopens a
    half-written code object."""
    import fcntl
    import tempfile
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not _stale(out) and not extra.get("force"):
                return out
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            fd, tmp = tempfile.mkstemp(prefix=os.path.basename(out) + ".", suffix=".tmp", dir=os.path.dirname(out))
            os.close(fd)
            cmd = [hipcc] + FLAGS + extra.get("flags", []) + ["-I", INCLUDE, "-I", CSRC] + sources() + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, out)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    return _compile(OUT, {"force": force, "flags": os.environ.get("DFU3D_EXTRA_HIPCC_FLAGS", "").split()}, verbose)


# Variant builds for tests: the same sources with a debug switch, next to the product library.
VARIANTS = {
    # packed min-(key|pixel) word with only 14 key bits: keys collide below the cut all the time, so the exact
    # repair of k_bp_vox / k_ovf_* / k_bp_fix (practically never taken in the product build) does the work
    "keybits14": ["-DDFU3D_DBG_COMBO_KEYBITS=14"],
    # no middle tier: what float32 leaves undecided goes to k_bp_bin_amb (pixels) and is parked for k_bp_vox_amb (first
    # pixels of voxels) -- the paths the product build takes for a handful of pixels per launch
    "no_mid": ["-DDFU3D_DBG_NO_MID"],
    # timing experiments (dev; wrong results): selected at load time with DFU3D_LIB_VARIANT=<name>
    "p1_noflush": ["-DDFU3D_DBG_P1_NO_FLUSH"],
    "p1_nocommit": ["-DDFU3D_DBG_P1_NO_COMMIT"],
    "grid_timing": ["-DDFU3D_DBG_GRID_TIMING"],
    "grid_split": ["-DDFU3D_GRID_SPLIT_LAUNCH"],
    # radius filter: cycles per phase of k_rf_stream / k_rf_pair / k_rf_resolve (tools/rf_timing.py)
    "rf_timing": ["-DDFU3D_DBG_RF_TIMING"],
    # radius filter, phase A: the streaming part alone / without its flag stores (wrong results: what is the ceiling?)
    "rf_notail": ["-DDFU3D_DBG_RF_NOTAIL"],
    "rf_notail_noflags": ["-DDFU3D_DBG_RF_NOTAIL", "-DDFU3D_DBG_RF_NOFLAGS"],
    # radius filter, tuning builds (correct results): the pairing of listed points inside k_rf_stream instead of k_rf_pair,
    # list neighbours lane^1 / lane^2 or across the whole wave, grids, occupancy of phase B, ranges in flight
    "rf_fused": ["-DDFU3D_RF_FUSED"],
    "rf_forward": ["-DDFU3D_RF_REVERSE=0"],
    # ball query: two / four consecutive query tiles per workgroup (one table build for them when they share the instance)
    "ball_tpw2": ["-DDFU3D_BALL_TPW=2"], "ball_tpw4": ["-DDFU3D_BALL_TPW=4"],
    # ball query: one record per workgroup -- start, duration, the instance's sizes, longest chain walk (tools/ball_timing.py)
    "ball_timing": ["-DDFU3D_DBG_BALL_TIMING"],
    "rf_quad": ["-DDFU3D_RF_NBR=0"],
    "rf_wave": ["-DDFU3D_RF_NBR=2"],
    "rf_g2048": ["-DDFU3D_RF_GRID=2048"],
    "rf_g4096": ["-DDFU3D_RF_GRID=4096"],
    "rfb_occ8": ["-DDFU3D_RFB_OCC=8"],
    "rfb_occ6": ["-DDFU3D_RFB_OCC=6"],
    "rf_ul4": ["-DDFU3D_RF_UL=4"],
    # k_bp_vox: which of its accesses cost what (wrong results; tools/vox_traffic.sh)
    "vox_skip1": ["-DDFU3D_DBG_VOX_SKIP=1"], "vox_skip2": ["-DDFU3D_DBG_VOX_SKIP=2"],
    "vox_skip8": ["-DDFU3D_DBG_VOX_SKIP=8"], "vox_skip11": ["-DDFU3D_DBG_VOX_SKIP=11"],
    # k_bp_vox: two / one workgroup per compute unit instead of three (correct results)
    "vox_2wg": ["-DDFU3D_VOX_LDS_PAD=40960"], "vox_1wg": ["-DDFU3D_VOX_LDS_PAD=65536"],
    "vox_occ6": ["-DDFU3D_VOX_OCC=6"], "vox_occ8": ["-DDFU3D_VOX_OCC=8"],
    # k_bp_bin: register budgets for 6 / 8 workgroups per compute unit (the short LDS list of undecided pixels makes room)
    "p1_occ6": ["-DDFU3D_P1_OCC=6"], "p1_occ8": ["-DDFU3D_P1_OCC=8"],
}


def variant_path(name):
    return os.path.join(CSRC, "libdfu3d_hip_%s.so" % name)


def build_variant(name, force=False, verbose=False):
    out = variant_path(name)
    if not force and not _stale(out):
        return out
    return _compile(out, {"force": force, "flags": VARIANTS[name]}, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
