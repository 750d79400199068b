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
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = (sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc"))
            + [os.path.join(INCLUDE, "dfu3d.h")])
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DFU3D_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + ["-I", INCLUDE, "-I", CSRC] + sources() + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


# Variant builds for tests: the same sources with a debug switch, next to the product library.
VARIANTS = {
    # packed min-(key|pixel) word with only 14 key bits: keys collide below the cut all the time, so the exact
    # repair of k_bp_vox / k_ovf_* / k_bp_fix (practically never taken in the product build) does the work
    "keybits14": ["-DDFU3D_DBG_COMBO_KEYBITS=14"],
    # timing experiments (dev; wrong results): selected at load time with DFU3D_LIB_VARIANT=<name>
    "p1_noflush": ["-DDFU3D_DBG_P1_NO_FLUSH"],
    "p1_nocommit": ["-DDFU3D_DBG_P1_NO_COMMIT"],
    "grid_timing": ["-DDFU3D_DBG_GRID_TIMING"],
    "grid_split": ["-DDFU3D_GRID_SPLIT_LAUNCH"],
    # radius filter, phase B: the points of two candidate ranges requested together (measurement for DESIGN §10 item 1)
    "rf2": ["-DDFU3D_RF_RANGES_PER_STEP=2"],
}


def variant_path(name):
    return os.path.join(CSRC, "libdfu3d_hip_%s.so" % name)


def build_variant(name, force=False, verbose=False):
    out = variant_path(name)
    deps = (sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc"))
            + [os.path.join(INCLUDE, "dfu3d.h")])
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + VARIANTS[name] + ["-I", INCLUDE, "-I", CSRC] + sources() + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
