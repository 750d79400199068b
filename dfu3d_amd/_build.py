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


# Variant builds: the same sources with ONE switch of csrc/dbg.hpp, next to the product library.  All of them give the
# product's results.  They are loaded only through _lib.load_variant() -- by tests/ and tools/, never by the package.
VARIANTS = {
    # packed min-(key|pixel) word with only 14 key bits: keys collide below the cut all the time, so the exact
    # repair of k_bp_vox / k_ovf_* / k_bp_fix (practically never taken in the product build) does the work
    "keybits14": ["-DDFU3D_DBG_COMBO_KEYBITS=14"],
    # no middle tier: what float32 leaves undecided goes to k_bp_bin_amb -- the path the product build takes for a handful
    # of pixels per launch
    "no_mid": ["-DDFU3D_DBG_NO_MID"],
    # cycles between the DBG_T marks of a kernel, summed over its workgroups (tools/p1_timing.py)
    "timing": ["-DDFU3D_DBG_TIMING"],
    # the HOST side of the C ABI (argument validation, geometry, workspace carve-up) under AddressSanitizer +
    # UndefinedBehaviorSanitizer; device code as in the product.  CPU only (tools/sanitize_cpu.sh): GPU ASan is not available
    "asan_host": ["-O1", "-g", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
                  "-Xarch_host", "-fno-sanitize-recover=undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-shared-libsan"],
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
