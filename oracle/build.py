"""Build recipe for the oracle's C part (TEST INFRASTRUCTURE ONLY).

`python -m oracle.build` compiles oracle/csrc/*.c into
oracle/libdfu3d_oracle.so with gcc.  The reference ships no native code on this
path, so there is no `oracle/_ref` build: the oracle is pinned
by golden vectors captured from the importable reference modules instead
(tests/golden/capture_goldens.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "dfu3d_oracle.c")
SRCS = [SRC]
DEPS = SRCS
OUT = os.path.join(HERE, "libdfu3d_oracle.so")
# `python -m oracle.build --sanitize`: the same source under AddressSanitizer + UndefinedBehaviorSanitizer (gcc), next to
# the ordinary build.  oracle/penet_oracle.py loads it when DFU3D_ORACLE_SANITIZE=1 (tools/sanitize_cpu.sh; the process
# needs gcc's libasan preloaded).  CPU only -- never on the GPU box.
OUT_SAN = os.path.join(HERE, "libdfu3d_oracle_san.so")


def _fresh(out):
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, sanitize: bool = False) -> str:
    out = OUT_SAN if sanitize else OUT
    if not force and _fresh(out):
        return out
    opt = (["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
           if sanitize else ["-O2"])
    cmd = ["gcc"] + opt + ["-mfma", "-ffp-contract=off", "-fno-fast-math",
                           "-shared", "-fPIC"] + SRCS + ["-o", out, "-lm"]
    subprocess.check_call(cmd)
    return out


def selected() -> str:
    """The library the oracle loads in this process."""
    return build(sanitize=os.environ.get("DFU3D_ORACLE_SANITIZE") == "1")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, sanitize="--sanitize" in sys.argv))
