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


def build(force: bool = False) -> str:
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)):
        return OUT
    cmd = ["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math",
           "-shared", "-fPIC"] + SRCS + ["-o", OUT, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
