"""Dev tool (no GPU needed): static instruction mix and resource use of the kernels of one source file, from the gfx950
assembly hipcc emits with the product's flags.

    python tools/isa_mix.py dfu3d_amd/csrc/pixel_stage.hip [kernel-name-substring ...]

Per kernel: vector / scalar / LDS / global-memory instruction counts of the whole body (static: both sides of every branch
count), the fp64 and transcendental share of the vector instructions, VGPRs, SGPRs, LDS bytes, scratch bytes and the waves
per SIMD the VGPR count allows (512 VGPRs per SIMD lane).  Used to check a change before it is sent to the GPU box:
scratch must stay 0, the VGPR count decides the occupancy."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dfu3d_amd import _build  # noqa: E402


def assembly(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-I", _build.INCLUDE, "-I", _build.CSRC, "-S",
                                                                    "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text):
    """-> {mangled name: (body, metadata block)}"""
    found = {}
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)", text, flags=re.M):
        name = m.group(1)
        start = text.find("\n%s:" % name)
        if start < 0:
            continue
        meta = re.search(r"\.name:\s+%s\n.*?\.wavefront_size" % re.escape(name), text, flags=re.S)
        pre = text.rfind("- .agpr_count", 0, meta.start()) if meta else -1
        block = text[pre:meta.end()] if meta and pre >= 0 else ""
        found[name] = (text[start:m.start()], block)
    return found


def demangle(name):
    """Kernel name out of an Itanium-mangled symbol of the anonymous namespace (enough for this code base)."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    at = m.end()
    tail = name[at + n:]
    targs = re.match(r"I(.*?)EE", tail)
    return name[at:at + n] + ("<" + targs.group(1) + ">" if targs else "")


def field(block, key):
    m = re.search(r"\.%s:\s+(\d+)" % key, block)
    return int(m.group(1)) if m else -1


def main():
    src = sys.argv[1]
    want = sys.argv[2:]
    text = assembly(src)
    print("%-44s %6s %6s %6s %5s %5s %5s %5s %5s %6s %7s %6s" % ("kernel", "VALU", "f64", "trans", "SALU", "LDS", "vmem", "VGPR",
                                                                "SGPR", "waves", "LDS B", "scratch"))
    for name, (body, block) in sorted(kernels(text).items()):
        short = re.sub(r"\(.*", "", demangle(name)).replace("(anonymous namespace)::", "").replace("void ", "")
        if want and not any(w in short for w in want):
            continue
        ins = re.findall(r"^\s+([a-z][a-z0-9_]+)", body, flags=re.M)
        valu = [i for i in ins if i.startswith("v_")]
        f64 = [i for i in valu if "_f64" in i]
        trans = [i for i in valu if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", i)]
        vg = field(block, "vgpr_count")
        print("%-44s %6d %6d %6d %5d %5d %5d %5d %5d %6d %7d %6d" % (
            short[:44], len(valu), len(f64), len(trans), sum(i.startswith("s_") for i in ins),
            sum(i.startswith("ds_") for i in ins), sum(i.startswith(("global_", "buffer_", "flat_", "scratch_")) for i in ins),
            vg, field(block, "sgpr_count"), min(8, 512 // max(vg, 1)) if vg > 0 else -1,
            field(block, "group_segment_fixed_size"), field(block, "private_segment_fixed_size")))


if __name__ == "__main__":
    main()
