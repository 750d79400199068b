"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_to_json.py <fetch_dir> <write_dir> <views_per_launch> <out.json> "<command>" [passes]
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KB; on gfx950
FETCH_SIZE tallies 64 B per 128 B request of a wide coalesced read, so it is doubled."""
import csv, glob, json, sys, collections


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "(anonymous namespace)::k_" not in name or r["Counter_Name"] != counter:
                continue
            acc[name.split("::")[1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fd, wd, views, out, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
passes = int(sys.argv[6]) if len(sys.argv) > 6 else 2        # --steps 1 --warmup 1
F, Wr = means(fd, "FETCH_SIZE"), means(wd, "WRITE_SIZE")
kern = {}
for k in sorted(set(F) | set(Wr)):
    f = F.get(k, (None, 0))
    w = Wr.get(k, (None, 0))
    kern[k] = {"fetch_bytes": None if f[0] is None else int(2 * f[0] * 1024),
               "write_bytes": None if w[0] is None else int(w[0] * 1024),
               "FETCH_SIZE_KB_raw": f[0], "WRITE_SIZE_KB_raw": w[0], "launches": max(f[1], w[1]),
               "launches_per_pass": max(1, round(max(f[1], w[1]) / passes))}
import hashlib, os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in sorted(glob.glob(_root + "/dfu3d_amd/csrc/*.hip") + glob.glob(_root + "/dfu3d_amd/csrc/*.hpp") + [_root + "/include/dfu3d.h"]):
    _h.update(open(_f, "rb").read())
json.dump({"command": cmd, "views_per_launch": views, "sources_sha16": _h.hexdigest()[:16],
           "units": "bytes per launch (mean over the launches of the run)",
           "correction": "FETCH_SIZE is in KB and, on gfx950, counts 64 B per 128 B request for wide coalesced "
                         "streaming reads: fetch_bytes = 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md, HBM section; exact for "
                         "the float4 depth stream of k_bp_bin; uncalibrated for the gather-heavy kernels). "
                         "WRITE_SIZE*1024 as is (includes L2 atomics).",
           "kernels": kern}, open(out, "w"), indent=1)
print("wrote", out, "kernels", len(kern))
