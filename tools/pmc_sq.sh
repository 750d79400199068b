#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass of SQ counters over a 1-step bench, summed per kernel.
# usage: bash tools/pmc_sq.sh r02 [extra bench args]   -> gpurun_out/<tag>_pmc_sq.json
set -e
R=${1:-r02}
shift || true
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/psq
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/psq -o psq -- python3 $ROOT/bench.py --single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --rf-big-frames 0 "$@" > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_sq.err
cd $ROOT
python3 - /tmp/psq gpurun_out/${R}_pmc_sq.json <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "(anonymous namespace)::k_" not in name:
            continue
        acc[name.split("::")[1].split("(")[0].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["launches"] = max(len(v) for v in c.values())
    w = m.get("SQ_WAVES", 0) or 1
    m["valu_per_wave"] = m.get("SQ_INSTS_VALU", 0) / w
    m["lds_per_wave"] = m.get("SQ_INSTS_LDS", 0) / w
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    m["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / wc
    m["active_valu_frac"] = m.get("SQ_ACTIVE_INST_VALU", 0) / wc
    out[k] = m
import hashlib
h = hashlib.sha256()
for f in sorted(glob.glob("dfu3d_amd/csrc/*.hip") + glob.glob("dfu3d_amd/csrc/*.hpp") + ["include/dfu3d.h"]):
    h.update(open(f, "rb").read())
out["_meta"] = {"sources_sha16": h.hexdigest()[:16]}        # bench.py: are these counters of the build it is timing?
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], len(out))
PY
