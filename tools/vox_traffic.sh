#!/bin/bash
# Run on the GPU box: duration, FETCH_SIZE and WRITE_SIZE of k_bp_vox (and k_bp_bin) for the product build and for the
# builds that leave out one group of its accesses (dfu3d_amd/_build.py: vox_skip*; wrong results, first pass only is clean).
set -e
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for v in ${@:-product vox_skip1 vox_skip2 vox_skip8 vox_skip11}; do
  if [ "$v" = "product" ]; then unset DFU3D_LIB_VARIANT; else export DFU3D_LIB_VARIANT=$v; fi
  rm -rf /tmp/vt /tmp/vf /tmp/vw
  ARGS="--single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -o t -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/vf -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/vw -o w -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
def per_dispatch(d, pat, col, cond=None):
    out = {}
    for f in glob.glob(d + "/**/*" + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "::k_bp_" not in n: continue
            if cond and not cond(r): continue
            out.setdefault(n.split("::")[1].split("(")[0], []).append(col(r))
    return out
T = per_dispatch("/tmp/vt", "kernel_trace.csv", lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
F = per_dispatch("/tmp/vf", "counter_collection.csv", lambda r: 2 * float(r["Counter_Value"]) / 1024, lambda r: r["Counter_Name"] == "FETCH_SIZE")
W = per_dispatch("/tmp/vw", "counter_collection.csv", lambda r: float(r["Counter_Value"]) / 1024, lambda r: r["Counter_Name"] == "WRITE_SIZE")
for k in sorted(T):
    print("%-12s %-10s us %s  fetch_MB %s  write_MB %s" % (v, k, ["%.0f" % x for x in T[k]], ["%.0f" % x for x in F.get(k, [])], ["%.0f" % x for x in W.get(k, [])]))
PY
done
