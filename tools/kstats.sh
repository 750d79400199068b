#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/<tag>_kstats.csv
set -e
R=${1:-r02}
shift || true
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o ks -- python3 $ROOT/bench.py --no-cpu-baseline --rf-big-frames 0 --steps 3 --warmup 1 "$@" > $ROOT/gpurun_out/${R}_kstats_bench.json 2> $ROOT/gpurun_out/${R}_kstats_bench.err
STATS=$(ls /tmp/pk/*kernel_stats.csv /tmp/pk/*/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - "$STATS" "$ROOT/gpurun_out/${R}_kstats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if "(anonymous namespace)::k_" in r[0]]
csv.writer(open(sys.argv[2], "w")).writerows(keep)
for r in keep[1:]:
    name = r[0].split("(anonymous namespace)::")[-1].split("(")[0]
    if float(r[3]) > 15000: print("%-40s calls %4s avg_us %9.1f" % (name[:40], r[1], float(r[3]) / 1e3))
PY
