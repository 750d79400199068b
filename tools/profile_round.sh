#!/bin/bash
# Run on the GPU box: kernel-trace stats of the default bench command + the two PMC passes.
# usage: bash tools/profile_round.sh r01
set -e
R=${1:-r01}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pp /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o ks -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/${R}_bench_under_rocprof.log 2>&1
STATS=$(ls /tmp/pp/*kernel_stats.csv /tmp/pp/*/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - "$STATS" "$ROOT/gpurun_out/${R}_kernel_stats_bench_default.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if "(anonymous namespace)::k_" in r[0]]
csv.writer(open(sys.argv[2], "w")).writerows(keep)
print("kernels", len(keep) - 1)
PY
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pf -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pw -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cd $ROOT
python3 tools/pmc_to_json.py /tmp/pf /tmp/pw 96 gpurun_out/${R}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing  (two separate passes, MI355X)"
tail -1 gpurun_out/${R}_bench_under_rocprof.log | cut -c1-300
# single-stream companion profile (clean per-kernel durations)
cd /tmp
rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o ks1 -- python3 $ROOT/bench.py --no-cpu-baseline --lanes 1 --chunk-frames 64 --timing-steps 10 > $ROOT/gpurun_out/${R}_bench_1lane_under_rocprof.log 2>&1
STATS=$(ls /tmp/p1/*kernel_stats.csv /tmp/p1/*/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - "$STATS" "$ROOT/gpurun_out/${R}_kernel_stats_bench_single_stream.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if "(anonymous namespace)::k_" in r[0]]
csv.writer(open(sys.argv[2], "w")).writerows(keep)
PY
tail -1 $ROOT/gpurun_out/${R}_bench_1lane_under_rocprof.log | cut -c1-200
