#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the default bench command + the two PMC passes.
# usage: bash tools/profile_round.sh r02 [extra bench args]
set -e
R=${1:-r02}
shift || true
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pp /tmp/pf /tmp/pw
# (1) the default command (2 streams x 32-frame chunks in the timed region): launches overlap, durations are contended
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o ks -- python3 $ROOT/bench.py --no-cpu-baseline --rf-big-frames 0 "$@" > $ROOT/gpurun_out/${R}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${R}_bench_under_rocprof.err
# (2) one stream, one chunk: the durations the kernel table / roofline block of bench.py report
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o ks -- python3 $ROOT/bench.py --no-cpu-baseline --rf-big-frames 0 --single-stream "$@" > $ROOT/gpurun_out/${R}_bench_single_stream_under_rocprof.json 2> $ROOT/gpurun_out/${R}_bench_single_stream_under_rocprof.err
for pair in "/tmp/pp ${R}_kernel_stats_bench_default.csv" "/tmp/ps ${R}_kernel_stats_bench_single_stream.csv"; do
set -- $pair
STATS=$(ls $1/*kernel_stats.csv $1/*/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - "$STATS" "$ROOT/gpurun_out/$2" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if "(anonymous namespace)::k_" in r[0]]
csv.writer(open(sys.argv[2], "w")).writerows(keep)
print("kernels", len(keep) - 1)
PY
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pf -- python3 $ROOT/bench.py --single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --rf-big-frames 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pw -- python3 $ROOT/bench.py --single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --rf-big-frames 0 > /dev/null 2>&1
cd $ROOT
python3 tools/pmc_to_json.py /tmp/pf /tmp/pw 384 gpurun_out/${R}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --single-stream --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --rf-big-frames 0  (two separate passes, MI355X)"
tail -c 400 gpurun_out/${R}_bench_under_rocprof.json
