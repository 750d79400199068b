run() {
  python bench.py --no-cpu-baseline --rf-big-frames 0 --no-kernel-timing --steps 30 $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - "$1 $2" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1]); print(sys.argv[1], "->", j["value"], j["ms_per_step"], flush=True)
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/sw.err").read()[-300:])
PY
}
for rep in 1 2; do
run 2x32 "--lanes 2 --chunk-frames 32"
run 1x64 "--single-stream"
run 4x16 "--lanes 4 --chunk-frames 16"
run 3x32-96 "--lanes 3 --chunk-frames 32 --frames 96"
run 2x64-128 "--lanes 2 --chunk-frames 64 --frames 128"
done
