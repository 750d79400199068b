python -m pytest tests/test_gpu_stages.py tests/test_gpu_engine.py -x -q -m gpu -k "backproject or selftest or classify or engine or timed or bin" 2>&1 | tail -3
bash tools/kstats.sh r3bin --single-stream > gpurun_out/r3bin.log 2>&1
python3 -c "
import csv
for r in csv.reader(open('gpurun_out/r3bin_kstats.csv')):
    if 'k_bp_' in r[0]: print(r[0].split('::')[1][:14], r[1], r[3])
"
tail -c 600 gpurun_out/r3bin_kstats_bench.json
