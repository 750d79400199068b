#!/bin/bash
# CPU sanitizer job ("ASan host build of the C-ABI layer").  BUILD CONTAINER ONLY -- never on the GPU box.
#   bash tools/sanitize_cpu.sh [log]        default log: profiles/r04_sanitize_cpu.log
# (1) the oracle's C restatement under gcc's AddressSanitizer + UndefinedBehaviorSanitizer: every CPU test that drives it
# (2) the HOST side of libdfu3d_hip.so (argument validation, bin geometry, workspace carve-up, struct layouts) under
#     clang's ASan + UBSan (build variant asan_host; device code as in the product): tests/test_abi.py
# Two processes: the two compilers' ASan runtimes cannot share one.  A finding aborts its run (halt_on_error).
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r04_sanitize_cpu.log}
: > "$LOG"
rc=0
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
echo "== (1) oracle/csrc/dfu3d_oracle.c: gcc -O1 -g -fsanitize=address,undefined" | tee -a "$LOG"
python -m oracle.build --sanitize --force >> "$LOG" 2>&1 || rc=1
LD_PRELOAD=$(gcc -print-file-name=libasan.so) DFU3D_ORACLE_SANITIZE=1 \
  python -m pytest tests/test_oracle_golden.py tests/test_oracle_voxel_down.py tests/test_oracle_gtdb.py tests/test_oracle_iou3d.py \
         tests/test_oracle_kitti_eval.py tests/test_oracle_la_sampling.py tests/test_host.py -q -x -m "not gpu" -p no:cacheprovider >> "$LOG" 2>&1 || rc=1
tail -2 "$LOG"
echo "== (2) libdfu3d_hip_asan_host.so: hipcc -Xarch_host -fsanitize=address,undefined (host side of the C ABI)" | tee -a "$LOG"
python - >> "$LOG" 2>&1 <<'PY' || rc=1
from dfu3d_amd import _build
print(_build.build_variant("asan_host", force=True))
PY
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | tail -1)
LD_PRELOAD=$RT python tools/sanitize_abi.py >> "$LOG" 2>&1 || rc=1
tail -2 "$LOG"
echo "sanitize_cpu: rc=$rc" | tee -a "$LOG"
exit $rc
