"""Dev tool: idle time between the kernels of a single-stream run, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python3 bench.py --single-stream --no-cpu-baseline --no-kernel-timing --rf-big-frames 0 --steps 5 --warmup 2
    python tools/gap_report.py /tmp/kt/<...>_kernel_trace.csv

Prints, for the passes of the timed region: the busy time (sum of kernel durations), the idle time between the end of a kernel and the
start of the next one, and the gaps grouped by the kernel that FOLLOWS them (what waits how long to start)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])


def short(n):
    m = re.search(r"(k_[A-Za-z0-9_]+)", n)
    return m.group(1) if m else n.split("(")[0][-40:]


starts = [i for i, e in enumerate(ev) if "k_fov_filter" in e[2]]
if len(starts) < 4:
    sys.exit("fewer than four passes in the trace")
lo, hi = starts[len(starts) // 2], starts[-1]            # the later half of the passes (warm)
seg = ev[lo:hi]
n_pass = sum(1 for e in seg if "k_fov_filter" in e[2])
busy = sum(e[1] - e[0] for e in seg)
span = seg[-1][1] - seg[0][0]
gaps = defaultdict(lambda: [0, 0])
prev_end = seg[0][1]
for s, e, n in seg[1:]:
    g = max(s - prev_end, 0)
    gaps[short(n)][0] += g
    gaps[short(n)][1] += 1
    prev_end = max(prev_end, e)
print("passes %d: span %.1f us / pass, kernels busy %.1f us / pass, idle %.1f us / pass (%d launches / pass)"
      % (n_pass, span / n_pass / 1e3, busy / n_pass / 1e3, (span - busy) / n_pass / 1e3, len(seg) // n_pass))
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  before %-34s %8.1f us / pass  (%5.1f us each, %d / pass)" % (k, t / n_pass / 1e3, t / c / 1e3, c // n_pass))
