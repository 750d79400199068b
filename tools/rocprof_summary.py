"""Print our kernels from a rocprofv3 *_kernel_stats.csv (names contain commas -> use csv)."""
import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.reader(open(f)))
for r in rows[1:]:
    if "(anonymous namespace)::k_" in r[0]:
        print("%-34s calls %5s  avg %10.1f us  total %10.1f us" % (r[0].split("::")[1].split("(")[0], r[1], float(r[3]) / 1e3, float(r[2]) / 1e3))
