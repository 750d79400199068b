"""Print the longest durations of one kernel from a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
name = sys.argv[1]
f = sys.argv[2] if len(sys.argv) > 2 else sorted(glob.glob("/tmp/pp/**/*kernel_trace.csv", recursive=True))[-1]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
print(name, "calls", len(d), "longest (us):", [round(v, 1) for v in sorted(d)[-12:]])
