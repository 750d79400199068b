"""Average a PMC counter per kernel from rocprofv3 --pmc output (counter_collection.csv)."""
import csv, sys, glob, json, collections
d = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            name = r.get("Kernel_Name", "")
            if "(anonymous namespace)::k_" not in name:
                continue
            k = name.split("::")[1].split("(")[0]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
