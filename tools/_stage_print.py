import json,sys
j=json.load(open(sys.argv[1]))
print(j["value"], j["ms_per_step"], {r["stage"]: round(r.get("ms_per_pass", r["avg_ms"]),4) for r in j["kernels"]})
