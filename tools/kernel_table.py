"""Pretty-print the per-kernel table of a bench.py JSON line (stdin)."""
import json, sys
d = json.loads(sys.stdin.read())
k = d["kernels"]
print("ms/step", d["ms_per_step"], "value", d["value"])
tot = 0.0
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("%-28s %7.3f ms" % (n, v["ms_per_step"]), {a: b for a, b in v.items() if a != "ms_per_step"})
    tot += v["ms_per_step"]
print("sum of kernels %.2f ms" % tot)
print(d["roofline"])
