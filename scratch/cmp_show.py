import json, sys
d = json.load(open(sys.argv[1]))
w = d.get("workloads", {})
line = [sys.argv[2]]
for k in ("c2", "c3", "c4"):
    if k in w:
        line.append(f"{k} {w[k]['ms_per_step']:.4f} ms")
for k, v in d.items():
    if isinstance(v, dict) and ("training" in k or "probe" in k):
        line.append(f"{k}: " + json.dumps({a: b for a, b in v.items() if "it_s" in a or "per_s" in a}))
print(" | ".join(line))
