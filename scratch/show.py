"""Prints the headline figures of a bench.py JSON line file."""
import json, sys
for path in sys.argv[1:]:
    d = json.load(open(path))
    p = d["roofline"]["parts"]
    print(f"{path}: step {d['ms_per_step']:.3f} ms  frac {d['roofline']['frac']:.4f}  forward {p['forward']['device_ms']:.3f}  "
          f"backward {p['backward']['device_ms']:.3f}  launches {d['launches_per_step']}")
