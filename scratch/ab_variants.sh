#!/bin/bash
# A/B of library builds: ./scratch/ab_variants.sh <suffix> ... runs the C4 bench with time-gan_amd/libtgru_v_<suffix>.so
for v in "$@"; do
  TGRU_LIB_PATH=$PWD/time-gan_amd/libtgru_v_$v.so timeout -k 10 100 python bench.py --only --no-cpu --steps 5 --warmup 2 > gpurun_out/v_$v.json 2> gpurun_out/v_$v.err || { echo "$v failed"; tail -3 gpurun_out/v_$v.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/v_{v}.json"))
p = d["roofline"]["parts"]
print(f"== {v}: step {d['ms_per_step']:.2f} ms  forward {p['forward']['device_ms']:.2f}  backward {p['backward']['device_ms']:.2f}")
PY
done
