#!/bin/bash
# C4 stack step with the resident-W_hh backward recurrence and with the streamed one
set -e
out=gpurun_out/${1:-ab}
mkdir -p $out
timeout -k 10 200 python bench.py --only --workload c4 --no-cpu --steps 10 --warmup 3 > $out/c4_res.json 2> $out/c4_res.err
TGRU_NO_STEP_RES=1 timeout -k 10 200 python bench.py --only --workload c4 --no-cpu --steps 10 --warmup 3 > $out/c4_sq.json 2> $out/c4_sq.err
python - <<PY
import json
for n in ("res", "sq"):
    d = json.loads(open("$out/c4_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["parts"])
PY
