#!/bin/bash
# the C4 step with other builds of the library (TGRU_LIB_PATH), after the large-hidden tests on each
for lib in "$@"; do
  export TGRU_LIB_PATH=$PWD/time-gan_amd/$lib
  timeout -k 10 300 python -u -m pytest tests/test_module_gpu.py -k "large_hidden" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python bench.py --only --workload c4 --no-cpu --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['roofline']['frac'], d['roofline']['parts']['backward']['device_ms'])" || exit 1
done
