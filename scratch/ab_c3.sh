#!/bin/bash
# A/B of library builds on the C3 stack: ./scratch/ab_c3.sh <suffix> ... (time-gan_amd/libtgru_v_<suffix>.so)
for v in "$@"; do
  TGRU_LIB_PATH=$PWD/time-gan_amd/libtgru_v_$v.so timeout -k 10 100 python bench.py --only --workload c3 --no-cpu --steps 100 --warmup 10 > gpurun_out/v_$v.json 2> gpurun_out/v_$v.err || { echo "$v failed"; tail -3 gpurun_out/v_$v.err; exit 1; }
  echo -n "$v: "; python scratch/show.py gpurun_out/v_$v.json
done
