#!/bin/bash
# A/B of one TGRU_* switch on the default bench (all workloads + training probes, no CPU legs):
#   ./scratch/cmp_switch.sh TGRU_NO_FUSE
sw=$1
for v in on off; do
  if [ $v = off ]; then export $sw=1; fi
  timeout -k 10 400 python bench.py --no-cpu --steps 5 --warmup 2 > gpurun_out/cmp_$v.json 2> gpurun_out/cmp_$v.err || { echo "$v failed"; tail -3 gpurun_out/cmp_$v.err; exit 1; }
  python scratch/cmp_show.py gpurun_out/cmp_$v.json "$sw path $v"
done
