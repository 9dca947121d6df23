#!/bin/bash
# large-hidden tests, then the C4 A/B and the per-phase stamps of the backward recurrence
out=gpurun_out/${1:-res}
mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_module_gpu.py -k "large_hidden or timed_out" -x -q --timeout 120 --timeout-method thread > $out/t.log 2>&1 || { tail -30 $out/t.log; exit 1; }
tail -2 $out/t.log
bash scratch/ab_c4.sh ${1:-res} || exit 1
[ -f time-gan_amd/libtgru_stamps.so ] && timeout -k 10 120 python scratch/stamps_step.py 2>&1 | grep "bwd"; true
