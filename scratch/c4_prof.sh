#!/bin/bash
# GRU parity tests (all kernel families), then a kernel-stats profile of two C4 bench steps and a C3 bench
set -o pipefail
export TMPDIR=/tmp
tag=${1:-c4}
timeout -k 10 400 python -u -m pytest tests/test_module_gpu.py tests/test_gru_gpu.py tests/test_fuzz_gpu.py -x -q --timeout 150 --timeout-method thread > gpurun_out/step_test.log 2>&1 || { tail -15 gpurun_out/step_test.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/step_test.log
rm -rf /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o c4 -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu > gpurun_out/prof_${tag}.log 2>&1 || { tail -5 gpurun_out/prof_${tag}.log; exit 1; }
cp /tmp/prof/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
cut -c1-160 gpurun_out/${tag}_kernel_stats.csv | head -7
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}.log
timeout -k 10 200 python -u bench.py --workload c3 --no-cpu > gpurun_out/bench_c3_${tag}.log 2>&1 || { tail -5 gpurun_out/bench_c3_${tag}.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_c3_${tag}.log
