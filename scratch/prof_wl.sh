#!/bin/bash
# kernel stats of the bench command for one workload: ./scratch/prof_wl.sh c3 [steps]
export TMPDIR=/tmp
wl=${1:-c3}; steps=${2:-20}
rm -rf /tmp/prof_$wl
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o $wl -- \
    python3 bench.py --only --workload $wl --steps $steps --warmup 2 --no-cpu > gpurun_out/prof_${wl}.json 2> gpurun_out/prof_${wl}.err || { tail -3 gpurun_out/prof_${wl}.err; exit 1; }
cp /tmp/prof_$wl/*kernel_stats.csv gpurun_out/${wl}_kernel_stats.csv
head -12 gpurun_out/${wl}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
