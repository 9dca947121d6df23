#!/bin/bash
# kernel trace of the stress-dimension training probe (two embedding steps, two supervisor steps, two joint iterations)
export TMPDIR=/tmp
out=$PWD/gpurun_out/r03_joint_c4
mkdir -p "$out"
rm -rf /tmp/prof_joint_c4
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_joint_c4 -o j -- python3 scratch/joint_c4_probe.py ${1:-2048} > "$out/probe.log" 2>&1 || { tail -5 "$out/probe.log"; exit 1; }
cp /tmp/prof_joint_c4/*kernel_stats.csv "$out/joint_c4_kernel_stats.csv"
tail -6 "$out/probe.log"
python3 scratch/stats_table.py "$out/joint_c4_kernel_stats.csv" 30
