#!/bin/bash
# kernel stats of graph-mode joint iterations at the energy-like config (28 features, H 96, L 3, T 24, batch 1024)
export TMPDIR=/tmp
rm -rf /tmp/jc3
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jc3 -o jc3 -- python3 scratch/joint_c3_probe.py > gpurun_out/joint_c3_trace.log 2>&1 || { tail -5 gpurun_out/joint_c3_trace.log; exit 1; }
cp /tmp/jc3/*kernel_stats.csv gpurun_out/joint_c3_kernel_stats.csv
tail -2 gpurun_out/joint_c3_trace.log
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/joint_c3_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot / 1e6, 'kernels', sum(int(r['Calls']) for r in rows))
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
