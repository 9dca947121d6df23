#!/bin/bash
# one PMC pass over one C4 step: matrix-pipe busy cycles and GPU-active cycles per kernel
export TMPDIR=/tmp
rm -rf /tmp/pmc
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -o c4 -- python3 bench.py --workload c4 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_c4.log 2>&1 || { tail -5 gpurun_out/pmc_c4.log; exit 1; }
ls /tmp/pmc/* | head
python3 - <<'PY'
import csv, glob, collections
f = [p for p in glob.glob('/tmp/pmc/**/*counter_collection.csv', recursive=True)][0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[k] += 1
out = open('gpurun_out/pmc_c4_summary.csv', 'w')
out.write('kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE\n')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0))[:12]:
    line = '"%s",%d,%.0f,%.0f' % (k, cnt[k], v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), v.get('GRBM_GUI_ACTIVE', 0))
    print(line); out.write(line + '\n')
PY
