#!/bin/bash
# one PMC pass over the wide head kernels: matrix-pipe busy cycles and GPU-active cycles per kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_hw
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_hw -o hw -- python3 scratch/head_wide_probe.py > gpurun_out/pmc_head_wide.log 2>&1 || { tail -5 gpurun_out/pmc_head_wide.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json
f = [p for p in glob.glob('/tmp/pmc_hw/**/*counter_collection.csv', recursive=True)][0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('void ', '').split('(')[0][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[k] += 1
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 scratch/head_wide_probe.py (24 576 rows: 96/96/96 with both LayerNorms, 28/192/96; 4 forward + backward calls each)",
       "normalisation": "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)", "kernels": {}}
for k, v in agg.items():
    if 'tgru::' not in k: continue
    out["kernels"][k] = {"dispatches": cnt[k], "mfma_busy_fraction": round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8), 4)}
json.dump(out, open('gpurun_out/r02_pmc_head_wide.json', 'w'), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
