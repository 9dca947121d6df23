#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root:
#   1. rocprofv3 --kernel-trace --stats of the bench command for each workload (c2, c3, c4)
#   2. PMC passes over one C4 pass (each counter set in a run of its own):
#        matrix-pipe busy cycles + GPU-active cycles, FETCH_SIZE, WRITE_SIZE
# Results: gpurun_out/r02_prof/*  (copy what is to be kept into profiles/).
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/r02_prof
mkdir -p "$out"
for wl in c2 c3 c4; do
  steps=20; [ $wl = c4 ] && steps=3
  rm -rf /tmp/prof_$wl
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o $wl -- \
      python3 bench.py --only --workload $wl --steps $steps --warmup 2 --no-cpu > "$out/bench_${wl}_under_rocprof.json" 2> "$out/prof_$wl.err" \
      || { echo "kernel trace $wl failed"; tail -3 "$out/prof_$wl.err"; exit 1; }
  cp /tmp/prof_$wl/*kernel_stats.csv "$out/${wl}_kernel_stats.csv"
  echo "== $wl"; head -6 "$out/${wl}_kernel_stats.csv" | cut -c1-150
done
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/pmc_$name
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o c4 -- \
      python3 bench.py --only --workload c4 --steps 1 --warmup 0 --no-cpu > "$out/pmc_$name.log" 2>&1 \
      || { echo "pmc pass $name failed"; tail -3 "$out/pmc_$name.log"; exit 1; }
  cp $(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1) "$out/pmc_${name}_raw.csv"
done
python3 - "$out" <<'PY'
import collections, csv, json, sys
out = sys.argv[1]
PASSES = 3        # --steps 1 --warmup 0: the pass that counts launches, one timed pass, the pass that times the halves apart
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for name in ("mfma", "fetch", "write"):
    for r in csv.DictReader(open(f"{out}/pmc_{name}_raw.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
            cnt[k] += 1
kern = {}
tot_f = tot_w = 0.0
for k, v in agg.items():
    if not k.startswith("tgru::"):
        continue
    e = {"dispatches_per_pass": cnt[k] / PASSES}
    if v.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_fraction"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8), 4)
    # FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950 counts half the bytes of 16-byte-per-lane reads:
    # the GEMM operand loads are all of that kind, so their fetches are doubled
    f = v.get("FETCH_SIZE", 0.0) * 1024 / PASSES
    w = v.get("WRITE_SIZE", 0.0) * 1024 / PASSES
    fc = 2 * f if "gemm_kernel" in k else f
    e.update(fetch_bytes_per_pass_raw=f, fetch_bytes_per_pass_corrected=fc, write_bytes_per_pass=w)
    tot_f += fc; tot_w += w
    kern[k] = e
res = {"workload": "c4 (B2048 T256 H512 L3)",
       "command": "rocprofv3 --pmc <set> -- python3 bench.py --only --workload c4 --steps 1 --warmup 0 --no-cpu, one run per counter set "
                  "(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE); three forward+backward passes per run",
       "normalisation": "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); bytes per pass = counter KB x 1024 / 3; "
                        "FETCH_SIZE doubled for the GEMM kernels (16-byte-per-lane operand loads are counted at half their bytes on gfx950), "
                        "taken as reported for the recurrent kernels (their HBM reads are mostly 4-byte-per-lane epilogue operands; the 16-byte tile "
                        "loads are served by the L2)",
       "kernels": kern, "hbm_bytes_per_step": tot_f + tot_w,
       "fetch_bytes_per_step_corrected": tot_f, "write_bytes_per_step": tot_w}
json.dump(res, open(f"{out}/r02_pmc_c4.json", "w"), indent=1)
for k, e in sorted(kern.items(), key=lambda kv: -kv[1].get("write_bytes_per_pass", 0)):
    print(k, e)
print("HBM bytes per step", tot_f + tot_w)
PY
