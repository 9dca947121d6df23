#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root:
#   1. rocprofv3 --kernel-trace --stats of the bench command for each workload (c2, c3, c4)
#   2. PMC passes over one pass of C3 and of C4 (each counter set in a run of its own):
#        matrix-pipe busy cycles + GPU-active cycles, FETCH_SIZE, WRITE_SIZE
# Results: gpurun_out/r03_final/*  (copy what is to be kept into profiles/).
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/r03_final
mkdir -p "$out"
for wl in c2 c3 c4; do
  steps=20; [ $wl = c4 ] && steps=3
  rm -rf /tmp/prof_$wl
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o $wl -- \
      python3 bench.py --only --workload $wl --steps $steps --warmup 2 --no-cpu > "$out/bench_${wl}_under_rocprof.json" 2> "$out/prof_$wl.err" \
      || { echo "kernel trace $wl failed"; tail -3 "$out/prof_$wl.err"; exit 1; }
  cp /tmp/prof_$wl/*kernel_stats.csv "$out/${wl}_kernel_stats.csv"
  echo "== $wl"; python3 scratch/stats_table.py "$out/${wl}_kernel_stats.csv" 8
done
for wl in c3 c4; do
  for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/pmc_${wl}_$name
    timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_${wl}_$name -o $wl -- \
        python3 bench.py --only --workload $wl --steps 1 --warmup 0 --no-cpu > "$out/pmc_${wl}_$name.log" 2>&1 \
        || { echo "pmc pass $wl $name failed"; tail -3 "$out/pmc_${wl}_$name.log"; exit 1; }
    cp $(find /tmp/pmc_${wl}_$name -name '*counter_collection.csv' | head -1) "$out/pmc_${wl}_${name}_raw.csv"
  done
  python3 scratch/pmc_summary.py "$out" $wl || exit 1
done
