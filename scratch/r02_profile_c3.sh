#!/bin/bash
# C3 only: kernel trace of the bench command, then the three PMC passes (see r02_profile_final.sh).
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/r02_final
mkdir -p "$out"
wl=c3
rm -rf /tmp/prof_$wl
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o $wl -- \
    python3 bench.py --only --workload $wl --steps 20 --warmup 2 --no-cpu > "$out/bench_${wl}_under_rocprof.json" 2> "$out/prof_$wl.err" \
    || { echo "kernel trace $wl failed"; tail -3 "$out/prof_$wl.err"; exit 1; }
cp /tmp/prof_$wl/*kernel_stats.csv "$out/${wl}_kernel_stats.csv"
python3 scratch/stats_table.py "$out/${wl}_kernel_stats.csv" 6
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/pmc_${wl}_$name
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_${wl}_$name -o $wl -- \
      python3 bench.py --only --workload $wl --steps 1 --warmup 0 --no-cpu > "$out/pmc_${wl}_$name.log" 2>&1 \
      || { echo "pmc pass $wl $name failed"; tail -3 "$out/pmc_${wl}_$name.log"; exit 1; }
  cp $(find /tmp/pmc_${wl}_$name -name '*counter_collection.csv' | head -1) "$out/pmc_${wl}_${name}_raw.csv"
done
python3 scratch/pmc_summary.py "$out" $wl
