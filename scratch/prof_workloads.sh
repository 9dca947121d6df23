# kernel stats of the three bench workloads, one profiler run each (headline workload only, no CPU legs)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in c2 c3 c4; do
  steps=20; [ $w = c4 ] && steps=3
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -o p -- python3 bench.py --workload $w --only --no-cpu --no-training --steps $steps --warmup 2 > gpurun_out/prof_bench_$w.json 2> gpurun_out/prof_bench_$w.err || exit 1
done
