# kernel trace of the C3 joint-iteration probe (eager + graph mode, 16 iterations in all)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_joint_c3 -o j -- python3 scratch/joint_c3_probe.py > gpurun_out/prof_joint_c3.log 2>&1
