# kernel trace of the C2 joint-iteration probe (graph mode, 23 iterations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_joint_c2 -o j -- python3 scratch/joint_c2_probe.py > gpurun_out/prof_joint_c2.log 2>&1
