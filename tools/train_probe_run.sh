python tools/train_step_probe.py 10 2>&1 | tail -2
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/train_probe -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py 5 > $GRAFT_REPO_ROOT/gpurun_out/train_probe.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/train_probe.log
f=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/train_probe/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/train_probe_kernel_stats.csv
