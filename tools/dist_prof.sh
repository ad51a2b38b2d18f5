# rocprofv3 kernel stats of tools/dist_probe.py (the distributed formulation with a world of one): per-kernel cost of pack / unpack / scale
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/distprof -o dp -- python $GRAFT_REPO_ROOT/tools/dist_probe.py 20 > $GRAFT_REPO_ROOT/gpurun_out/distprof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/distprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/distprof_kernel_stats.csv \;
find gpurun_out/distprof -name "*kernel_trace.csv" -size +20M -delete
head -40 gpurun_out/distprof_kernel_stats.csv | cut -c1-200
