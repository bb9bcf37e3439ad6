set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p2048 -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/p2048.log 2>&1
cp $R/gpurun_out/p2048/run_kernel_stats.csv $R/gpurun_out/r04_o_kernel_stats_train.csv
rm -f $R/gpurun_out/p2048/run_kernel_trace.csv
