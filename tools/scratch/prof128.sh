set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p128 -o run -- python3 $R/bench.py --batch 128 --no-extras --no-cpu-baseline --steps 40 --warmup 4 > $R/gpurun_out/p128.log 2>&1
cp $R/gpurun_out/p128/run_kernel_stats.csv $R/gpurun_out/r04_i_kernel_stats_b128.csv
rm -f $R/gpurun_out/p128/run_kernel_trace.csv
tail -c 300 $R/gpurun_out/p128.log
