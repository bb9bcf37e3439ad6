#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_artefacts.sh [prefix]'): the measurement set kept under profiles/ -- default bench
# line, inference line, graph bench, 2-rank gloo rehearsal of `bench.py --gpus 2` (self-launched) on one card, rocprofv3 kernel stats
# of the training / inference / 128-question benches, row-kernel byte accounting.  Outputs go to gpurun_out/ under the prefix
# (default r04_a); copy what is to be judged into profiles/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-r04_a}
mkdir -p gpurun_out/fin3
timeout -k 10 900 python bench.py > gpurun_out/${P}_bench_default.json 2> gpurun_out/fin3/bench.err
echo "bench default done" >> gpurun_out/fin3/progress.txt
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --steps 6 > gpurun_out/${P}_bench_infer.json 2> gpurun_out/fin3/infer.err
timeout -k 10 300 python tools/graph_bench.py > gpurun_out/${P}_graph_bench.txt 2>&1
echo "graph bench done" >> gpurun_out/fin3/progress.txt
STAIR_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${P}_bench_n2_gloo_rehearsal.json 2> gpurun_out/fin3/n2.err || echo "n2 rehearsal failed" >> gpurun_out/fin3/progress.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin3/train -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/fin3/train.log 2>&1
cp $R/gpurun_out/fin3/train/run_kernel_stats.csv $R/gpurun_out/${P}_kernel_stats_bench_train.csv
rm -f $R/gpurun_out/fin3/train/run_kernel_trace.csv
echo "train stats done" >> $R/gpurun_out/fin3/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin3/infer -o run -- python3 $R/bench.py --mode infer --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/fin3/infer.log 2>&1
cp $R/gpurun_out/fin3/infer/run_kernel_stats.csv $R/gpurun_out/${P}_kernel_stats_bench_infer.csv
rm -f $R/gpurun_out/fin3/infer/run_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin3/b128 -o run -- python3 $R/bench.py --batch 128 --no-extras --no-cpu-baseline --steps 40 --warmup 4 > $R/gpurun_out/fin3/b128.log 2>&1
cp $R/gpurun_out/fin3/b128/run_kernel_stats.csv $R/gpurun_out/${P}_kernel_stats_bench_train_b128.csv
rm -f $R/gpurun_out/fin3/b128/run_kernel_trace.csv
echo "b128 stats done" >> $R/gpurun_out/fin3/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin3/rows -o run -- python3 $R/tools/row_kernels.py $R/gpurun_out/fin3/acct.json > $R/gpurun_out/fin3/rows.log 2>&1
rm -f $R/gpurun_out/fin3/rows/run_kernel_trace.csv
cd $R
timeout -k 10 200 python3 tools/supervised_host_time.py > gpurun_out/${P}_supervised_host_time.txt 2>&1 || true
timeout -k 10 200 python3 tools/determinism_probe.py 64 3 all > gpurun_out/${P}_determinism_probe.txt 2>&1 || true
timeout -k 10 200 python3 tools/determinism_probe.py 2048 2 paper >> gpurun_out/${P}_determinism_probe.txt 2>&1 || true
timeout -k 10 200 python3 tools/determinism_probe.py 64 4 all 1 >> gpurun_out/${P}_determinism_probe.txt 2>&1 || true      # the supervised step (configs[4])
timeout -k 10 200 python3 tools/chain_bench.py > gpurun_out/${P}_chain_bench.txt 2>&1 || true
python3 tools/row_kernels.py --merge gpurun_out/fin3/acct.json gpurun_out/fin3/rows/run_kernel_stats.csv gpurun_out/${P}_row_kernels.json
tail -c 600 gpurun_out/${P}_bench_default.json
