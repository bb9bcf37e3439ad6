#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_artefacts.sh'): the measurement set kept under profiles/ -- default bench line,
# graph bench, 2-rank gloo rehearsal of bench.py on one card, rocprofv3 kernel stats of the training bench, row-kernel byte accounting.
# Outputs go to gpurun_out/ (prefix r02_l_); copy what is to be judged into profiles/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin2
timeout -k 10 900 python bench.py > gpurun_out/r02_l_bench_default.json 2> gpurun_out/fin2/bench.err
timeout -k 10 300 python tools/graph_bench.py > gpurun_out/r02_l_graph_bench.txt 2>&1
HSA_ENABLE_IPC_MODE_LEGACY=0 STAIR_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/r02_l_bench_n2_gloo_rehearsal.json 2> gpurun_out/fin2/n2.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin2/train -o run -- python3 $R/bench.py --no-extras --steps 10 --warmup 2 > $R/gpurun_out/fin2/train.log 2>&1
cp $R/gpurun_out/fin2/train/run_kernel_stats.csv $R/gpurun_out/r02_l_kernel_stats_bench_train.csv
rm -f $R/gpurun_out/fin2/train/run_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin2/rows -o run -- python3 $R/tools/row_kernels.py $R/gpurun_out/fin2/acct.json > $R/gpurun_out/fin2/rows.log 2>&1
rm -f $R/gpurun_out/fin2/rows/run_kernel_trace.csv
cd $R
python3 tools/row_kernels.py --merge gpurun_out/fin2/acct.json gpurun_out/fin2/rows/run_kernel_stats.csv gpurun_out/r02_l_row_kernels.json
grep "^B=" gpurun_out/r02_l_graph_bench.txt
