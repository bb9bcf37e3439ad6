#!/bin/bash
# Whole-step memory-side traffic: FETCH_SIZE and WRITE_SIZE (separate passes, --pmc only) summed over every kernel of K training
# steps (tools/pmc_step.py) -> gpurun_out/<prefix>_pmc_whole_step.json.  Also the N = 256 calibration launch of the plane GEMM
# (one column tile per A panel: A is read exactly once) -> gpurun_out/<prefix>_pmc_planes_calibration.json.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-r04}
K=${2:-4}
mkdir -p gpurun_out/pmcs
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcs/step_$c -o run -- python3 $R/tools/pmc_step.py $K > $R/gpurun_out/pmcs/step_$c.log 2>&1
  echo "step $c done" >> $R/gpurun_out/pmcs/progress.txt
  STAIR_PLANES_N=256 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcs/cal_$c -o run -- python3 $R/tools/pmc_planes.py > $R/gpurun_out/pmcs/cal_$c.log 2>&1
  echo "cal $c done" >> $R/gpurun_out/pmcs/progress.txt
done
cd $R
python3 tools/summarize_prof.py step $K gpurun_out/pmcs/step_FETCH_SIZE/run_counter_collection.csv gpurun_out/pmcs/step_WRITE_SIZE/run_counter_collection.csv gpurun_out/${P}_pmc_whole_step.json
STAIR_PMC_SHAPE=131072,256,2048 python3 tools/summarize_prof.py pmc gpurun_out/pmcs/cal_FETCH_SIZE/run_counter_collection.csv gpurun_out/pmcs/cal_WRITE_SIZE/run_counter_collection.csv gpurun_out/${P}_pmc_planes_calibration.json
python3 - <<PY
import json
d = json.load(open('gpurun_out/${P}_pmc_whole_step.json'))
print({k: d[k] for k in ('steps', 'fetch_bytes_per_step', 'write_bytes_per_step', 'bytes_per_question')})
for r in json.load(open('gpurun_out/${P}_pmc_planes_calibration.json')):
    print(r['kernel'][:44], r['counter'], r['dispatches'], round(r['mean_per_dispatch']), r.get('shape'))
PY
