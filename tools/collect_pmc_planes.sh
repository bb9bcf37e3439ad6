#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --pmc only) for the dominant launch (tools/pmc_planes.py) -> gpurun_out/<prefix>_pmc_dominant.json
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-r03}
mkdir -p gpurun_out/pmcp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcp/$c -o run -- python3 $R/tools/pmc_planes.py > $R/gpurun_out/pmcp/$c.log 2>&1
  echo "$c done" >> $R/gpurun_out/pmcp/progress.txt
done
cd $R
STAIR_PMC_SHAPE=131072,2048,2048 python3 tools/summarize_prof.py pmc gpurun_out/pmcp/FETCH_SIZE/run_counter_collection.csv gpurun_out/pmcp/WRITE_SIZE/run_counter_collection.csv gpurun_out/${P}_pmc_dominant.json
python3 - <<PY
import json
for r in json.load(open('gpurun_out/${P}_pmc_dominant.json')):
    print(r['kernel'][:44], r['counter'], r['dispatches'], round(r['mean_per_dispatch']), r.get('shape'))
PY
