#!/bin/bash
# PMC passes (separate runs per counter, --pmc only: no tracing domains) for the fused tile operator; summaries under gpurun_out/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-r03}
mkdir -p gpurun_out/pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc/tile_$c -o run -- python3 $R/tools/pmc_tile.py > $R/gpurun_out/pmc/tile_$c.log 2>&1
  echo "$c done" >> $R/gpurun_out/pmc/progress.txt
done
cd $R
python3 tools/summarize_prof.py pmc gpurun_out/pmc/tile_FETCH_SIZE/run_counter_collection.csv gpurun_out/pmc/tile_WRITE_SIZE/run_counter_collection.csv gpurun_out/${P}_pmc_tile.json
tail -2 gpurun_out/pmc/tile_FETCH_SIZE.log
python3 - <<PY
import json
for r in json.load(open('gpurun_out/${P}_pmc_tile.json')):
    print(r['kernel'][:40], r['counter'], r['dispatches'], round(r['mean_per_dispatch']), r['min'], r['max'])
PY
