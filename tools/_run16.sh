set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/graph_bench.py 2>&1 | grep "^B="
timeout -k 10 300 python bench.py --no-extras --supervision --steps 10 --warmup 2 | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('supervised', l['value'], l['ms_per_step'])"
timeout -k 10 300 python bench.py --no-extras --mode infer --steps 20 --warmup 4 | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer', l['value'], l['ms_per_step'])"
