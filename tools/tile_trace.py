#!/usr/bin/env python3
"""Per-launch durations of the fused tile operator in one training / inference step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2
    python3 tools/tile_trace.py DIR/run_kernel_trace.csv [launches per step]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
tile = [r for r in rows if 'tile_mlp_kernel' in r['Kernel_Name']]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6        # steps in the trace (warm-up included)
per = len(tile) // steps
last = tile[-per:]
tot = 0.0
for r in last:
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += us
    print('grid %4d  %8.1f us' % (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), us))
print('%d tile launches per step, %.3f ms' % (per, tot / 1e3))
