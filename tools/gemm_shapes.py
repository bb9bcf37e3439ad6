#!/usr/bin/env python3
"""Which GEMM shapes does a training step launch, and what does each cost?

  STAIR_GEMM_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/gemm_shapes.py run
  python3 tools/gemm_shapes.py join DIR/run_kernel_trace.csv SHAPES.txt OUT.csv

`run` executes `--steps` training steps of the bench workload with the launchers' shape trace on (stderr lines
"STAIR_GEMM nt|tn|planes M= N= K= ...", one per GEMM launch, in launch order) and writes them to gpurun_out/gemm_shapes.txt;
`join` walks the kernel trace in start order, pairs every GEMM main kernel (gemm_* except the split-K reduction and the
plane split passes) with the next shape line and prints per (kind, M, N, K): launches per step, mean time, executed and
algorithmic TFLOP/s."""
import sys, os, csv, collections, re

if sys.argv[1] == 'run':
    # in-process (a profiler follows this process): stderr goes to the shape file, the launchers' trace is switched on
    # before the library loads, then bench.py runs a few training steps
    import runpy
    os.makedirs('gpurun_out', exist_ok=True)
    os.environ['STAIR_GEMM_TRACE'] = '1'
    fd = os.open('gpurun_out/gemm_shapes.txt', os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 2)
    sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'), '--no-extras', '--steps', '3', '--warmup', '0'] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name='__main__')
    sys.exit(0)

trace, shapes_file, out = sys.argv[2:5]
shapes = [dict([kv.split('=') for kv in l.split()[2:]], kind=l.split()[1]) for l in open(shapes_file) if l.startswith('STAIR_GEMM')]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
main = [r for r in rows if re.search(r'stair::gemm_', r['Kernel_Name']) and 'reduce' not in r['Kernel_Name'] and 'split_planes' not in r['Kernel_Name']]
print(len(main), 'GEMM kernels in the trace,', len(shapes), 'shape lines')
n = min(len(main), len(shapes))
agg = collections.defaultdict(list)
for r, sh in zip(main[-n:], shapes[-n:]):        # align at the end: warm-up launches before the trace began are dropped
    kern = r['Kernel_Name'].split('(')[0].replace('void stair::', '')[:44]
    agg[(sh['kind'], int(sh['M']), int(sh['N']), int(sh['K']), sh['act'], sh['acc'], kern)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kind', 'M', 'N', 'K', 'act', 'acc', 'kernel', 'launches', 'mean_us', 'total_us', 'algorithmic_TFLOPs'])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fl = 2.0 * k[1] * k[2] * k[3]
        w.writerow(list(k) + [len(v), round(sum(v) / len(v), 1), round(sum(v), 1), round(fl / (sum(v) / len(v)) / 1e6, 1)])
