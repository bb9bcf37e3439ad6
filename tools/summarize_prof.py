#!/usr/bin/env python3
"""Turn rocprofv3 outputs under gpurun_out/ into the small summaries kept in profiles/.
  summarize_prof.py stats <results.db> <out.csv>            per-kernel calls / total / average / share
  summarize_prof.py pmc <counter_collection.csv>... <out.json>   mean counter value per kernel
  summarize_prof.py bygrid <kernel_trace.csv> <out.csv>      per (kernel, launch grid): calls, total, average, min"""
import os, sys, csv, json, sqlite3, collections


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name[:160], calls, round(tot, 3), round(avg, 3), round(pct, 3)])
        # the GEMM kernels serve many shapes: break them down by launch grid so that one shape's average can be read off
        w.writerow([])
        w.writerow(['Name', 'GridWorkgroups', 'Calls', 'TotalDurationUs', 'AverageUs'])
        for name, wg, calls, tot in c.execute(
                "select name, grid_x / workgroup_x, count(*), sum(end - start) / 1e3 from kernels where name like '%gemm%' "
                "group by name, grid_x / workgroup_x order by 4 desc limit 24"):
            w.writerow([name.split('(')[0][:80], wg, calls, round(tot, 3), round(tot / calls, 3)])


def bygrid(trace, out):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        wg = [max(1, int(r['Workgroup_Size_' + a])) for a in 'XYZ']
        grid = tuple(int(r['Grid_Size_' + a]) // w for a, w in zip('XYZ', wg))
        agg[(r['Kernel_Name'].split('(')[0][:90], grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'GridWorkgroups', 'Calls', 'TotalUs', 'AverageUs', 'MinUs'])
        for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, 'x'.join(map(str, grid)), len(v), round(sum(v), 1), round(sum(v) / len(v), 2), round(min(v), 2)])


def pmc(files, out):
    res = []
    for fn in files:
        agg = collections.defaultdict(list)
        grid = {}
        for r in csv.DictReader(open(fn)):
            if 'stair::' not in r['Kernel_Name']:
                continue
            k = (r['Kernel_Name'].split('(')[0], r['Counter_Name'])
            agg[k].append(float(r['Counter_Value']))
            grid[k] = int(r['Grid_Size'])
        shape = [int(x) for x in os.environ['STAIR_PMC_SHAPE'].split(',')] if os.environ.get('STAIR_PMC_SHAPE') else None
        for (kern, ctr), v in agg.items():
            row = {'file': fn.split('/')[-3] if fn.split('/')[-2].startswith('run') else fn.split('/')[-2], 'kernel': kern, 'counter': ctr,
                   'dispatches': len(v), 'grid_threads': grid[(kern, ctr)], 'mean_per_dispatch': sum(v) / len(v), 'min': min(v), 'max': max(v)}
            if shape and ('gemm' in kern):
                row['shape'] = shape            # M, N, K of the launches in this run (tools/pmc_*.py print it)
            res.append(row)
    json.dump(res, open(out, 'w'), indent=1)


def step(k, fetch_csv, write_csv, out):
    """Whole-step traffic: both counters summed over every kernel of the run (setup kernels of torch included: they are listed),
    per kernel family and in total, divided by the K steps of tools/pmc_step.py."""
    doc = {'steps': k, 'unit': 'KiB as rocprofv3 reports them; bytes = x 1024', 'per_kernel': {}}
    tot = {}
    for ctr, fn in (('FETCH_SIZE', fetch_csv), ('WRITE_SIZE', write_csv)):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(fn)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
            agg[name][0] += 1
            agg[name][1] += float(r['Counter_Value'])
        tot[ctr] = sum(v[1] for n, v in agg.items() if 'stair::' in n)
        for n, v in agg.items():
            doc['per_kernel'].setdefault(n, {})[ctr] = {'dispatches': v[0], 'sum_kib': round(v[1], 1)}
    q = 2048
    doc['fetch_bytes_per_step'] = int(tot['FETCH_SIZE'] * 1024 / k)
    doc['write_bytes_per_step'] = int(tot['WRITE_SIZE'] * 1024 / k)
    doc['questions_per_step'] = q
    doc['bytes_per_question'] = int((tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / k / q)
    doc['note'] = ('stair:: kernels only, K steps incl. the first (lazy one-time weight packing is part of every step anyway); FETCH_SIZE as '
                   'reported: MI355X_MICROARCH.md says 16-byte-per-lane streaming reads are tallied at half their bytes on gfx950, so the '
                   'true read side lies between fetch_bytes_per_step and twice that')
    json.dump(doc, open(out, 'w'), indent=1)


if __name__ == '__main__':
    if sys.argv[1] == 'step':
        step(int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]); sys.exit(0)
    if sys.argv[1] == 'bygrid':
        bygrid(sys.argv[2], sys.argv[3]); sys.exit(0)
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
