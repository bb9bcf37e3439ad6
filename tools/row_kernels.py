#!/usr/bin/env python3
"""Per-kernel HBM accounting of the row-kernel family (north_star: "rocprof HBM GB/s against gfx950 peak").

Run under rocprofv3:   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/row_kernels.py <dir>/acct.json
The script runs STEPS training steps of the default bench batch (2048 questions, bf16 clips) with the library's
algorithmic-byte accounting on and writes {kernel: [launches, bytes]} per step; then
    python3 tools/row_kernels.py --merge <dir>/acct.json <kernel_stats.csv> profiles/r02_row_kernels.json
divides those bytes by the kernel durations of the SAME run: GB/s per kernel and its fraction of the 8 TB/s HBM peak."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS, WARM = 6, 2
HBM_PEAK_GBS = 8000.0
# accounting key -> kernel functions that serve it (float4 / fused forms of the same operator)
ALIASES = {'mask_relu_kernel': ('mask_relu_kernel', 'mask_relu_v4_kernel'),
           'bcast_mask_relu_kernel': ('bcast_mask_relu_kernel', 'bcast_mask_relu_v4_kernel'),
           'sum_rows_kernel': ('sum_rows_kernel', 'sum_rows_v4_kernel'),
           'layernorm_bwd_kernel+param_grad': ('layernorm_bwd_kernel', 'layernorm_param_grad_kernel', 'layernorm_bwd_fused_kernel')}


def run(out):
    import ctypes as C
    import torch
    from stair_amd import spec, synth, losses as L
    from stair_amd._lib import lib
    from stair_amd.module_net import VideoNMN
    from stair_amd.train import Trainer
    dev = torch.device('cuda', 0)
    config = dict(spec.DEFAULT_CONFIG)
    w = synth.make_weights(config, 0)
    m = VideoNMN(config, pretrain_modules=set(L.CRITERION_MODULES))
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    m = m.to(dev)
    B = 2048
    qs = [synth.make_question(config, 0, i, T=64, with_video=False) for i in range(B)]
    g = torch.Generator(device=dev).manual_seed(1234)
    video = torch.randn(B, 64, 2048, device=dev, generator=g).to(torch.bfloat16)
    q_lens = [q['question'].shape[0] for q in qs]
    question = torch.randn(sum(q_lens), 300, device=dev, generator=g)
    answers = torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=dev)
    progs = [q['nmn_program_list'] for q in qs]; spans = [q['prog_str_to_question_tokens'] for q in qs]
    tr = Trainer(m, dropout=0.0)
    for _ in range(WARM):
        tr.step(progs, spans, video, question, q_lens, answers)
    torch.cuda.synchronize()
    lib.stair_acct_enable(1)
    for _ in range(STEPS):
        tr.step(progs, spans, video, question, q_lens, answers)
    torch.cuda.synchronize()
    n = lib.stair_acct_dump(None, 0)
    buf = C.create_string_buffer(n)
    lib.stair_acct_dump(buf, n)
    lib.stair_acct_enable(0)
    table = {}
    for line in buf.value.decode().splitlines():
        k, c, b, f = line.split()
        if int(f) == 0 and int(b) > 0:                     # the HBM-bound row kernels (MFMA kernels carry flops)
            table[k] = [int(c) / STEPS, int(b) / STEPS]
    json.dump({'steps_accounted': STEPS, 'steps_profiled': STEPS + WARM, 'per_step': table}, open(out, 'w'), indent=1)


def merge(acct, stats, out):
    a = json.load(open(acct))
    nprof = a['steps_profiled']
    dur = {}
    for r in csv.DictReader(open(stats)):
        name = r['Name']
        m = re.search(r'stair::([A-Za-z0-9_]+)', name)
        fn = m.group(1) if m else ''
        for k in a['per_step']:
            if fn in ALIASES.get(k, (k,)):
                dur[k] = dur.get(k, 0.0) + float(r['TotalDurationNs']) / nprof / 1e3          # us per step
    rows = []
    for k, (calls, nbytes) in sorted(a['per_step'].items(), key=lambda kv: -dur.get(kv[0], 0)):
        us = dur.get(k)
        rows.append({'kernel': k, 'launches_per_step': calls, 'algorithmic_bytes_per_step': nbytes, 'us_per_step': round(us, 1) if us else None,
                     'GBps': round(nbytes / us / 1e3, 1) if us and nbytes else None,
                     'frac_of_hbm_peak': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 3) if us and nbytes else None})
    tot_us = sum(r['us_per_step'] or 0 for r in rows)
    json.dump({'note': 'algorithmic bytes (inputs once + outputs once, as each launcher reports them: csrc STAIR_ACCT) / rocprofv3 kernel time of the '
                       'same run; 2048-question training step, bf16 clips; HBM peak 8000 GB/s (6300 achievable, MI355X_MICROARCH.md)',
               'row_family_accounted_us_per_step': round(tot_us, 1), 'kernels': rows}, open(out, 'w'), indent=1)
    for r in rows:
        print('%-34s x%-5.1f %9.1f MB %8s us %8s GB/s  %s' % (r['kernel'], r['launches_per_step'], r['algorithmic_bytes_per_step'] / 1e6,
                                                          r['us_per_step'], r['GBps'], r['frac_of_hbm_peak']))


if __name__ == '__main__':
    if sys.argv[1] == '--merge':
        merge(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        run(sys.argv[1])
